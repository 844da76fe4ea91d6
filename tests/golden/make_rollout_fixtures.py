#!/usr/bin/env python
"""DoNothing rollouts of the UNMODIFIED reference Environment (build container only) -> tests/golden/rollout_*.npz.

    python tests/golden/make_rollout_fixtures.py

What the batched device-side step (gpf_step_n: chronics row -> injections -> power flow -> overflow counters / cascade,
maintenance) must reproduce step for step: for every scenario of the environment's chronics folder (grid2op/Chronics/
multiFolder.py picks them in sorted order; environment.py:431-437 feeds them to the backend) the observation's rho,
line_status, time-step overflow counters and line cooldowns (time_before_cooldown_line) of N consecutive ``env.step(do_nothing)``, with the environment's DEFAULT
parameters (overflow disconnections on) and its thermal limits.  The backend under the environment is the façade over the CPU
oracle (tests/conformance_backend.py); the opponent of l2rpn_neurips_2020_track1 is switched off (it is outside the Backend
boundary and would attack lines at random)."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REFERENCE = os.environ.get("GRID2OP_REFERENCE", "/root/reference")
for p in (ROOT, os.path.join(ROOT, "tests"), REFERENCE, os.path.join(ROOT, "tests", "_refshim")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("_GRID2OP_FORCE_TEST", "1")
warnings.filterwarnings("ignore")

import grid2op  # noqa: E402
from grid2op.Opponent import BaseOpponent  # noqa: E402
from grid2op.Action import DontAct  # noqa: E402

from conformance_backend import OracleHipBackend  # noqa: E402


def rollout(env_name, n_steps, n_scen, tight_lines=None):
    """`tight_lines`: the thermal limits of these lines are set to 0.8 x their initial flow, so that they overflow, are disconnected by the
    protections after NB_TIMESTEP_OVERFLOW_ALLOWED steps and start a reconnection cooldown (the default limits never trip anything)."""
    env = grid2op.make(env_name, test=True, backend=OracleHipBackend(), opponent_class=BaseOpponent, opponent_action_class=DontAct,
                       opponent_init_budget=0.0, opponent_budget_per_ts=0.0)
    th_tight = None
    if tight_lines is not None:
        env.set_id(0)
        o0 = env.reset()
        th_tight = env.get_thermal_limit().copy()
        th_tight[tight_lines] = 0.8 * o0.a_or[tight_lines]
        env.set_thermal_limit(th_tight)
    out = {"thermal_limit": env.get_thermal_limit().astype(np.float32), "hard_overflow": np.float32(env.parameters.HARD_OVERFLOW_THRESHOLD),
           "nb_ts_allowed": np.int32(env.parameters.NB_TIMESTEP_OVERFLOW_ALLOWED), "scenarios": []}
    out["nb_ts_reco"] = np.int32(env.parameters.NB_TIMESTEP_RECONNECTION)
    rho, ls, ovc, done_at, cool = [], [], [], [], []
    for k in range(n_scen):
        env.set_id(k)
        obs = env.reset()
        if th_tight is not None:
            env.set_thermal_limit(th_tight)
        out["scenarios"].append(os.path.basename(env.chronics_handler.get_id()))
        r_, l_, o_, c_ = [], [], [], []
        d_at = -1
        for t in range(n_steps):
            obs, _, done, info = env.step(env.action_space())
            if done:
                d_at = t
                break
            r_.append(obs.rho.copy())
            l_.append(obs.line_status.copy())
            o_.append(obs.timestep_overflow.copy())
            c_.append(obs.time_before_cooldown_line.copy())      # BaseEnv._times_before_line_status_actionable (baseEnv.py:3352-3358, 2590-2597)
        pad = n_steps - len(r_)
        nl = type(env).n_line
        rho.append(np.concatenate([np.array(r_, np.float32).reshape(-1, nl), np.full((pad, nl), np.nan, np.float32)]))
        ls.append(np.concatenate([np.array(l_, bool).reshape(-1, nl), np.zeros((pad, nl), bool)]))
        ovc.append(np.concatenate([np.array(o_, np.int32).reshape(-1, nl), np.zeros((pad, nl), np.int32)]))
        cool.append(np.concatenate([np.array(c_, np.int32).reshape(-1, nl), np.zeros((pad, nl), np.int32)]))
        done_at.append(d_at)
    env.close()
    out.update(rho=np.stack(rho), line_status=np.stack(ls), timestep_overflow=np.stack(ovc), time_before_cooldown_line=np.stack(cool),
               done_at=np.array(done_at, np.int32),
               scenarios=np.array(out["scenarios"]))
    return out


def chronics_rows(env_name, n_rows):
    """the first rows of every scenario, decoded by the product loader (checked against the reference readers in
    tests/test_chronics_loader.py)"""
    from grid2op_amd.chronics import load_chronics_multifolder
    from grid2op_amd.grid_model import GridModel
    m = GridModel.load_npz(os.path.join(HERE, f"{env_name}.grid.npz"))
    base = os.path.join(REFERENCE, "grid2op", "data", env_name)
    names, ch = load_chronics_multifolder(os.path.join(base, "chronics"), m, prods_charac=os.path.join(base, "prods_charac.csv"),
                                          max_rows=n_rows, truncate=True)
    return names, ch


def main():
    for env_name, n_steps, n_scen, tight, suffix in (("l2rpn_case14_sandbox", 60, 3, None, ""), ("l2rpn_neurips_2020_track1", 130, 2, None, ""),
                                                     ("l2rpn_case14_sandbox", 50, 3, [3, 12], "_trips")):
        d = rollout(env_name, n_steps, n_scen, tight)
        names, ch = chronics_rows(env_name, n_steps + 2)
        assert names[:n_scen] == [str(x) for x in d["scenarios"]]
        for k, v in ch.items():
            d["chron_" + k] = v[:n_scen]
        if "maintenance" in ch:
            # the fixture keeps a WINDOW of the chronics: how long a maintenance at its end lasts is only known from the full column -- the
            # reference's own GridValue.get_maintenance_duration_1d on it (what the environment's duration_next_maintenance is made of)
            from grid2op.Chronics import GridValue
            _, full = chronics_rows(env_name, None)
            mt = full["maintenance"][:n_scen]
            dur = np.zeros(mt.shape, np.uint16)
            for s_ in range(mt.shape[0]):
                for l in range(mt.shape[2]):
                    dur[s_, :, l] = np.minimum(GridValue.get_maintenance_duration_1d(mt[s_, :, l]), 65535)
            d["chron_maintenance_duration"] = np.where(mt > 0, dur, 0)[:, :n_steps + 2].astype(np.uint16)
        path = os.path.join(HERE, f"rollout_{env_name}{suffix}.npz")
        np.savez_compressed(path, **d)
        print(env_name, d["scenarios"], "done_at", d["done_at"], "lines off at the end", (~d["line_status"][:, -1]).sum(axis=-1),
              "max rho", np.nanmax(d["rho"]), f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()

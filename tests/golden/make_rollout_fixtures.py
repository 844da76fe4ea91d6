#!/usr/bin/env python
"""DoNothing rollouts of the UNMODIFIED reference Environment (build container only) -> tests/golden/rollout_*.npz.

    python tests/golden/make_rollout_fixtures.py

What the batched device-side step (gpf_step_n: chronics row -> injections -> power flow -> overflow counters / cascade,
maintenance) must reproduce step for step: for every scenario of the environment's chronics folder (grid2op/Chronics/
multiFolder.py picks them in sorted order; environment.py:431-437 feeds them to the backend) the observation's rho,
line_status and time-step overflow counters of N consecutive ``env.step(do_nothing)``, with the environment's DEFAULT
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


def rollout(env_name, n_steps, n_scen):
    env = grid2op.make(env_name, test=True, backend=OracleHipBackend(), opponent_class=BaseOpponent, opponent_action_class=DontAct,
                       opponent_init_budget=0.0, opponent_budget_per_ts=0.0)
    out = {"thermal_limit": env.get_thermal_limit().astype(np.float32), "hard_overflow": np.float32(env.parameters.HARD_OVERFLOW_THRESHOLD),
           "nb_ts_allowed": np.int32(env.parameters.NB_TIMESTEP_OVERFLOW_ALLOWED), "scenarios": []}
    rho, ls, ovc, done_at = [], [], [], []
    for k in range(n_scen):
        env.set_id(k)
        obs = env.reset()
        out["scenarios"].append(os.path.basename(env.chronics_handler.get_id()))
        r_, l_, o_ = [], [], []
        d_at = -1
        for t in range(n_steps):
            obs, _, done, info = env.step(env.action_space())
            if done:
                d_at = t
                break
            r_.append(obs.rho.copy())
            l_.append(obs.line_status.copy())
            o_.append(obs.timestep_overflow.copy())
        pad = n_steps - len(r_)
        nl = type(env).n_line
        rho.append(np.concatenate([np.array(r_, np.float32).reshape(-1, nl), np.full((pad, nl), np.nan, np.float32)]))
        ls.append(np.concatenate([np.array(l_, bool).reshape(-1, nl), np.zeros((pad, nl), bool)]))
        ovc.append(np.concatenate([np.array(o_, np.int32).reshape(-1, nl), np.zeros((pad, nl), np.int32)]))
        done_at.append(d_at)
    env.close()
    out.update(rho=np.stack(rho), line_status=np.stack(ls), timestep_overflow=np.stack(ovc), done_at=np.array(done_at, np.int32),
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
    for env_name, n_steps, n_scen in (("l2rpn_case14_sandbox", 60, 3), ("l2rpn_neurips_2020_track1", 130, 2)):
        d = rollout(env_name, n_steps, n_scen)
        names, ch = chronics_rows(env_name, n_steps + 2)
        assert names[:n_scen] == [str(x) for x in d["scenarios"]]
        for k, v in ch.items():
            d["chron_" + k] = v[:n_scen]
        path = os.path.join(HERE, f"rollout_{env_name}.npz")
        np.savez_compressed(path, **d)
        print(env_name, d["scenarios"], "done_at", d["done_at"], "lines off at the end", (~d["line_status"][:, -1]).sum(axis=-1),
              "max rho", np.nanmax(d["rho"]), f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()

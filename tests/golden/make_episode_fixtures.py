#!/usr/bin/env python
"""Record episodes of the UNMODIFIED reference Environment at the Backend plugin boundary (build container only).

    python tests/golden/make_episode_fixtures.py        # writes tests/golden/episodes/*.npz

The reference package is imported from /root/reference (with the pandapower stand-in of tests/_refshim: the reference
imports pandapower at module level but these episodes never call it); the backend under the environment is the façade
`grid2op_amd.backend.HipBackend` over the CPU oracle (tests/conformance_backend.py), wrapped so that EVERY call the
framework makes on EVERY backend instance is logged (format: tests/replay.py).  The ``-m gpu`` test
tests/test_episode_replay.py replays the logs through the façade over the real HIP engine.

Episodes (reference call sites exercised: Environment/environment.py:300-349 init, Environment/baseEnv.py:3562-3931 step,
Backend/backend.py:1433-1521 next_grid_state, Observation/observationSpace.py:218-254 + _obsEnv.py:321-503 simulate,
Reward/n1Reward.py:70-99, Runner/runner.py:739-766):
"""
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
from grid2op.Parameters import Parameters  # noqa: E402
from grid2op.Reward import N1Reward, L2RPNReward  # noqa: E402
from grid2op.Runner import Runner  # noqa: E402
from grid2op.Agent import DoNothingAgent  # noqa: E402

import replay as R  # noqa: E402


def _default_base():
    from conformance_backend import OracleHipBackend      # facade over the CPU oracle (build container: no GPU)
    return OracleHipBackend


class Trace:
    def __init__(self):
        self.ev = []                 # (kind, bid, arg, row)
        self.acts = []
        self.pfs = []
        self.obs = []
        self.n_bid = 0
        self.last_pf_row_of = {}
        self.last_pf_row = -1

    def new_bid(self):
        self.n_bid += 1
        return self.n_bid - 1

    def add(self, kind, bid, arg=0, row=-1):
        self.ev.append((kind, bid, arg, row))

    def mark_obs(self, obs, pf_row, thermal_limit):
        d = {f: np.asarray(getattr(obs, f)).copy() for f in R.OBS_FLOAT + R.OBS_INT}
        d["thermal_limit"] = np.asarray(thermal_limit, dtype=np.float32).copy()
        d["pf_row"] = pf_row
        self.obs.append(d)

    def save(self, path, meta):
        out = {"ev_kind": np.array([e[0] for e in self.ev], np.int32), "ev_bid": np.array([e[1] for e in self.ev], np.int32),
               "ev_arg": np.array([e[2] for e in self.ev], np.int32), "ev_row": np.array([e[3] for e in self.ev], np.int32)}
        for f in R.ACT_FIELDS:
            out[f"act_{f}_values"] = np.stack([a[f][0] for a in self.acts])
            out[f"act_{f}_changed"] = np.stack([a[f][1] for a in self.acts])
        out["pf_ok"] = np.array([p["ok"] for p in self.pfs], bool)
        for f in R.PF_INT + R.PF_FLOAT:
            out[f"pf_{f}"] = np.stack([p[f] for p in self.pfs])
        if self.obs:
            out["obs_pf_row"] = np.array([o["pf_row"] for o in self.obs], np.int32)
            for f in R.OBS_FLOAT + R.OBS_INT + ("thermal_limit",):
                out[f"obs_{f}"] = np.stack([o[f] for o in self.obs])
        for k, v in meta.items():
            out[f"meta_{k}"] = np.array(v)
        np.savez_compressed(path, **out)
        return out


def recording_backend(base):
    """`base`: a grid2op Backend class -- the facade over the oracle engine in the build container, `HipBackend` over the HIP
    engine on the GPU box (tests/reference_on_hip.py) -- wrapped so that every call of the framework is logged."""

    class RecordingBackend(base):
        trace = None

        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self._bid = type(self).trace.new_bid()

        def load_grid(self, path, filename=None):
            super().load_grid(path, filename)
            type(self).trace.add(R.EV_LOAD, self._bid)

        def copy(self):
            res = super().copy()
            type(self).trace.add(R.EV_COPY, self._bid, res._bid)
            return res

        def apply_action(self, backend_action):
            if backend_action is not None:
                tr = type(self).trace
                (_ab, (prod_p, prod_v, load_p, load_q, storage), topo__, shunts__) = backend_action()
                cls = type(self)
                rec = {}
                for f, vs in (("prod_p", prod_p), ("prod_v", prod_v), ("load_p", load_p), ("load_q", load_q), ("storage", storage),
                              ("topo", topo__)):
                    rec[f] = (np.asarray(vs.values).copy(), np.asarray(vs.changed).copy())
                sp, sq, sb = shunts__
                rec["shunt_p"] = (np.asarray(sp.values).copy(), np.asarray(sp.changed).copy())
                rec["shunt_q"] = (np.asarray(sq.values).copy(), np.asarray(sq.changed).copy())
                rec["shunt_bus"] = (np.asarray(sb.values).copy(), np.asarray(sb.changed).copy())
                if cls.n_storage > 0:
                    stb = backend_action.get_storages_bus()
                    rec["storage_bus"] = (np.asarray(stb.values).copy(), np.asarray(stb.changed).copy())
                else:
                    rec["storage_bus"] = (np.zeros(0, np.int32), np.zeros(0, bool))
                tr.add(R.EV_APPLY, self._bid, 0, len(tr.acts))
                tr.acts.append(rec)
            return super().apply_action(backend_action)

        def runpf(self, is_dc=False):
            ok, exc = super().runpf(is_dc=is_dc)
            tr = type(self).trace
            d = R.read_backend(self)
            d["ok"] = bool(ok)
            tr.add(R.EV_RUNPF, self._bid, int(bool(is_dc)), len(tr.pfs))
            tr.last_pf_row_of[self._bid] = len(tr.pfs)
            tr.last_pf_row = len(tr.pfs)
            tr.pfs.append(d)
            return ok, exc

        def reset(self, path=None, grid_filename=None):
            type(self).trace.add(R.EV_RESET, self._bid)
            return super().reset(path, grid_filename)

        def close(self):
            if self._engine is not None:
                type(self).trace.add(R.EV_CLOSE, self._bid)
            return super().close()

        def _disconnect_line(self, id_):
            type(self).trace.add(R.EV_DISCO, self._bid, int(id_))
            return super()._disconnect_line(id_)

        def _reconnect_line(self, id_):
            type(self).trace.add(R.EV_RECO, self._bid, int(id_))
            return super()._reconnect_line(id_)

    return RecordingBackend


RecordingBackend = None       # built on first use (set_backend_base selects the engine underneath)


def set_backend_base(base):
    global RecordingBackend
    RecordingBackend = recording_backend(base)


def _make(env_name, **kw):
    if RecordingBackend is None:
        set_backend_base(_default_base())
    RecordingBackend.trace = Trace()
    env = grid2op.make(env_name, test=True, backend=RecordingBackend(), **kw)
    return env, RecordingBackend.trace


def _params(**kw):
    p = Parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _step(env, tr, act, simulate=None):
    obs, rew, done, info = env.step(act)
    if not done:
        tr.mark_obs(obs, tr.last_pf_row_of[env.backend._bid], env.get_thermal_limit())
        if simulate is not None:
            sim_obs, _, sim_done, _ = obs.simulate(simulate)
            if not sim_done:
                tr.mark_obs(sim_obs, tr.last_pf_row, env.get_thermal_limit())
    return obs, done, info


def ep_donothing(env_name, n_steps, **kw):
    env, tr = _make(env_name, **kw)
    env.seed(0)
    env.reset()
    for _ in range(n_steps):
        _, done, _ = _step(env, tr, env.action_space(), simulate=env.action_space())
        if done:
            break
    env.close()
    return tr


def ep_topology(env_name, n_steps):
    """bus splits, line disconnection / reconnection, simulate of the next action before playing it"""
    env, tr = _make(env_name, param=_params(NB_TIMESTEP_COOLDOWN_LINE=0, NB_TIMESTEP_COOLDOWN_SUB=0, MAX_SUB_CHANGED=2,
                                             MAX_LINE_STATUS_CHANGED=2, NO_OVERFLOW_DISCONNECTION=True))
    env.seed(1)
    obs = env.reset()
    rng = np.random.default_rng(7)
    sp = env.action_space
    n_sub, n_line = type(env).n_sub, type(env).n_line
    big = [s for s in range(n_sub) if type(env).sub_info[s] >= 4]
    line_off = None
    for t in range(n_steps):
        if t % 4 == 0 and big:                                      # split a substation: alternate its elements
            s = int(rng.choice(big))
            n_el = int(type(env).sub_info[s])
            tgt = 1 + (np.arange(n_el) + int(rng.integers(0, 2))) % 2
            act = sp({"set_bus": {"substations_id": [(s, tgt.astype(int))]}})
        elif t % 4 == 1:                                            # ... and merge it back
            act = sp({"set_bus": {"substations_id": [(s, np.ones(n_el, dtype=int))]}})
        elif t % 4 == 2 and line_off is None:
            line_off = int(rng.integers(0, n_line))
            act = sp({"set_line_status": [(line_off, -1)]})
        elif t % 4 == 3 and line_off is not None:
            act = sp({"set_line_status": [(line_off, +1)]})
            line_off = None
        else:
            act = sp()
        obs, done, info = _step(env, tr, act, simulate=act)
        if done:
            obs = env.reset()
            line_off = None
    env.close()
    return tr


def ep_three_busbars(env_name, n_steps, n_busbar=3):
    """``grid2op.make(..., n_busbar=3)`` (Backend.can_handle_more_than_2_busbar, backend.py:212-260): substations split three ways;
    ``n_busbar=6``: what grid2op/tests/test_issue_l2g_128.py:218 asks of a backend"""
    env, tr = _make(env_name, n_busbar=n_busbar, param=_params(NB_TIMESTEP_COOLDOWN_LINE=0, NB_TIMESTEP_COOLDOWN_SUB=0, MAX_SUB_CHANGED=2,
                                                        NO_OVERFLOW_DISCONNECTION=True))
    env.seed(6)
    env.reset()
    rng = np.random.default_rng(13)
    sp = env.action_space
    big = [s for s in range(type(env).n_sub) if type(env).sub_info[s] >= 5]
    for t in range(n_steps):
        s = int(rng.choice(big))
        n_el = int(type(env).sub_info[s])
        if t % 3 == 2:
            tgt = np.ones(n_el, dtype=int)
        else:
            tgt = 1 + (np.arange(n_el) + int(rng.integers(0, n_busbar))) % n_busbar
        act = sp({"set_bus": {"substations_id": [(s, tgt.astype(int))]}})
        _, done, _ = _step(env, tr, act, simulate=act)
        if done:
            env.reset()
    env.close()
    return tr


def ep_n1reward(env_name, n_steps, lines):
    env, tr = _make(env_name, reward_class=L2RPNReward, other_rewards={f"n1_{l}": N1Reward(l_id=l) for l in lines})
    env.seed(2)
    env.reset()
    for _ in range(n_steps):
        _, done, _ = _step(env, tr, env.action_space())
        if done:
            break
    env.close()
    return tr


def ep_cascade(env_name, n_steps, n_line_pick=None):
    """thermal limits squeezed below the initial flows on the most loaded lines: soft overflows for 2 steps, then trips,
    hard overflows, re-solves inside Backend.next_grid_state -- possibly down to a game over"""
    env, tr = _make(env_name, param=_params(NB_TIMESTEP_OVERFLOW_ALLOWED=2, HARD_OVERFLOW_THRESHOLD=1.5))
    env.seed(3)
    obs = env.reset()
    lim = env.get_thermal_limit().copy()
    order = np.argsort(-obs.rho)
    a = obs.a_or
    k_hard, k_soft = (6, 1) if n_line_pick is None else n_line_pick
    lim[order[k_hard]] = 0.6 * a[order[k_hard]]     # hard overflow (> 1.5): trips at the first step
    lim[order[k_soft]] = 0.93 * a[order[k_soft]]    # soft overflow: trips after NB_TIMESTEP_OVERFLOW_ALLOWED steps
    env.set_thermal_limit(lim)
    for _ in range(n_steps):
        _, done, _ = _step(env, tr, env.action_space())
        if done:                                # game over (the cascade islanded the grid): the episode ends here
            break
    env.close()
    return tr


def ep_storage(env_name, n_steps):
    env, tr = _make(env_name, param=_params(NO_OVERFLOW_DISCONNECTION=True, ACTIVATE_STORAGE_LOSS=False))
    env.seed(4)
    env.reset()
    rng = np.random.default_rng(11)
    n_sto = type(env).n_storage
    for t in range(n_steps):
        act = env.action_space({"set_storage": [(i, float(rng.uniform(-3, 3))) for i in range(n_sto)]}) if t % 2 == 0 else env.action_space()
        if t == 5:
            act = env.action_space({"set_bus": {"storages_id": [(0, 2)]}})       # moves a storage unit alone to busbar 2
        if t == 7:
            act = env.action_space({"set_bus": {"storages_id": [(0, -1)]}})      # switches it off
        if t == 9:
            act = env.action_space({"set_bus": {"storages_id": [(0, 1)]}})
        _, done, _ = _step(env, tr, act, simulate=act)
        if done:
            env.reset()
    env.close()
    return tr


def ep_dc(env_name, n_steps):
    """Parameters.ENV_DC (grid2op/Parameters.py:273): the environment runs runpf(is_dc=True)"""
    env, tr = _make(env_name, param=_params(ENV_DC=True, FORECAST_DC=True, NO_OVERFLOW_DISCONNECTION=True))
    env.seed(5)
    env.reset()
    for t in range(n_steps):
        act = env.action_space({"set_line_status": [(3, -1)]}) if t == 2 else env.action_space()
        _, done, _ = _step(env, tr, act, simulate=env.action_space())
        if done:
            env.reset()
    env.close()
    return tr


def ep_runner(env_name, n_steps):
    """Runner re-instantiates the backend class from its kwargs (Runner/runner.py:739-766)"""
    env, tr = _make(env_name)
    runner = Runner(**env.get_params_for_runner(), agentClass=DoNothingAgent)
    runner.run(nb_episode=2, max_iter=n_steps, env_seeds=[0, 1], nb_process=1)
    env.close()
    return tr


EPISODES = {
    "case14_donothing_simulate": ("l2rpn_case14_sandbox", lambda: ep_donothing("l2rpn_case14_sandbox", 12)),
    "case14_topology": ("l2rpn_case14_sandbox", lambda: ep_topology("l2rpn_case14_sandbox", 24)),
    "case14_n1reward": ("l2rpn_case14_sandbox", lambda: ep_n1reward("l2rpn_case14_sandbox", 4, range(20))),
    "case14_cascade": ("l2rpn_case14_sandbox", lambda: ep_cascade("l2rpn_case14_sandbox", 12, (10, 3))),
    "case14_dc": ("l2rpn_case14_sandbox", lambda: ep_dc("l2rpn_case14_sandbox", 6)),
    "case14_runner": ("l2rpn_case14_sandbox", lambda: ep_runner("l2rpn_case14_sandbox", 5)),
    "case14_three_busbars": ("l2rpn_case14_sandbox", lambda: ep_three_busbars("l2rpn_case14_sandbox", 12)),
    "case14_six_busbars": ("l2rpn_case14_sandbox", lambda: ep_three_busbars("l2rpn_case14_sandbox", 12, n_busbar=6)),
    "storage14_actions": ("educ_case14_storage", lambda: ep_storage("educ_case14_storage", 12)),
    "neurips36_topology": ("l2rpn_neurips_2020_track1", lambda: ep_topology("l2rpn_neurips_2020_track1", 16)),
    "neurips36_cascade": ("l2rpn_neurips_2020_track1", lambda: ep_cascade("l2rpn_neurips_2020_track1", 8)),
    "wcci118_donothing": ("l2rpn_wcci_2022_dev", lambda: ep_donothing("l2rpn_wcci_2022_dev", 4)),
}


def main():
    out_dir = os.path.join(HERE, "episodes")
    os.makedirs(out_dir, exist_ok=True)
    only = sys.argv[1:]
    for name, (grid, fn) in EPISODES.items():
        if only and name not in only:
            continue
        tr = fn()
        d = tr.save(os.path.join(out_dir, f"{name}.npz"), {"grid": grid, "n_busbar": 3 if "three_busbars" in name else 6 if "six_busbars" in name else 2})
        n_div = int((~d["pf_ok"]).sum())
        print(f"{name}: {len(tr.ev)} events, {tr.n_bid} backend instances, {len(tr.pfs)} power flows ({n_div} diverged), "
              f"{len(tr.obs)} observations, {os.path.getsize(os.path.join(out_dir, name + '.npz')) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()

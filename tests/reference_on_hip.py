#!/usr/bin/env python
"""TEST INFRASTRUCTURE (GPU box, staged reference): the UNMODIFIED reference framework on top of `HipBackend` + libgridpf.so.

    GRID2OP_REFERENCE=$PWD/_stage python tests/reference_on_hip.py episodes|aaa|long|wcci|timing [engine]

``tools/stage_reference.py`` copies the reference package into the git-ignored scratch directory ``_stage/`` for one gpurun
call (nothing of the reference enters the tree); this script then puts it on ``sys.path`` together with the pandapower
import stub (tests/_refshim) and drives the REAL ``grid2op.make(..., backend=HipBackend())`` -> ``Environment.step``
(grid2op/Environment/baseEnv.py:3562-3931) -> ``Backend.next_grid_state`` (Backend/backend.py:1433-1521) ->
``obs.simulate`` (Observation/baseObservation.py:3365-3670) / ``N1Reward`` (Reward/n1Reward.py:70-99) / ``Runner``
(Runner/runner.py:739-756) with the HIP engine underneath:

  episodes  the 12 episodes of tests/golden/make_episode_fixtures.py re-run on HipBackend; every call the framework makes on
            every backend instance, every power-flow result and every observation must equal the committed recordings
            (tests/golden/episodes/*.npz; integers bit-exact, floats 2e-4 + 5e-6 |x|)
  aaa       the reference's own backend API kit, AAATestBackendAPI (grid2op/tests/aaa_test_backend_interface.py, 41 tests)
  long      a 288-step DoNothing episode with DEFAULT parameters (protections on) + obs.simulate on 13 candidates at 6 steps,
            HIP engine vs oracle engine step by step
  wcci      one l2rpn_wcci_2022 episode with storage + redispatch + curtailment actions, HIP engine vs oracle engine
  timing    the reference's DoNothing profiler loop (_profiling/profiler_do_nothing.py:42-61) with BOTH engines:
            env._time_powerflow / _time_apply_act / _time_extract_obs / _time_step per step -> one JSON line

The oracle engine (tests/oracle_engine.py) is the checker here, never the thing shipped."""
import json
import os
import sys
import time
import unittest
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = os.environ.get("GRID2OP_REFERENCE", os.path.join(ROOT, "_stage"))
for p in (ROOT, HERE, os.path.join(HERE, "golden"), REFERENCE, os.path.join(HERE, "_refshim")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("_GRID2OP_FORCE_TEST", "1")
os.environ["GRID2OP_REFERENCE"] = REFERENCE
warnings.filterwarnings("ignore")

try:
    import grid2op  # noqa: E402
except Exception as exc:          # say exactly which import fails on the box
    print(f"IMPORT FAILED: {type(exc).__name__}: {exc}")
    raise
from grid2op.Parameters import Parameters  # noqa: E402

import replay as R  # noqa: E402

ABS_TOL, REL_TOL = R.ABS_TOL, R.REL_TOL


def hip_backend_class():
    if os.environ.get("REFERENCE_ON_HIP_DRYRUN") == "1":      # build container (no GPU): exercise this script's own logic only
        class Dry(oracle_backend_class()):
            engines = ["dry run: oracle engine"]
        return Dry
    from grid2op_amd.backend import HipBackend
    from grid2op_amd.engine import PowerFlowEngine
    made = []

    class HipOnDevice(HipBackend):
        """HipBackend as shipped; only records the engines it creates so that the caller can assert they are HIP engines."""
        engines = made

        def _make_engine(self, model, n_busbar, n_lanes=1):
            eng = super()._make_engine(model, n_busbar, n_lanes)
            assert isinstance(eng, PowerFlowEngine), type(eng)
            made.append(eng)
            return eng
    return HipOnDevice


def oracle_backend_class():
    from conformance_backend import OracleHipBackend
    return OracleHipBackend


def _close(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind in "biu":
        assert np.array_equal(a, b), (what, a, b)
        return 0.0
    a, b = a.astype(np.float64), b.astype(np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), (what, "NaN pattern")
    ok = ~nb
    err = np.abs(a[ok] - b[ok])
    assert np.all(err <= ABS_TOL + REL_TOL * np.abs(b[ok])), (what, float(err.max()))
    return float(err.max()) if err.size else 0.0


def cmd_episodes():
    import make_episode_fixtures as F
    Hip = hip_backend_class()
    F.set_backend_base(Hip)
    worst_all = 0.0
    for name, (grid, fn) in F.EPISODES.items():
        ref = dict(np.load(os.path.join(HERE, "golden", "episodes", f"{name}.npz"), allow_pickle=False))
        tr = fn()
        ev = np.array(tr.ev, dtype=np.int64).reshape(-1, 4)
        for k, col in (("ev_kind", 0), ("ev_bid", 1), ("ev_arg", 2), ("ev_row", 3)):
            assert np.array_equal(ev[:, col], ref[k].astype(np.int64)), (name, k, "the framework made a different call sequence")
        assert len(tr.pfs) == len(ref["pf_ok"])
        worst = 0.0
        for i, p in enumerate(tr.pfs):
            assert bool(p["ok"]) == bool(ref["pf_ok"][i]), (name, i)
            for f in R.PF_INT:
                assert np.array_equal(np.asarray(p[f]).astype(np.int64), ref[f"pf_{f}"][i].astype(np.int64)), (name, i, f)
            for f in R.PF_FLOAT:
                worst = max(worst, R._cmp_float(p[f], ref[f"pf_{f}"][i], (name, i, f)))
        n_obs = len(ref["obs_pf_row"]) if "obs_pf_row" in ref else 0
        assert len(tr.obs) == n_obs, (name, len(tr.obs), n_obs)
        for i, o in enumerate(tr.obs):
            assert int(o["pf_row"]) == int(ref["obs_pf_row"][i])
            for f in R.OBS_INT:
                assert np.array_equal(np.asarray(o[f]).astype(np.int64), ref[f"obs_{f}"][i].astype(np.int64)), (name, "obs", i, f)
            for f in R.OBS_FLOAT:
                if f == "rho":
                    g, r_ = np.asarray(o[f], np.float64), ref["obs_rho"][i].astype(np.float64)
                    assert np.all(np.abs(g - r_) <= 1e-5 + 1e-5 * np.abs(r_)), (name, "obs", i, "rho")
                else:
                    worst = max(worst, R._cmp_float(o[f], ref[f"obs_{f}"][i], (name, "obs", i, f)))
        worst_all = max(worst_all, worst)
        print(f"{name}: reproduced on the real framework + HIP engine: {len(tr.ev)} backend calls, {tr.n_bid} backend instances, "
              f"{len(tr.pfs)} power flows, {n_obs} observations, worst |d| = {worst:.2e}", flush=True)
    assert Hip.engines, "no HIP engine was created"
    print(f"EPISODES OK: {len(F.EPISODES)} episodes, {len(Hip.engines)} HIP engines, worst |d| = {worst_all:.2e}")


def cmd_aaa():
    from grid2op.tests.aaa_test_backend_interface import AAATestBackendAPI
    Hip = hip_backend_class()

    class TestBackendAPI_HipBackendOnDevice(AAATestBackendAPI, unittest.TestCase):
        def make_backend(self, detailed_infos_for_cascading_failures=False):
            return Hip(detailed_infos_for_cascading_failures=detailed_infos_for_cascading_failures)
    suite = unittest.defaultTestLoader.loadTestsFromTestCase(TestBackendAPI_HipBackendOnDevice)
    res = unittest.TextTestRunner(verbosity=1, stream=sys.stdout).run(suite)
    n_ok = res.testsRun - len(res.failures) - len(res.errors) - len(res.skipped)
    print(f"AAA: ran {res.testsRun}, passed {n_ok}, failures {len(res.failures)}, errors {len(res.errors)}, skipped {len(res.skipped)}; "
          f"HIP engines created {len(Hip.engines)}")
    for t, msg in res.skipped:
        print("  skipped:", t, msg)
    assert res.wasSuccessful() and Hip.engines
    print("AAA OK")


def _obs_attrs(obs):
    return {a: np.asarray(getattr(obs, a)).copy() for a in type(obs).attr_list_vect}


def _cmp_obs(oa, ob, what):
    worst = 0.0
    for a in oa:
        worst = max(worst, _close(oa[a], ob[a], (what, a)))
    return worst


def _run_long(Bk, n_steps, sim_steps):
    env = grid2op.make("l2rpn_case14_sandbox", test=True, backend=Bk())          # DEFAULT parameters: protections on
    env.seed(0)
    env.set_id(0)
    obs = env.reset()
    sp = env.action_space
    cands = [sp()] + [sp({"set_line_status": [(l, -1)]}) for l in range(10)] + \
            [sp({"set_bus": {"substations_id": [(1, [1, 2, 1, 2, 1, 2])]}}), sp({"set_bus": {"substations_id": [(4, [1, 2, 2, 1, 1])]}})]
    rec, sims = [], []
    for t in range(n_steps):
        if t in sim_steps:
            for c in cands:
                so, sr, sd, si = obs.simulate(c)
                sims.append((t, bool(sd), _obs_attrs(so) if not sd else None, float(sr)))
        obs, rew, done, info = env.step(sp())
        rec.append((bool(done), float(rew), _obs_attrs(obs), list(np.nonzero(info["disc_lines"] >= 0)[0])))
        if done:
            break
    n = len(rec)
    env.close()
    return rec, sims, n


def cmd_long():
    n_steps, sim_steps = 288, (0, 3, 50, 100, 200, 287)
    a, sa, na = _run_long(hip_backend_class(), n_steps, sim_steps)
    b, sb, nb = _run_long(oracle_backend_class(), n_steps, sim_steps)
    assert na == nb, (na, nb)
    worst = 0.0
    for t, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[3] == y[3], (t, x[0], y[0], x[3], y[3])
        assert abs(x[1] - y[1]) <= 1e-3 * max(1.0, abs(y[1])), (t, "reward", x[1], y[1])
        worst = max(worst, _cmp_obs(x[2], y[2], ("step", t)))
    assert len(sa) == len(sb) == 13 * len([s for s in sim_steps if s < na])
    for (t, d1, o1, r1), (_, d2, o2, r2) in zip(sa, sb):
        assert d1 == d2, ("simulate done", t)
        if not d1:
            worst = max(worst, _cmp_obs(o1, o2, ("simulate", t)))
    print(f"LONG OK: {na} env.step of the unmodified Environment (default parameters) + {len(sa)} obs.simulate calls on the HIP engine "
          f"equal the oracle-engine run, worst |d| = {worst:.2e}")


def _run_wcci(Bk, n_steps):
    env = grid2op.make("l2rpn_wcci_2022", test=True, backend=Bk())
    cls = type(env)
    env.seed(6)
    env.set_id(0)
    obs = env.reset()
    rng = np.random.default_rng(6)
    disp = np.nonzero(cls.gen_redispatchable)[0]
    ren = np.nonzero(cls.gen_renewable)[0]
    rec = []
    for t in range(n_steps):
        act = {}
        if t % 4 == 0:
            k = rng.choice(disp, size=2, replace=False)
            amp = cls.gen_max_ramp_up[k] * rng.uniform(0.2, 0.6, 2) * np.array([1.0, -1.0])
            act["redispatch"] = [(int(g), float(a_)) for g, a_ in zip(k, amp)]
            act["set_storage"] = [(int(i), float(rng.uniform(-4.0, 4.0))) for i in range(cls.n_storage)]
        if t % 8 == 2:
            k = rng.choice(ren, size=2, replace=False)
            act["curtail"] = [(int(g), float(rng.uniform(0.3, 0.8))) for g in k]
        if t == 5:
            act["set_bus"] = {"substations_id": [(int(np.argmax(cls.sub_info)), (1 + np.arange(int(cls.sub_info.max())) % 2).astype(int))]}
        obs, rew, done, info = env.step(env.action_space(act))
        rec.append((bool(done), _obs_attrs(obs), bool(info["failed_redispatching"]) if "failed_redispatching" in info else False))
        if done:
            break
    env.close()
    return rec


def cmd_wcci():
    n = 24
    a = _run_wcci(hip_backend_class(), n)
    b = _run_wcci(oracle_backend_class(), n)
    assert len(a) == len(b)
    worst = 0.0
    for t, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[2] == y[2], (t, x[0], y[0])
        worst = max(worst, _cmp_obs(x[1], y[1], ("wcci step", t)))
    print(f"WCCI OK: {len(a)} env.step of l2rpn_wcci_2022 (118 substations) with redispatch + storage + curtailment + a bus split on the HIP "
          f"engine equal the oracle-engine run, worst |d| = {worst:.2e}")


def _time_loop(Bk, env_name, n_steps, label):
    from grid2op.Agent import DoNothingAgent
    from grid2op.Rules import AlwaysLegal
    param = Parameters()
    param.init_from_dict({"NO_OVERFLOW_DISCONNECTION": True})
    env = grid2op.make(env_name, test=True, backend=Bk(), param=param, gamerules_class=AlwaysLegal)
    agent = DoNothingAgent(action_space=env.action_space)
    obs = env.reset()
    for _ in range(20):                                  # warm-up (kernel load, clocks)
        obs, reward, done, info = env.step(agent.act(obs, 0.0, False))
    keys = ("_time_powerflow", "_time_apply_act", "_time_extract_obs", "_time_step")
    acc = {k: 0.0 for k in keys + ("comp_time",)}

    def harvest():                                       # env.reset() clears the environment's timers: collect them before
        for k in keys:
            acc[k] += float(getattr(env, k))
            setattr(env, k, 0.0)
        acc["comp_time"] += float(env.backend.comp_time)
        env.backend.comp_time = 0.0
    harvest()
    acc = {k: 0.0 for k in acc}
    done, reward, n, n_reset = False, env.reward_range[0], 0, 0
    t_reset = 0.0
    t0 = time.perf_counter()
    while n < n_steps:
        act = agent.act(obs, reward, done)
        obs, reward, done, info = env.step(act)
        n += 1
        if done:
            harvest()
            tr = time.perf_counter()
            obs = env.reset()
            t_reset += time.perf_counter() - tr
            n_reset += 1
            for k in keys:                               # (the power flow of the reset is not an env.step)
                setattr(env, k, 0.0)
            env.backend.comp_time = 0.0
    harvest()
    el = time.perf_counter() - t0 - t_reset
    res = {"env": env_name, "backend": label, "steps": n, "episode_resets_excluded": n_reset, "env_steps_per_sec": n / el, "ms_per_step": el / n * 1e3,
           "time_powerflow_ms_per_step": acc["_time_powerflow"] / n * 1e3, "time_apply_act_ms_per_step": acc["_time_apply_act"] / n * 1e3,
           "time_extract_obs_ms_per_step": acc["_time_extract_obs"] / n * 1e3, "time_step_ms_per_step": acc["_time_step"] / n * 1e3,
           "backend_comp_time_ms_per_step": acc["comp_time"] / n * 1e3}
    env.close()
    return {k: (float(v) if hasattr(v, "dtype") else v) for k, v in res.items()}


def cmd_timing():
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    out = {"what": "reference Environment.step loop, DoNothingAgent, NO_OVERFLOW_DISCONNECTION, AlwaysLegal (_profiling/profiler_do_nothing.py:42-61), "
                   "measured on the GPU box with the staged unmodified reference package", "host_cpus": os.cpu_count(), "rows": []}
    for env_name in ("l2rpn_case14_sandbox", "l2rpn_wcci_2022"):
        out["rows"].append(_time_loop(hip_backend_class(), env_name, n, "HipBackend + libgridpf.so (MI355X), one gpf_solve_lane per runpf"))
        if os.environ.get("REFERENCE_ON_HIP_DRYRUN") != "1":
            import functools
            out["rows"].append(_time_loop(functools.partial(hip_backend_class(), specialize=True), env_name, n,
                                          "HipBackend(specialize=True) + libgridpf.so (MI355X): kernels compiled at run time for the grid"))
        out["rows"].append(_time_loop(oracle_backend_class(), env_name, n, "HipBackend facade over the CPU oracle engine (oracle/pf_oracle.c, 1 core)"))
    print("TIMING " + json.dumps(out))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "episodes"
    {"episodes": cmd_episodes, "aaa": cmd_aaa, "long": cmd_long, "wcci": cmd_wcci, "timing": cmd_timing}[cmd]()

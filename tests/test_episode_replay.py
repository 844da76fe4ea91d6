"""Recorded episodes of the unmodified reference Environment (tests/golden/episodes, recorded by
tests/golden/make_episode_fixtures.py) replayed at the Backend plugin boundary.

* ``-m gpu``: through `grid2op_amd.backend.HipBackend` + the real `PowerFlowEngine` (HIP) + `_LanePool` -- the façade, its
  lane sharing (environment backend, obs.simulate copy, N-1 reward copies on ONE engine) and the engine seam, on hardware.
  grid2op is not installable on the GPU box: `HipBackend` is imported on top of the stand-in tests/grid2op_stub.
* CPU: the same replay with the oracle engine, once on the real grid2op ``Backend`` base class (in-process, build container)
  and once on the stand-in in a fresh interpreter (proves the stand-in carries the façade)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, have_reference

EP_DIR = os.path.join(ROOT, "tests", "golden", "episodes")
EPISODES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(EP_DIR, "*.npz")))


def _load(name):
    tr = dict(np.load(os.path.join(EP_DIR, f"{name}.npz")))
    return tr, os.path.join(ROOT, "tests", "golden", f"{str(tr['meta_grid'])}.grid.npz")


def test_fixtures_cover_the_contract():
    assert len(EPISODES) >= 8
    import replay as R
    kinds = set()
    n_div = n_dc = 0
    for name in EPISODES:
        tr, _ = _load(name)
        kinds |= set(tr["ev_kind"].tolist())
        n_div += int((~tr["pf_ok"]).sum())
        n_dc += int(((tr["ev_kind"] == R.EV_RUNPF) & (tr["ev_arg"] == 1)).sum())
    assert {R.EV_LOAD, R.EV_COPY, R.EV_APPLY, R.EV_RUNPF, R.EV_RESET, R.EV_CLOSE, R.EV_DISCO} <= kinds
    assert n_div >= 3 and n_dc >= 5                                 # diverging / islanded power flows and DC mode are in there
    tr, _ = _load("case14_topology")
    assert (tr["pf_topo_vect"] == 2).any() and (tr["pf_topo_vect"] == -1).any()      # bus splits and open lines


@pytest.mark.skipif(not have_reference(), reason="reference checkout not available")
@pytest.mark.parametrize("name", EPISODES)
def test_replay_on_real_grid2op_base_class_with_oracle_engine(name):
    import test_backend_conformance  # noqa: F401  (puts the reference + pandapower stand-in on sys.path)
    from grid2op_amd.backend import HipBackend
    from oracle_engine import OracleEngine
    import replay as R
    from conftest import REFERENCE

    class ReplayOracleBackend(HipBackend):           # a fresh class: the conformance kit leaves grid attributes on the one it used
        def _make_engine(self, model, n_busbar, n_lanes=1):
            return OracleEngine(model, n_lanes=n_lanes, n_busbar=n_busbar)
    tr, grid = _load(name)
    n_pf, n_obs, worst = R.replay(tr, ReplayOracleBackend, grid, env_dir=os.path.join(REFERENCE, "grid2op", "data", str(tr["meta_grid"])))
    assert n_pf == len(tr["pf_ok"]) and worst < 1e-3


def test_replay_on_the_grid2op_stand_in_with_oracle_engine():
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "replay_cli.py"), "oracle"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    assert len([l for l in p.stdout.splitlines() if "reproduced" in l]) == len(EPISODES)


@pytest.mark.gpu
@pytest.mark.parametrize("name", EPISODES)
def test_replay_through_hipbackend_on_the_hip_engine(name):
    try:
        import grid2op  # noqa: F401
    except ImportError:
        sys.path.insert(0, os.path.join(ROOT, "tests", "grid2op_stub"))
    from grid2op_amd.backend import HipBackend, _LanePool
    from grid2op_amd.engine import PowerFlowEngine
    import replay as R
    tr, grid = _load(name)
    seen = []

    class Bk(HipBackend):
        def _make_engine(self, model, n_busbar, n_lanes=1):
            eng = super()._make_engine(model, n_busbar, n_lanes)
            seen.append(eng)
            return eng
    n_pf, n_obs, worst = R.replay(tr, Bk, grid)
    assert n_pf == len(tr["pf_ok"])
    assert n_obs == (len(tr["obs_pf_row"]) if "obs_pf_row" in tr else 0)
    assert seen and all(isinstance(e, PowerFlowEngine) for e in seen)           # the HIP engine, not a stand-in
    n_bk = int((tr["ev_kind"] == R.EV_LOAD).sum() + (tr["ev_kind"] == R.EV_COPY).sum())
    # lane sharing: copies AND re-loads of the same grid file (Runner) take lanes of the same engine
    assert len(seen) <= 1 + n_bk // _LanePool.LANES
    assert not _LanePool._pools


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["case14_topology", "storage14_actions"])
def test_replay_through_hipbackend_with_run_time_specialised_kernels(name):
    """`HipBackend(specialize=True)`: the recorded calls of the reference framework replayed on the engine's kernels compiled at run time
    for the grid (PowerFlowEngine.specialize: self-test against the shipped kernels, then every runpf on the specialised one-lane kernel)"""
    try:
        import grid2op  # noqa: F401
    except ImportError:
        sys.path.insert(0, os.path.join(ROOT, "tests", "grid2op_stub"))
    from grid2op_amd.backend import HipBackend, _LanePool
    import functools
    import replay as R
    tr, grid = _load(name)
    seen = []

    class Bk(HipBackend):
        def _make_engine(self, model, n_busbar, n_lanes=1):
            eng = super()._make_engine(model, n_busbar, n_lanes)
            close = eng.close

            def close_and_report():                       # (the lane pool closes an engine when its last backend is gone)
                seen.append(eng.specialization())
                close()
            eng.close = close_and_report
            return eng
    n_pf, n_obs, worst = R.replay(tr, functools.partial(Bk, specialize=True), grid)
    assert n_pf == len(tr["pf_ok"])
    infos = seen
    assert infos and all(i["enabled"] and i["failed"] == 0 for i in infos), infos
    assert sum(i["launches"] for i in infos) >= n_pf and all("runpf<" in i["variants"] for i in infos if i["launches"]), infos
    assert not _LanePool._pools

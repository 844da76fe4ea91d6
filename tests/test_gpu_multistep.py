"""GPU tests of the multi-step launch (gpf_step_n): n consecutive env steps in one launch must reproduce n single-step
launches (integers bit-exact, floats to rounding of the LDS atomics' summation order), with and without line trips, in DC
mode, with a redispatch delta, with failing lanes and auto-reset; plus the zero-copy device views and the episode counters.
The single-step path itself is checked against the oracle in tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle.pf_oracle import LaneState, solve

pytestmark = pytest.mark.gpu

from test_gpu_parity import _compare  # noqa: E402


def _setup(load_model, load_npz, name, B, seed=0, env=None):
    from grid2op_amd.engine import PowerFlowEngine
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    if "load_p" not in ch:               # no chronics fixture for this grid: 40 jittered copies of the stored state
        rg = np.random.default_rng(99)
        j = lambda v: (np.asarray(v, np.float64)[None, :] * (1 + 0.03 * rg.standard_normal((40, len(v))))).astype(np.float32)  # noqa: E731
        ch.update(load_p=j(m.load_p0), load_q=j(m.load_q0), prod_p=j(m.gen_p0))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        eng = PowerFlowEngine(m, n_lanes=B, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"])
    eng.upload_chronics(tab)
    rng = np.random.default_rng(seed)
    off = (11 * np.arange(B)) % tab.shape[0]
    scale = (1 + 0.05 * rng.standard_normal((B, 2 * m.n_load))).astype(np.float32)
    eng.set_lane_chronics(lane_offset=off, lane_scale=scale)
    if "thermal_limits" in ch:
        eng.set_thermal_limits(ch["thermal_limits"])
    return m, ch, eng, tab, off, scale


def _snapshot(eng):
    r = eng.results()
    rho, ovc, dr = eng.step_outputs()
    return dict(out=r.out.copy(), status=r.status.copy(), topo=r.topo_vect.copy(), ls=r.line_status.copy(), rho=rho, ovc=ovc, dr=dr,
                bus_vm=r.bus_vm.copy(), inj=eng.get_injections(), sb=r.shunt_bus.copy())


def _same(a, b, what):
    for k in ("status", "topo", "ls", "ovc", "dr", "sb"):
        assert np.array_equal(a[k], b[k]), (what, k)
    assert np.array_equal(np.isnan(a["out"]), np.isnan(b["out"])), what
    assert np.allclose(a["out"], b["out"], rtol=2e-6, atol=2e-5, equal_nan=True), (what, np.nanmax(np.abs(a["out"] - b["out"])))
    assert np.allclose(a["rho"], b["rho"], rtol=2e-6, atol=1e-6, equal_nan=True), what
    assert np.allclose(a["bus_vm"], b["bus_vm"], rtol=0, atol=1e-11, equal_nan=True), what
    assert np.array_equal(a["inj"], b["inj"]), what


@pytest.mark.parametrize("name,B,kw", [
    ("rte_case5_example", 37, dict(rebalance=1.02)),                       # 4 instances per wavefront, ragged tail
    ("l2rpn_case14_sandbox", 66, dict(rebalance=1.02)),                    # 2 instances per wavefront
    ("l2rpn_case14_sandbox", 64, dict(rebalance=1.02, is_dc=True)),
    ("l2rpn_neurips_2020_track1", 33, dict(rebalance=1.02)),               # 1 instance per wavefront
    ("l2rpn_wcci_2022_dev", 8, dict(rebalance=1.02)),                      # 2 wavefronts per instance, no room for the DC factors
    ("educ_case14_storage", 16, dict(rebalance=1.0)),
])
def test_multi_step_launch_equals_single_steps(name, B, kw, load_model, load_npz):
    m, ch, e1, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e2, _, _, _ = _setup(load_model, load_npz, name, B)
    n = 7
    e2.set_trajectory(n)
    snaps = []
    for t in range(3, 3 + n):
        e1.step(t, **kw)
        snaps.append(_snapshot(e1))
    e2.step(3, n_steps=n, **kw)
    _same(_snapshot(e2), snaps[-1], name)
    rho, st = e2.trajectory(n)
    for k in range(n):
        assert np.array_equal(st[k], snaps[k]["status"][:, 0]), k
        assert np.allclose(rho[k], snaps[k]["rho"], rtol=2e-6, atol=1e-6, equal_nan=True), k
    assert (snaps[-1]["status"][:, 0] == 0).all()
    # a second multi-step launch continues from the state the first one left (injection rows, counters)
    e1.step(3 + n, **kw)
    e1.step(4 + n, **kw)
    e2.step(3 + n, n_steps=2, **kw)
    _same(_snapshot(e2), _snapshot(e1), name + " second launch")
    done, steps, resets = e2.episode()
    assert not done.any() and (steps == n + 2).all() and (resets == 0).all()
    e1.close()
    e2.close()


@pytest.mark.parametrize("name,B,kw", [
    ("rte_case5_example", 20, dict(rebalance=1.02)),
    ("l2rpn_case14_sandbox", 66, dict(rebalance=1.02)),
    ("l2rpn_case14_sandbox", 32, dict(rebalance=1.02, cascade=True, hard_overflow=1.05, nb_ts_allowed=1)),
    ("l2rpn_neurips_2020_track1", 17, dict(rebalance=1.02)),
    ("l2rpn_wcci_2022_dev", 8, dict(rebalance=1.02)),
])
def test_warm_start_option_same_solution_fewer_iterations(name, B, kw, load_model, load_npz):
    """The opt-in warm start (NOT the reference's algorithm: pandapower re-initialises from the DC solution on every call)
    converges to the same power flow within the solver tolerance, in fewer Newton iterations."""
    m, ch, e1, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e2, _, _, _ = _setup(load_model, load_npz, name, B)
    n = 6
    e1.set_trajectory(n)
    e2.set_trajectory(n)
    e1.step(3, n_steps=n, **kw)
    e2.step(3, n_steps=n, warm_start=True, **kw)
    a, b = _snapshot(e1), _snapshot(e2)
    for k in ("topo", "ls", "ovc", "dr", "sb"):
        assert np.array_equal(a[k], b[k]), (name, k)
    assert np.array_equal(a["status"][:, [0, 2, 3]], b["status"][:, [0, 2, 3]])      # column 1 is n_iter
    ok = a["status"][:, 0] == 0
    assert ok.sum() >= B // 2
    err = np.abs(a["out"] - b["out"])
    assert np.array_equal(np.isnan(a["out"]), np.isnan(b["out"]))
    assert np.all(err[~np.isnan(err)] <= (2e-4 + 5e-6 * np.abs(a["out"]))[~np.isnan(err)]), np.nanmax(err)
    assert np.allclose(a["bus_vm"], b["bus_vm"], rtol=0, atol=1e-8, equal_nan=True)     # pu: both within tol_mva = 1e-8 of the solution
    ra, sa_ = e1.trajectory(n)
    rb, sb_ = e2.trajectory(n)
    assert np.array_equal(sa_, sb_)
    assert np.allclose(ra, rb, rtol=1e-5, atol=1e-6, equal_nan=True)
    it_cold, it_warm = a["status"][ok, 1], b["status"][ok, 1]
    assert (it_warm <= it_cold).all() and it_warm.mean() < it_cold.mean() - 0.5, (it_cold.mean(), it_warm.mean())
    e1.close()
    e2.close()


def test_ybus_in_registers_variant(load_model, load_npz):
    """118 substations with the tables left in global memory (what a 1 024-lane batch runs; GRIDPF_STAGE=0 forces it on a small
    one): the Ybus blocks live in registers and the freed LDS keeps the factored DC matrix across the steps of a launch.  Same
    results as the LDS variant (GRIDPF_YREG=0), multi-step and single-step, and as the oracle."""
    name, B, kw = "l2rpn_wcci_2022_dev", 10, dict(rebalance=1.02)
    m, ch, e_lds, tab, off, scale = _setup(load_model, load_npz, name, B, env={"GRIDPF_STAGE": "0", "GRIDPF_YREG": "0"})
    _, _, e_reg, _, _, _ = _setup(load_model, load_npz, name, B, env={"GRIDPF_STAGE": "0"})
    _, _, e_one, _, _, _ = _setup(load_model, load_npz, name, B, env={"GRIDPF_STAGE": "0"})
    pl, pr_ = e_lds.plan(), e_reg.plan()
    print("plan LDS variant:", pl, "\nplan register variant:", pr_)
    if os.environ.get("GRIDPF_DETERMINISTIC") == "1" or os.environ.get("GRIDPF_WPI") == "1":
        pytest.skip("two wavefronts per instance are switched off by the environment (GRIDPF_DETERMINISTIC / GRIDPF_WPI): no register variant")
    assert pl["wavefronts_per_instance"] == 2 and pr_["wavefronts_per_instance"] == 2, (pl, pr_)
    assert pl["staging_tier"] == 0 and pl["wavefronts_per_instance"] == 2 and not pl["ybus_in_registers"]
    assert pr_["staging_tier"] == 0 and pr_["ybus_in_registers"] == 1 and pr_["dc_factors_kept"] == 1 and pr_["lds_bytes"] < pl["lds_bytes"]
    n = 5
    e_lds.step(2, n_steps=n, **kw)
    e_reg.step(2, n_steps=n, **kw)
    for t in range(2, 2 + n):
        e_one.step(t, **kw)
    a, b, c1 = _snapshot(e_lds), _snapshot(e_reg), _snapshot(e_one)
    _same(b, a, "registers vs LDS")
    _same(b, c1, "registers: multi-step vs single steps")
    assert (b["status"][:, 0] == 0).all()
    # a topology change in the middle of a launch sequence rebuilds the register copy
    topo = np.tile(m.initial_topo_vect(), (B, 1))
    l_out = 3
    topo[:, m.line_or_pos_topo_vect[l_out]] = -1
    topo[:, m.line_ex_pos_topo_vect[l_out]] = -1
    for e in (e_lds, e_reg):
        e.set_topology(topo)
        e.step(2 + n, n_steps=3, **kw)
    _same(_snapshot(e_reg), _snapshot(e_lds), "registers vs LDS after a line outage")
    T = tab.shape[0]
    r = e_reg.results()
    t = 2 + n + 2
    for k in (0, B - 1):
        row = (t + off[k]) % T
        s = LaneState.from_model(m)
        s.topo = topo[k].copy()
        lp = ch["load_p"][row] * scale[k, :m.n_load]
        lq = ch["load_q"][row] * scale[k, m.n_load:]
        pp = ch["prod_p"][row].copy()
        ns = ~m.gen_slack
        pp[ns] = pp[ns] * np.float32(1.02 * lp.astype(np.float64).sum() / pp[ns].astype(np.float64).sum())
        s.load_p, s.load_q, s.gen_p = lp.astype(np.float64), lq.astype(np.float64), pp.astype(np.float64)
        s.gen_vm = (ch["prod_v"][row] / m.sub_vn_kv[m.gen_sub].astype(np.float32)).astype(np.float64)
        _compare(m, r, k, solve(m, s))
    for e in (e_lds, e_reg, e_one):
        e.close()


def test_multi_step_last_step_matches_oracle(load_model, load_npz):
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 32)
    T = tab.shape[0]
    eng.step(5, n_steps=4, rebalance=1.02)
    r = eng.results()
    t = 8
    for k in range(32):
        row = (t + off[k]) % T
        s = LaneState.from_model(m)
        lp = ch["load_p"][row] * scale[k, :m.n_load]
        lq = ch["load_q"][row] * scale[k, m.n_load:]
        pp = ch["prod_p"][row].copy()
        ns = ~m.gen_slack
        pp[ns] = pp[ns] * np.float32(1.02 * lp.astype(np.float64).sum() / pp[ns].astype(np.float64).sum())
        s.load_p, s.load_q, s.gen_p = lp.astype(np.float64), lq.astype(np.float64), pp.astype(np.float64)
        s.gen_vm = (ch["prod_v"][row] / m.sub_vn_kv[m.gen_sub].astype(np.float32)).astype(np.float64)
        _compare(m, r, k, solve(m, s))
    eng.close()


@pytest.mark.parametrize("name,B", [("l2rpn_case14_sandbox", 24), ("l2rpn_neurips_2020_track1", 9)])
def test_multi_step_with_cascade_equals_single_steps(name, B, load_model, load_npz):
    """Tight thermal limits: lines trip at different steps in different lanes (soft overflows counted over steps, hard
    overflows at once), some lanes end islanded.  The topology-derived state kept on chip must be dropped exactly then."""
    m, ch, e1, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e2, _, _, _ = _setup(load_model, load_npz, name, B)
    e1.step(0, rebalance=1.02)
    a0 = e1.results().a_or.copy()
    lim = (np.median(a0, axis=0) * 1.02).astype(np.float32) + 1.0          # about half of the lanes overflow softly on every line
    hot = np.argsort(-np.median(a0, axis=0))[:2]
    lim[hot] = np.median(a0, axis=0)[hot] * 0.45                           # hard overflows
    kw = dict(rebalance=1.02, cascade=True, hard_overflow=2.0, soft_overflow=1.0, nb_ts_allowed=2)
    n = 6
    for e in (e1, e2):
        e.set_thermal_limits(lim)
        e.reset()
    e2.set_trajectory(n)
    snaps = []
    for t in range(n):
        e1.step(t, **kw)
        snaps.append(_snapshot(e1))
    e2.step(0, n_steps=n, **kw)
    _same(_snapshot(e2), snaps[-1], name)
    _, st = e2.trajectory(n)
    for k in range(n):
        assert np.array_equal(st[k], snaps[k]["status"][:, 0]), k
    tripped = (~snaps[-1]["ls"]).any(axis=1)
    assert tripped.any() and not tripped.all() or (snaps[-1]["status"][:, 0] != 0).any()
    t1, _ = e1.get_topology()
    t2, _ = e2.get_topology()
    assert np.array_equal(t1, t2)
    e1.close()
    e2.close()


def test_failing_lanes_done_flags_and_auto_reset(load_model, load_npz):
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 16)
    # lanes 3 and 10: loads x 6 -> the Newton iteration cannot converge (game over for these lanes)
    scale2 = scale.copy()
    scale2[[3, 10]] *= 6.0
    eng.set_lane_chronics(lane_offset=off, lane_scale=scale2)
    # every lane starts an episode with line 2 out of service (the topology an auto-reset must come back to)
    topo = np.tile(m.initial_topo_vect(), (16, 1))
    topo[:, m.line_or_pos_topo_vect[2]] = -1
    topo[:, m.line_ex_pos_topo_vect[2]] = -1
    eng.set_topology(topo)
    eng.set_trajectory(5)
    eng.step(0, n_steps=5, rebalance=1.02, auto_reset=True)
    r = eng.results()
    done, steps, resets = eng.episode()
    bad = np.zeros(16, bool)
    bad[[3, 10]] = True
    assert np.array_equal(done, bad) and np.array_equal(r.converged, ~bad)
    assert (steps[~bad] == 5).all() and (steps[bad] == 0).all() and (resets[bad] == 5).all() and (resets[~bad] == 0).all()
    assert np.isnan(r.out[bad]).all() and (r.topo_vect[bad] == -1).all() and not np.isnan(r.out[~bad]).any()
    _, st = eng.trajectory(5)
    assert (st[:, bad] != 0).all() and (st[:, ~bad] == 0).all()
    t_dev, _ = eng.get_topology()
    assert np.array_equal(t_dev, topo)                                  # restored for the failed lanes, untouched for the others
    # the healthy lanes are not disturbed by their failing neighbours (same wavefront): compare with a clean engine
    _, _, ref, _, _, _ = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 16)
    ref.set_lane_chronics(lane_offset=off, lane_scale=scale)
    ref.set_topology(topo)
    ref.step(0, n_steps=5, rebalance=1.02)
    rr = ref.results()
    assert np.allclose(r.out[~bad], rr.out[~bad], rtol=2e-6, atol=2e-5)
    # once the overload is gone the reset lanes converge again and survive
    eng.set_lane_chronics(lane_offset=off, lane_scale=scale)
    eng.step(5, n_steps=3, rebalance=1.02, auto_reset=True)
    done, steps, resets = eng.episode()
    assert not done.any() and (steps[bad] == 3).all() and (steps[~bad] == 8).all()
    eng.close()
    ref.close()


def test_redispatch_delta_matches_oracle(load_model, load_npz):
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_wcci_2022_dev", 6)
    rng = np.random.default_rng(3)
    disp = np.nonzero(~m.gen_slack)[0]
    delta = np.zeros((6, m.n_gen), np.float32)
    for k in range(6):
        a, b = rng.choice(disp, 2, replace=False)
        delta[k, a], delta[k, b] = 1.0, -1.0
    eng.set_lane_redispatch(delta)
    eng.step(2, n_steps=2)
    r = eng.results()
    T = tab.shape[0]
    for k in range(6):
        row = (3 + off[k]) % T
        s = LaneState.from_model(m)
        s.load_p = (ch["load_p"][row] * scale[k, :m.n_load]).astype(np.float64)
        s.load_q = (ch["load_q"][row] * scale[k, m.n_load:]).astype(np.float64)
        s.gen_p = (ch["prod_p"][row] + delta[k]).astype(np.float64)
        s.gen_vm = (ch["prod_v"][row] / m.sub_vn_kv[m.gen_sub].astype(np.float32)).astype(np.float64)
        _compare(m, r, k, solve(m, s))
    eng.set_lane_redispatch(None)
    eng.step(3)
    assert np.abs(eng.results().gen_p - r.gen_p)[:, ~m.gen_slack].max() > 0.5
    eng.close()


def test_zero_copy_device_views(load_model, load_npz):
    import torch
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 130)
    eng.step(1, n_steps=3, rebalance=1.02)
    v = eng.device_views()
    eng.sync()
    r = eng.results()
    rho, ovc, _ = eng.step_outputs()
    assert v["out"].is_cuda and v["out"].shape == (130, eng.n_out)
    assert np.array_equal(v["out"].cpu().numpy(), r.out)
    assert np.array_equal(v["rho"].cpu().numpy(), rho) and np.array_equal(v["overflow_count"].cpu().numpy(), ovc)
    assert np.array_equal(v["status"].cpu().numpy(), r.status) and np.array_equal(v["topo_vect"].cpu().numpy(), r.topo_vect)
    assert np.array_equal(v["line_status"].cpu().numpy().astype(bool), r.line_status)
    assert np.array_equal(v["inj"].cpu().numpy(), eng.get_injections())
    # the views alias the engine's memory: the next launch shows through without any copy
    before = v["rho"].clone()
    eng.step(40, rebalance=1.02)
    with torch.cuda.stream(v["stream"]):
        pass
    eng.sync()
    assert not torch.equal(before, v["rho"])
    assert np.array_equal(v["rho"].cpu().numpy(), eng.step_outputs()[0])
    # a consumer that never leaves the device: worst loading per lane
    worst = v["rho"].max(dim=1).values
    assert torch.allclose(worst.cpu(), torch.from_numpy(eng.step_outputs()[0].max(axis=1)))
    eng.close()


@pytest.mark.parametrize("mode", ["GRIDPF_NO_CLASSES", "GRIDPF_NO_PARTITION", "GRIDPF_DCF0"])
def test_split_batches_in_every_kernel_mode(mode, load_model, load_npz):
    """A batch with split substations stepped through the fallback kernel modes (NB = n_busbar blocks instead of topology
    classes, one launch instead of a partition, DC factors not kept) must agree with the default mode."""
    env = {"GRIDPF_DCF": "0"} if mode == "GRIDPF_DCF0" else {mode: "1"}
    name, B = "l2rpn_case14_sandbox", 40
    m, ch, e1, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e2, _, _, _ = _setup(load_model, load_npz, name, B, env=env)
    topo = np.tile(m.initial_topo_vect(), (B, 1))
    sub = int(np.argmax(m.sub_info))
    start = int(np.concatenate(([0], np.cumsum(m.sub_info)))[sub])
    pos = np.arange(start, start + m.sub_info[sub])
    split = np.random.default_rng(1).random(B) < (1.0 if mode == "GRIDPF_NO_PARTITION" else 0.4)
    for q in pos[::2]:
        topo[split, q] = 2
    for e in (e1, e2):
        e.set_topology(topo)
    n = 1 if mode == "GRIDPF_NO_CLASSES" else 3           # two launches per step (split / unsplit lanes): single steps only
    for t in range(3):
        e1.step(t, n_steps=n, rebalance=1.02)
        e2.step(t, n_steps=n, rebalance=1.02)
        _same(_snapshot(e2), _snapshot(e1), (mode, t))
    assert (e1.results().status[:, 0] == 0).all()
    if mode == "GRIDPF_NO_CLASSES":
        from grid2op_amd.engine import GridPFError
        with pytest.raises(GridPFError):
            e2.step(0, n_steps=2)
    e1.close()
    e2.close()


@pytest.mark.parametrize("name,B", [("l2rpn_case14_sandbox", 26), ("rte_case5_example", 12), ("educ_case14_storage", 8),
                                    ("l2rpn_neurips_2020_track1", 13), ("l2rpn_wcci_2022_dev", 7)])
def test_static_dc_inverse_equals_the_dc_solve(name, B, load_model, load_npz):
    """Lanes in the reference topology get their DC initialisation from the static inverse of B' (one matrix-vector phase; round 4:
    on every grid size -- the table of the larger grids is read from global memory); lanes with exactly ONE line out -- here every
    third lane, so that wavefronts mix both kinds -- from the same table + a Sherman-Morrison rank-1 correction (single-wavefront
    kernels; the two-wavefront kernels of the 118-substation grids factorise those lanes).  GRIDPF_NO_DCINV=1 (always assemble and
    factorise the DC system) must give the same power flows, iteration counts included."""
    m, ch, e1, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e2, _, _, _ = _setup(load_model, load_npz, name, B, env={"GRIDPF_NO_DCINV": "1"})
    topo = np.tile(m.initial_topo_vect(), (B, 1))
    for k in range(0, B, 3):
        l = (k // 3) % m.n_line
        topo[k, m.line_or_pos_topo_vect[l]] = -1
        topo[k, m.line_ex_pos_topo_vect[l]] = -1
    for e in (e1, e2):
        e.set_topology(topo)
    for t, n in ((0, 1), (1, 4), (5, 1), (6, 3)):
        e1.step(t, n_steps=n, rebalance=1.02)
        e2.step(t, n_steps=n, rebalance=1.02)
        _same(_snapshot(e1), _snapshot(e2), (name, t))
    st = e1.results().status
    assert (st[np.arange(B) % 3 != 0, 0] == 0).all()                # (some single-line outages may island a bus: those lanes may fail in both)
    e1.runpf(0, B, is_dc=True)
    e2.runpf(0, B, is_dc=True)
    a, b = e1.results(), e2.results()
    assert np.array_equal(a.status[:, 0], b.status[:, 0])
    assert np.allclose(a.out, b.out, rtol=2e-6, atol=2e-5, equal_nan=True)
    e1.close()
    e2.close()


@pytest.mark.parametrize("name", ["l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_case14_sandbox_trips"])
def test_device_rollout_reproduces_the_reference_environment(name, load_model, load_npz):
    """A DoNothing rollout of the UNMODIFIED reference Environment (default parameters: overflow disconnections on; all the
    scenarios of its chronics folder, scheduled maintenance included) recorded by tests/golden/make_rollout_fixtures.py, against ONE
    multi-step launch of the engine with one lane per scenario: rho of every step, final line status and overflow counters, and -- round 5 --
    the LINE COOLDOWNS of every step (obs.time_before_cooldown_line: `_trips` has two thermal limits below the initial flows, so that the
    protections trip lines 3, 12 and then 4 and their reconnection cooldowns run 10, 9, ...; the 36-substation rollout has a 96-step
    scheduled maintenance that holds the counter at its remaining duration)."""
    from grid2op_amd.engine import PowerFlowEngine
    from grid2op_amd.chronics import chronics_table
    m = load_model(name.replace("_trips", ""))
    fx = load_npz(f"rollout_{name}.npz")
    n_scen, n_steps = fx["rho"].shape[:2]
    reps = 5                                              # every scenario on several lanes (different wavefront positions)
    B = n_scen * reps
    eng = PowerFlowEngine(m, n_lanes=B, device=0)
    ch = {k[len("chron_"):]: fx[k] for k in fx if k.startswith("chron_")}
    eng.upload_chronics(chronics_table(ch))
    if "maintenance" in ch:
        eng.upload_maintenance(ch["maintenance"])
        # the fixture is a 132-row WINDOW of the chronics: the maintenance that starts at row 108 lasts 96 steps, of which the window shows 24
        eng.upload_outage_durations(ch["maintenance_duration"])
    eng.set_lane_chronics(lane_table=np.tile(np.arange(n_scen), reps), lane_offset=np.ones(B, np.int32))   # env step k reads row k
    eng.set_thermal_limits(fx["thermal_limit"])
    eng.set_trajectory(n_steps)
    kw = dict(cascade=True, hard_overflow=float(fx["hard_overflow"]), nb_ts_allowed=int(fx["nb_ts_allowed"]), nb_ts_reco=int(fx["nb_ts_reco"]))
    eng.step(0, n_steps=n_steps, **kw)
    rho, st = eng.trajectory(n_steps)
    cool = eng.trajectory_cooldown(n_steps)
    r = eng.results()
    _, ovc, _ = eng.step_outputs()
    assert (fx["done_at"] < 0).all() and (st == 0).all()
    for lane in range(B):
        k = lane % n_scen
        ok = ~np.isnan(fx["rho"][k])                      # (a tripped line has rho 0 in the recording and on the device)
        assert np.allclose(rho[:, lane][ok], fx["rho"][k][ok], rtol=2e-5, atol=2e-6), (lane, np.abs(rho[:, lane] - fx["rho"][k]).max())
        assert np.array_equal(r.line_status[lane], fx["line_status"][k, -1]), lane
        assert np.array_equal(ovc[lane], fx["timestep_overflow"][k, -1]), lane
        assert np.array_equal(cool[:, lane], fx["time_before_cooldown_line"][k]), (lane, np.argwhere(cool[:, lane] != fx["time_before_cooldown_line"][k])[:4])
    assert np.array_equal(eng.cooldown(), cool[-1])
    if name.endswith("_trips"):
        assert cool.max() == int(fx["nb_ts_reco"]) and (~r.line_status).sum() >= 2 * reps
    if name == "l2rpn_neurips_2020_track1":
        assert cool.max() == 96                           # the scheduled maintenance of the first scenario
        eng.upload_outage_durations(None)                 # derived from the window alone: the outage looks 24 steps long
        eng.reset()
        eng.set_lane_chronics(lane_table=np.tile(np.arange(n_scen), reps), lane_offset=np.ones(B, np.int32))
        eng.step(0, n_steps=n_steps, **kw)
        c2 = eng.trajectory_cooldown(n_steps)
        assert c2.max() == 24 and np.array_equal(c2 > 0, cool > 0)
        eng.upload_outage_durations(ch["maintenance_duration"])
    if "maintenance" in ch:
        assert (~r.line_status).any()                     # the scheduled outage really happened
    # the same rollout one launch per step, and in launches of 7
    for spl in (1, 7):
        eng.reset()
        eng.set_lane_chronics(lane_table=np.tile(np.arange(n_scen), reps), lane_offset=np.ones(B, np.int32))
        t = 0
        while t < n_steps:
            k = min(spl, n_steps - t)
            eng.step(t, n_steps=k, **kw)
            t += k
        r2 = eng.results()
        assert np.array_equal(r2.line_status, r.line_status) and np.allclose(r2.out, r.out, rtol=2e-6, atol=2e-5, equal_nan=True)
        assert np.array_equal(eng.cooldown(), cool[-1])
    eng.close()


def test_sharded_engine_on_real_engines(load_model, load_npz):
    """`ShardedEngine` with two REAL engines (both on device 0: the routing, not the device count, is what is checked): a
    stepped and solved batch equals the same batch on one engine, lane for lane."""
    from grid2op_amd.sharding import ShardedEngine
    m, ch, one, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 70)
    se = ShardedEngine(m, 70, devices=[0, 0])
    assert [e.n_lanes for e in se.engines] == [35, 35]
    se.upload_chronics(tab)
    se.set_lane_chronics(lane_offset=off, lane_scale=scale)
    se.set_thermal_limits(ch["thermal_limits"])
    topo = np.tile(m.initial_topo_vect(), (6, 1))
    topo[:, m.line_or_pos_topo_vect[4]] = -1
    topo[:, m.line_ex_pos_topo_vect[4]] = -1
    for e in (one, se):
        e.set_topology(topo, lane0=32)                     # lanes 32..37 straddle the shard boundary
        e.step(2, n_steps=5, rebalance=1.02)
    a, b = one.results(), se.results()
    assert np.array_equal(a.status, b.status) and np.array_equal(a.topo_vect, b.topo_vect)
    assert np.array_equal(a.out, b.out, equal_nan=True)
    assert np.array_equal(one.step_outputs()[0], se.step_outputs()[0])
    inj = one.get_injections(30, 10)
    inj[:, :m.n_gen] *= 1.01
    for e in (one, se):
        e.set_injections(inj, lane0=30)
        e.runpf(30, 10)
    assert np.array_equal(one.results(28, 14).out, se.results(28, 14).out, equal_nan=True)
    d1, s1, r1 = one.episode()
    d2, s2, r2 = se.episode()
    assert np.array_equal(s1, s2) and (s2 == 5).all()
    # the rest of the batched API is forwarded too: observation trajectory, maintenance, redispatch delta, fan-out, copies
    maint = np.zeros((1, tab.shape[0], m.n_line), np.uint8)
    maint[0, 9:12, 7] = 1
    delta = np.zeros((70, m.n_gen), np.float32)
    delta[:, 1], delta[:, 2] = 0.5, -0.5
    for e in (one, se):
        e.reset()
        e.upload_maintenance(maint)
        e.set_lane_redispatch(delta)
        e.set_trajectory(6, e.TRAJ_OBS)
        e.step(7, n_steps=6, rebalance=1.02)
    oa, ob = one.trajectory_obs(6), se.trajectory_obs(6)
    ra, rb = one.trajectory(6), se.trajectory(6)
    assert np.array_equal(ra[0], rb[0], equal_nan=True) and np.array_equal(ra[1], rb[1])
    for k in range(6):
        assert np.array_equal(oa[k].out, ob[k].out, equal_nan=True) and np.array_equal(oa[k].line_status, ob[k].line_status)
    assert not ob[5].line_status[0, 7] and ob[0].line_status[0, 7] and ob[5].line_status[:, 7].any()   # lane 0 crosses the outage rows
    part = se.trajectory_obs(2, step0=1, lane0=33, n=4)                              # lanes straddling the shard boundary
    assert np.array_equal(part[1].out, oa[2].out[33:37], equal_nan=True)
    for e in (one, se):
        e.set_trajectory(0)
        e.upload_maintenance(None)
        e.fanout_n1(36, 40, [0, 3, -1])
        e.copy_lanes(2, 60, 3)                                                        # crosses devices on the sharded engine
        e.runpf()
    assert np.array_equal(one.results().out, se.results().out, equal_nan=True)
    assert np.array_equal(one.get_topology()[0], se.get_topology()[0])
    with pytest.raises(ValueError):
        se.fanout_n1(2, 50, [1])                                                      # source and destinations on different devices
    assert len(se.device_views()) == 2 and len(se.plan()) == 2
    # run-time specialised kernels on every shard (self-test on the first, code objects shared through the cache): same results again
    one_before = one.specialization()                 # (already on when the suite itself runs with GRIDPF_JIT=1)
    infos = se.specialize(True)
    assert len(infos) == 2 and all(i["enabled"] for i in infos)
    for e in (one, se):
        e.step(13, n_steps=4, rebalance=1.02)
    assert np.array_equal(one.results().out, se.results().out, equal_nan=True)
    assert all(b["launches"] == a["launches"] + 1 and b["failed"] == 0 for a, b in zip(infos, se.specialization()))
    assert one.specialization()["launches"] == one_before["launches"] + (1 if one_before["enabled"] else 0)
    one.close()
    se.close()


@pytest.mark.parametrize("name", ["l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1"])
def test_batched_simulate_of_candidate_actions(name, load_model):
    """`simulate_candidates`: a list of grid2op-style candidate actions (line switching, bus splits by element, raw set_bus,
    change_bus, do-nothing) fanned out from one lane and solved in one launch, each against the oracle on the same topology."""
    from grid2op_amd.engine import PowerFlowEngine
    m = load_model(name)
    rng = np.random.default_rng(5)
    sub = int(np.argmax(m.sub_info))
    start = int(np.concatenate(([0], np.cumsum(m.sub_info)))[sub])
    pos = list(range(start, start + int(m.sub_info[sub])))
    lines_at = [int(l) for l in np.nonzero(m.line_or_sub == sub)[0]]
    acts = [{}]
    acts += [{"set_line_status": [(int(l), -1)]} for l in rng.choice(m.n_line, 6, replace=False)]
    acts += [{"set_bus": {p: 2 for p in pos[::2]}}, {"change_bus": pos[1::2]},
             {"lines_or_bus": [(l, 2) for l in lines_at[:2]], "set_line_status": [(int((lines_at[0] + 1) % m.n_line), -1)]},
             {"loads_bus": [(0, 2)], "gens_bus": [(0, 2)]}]
    eng = PowerFlowEngine(m, n_lanes=1 + len(acts), device=0)
    base = LaneState.from_model(m)
    base.load_p = base.load_p * 1.07
    inj, topo, sb = __import__("helpers").pack_states(m, [base])
    eng.set_injections(inj, lane0=0)
    eng.set_topology(topo, sb, lane0=0)
    n = eng.simulate_candidates(0, 1, acts)
    assert n == len(acts)
    r = eng.results(1, n)
    cand = eng.candidate_topologies(base.topo, acts)
    assert (cand[0] == base.topo).all() and (cand[1:] != base.topo).any(axis=1).all()
    n_conv = 0
    for k in range(n):
        s = base.copy()
        s.topo = cand[k].copy()
        o = solve(m, s)
        _compare(m, r, k, o)
        n_conv += int(o.converged)
    assert n_conv >= n - 2
    eng.close()


def test_api_error_paths_and_edge_cases(load_model, load_npz):
    """Bad arguments are refused with GPF_E_INVALID and leave the engine state untouched; odd shapes work: a 1-lane engine, more
    steps per launch than the chronics table has rows (the cursor wraps), a trajectory buffer shorter than the launch."""
    from grid2op_amd.engine import GridPFError, PowerFlowEngine
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 5)
    eng.step(0, rebalance=1.02)
    before = _snapshot(eng)
    t_before, sb_before = eng.get_topology()
    bad = np.tile(m.initial_topo_vect(), (5, 1))
    bad[3, 7] = 0                                            # local bus ids are -1 or 1..n_busbar
    with pytest.raises(GridPFError):
        eng.set_topology(bad)
    bad[3, 7] = 3
    with pytest.raises(GridPFError):
        eng.set_topology(bad)
    good = np.tile(m.initial_topo_vect(), (5, 1))
    with pytest.raises(GridPFError):
        eng.set_topology(good, np.full((5, m.n_shunt), 5, np.int32))
    t_after, sb_after = eng.get_topology()
    assert np.array_equal(t_before, t_after) and np.array_equal(sb_before, sb_after)      # nothing was half-applied
    with pytest.raises(GridPFError):
        eng.step(0, n_steps=0)
    with pytest.raises(GridPFError):
        eng.runpf(3, 10)                                      # range past the last lane
    with pytest.raises(GridPFError):
        eng.upload_maintenance(np.zeros((1, 7, m.n_line), np.uint8))          # shape must match the chronics tables
    with pytest.raises(GridPFError):
        eng.redispatch(np.zeros((5, m.n_gen)), np.zeros((5, m.n_gen)), np.zeros((5, m.n_gen)), np.zeros((5, m.n_gen)),
                       np.zeros((5, m.n_gen), bool), np.zeros(5))             # generator limits not set
    with pytest.raises(GridPFError):
        eng.trajectory(1)                                     # no trajectory buffer requested
    eng.step(0, rebalance=1.02)
    _same(_snapshot(eng), before, "state after the refused calls")
    # the chronics cursor wraps inside a launch: T + 3 steps from row 0 end on row 2
    T = tab.shape[0]
    eng.set_lane_chronics(lane_offset=np.zeros(5, np.int32), lane_scale=scale)
    eng.set_trajectory(4)                                     # shorter than the launch: refused (steps would be dropped silently)
    with pytest.raises(GridPFError):
        eng.step(0, n_steps=T + 3, rebalance=1.02)
    eng.set_trajectory(T + 3)
    with pytest.raises(GridPFError):
        eng.trajectory(1)                                     # nothing written yet by a launch into the new buffer
    eng.step(0, n_steps=T + 3, rebalance=1.02)
    a = _snapshot(eng)
    rho4, st4 = eng.trajectory(4)
    eng.step(5, n_steps=2, rebalance=1.02)
    with pytest.raises(GridPFError):
        eng.trajectory(4)                                     # only the 2 steps of the LAST launch are retrievable
    with pytest.raises(GridPFError):
        eng.trajectory_obs(1)                                 # no observation trajectory requested
    eng.set_trajectory(0)
    eng.step(2, rebalance=1.02)
    b = _snapshot(eng)
    assert np.allclose(a["out"], b["out"], rtol=2e-6, atol=2e-5) and np.array_equal(a["status"], b["status"])
    eng.step(3, rebalance=1.02)
    assert np.allclose(rho4[3], eng.step_outputs()[0], rtol=2e-6, atol=1e-6)
    eng.close()
    one = PowerFlowEngine(m, n_lanes=1, device=0)             # a single lane (what a HipBackend without copies uses)
    one.upload_chronics(tab)
    one.step(5, n_steps=2)
    one.runpf()
    assert one.results().converged.all()
    one.close()


def test_solve_lane_equals_the_four_call_sequence(load_model):
    """`gpf_solve_lane` (one call, pinned staging, one sync) against set_injections + set_topology + runpf + results on another
    lane of the same engine, for random states incl. splits, outages, diverging lanes and DC mode."""
    from grid2op_amd.engine import PowerFlowEngine
    from helpers import pack_states, random_states
    m = load_model("educ_case14_storage")
    eng = PowerFlowEngine(m, n_lanes=5, device=0)
    rng = np.random.default_rng(9)
    for s in [LaneState.from_model(m)] + random_states(m, 25, rng):
        inj, topo, sb = pack_states(m, [s])
        for dc in (False, True):
            a = eng.solve_lane(3, inj[0], topo[0], sb[0], is_dc=dc)
            eng.set_injections(inj, lane0=1)
            eng.set_topology(topo, sb, lane0=1)
            eng.runpf(1, 1, is_dc=dc)
            b = eng.results(1, 1)
            assert np.array_equal(a.status, b.status) and np.array_equal(a.topo_vect, b.topo_vect)
            assert np.array_equal(a.line_status, b.line_status) and np.array_equal(a.shunt_bus, b.shunt_bus)
            assert np.array_equal(a.out, b.out, equal_nan=True)
            assert np.array_equal(a.bus_vm, b.bus_vm, equal_nan=True) and np.array_equal(a.bus_va, b.bus_va, equal_nan=True)
            assert np.array_equal(eng.get_injections(3, 1), inj) and np.array_equal(eng.get_topology(3, 1)[0], topo)
    eng.close()


def _tight_limits(e1, others, kw0):
    """thermal limits that make about half of the lanes overflow softly on every line and two lines overflow hard"""
    e1.step(0, **kw0)
    a0 = e1.results().a_or.copy()
    lim = (np.median(a0, axis=0) * 1.02).astype(np.float32) + 1.0
    hot = np.argsort(-np.median(a0, axis=0))[:2]
    lim[hot] = np.median(a0, axis=0)[hot] * 0.45
    for e in [e1] + list(others):
        e.set_thermal_limits(lim)
        e.reset()


@pytest.mark.parametrize("name,B,kw,tight", [
    ("rte_case5_example", 37, dict(rebalance=1.02), False),                # 4 instances per wavefront, ragged tail
    ("l2rpn_case14_sandbox", 66, dict(rebalance=1.02), False),             # 2 instances per wavefront
    ("l2rpn_case14_sandbox", 24, dict(rebalance=1.02, cascade=True, hard_overflow=2.0, soft_overflow=1.0, nb_ts_allowed=2), True),
    ("l2rpn_case14_sandbox", 64, dict(rebalance=1.02, is_dc=True), False),
    ("l2rpn_neurips_2020_track1", 33, dict(rebalance=1.02), False),        # 1 instance per wavefront
    ("l2rpn_neurips_2020_track1", 9, dict(rebalance=1.02, cascade=True, hard_overflow=2.0, soft_overflow=1.0, nb_ts_allowed=2), True),
    ("l2rpn_wcci_2022_dev", 8, dict(rebalance=1.02), False),               # 2 wavefronts per instance, Ybus in registers
    ("educ_case14_storage", 16, dict(rebalance=1.0), False),
])
def test_observation_trajectory_holds_every_step(name, B, kw, tight, load_model, load_npz):
    """gpf_set_trajectory(.., GPF_TRAJ_OBS): EVERY step of a multi-step launch leaves its complete backend observation in HBM
    (what BaseEnv.step hands to the observation after each env.step, Environment/baseEnv.py:3562-3931) -- equal to what n
    single-step launches return one after the other (integers bit-exact), incl. steps in which lines trip; the lane's own rows
    still hold the last step."""
    m, ch, e1, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e2, _, _, _ = _setup(load_model, load_npz, name, B)
    if tight:
        _tight_limits(e1, [e2], dict(rebalance=1.02))
    n = 6
    e2.set_trajectory(n, e2.TRAJ_OBS)
    snaps = []
    for t in range(2, 2 + n):
        e1.step(t, **kw)
        snaps.append(_snapshot(e1))
    e2.step(2, n_steps=n, **kw)
    _same(_snapshot(e2), snaps[-1], name)
    obs = e2.trajectory_obs(n)
    rho, st = e2.trajectory(n)
    v = e2.device_views()
    e2.sync()
    for k in range(n):
        a = snaps[k]
        assert np.array_equal(st[k], a["status"][:, 0]), k
        assert np.array_equal(obs[k].topo_vect, a["topo"]) and np.array_equal(obs[k].line_status, a["ls"]), k
        assert np.array_equal(obs[k].shunt_bus, a["sb"]), k
        assert np.array_equal(np.isnan(obs[k].out), np.isnan(a["out"])), k
        assert np.allclose(obs[k].out, a["out"], rtol=2e-6, atol=2e-5, equal_nan=True), (k, np.nanmax(np.abs(obs[k].out - a["out"])))
        assert np.allclose(rho[k], a["rho"], rtol=2e-6, atol=1e-6, equal_nan=True), k
        assert np.array_equal(v["traj_out"][k].cpu().numpy(), obs[k].out, equal_nan=True)        # zero-copy view of the same rows
        assert np.array_equal(v["traj_topo_vect"][k].cpu().numpy(), obs[k].topo_vect)
        assert np.array_equal(v["traj_line_status"][k].cpu().numpy().astype(bool), obs[k].line_status)
    if tight:
        assert (~snaps[-1]["ls"]).any(), "the scenario is meant to trip lines"
        if name == "l2rpn_case14_sandbox":
            assert any(not np.array_equal(snaps[k]["ls"], snaps[k + 1]["ls"]) for k in range(n - 1))   # ... at different steps
    # a sub-range of steps / lanes
    part = e2.trajectory_obs(2, step0=3, lane0=1, n=3)
    assert np.array_equal(part[1].out, obs[4].out[1:4], equal_nan=True) and np.array_equal(part[0].topo_vect, obs[3].topo_vect[1:4])
    # without the buffer the same launch gives the same last step (the trajectory only adds stores)
    e2.set_trajectory(0)
    e2.reset(); e1.reset()
    e2.step(2, n_steps=n, **kw)
    e1.set_trajectory(n, e1.TRAJ_OBS)
    e1.step(2, n_steps=n, **kw)
    _same(_snapshot(e2), _snapshot(e1), name + " with / without trajectory")
    e1.close()
    e2.close()


def test_observation_trajectory_failing_lanes_and_auto_reset(load_model, load_npz):
    """A lane whose step fails leaves an all-NaN observation (topo_vect -1) for THAT step only; after the auto-reset its next
    steps are regular observations again."""
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 16)
    scale2 = scale.copy()
    scale2[[3, 10]] *= 6.0
    eng.set_lane_chronics(lane_offset=off, lane_scale=scale2)
    eng.set_trajectory(5, eng.TRAJ_OBS)
    eng.step(0, n_steps=5, rebalance=1.02, auto_reset=True)
    obs = eng.trajectory_obs(5)
    _, st = eng.trajectory(5)
    for k in range(5):
        bad = st[k] != 0
        assert bad[[3, 10]].all() and bad.sum() == 2
        assert np.isnan(obs[k].out[bad]).all() and (obs[k].topo_vect[bad] == -1).all() and not obs[k].line_status[bad].any()
        assert not np.isnan(obs[k].out[~bad]).any() and (obs[k].topo_vect[~bad] >= 1).all()
    eng.close()


def test_hazards_table_forces_lines_out_like_maintenance(load_model, load_npz):
    """gpf_upload_hazards (hazards.csv, Chronics/gridStateFromFile.py:478-490): the union of the hazards and maintenance tables is
    applied; each table can be replaced / removed independently."""
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 8)
    T = tab.shape[0]
    eng.set_lane_chronics(lane_offset=np.zeros(8, np.int32), lane_scale=scale)
    maint = np.zeros((1, T, m.n_line), np.uint8)
    haz = np.zeros((1, T, m.n_line), np.uint8)
    maint[0, 3:5, 4] = 1
    haz[0, 4:6, 9] = 1
    eng.upload_maintenance(maint)
    eng.upload_hazards(haz)
    eng.set_trajectory(8, eng.TRAJ_OBS)
    eng.step(0, n_steps=8, rebalance=1.02)
    obs = eng.trajectory_obs(8)
    for t in range(8):
        ls = obs[t].line_status
        assert (ls[:, 4] == (t < 3)).all(), t            # out from its maintenance on (nothing reconnects it for a DoNothing agent)
        assert (ls[:, 9] == (t < 4)).all(), t            # out from its hazard on
        assert ls[:, [l for l in range(m.n_line) if l not in (4, 9)]].all()
    eng.reset()
    eng.upload_maintenance(None)                         # only the hazards are left
    eng.step(0, n_steps=8, rebalance=1.02)
    ls = eng.results().line_status
    assert ls[:, 4].all() and not ls[:, 9].any()
    eng.reset()
    eng.upload_hazards(None)
    eng.step(0, n_steps=8, rebalance=1.02)
    assert eng.results().line_status.all()
    eng.close()


def test_pandapower_recorded_do_nothing_episodes_in_multi_step_launches(load_model, load_npz):
    """The reference's OWN recording of the headline workload (grid2op/data/rte_case5_example/_statistics: DoNothingAgent, default
    parameters = protections ON, PandaPowerBackend, 20 scenarios, 7 930 observations; tests/golden/make_statistics_fixtures.py),
    replayed with one lane per scenario in launches of 64 env steps (4 instances per wavefront on this grid): line status at every
    step, flows / voltages / generator results at the fixture's sub-sampled rows, protection counters at every launch boundary,
    and the step at which each episode ends (19 game overs after overloaded lines tripped) -- against pandapower's values."""
    from grid2op_amd.engine import PowerFlowEngine
    m = load_model("rte_case5_example")
    fx = load_npz("stats_case5.npz")
    start = fx["episode_start"]
    n_ep = len(start) - 1
    n_rows = np.diff(start).astype(int)
    T = int(n_rows.max())
    n_chron = 2 * m.n_load + 2 * m.n_gen
    tables = np.zeros((n_ep, T, n_chron), np.float32)
    for e in range(n_ep):
        a, b = int(fx["chronics_start"][e]), int(fx["chronics_start"][e + 1])
        rows = fx["chronics_rows"][a:b][:T]
        tables[e, :rows.shape[0]] = rows
        tables[e, rows.shape[0]:] = rows[-1]
    pos = {int(r): i for i, r in enumerate(fx["row_idx"])}
    lim = fx["thermal_limit"]
    eng = PowerFlowEngine(m, n_lanes=n_ep, device=0)
    eng.upload_chronics(tables)
    eng.set_lane_chronics(lane_table=np.arange(n_ep, dtype=np.int32))
    eng.set_thermal_limits(lim)
    spl = 64
    eng.set_trajectory(spl, eng.TRAJ_OBS)
    n_cmp = 0
    ended = np.full(n_ep, -1)

    def compare(r, e, s):
        nonlocal n_cmp
        g = int(start[e]) + s
        assert np.array_equal(r.line_status[e], fx["line_status_all"][g]), (e, s)
        if g not in pos:
            return
        i = pos[g]
        assert np.array_equal(r.topo_vect[e], fx["topo_vect"][i]), (e, s)
        for f, tol in [("p_or", 1e-4), ("q_or", 3e-4), ("p_ex", 1e-4), ("q_ex", 3e-4), ("v_or", 1e-4), ("v_ex", 1e-4)]:
            assert np.abs(getattr(r, f)[e] - fx[f][i]).max() < tol, (e, s, f, np.abs(getattr(r, f)[e] - fx[f][i]).max())
        on = fx["a_or"][i] > 1e-6
        assert np.abs(r.a_or[e][on] / fx["a_or"][i][on] - 1).max() < 1e-5, (e, s)
        assert np.abs(r.gen_p[e] - fx["prod_p"][i]).max() < 1e-4 and np.abs(r.gen_q[e] - fx["prod_q"][i]).max() < 3e-4, (e, s)
        assert np.abs(r.load_p[e] - fx["load_p"][i]).max() < 1e-5 and np.abs(r.load_v[e] - fx["load_v"][i]).max() < 1e-4, (e, s)
        n_cmp += 1

    # row 0 = the reset observation: a power flow without the protections, counters untouched
    eng.step(0, n_steps=1, cascade=False)
    r0 = eng.trajectory_obs(1)[0]
    assert r0.converged.all()
    for e in range(n_ep):
        compare(r0, e, 0)
    eng.set_overflow_count(np.zeros((n_ep, m.n_line), np.int32))
    t = 1
    while t < T:
        n = min(spl, T - t)
        eng.step(t, n_steps=n, cascade=True)
        obs = eng.trajectory_obs(n)
        rho, oc, _ = eng.step_outputs()
        for j in range(n):
            s = t + j
            for e in range(n_ep):
                if ended[e] >= 0 or s >= n_rows[e]:
                    continue
                if s == n_rows[e] - 1 and n_rows[e] < T:
                    # the runner's final row repeats the last valid observation: this is the step the episode ends at
                    assert not obs[j].converged[e], (e, s)
                    ended[e] = s
                    continue
                assert obs[j].converged[e], (e, s, obs[j].status[e])
                compare(obs[j], e, s)
        s_last = t + n - 1
        for e in range(n_ep):
            if ended[e] < 0 and s_last < n_rows[e]:
                g = int(start[e]) + s_last
                assert np.array_equal(oc[e], fx["timestep_overflow_all"][g]), (e, s_last, oc[e], fx["timestep_overflow_all"][g])
                on = fx["line_status_all"][g]
                if g in pos:
                    assert np.abs(rho[e][on] - fx["rho"][pos[g]][on]).max() < 1e-5, (e, s_last)
        t += n
    assert n_cmp >= 2000
    assert (ended >= 0).sum() == 19 and all(ended[e] == n_rows[e] - 1 for e in range(n_ep) if n_rows[e] < T)
    eng.close()


def test_line_cooldowns_decrement_copy_reset_and_opt_out(load_model, load_npz):
    """The line cooldowns outside the recorded rollouts: an environment restored from an observation hands its counters over
    (`set_cooldown`), every tracked step takes one off (never below 0), `copy_lanes` / `reset` treat them like the other per-lane counters,
    an untracked step (nb_ts_reco < 0: the default when nothing can take a line out) leaves them alone."""
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 9)
    c0 = np.zeros((9, m.n_line), np.int32)
    c0[0, :5] = (5, 1, 0, 3, 12)
    c0[1, 7] = 2
    eng.set_cooldown(c0)
    assert np.array_equal(eng.cooldown(), c0)
    eng.set_trajectory(4)
    eng.step(0, n_steps=3, rebalance=1.02, nb_ts_reco=10)
    want = np.maximum(c0 - 3, 0)
    assert np.array_equal(eng.cooldown(), want)
    tr = eng.trajectory_cooldown(3)
    for k in range(3):
        assert np.array_equal(tr[k], np.maximum(c0 - (k + 1), 0)), k
    eng.step(3, n_steps=2, rebalance=1.02)                       # default: not tracked (no cascade, no outage tables)
    assert np.array_equal(eng.cooldown(), want)
    eng.copy_lanes(0, 5, 2)
    assert np.array_equal(eng.cooldown(5, 2), want[:2])
    eng.reset(0, 1)
    assert (eng.cooldown(0, 1) == 0).all() and np.array_equal(eng.cooldown(5, 1), want[:1])
    with pytest.raises(Exception):
        eng.set_cooldown(-np.ones((1, m.n_line), np.int32))
    eng.close()

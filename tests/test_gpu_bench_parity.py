"""GPU parity at the sizes and in the variants bench.py times (BASELINE.json configs[1..4]): the bench-size launches pick other
kernel variants (instances per wavefront, staging tier, wavefronts per instance, Ybus in registers) than the small batches of
the other test files, so every configuration of the bench line is solved here as bench.py sets it up and a sample of >= 64
lanes is re-solved by the C oracle (oracle/pf_oracle.c): convergence status and Newton iteration count bit-exact, integer vectors
bit-exact, float32 API outputs within 2e-4 + 5e-6 |x| (MW, MVAr, kV, A, deg).  The north star's bar -- max line-flow error < 1e-4 pu --
is stated on each grid's OWN base: 1e-2 MW on the 100 MVA grids (case14), 1e-4 MW on l2rpn_neurips_2020_track1 / l2rpn_wcci_2022_dev /
l2rpn_idf_2023 whose sn_mva is 1 -- there a float32 output cannot resolve it (1 ulp of a 500 MW flow is 3e-5 MW), so `check_lanes`
also recomputes the line flows in float64 from the engine's pre-cast bus voltages (bus_vm / bus_va) and holds THEM to 1e-4 pu
(`max_flow_err_pu_f64`, oracle/spot_check.py)."""
import numpy as np
import pytest

from oracle.spot_check import check_lanes, check_step

pytestmark = pytest.mark.gpu


def _bench_engine(load_model, load_npz, name, n_envs, fan=1):
    from grid2op_amd.engine import PowerFlowEngine
    from grid2op_amd.sharding import synthetic_lane_inputs
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    B = n_envs * fan
    eng = PowerFlowEngine(m, n_lanes=B, device=0)
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"])
    eng.upload_chronics(tab)
    off, sc = synthetic_lane_inputs(m.n_load, tab.shape[0], np.arange(n_envs))
    if fan > 1:
        off, sc = np.repeat(off, fan), np.repeat(sc, fan, axis=0)
    eng.set_lane_chronics(lane_offset=off, lane_scale=sc)
    if "thermal_limits" in ch:
        eng.set_thermal_limits(ch["thermal_limits"])
    return m, eng, tab, off, sc


def _headline_launch_lengths():
    """The launch lengths bench.py's headline really uses: its default, and what the driver's command (--steps 20) turns it into."""
    import bench
    return sorted({bench.launch_length(bench.DRIVER_STEPS), bench.launch_length(2000)})


@pytest.mark.parametrize("n", _headline_launch_lengths())
def test_headline_launch_4096_lanes_every_step_vs_oracle(load_model, load_npz, n):
    """configs[1] as bench.py's headline runs it: l2rpn_case14_sandbox, 4 096 lanes, ``n`` env steps per launch (bench.py's own launch
    lengths: 16 by default, 20 under the driver's command) with the observation trajectory on.  64 sampled lanes x 4 steps of the launch are
    recomputed by the C oracle FROM THE CHRONICS TABLE (this covers the device-side chronics gather, jitter and rebalancing too), and all
    4 096 x n observations obey KCL."""
    m, eng, tab, off, sc = _bench_engine(load_model, load_npz, "l2rpn_case14_sandbox", 4096)
    assert eng.plan()["instances_per_wavefront"] == 2
    t0 = 32
    eng.set_trajectory(n, eng.TRAJ_OBS)
    eng.step(0, n_steps=n, rebalance=1.02)            # a first launch, so that the checked one starts from a used state
    eng.step(t0, n_steps=n, rebalance=1.02)
    obs = eng.trajectory_obs(n)
    _, st = eng.trajectory(n)
    lanes = np.sort(np.random.default_rng(5).choice(4096, 64, replace=False))
    for k in (0, 5, 10, n - 1):
        res = check_step(m, tab, off, sc, 1.02, t0 + k, lanes, obs[k].out[lanes], st[k][lanes])
        assert res["ok"] and res["n_converged"] == 64, (k, res)
        r = obs[k]
        assert (st[k] == 0).all()
        p_bus = np.zeros((4096, m.n_sub))
        for sub, val in [(m.line_or_sub, r.p_or), (m.line_ex_sub, r.p_ex), (m.load_sub, r.load_p), (m.gen_sub, -r.gen_p),
                         (m.shunt_sub, r.shunt_p)]:
            np.add.at(p_bus, (slice(None), sub), val.astype(np.float64))
        assert np.abs(p_bus).max() < 1e-2, k
        assert (r.topo_vect == m.initial_topo_vect()[None, :]).all() and r.line_status.all()
    # the lane's own rows = the last step, and n_iter of the sampled lanes equals the oracle's
    last = eng.results(with_bus=False)
    assert np.array_equal(last.out, obs[n - 1].out)
    res = check_step(m, tab, off, sc, 1.02, t0 + n - 1, lanes, last.out[lanes], last.status[lanes])
    assert res["ok"] and res.get("n_iter_mismatch", 1) == 0, res
    eng.close()


def test_n1_fanout_1024_envs_x_60_lanes_vs_oracle(load_model, load_npz):
    """configs[2] as bench.py runs it: l2rpn_neurips_2020_track1, 1 024 envs x (1 intact + 59 single-line outages) = 61 440
    lanes stepped together.  All 60 lanes of two envs and 80 random lanes are re-solved by the C oracle from the inputs the lanes
    hold on the device, incl. the contingencies that island the grid or diverge."""
    name, n_envs = "l2rpn_neurips_2020_track1", 1024
    m0 = load_model(name)
    fan = 1 + m0.n_line
    m, eng, tab, off, sc = _bench_engine(load_model, load_npz, name, n_envs, fan)
    B = n_envs * fan
    topo = np.tile(m.initial_topo_vect(), (B, 1))
    for c in range(1, fan):
        topo[c::fan, m.line_or_pos_topo_vect[c - 1]] = -1
        topo[c::fan, m.line_ex_pos_topo_vect[c - 1]] = -1
    eng.set_topology(topo)
    eng.set_trajectory(4, eng.TRAJ_OBS)               # as bench.py times it: every env step leaves its observation in HBM
    eng.step(3, n_steps=4, rebalance=1.02)
    r = eng.results(with_bus=False)
    obs1 = eng.trajectory_obs(1, step0=1)[0]          # an earlier step of the launch: same verdicts, its own flows
    assert np.array_equal(obs1.converged, r.converged) and np.array_equal(obs1.topo_vect, r.topo_vect)
    assert not np.array_equal(obs1.out[:fan], r.out[:fan])
    conv = r.converged.reshape(n_envs, fan)
    assert conv[:, 0].all()                                       # the intact grid converges in every env
    lanes = np.unique(np.concatenate([np.arange(fan), 517 * fan + np.arange(fan), np.random.default_rng(2).choice(B, 80, replace=False)]))
    res = check_lanes(eng, lanes, results=r)
    assert res["ok"], res
    assert res["n_converged"] < res["n"], "the sample is meant to contain islanding / diverging contingencies"
    # the same contingency gives the same verdict in every env (same topology, similar injections) for the islanding ones
    isl = r.status.reshape(n_envs, fan, 4)[:, :, 0] == 2
    assert (isl == isl[:1]).all()
    eng.close()


def test_n1_fanout_118_substations_1024_envs_x_187_lanes_vs_oracle(load_model, load_npz):
    """configs[2] on a REAL 118-substation grid (BASELINE.json says "IEEE 118-bus"; the bundled l2rpn_neurips_2020_track1 is a
    36-substation sub-area, grid2op/tests/test_attached_envs.py:31-35): l2rpn_wcci_2022_dev, 1 024 envs x (1 intact + 186 single-line
    outages) = 191 488 lanes stepped together as bench.py's `n1_fanout_118` does (observation per step; 4 steps per launch here, 16 in the bench).  All 187
    lanes of two envs and 80 random lanes are re-solved by the C oracle from the inputs the lanes hold on the device, incl. the
    contingencies that island the grid (grid2op/Reward/n1Reward.py:70-99, Environment/_obsEnv.py:321-428 is what the fan-out replaces)."""
    name, n_envs = "l2rpn_wcci_2022_dev", 1024
    m0 = load_model(name)
    fan = 1 + m0.n_line
    m, eng, tab, off, sc = _bench_engine(load_model, load_npz, name, n_envs, fan)
    B = n_envs * fan
    topo = np.tile(m.initial_topo_vect().astype(np.int32), (B, 1))
    for c in range(1, fan):
        topo[c::fan, m.line_or_pos_topo_vect[c - 1]] = -1
        topo[c::fan, m.line_ex_pos_topo_vect[c - 1]] = -1
    eng.set_topology(topo)
    eng.set_trajectory(4, eng.TRAJ_OBS)
    eng.step(3, n_steps=4, rebalance=1.02)
    r = eng.results(with_bus=False)
    conv = r.converged.reshape(n_envs, fan)
    assert conv[:, 0].all()                                       # the intact grid converges in every env
    obs1 = eng.trajectory_obs(1, step0=1, lane0=0, n=fan)[0]      # an earlier step of the launch (first env): same verdicts, its own flows
    assert np.array_equal(obs1.converged, r.converged[:fan]) and not np.array_equal(obs1.out, r.out[:fan])
    lanes = np.unique(np.concatenate([np.arange(fan), 517 * fan + np.arange(fan), np.random.default_rng(2).choice(B, 80, replace=False)]))
    res = check_lanes(eng, lanes, results=r)
    assert res["ok"], res
    assert res["n_converged"] < res["n"], "the sample is meant to contain islanding contingencies"
    isl = r.status.reshape(n_envs, fan, 4)[:, :, 0] == 2
    assert (isl == isl[:1]).all() and isl[0].any()
    eng.close()


def test_wcci_1024_lanes_storage_and_redispatch_16_step_launch_vs_oracle(load_model, load_npz):
    """configs[3] as bench.py runs it: l2rpn_wcci_2022_dev (118 substations), 1 024 lanes, storage set-points U(-2, 2) MW and a
    zero-sum +-1 MW redispatch per lane, one 16-step launch (2 wavefronts per instance, Ybus blocks in registers)."""
    name, B = "l2rpn_wcci_2022_dev", 1024
    m, eng, tab, off, sc = _bench_engine(load_model, load_npz, name, B)
    inj = eng.get_injections()
    lay = eng.layout
    for k in range(B):
        inj[k, lay.inj_storage_p:lay.inj_storage_p + m.n_storage] = np.random.default_rng(k).uniform(-2.0, 2.0, m.n_storage)
    eng.set_injections(inj)
    disp = np.nonzero(~m.gen_slack)[0]
    delta = np.zeros((B, m.n_gen), dtype=np.float32)
    for k in range(B):
        a, b = np.random.default_rng(10_000_000 + k).choice(disp, size=2, replace=False)
        delta[k, a], delta[k, b] = 1.0, -1.0
    eng.set_lane_redispatch(delta)
    p = eng.plan()
    assert p["wavefronts_per_instance"] == 2 and p["ybus_in_registers"] == 1, p
    n, t0 = 16, 7
    eng.set_trajectory(n, eng.TRAJ_OBS)
    eng.step(t0, n_steps=n, rebalance=1.02)
    r = eng.results(with_bus=False)
    assert r.converged.all()
    lanes = np.sort(np.random.default_rng(8).choice(B, 64, replace=False))
    res = check_lanes(eng, lanes, results=r)
    assert res["ok"] and res["n_converged"] == 64, res
    # the injections the last step left on the device are the chronics row + jitter + rebalancing + redispatch + storage
    T = tab.shape[0]
    nl, ng = m.n_load, m.n_gen
    inj_d = eng.get_injections()
    for k in lanes[:16]:
        row = tab[(t0 + n - 1 + off[k]) % T]
        lp = row[:nl] * sc[k, :nl]
        pp = row[2 * nl:2 * nl + ng].copy()
        ns = ~m.gen_slack
        pp[ns] = pp[ns] * np.float32(1.02 * lp.astype(np.float64).sum() / row[2 * nl:2 * nl + ng][ns].astype(np.float64).sum())
        pp = pp + delta[k]
        assert np.array_equal(inj_d[k, lay.inj_load_p:lay.inj_load_p + nl], lp.astype(np.float64))
        assert np.allclose(inj_d[k, lay.inj_gen_p:lay.inj_gen_p + ng][ns], pp[ns].astype(np.float64), rtol=0, atol=2e-5)
        assert np.array_equal(inj_d[k, lay.inj_storage_p:lay.inj_storage_p + m.n_storage], inj[k, lay.inj_storage_p:lay.inj_storage_p + m.n_storage])
    # an earlier step of the same launch, straight from the observation trajectory: storage power as set, flows obey KCL
    obs = eng.trajectory_obs(1, step0=6)[0]
    assert np.allclose(obs.storage_p, inj[:, lay.inj_storage_p:lay.inj_storage_p + m.n_storage], atol=1e-6)
    p_bus = np.zeros((B, m.n_sub))
    for sub, val in [(m.line_or_sub, obs.p_or), (m.line_ex_sub, obs.p_ex), (m.load_sub, obs.load_p), (m.gen_sub, -obs.gen_p),
                     (m.shunt_sub, obs.shunt_p), (m.storage_sub, obs.storage_p)]:
        np.add.at(p_bus, (slice(None), sub), val.astype(np.float64))
    assert np.abs(p_bus).max() < 1e-2
    eng.close()


def test_idf_2023_ac_env_steps_and_ptdf_on_the_same_chronics_rows_vs_oracle(load_model, load_npz):
    """configs[4] as bench.py runs it: l2rpn_idf_2023, chronics 2035-01-15_0, 2 048 lanes.  One 16-step launch with
    the observation trajectory: 64 sampled lanes x 4 steps recomputed by the C oracle FROM THE CHRONICS TABLE; then the PTDF launch on
    the injection rows the last step left (= the same chronics rows) against the oracle's DC power flow of those rows."""
    from oracle.pf_oracle_c import COracle
    name, B = "l2rpn_idf_2023", 2048
    m, eng, tab, off, sc = _bench_engine(load_model, load_npz, name, B)
    assert tab.shape == (576, 2 * 99 + 2 * 62)
    assert eng.plan()["wavefronts_per_instance"] == 2
    n, t0 = 16, 9
    eng.set_trajectory(n, eng.TRAJ_OBS)
    eng.step(t0, n_steps=n, rebalance=1.02)
    obs = eng.trajectory_obs(n)
    _, st = eng.trajectory(n)
    lanes = np.sort(np.random.default_rng(23).choice(B, 64, replace=False))
    for k in (0, 6, 11, 15):
        assert (st[k] == 0).all(), k
        res = check_step(m, tab, off, sc, 1.02, t0 + k, lanes, obs[k].out[lanes], st[k][lanes])
        assert res["ok"] and res["n_converged"] == 64, (k, res)
    last = eng.results(with_bus=False)
    res = check_step(m, tab, off, sc, 1.02, t0 + n - 1, lanes, last.out[lanes], last.status[lanes])
    assert res["ok"] and res.get("n_iter_mismatch", 1) == 0, res
    # DC sensitivity path on the rows the lanes hold now
    inj = eng.get_injections()
    lay = eng.layout
    nl = m.n_load
    for k in lanes[:8]:                                # ... which ARE the chronics rows of the last step (float32 values, jittered)
        row = tab[(t0 + n - 1 + off[k]) % tab.shape[0]]
        assert np.array_equal(inj[k, lay.inj_load_p:lay.inj_load_p + nl], (row[:nl] * sc[k, :nl]).astype(np.float64))
    eng.ptdf_build(0)
    flows = eng.ptdf_flows()
    topo, sb = eng.get_topology(0, 1)
    ref = COracle(m).solve_rows(inj[lanes], np.tile(topo, (64, 1)), np.tile(sb, (64, 1)) if m.n_shunt else None, is_dc=True)
    assert (ref["status"][:, 0] == 0).all()
    p_ref = ref["out"][:, lay.out_p_or:lay.out_p_or + m.n_line]
    err = np.abs(flows[lanes] - p_ref)
    assert np.all(err <= 2e-4 + 5e-6 * np.abs(p_ref)), float(err.max())
    # ... and 16 consecutive chronics rows of every lane as ONE GEMM (gpf_ptdf_flows_rows gathers the injections from the chronics table
    # itself): row n - 1 of the launch that starts at t0 is the row the lanes hold -> the two device paths agree; 64 (lane, row) pairs
    # against the oracle's DC power flow of the K9 injections of that row
    rows = eng.ptdf_flows_rows(t0, n, rebalance=1.02)
    assert rows.shape == (n, B, m.n_line)
    assert np.abs(rows[n - 1] - flows).max() < 2e-4
    rg = np.random.default_rng(29)
    ls_, rs_ = rg.choice(B, 64, replace=False), rg.integers(0, n, 64)
    ng = m.n_gen
    xs = []
    for k, j in zip(ls_, rs_):
        row = tab[(t0 + j + off[k]) % tab.shape[0]]
        lp = row[:nl] * sc[k, :nl]
        pp = row[2 * nl:2 * nl + ng].copy()
        ns = ~m.gen_slack
        pp[ns] = pp[ns] * np.float32(1.02 * lp.astype(np.float64).sum() / row[2 * nl:2 * nl + ng][ns].astype(np.float64).sum())
        x = inj[k].copy()
        x[lay.inj_load_p:lay.inj_load_p + nl] = lp
        x[lay.inj_gen_p:lay.inj_gen_p + ng] = pp
        xs.append(x)
    ref = COracle(m).solve_rows(np.asarray(xs), np.tile(topo, (64, 1)), np.tile(sb, (64, 1)) if m.n_shunt else None, is_dc=True)
    p_ref = ref["out"][:, lay.out_p_or:lay.out_p_or + m.n_line]
    err = np.abs(rows[rs_, ls_] - p_ref)
    assert np.all(err <= 2e-4 + 5e-6 * np.abs(p_ref)), float(err.max())
    # a ragged tail (a lane count that is not a multiple of the 64-pair blocks) and a single row
    one = eng.ptdf_flows_rows(t0 + n - 1, 1, rebalance=1.02, lane0=3, n=101)
    assert np.abs(one[0] - flows[3:104]).max() < 2e-4
    eng.close()


@pytest.mark.parametrize("B", [2048, 32768])
def test_ptdf_path_at_bench_size_vs_dc_oracle(B, load_model):
    """configs[4] as bench.py runs it: l2rpn_idf_2023, one FP64-MFMA PTDF launch over 2 048 (and 32 768) lanes; 64 sampled lanes
    against the C oracle's DC power flow (pp.rundcpp restated) of the same injections."""
    from grid2op_amd.engine import PowerFlowEngine
    from oracle.pf_oracle_c import COracle
    m = load_model("l2rpn_idf_2023")
    eng = PowerFlowEngine(m, n_lanes=B, device=0)
    base = eng.get_injections(0, 1)
    lay = eng.layout
    inj = np.tile(base, (2048, 1))
    for k in range(2048):
        f = 1.0 + 0.05 * np.random.default_rng(k).standard_normal(m.n_load)
        lp = inj[k, lay.inj_load_p:lay.inj_load_p + m.n_load]
        gp = inj[k, lay.inj_gen_p:lay.inj_gen_p + m.n_gen]
        gp *= (lp * f).sum() / lp.sum()
        lp *= f
    inj = np.tile(inj, (B // 2048, 1))
    eng.set_injections(inj)
    eng.ptdf_build(0)
    flows = eng.ptdf_flows()
    lanes = np.sort(np.random.default_rng(B).choice(B, 64, replace=False))
    topo, sb = eng.get_topology(0, 1)
    ref = COracle(m).solve_rows(inj[lanes], np.tile(topo, (64, 1)), np.tile(sb, (64, 1)) if m.n_shunt else None, is_dc=True)
    assert (ref["status"][:, 0] == 0).all()
    p_ref = ref["out"][:, lay.out_p_or:lay.out_p_or + m.n_line]
    err = np.abs(flows[lanes] - p_ref)
    assert np.all(err <= 2e-4 + 5e-6 * np.abs(p_ref)), float(err.max())
    if B > 2048:                                                # tiled inputs: identical lanes give identical flows
        assert np.array_equal(flows[:2048], flows[B - 2048:])
    eng.close()


@pytest.mark.parametrize("name", ["l2rpn_wcci_2022_dev", "l2rpn_idf_2023"])
def test_118_substations_every_step_of_a_launch_vs_oracle_from_the_chronics_table(name, load_model, load_npz):
    """Round 6: on the two-wavefront kernels the chronics-driven injections of the steps INSIDE a launch no longer pass through the lane's
    injection row (K9's owner lanes hand them to the results phase in registers), the results row goes to HBM through a staged copy in
    LDS, and the narrow LU passes run on one wavefront.  512 lanes, one 20-step launch with the observation trajectory (the launch length of
    the driver's bench): steps 0, 1, 9, 18 and 19 of 48 sampled lanes are recomputed by the C oracle FROM THE CHRONICS TABLE (gather, jitter,
    rebalancing, power flow, result row) -- status and iteration count bit for bit --, every observation of the launch obeys KCL, and the
    injection row the launch leaves behind is the LAST step's."""
    import bench
    B, n, t0 = 512, bench.launch_length(bench.DRIVER_STEPS), 11
    m, eng, tab, off, sc = _bench_engine(load_model, load_npz, name, B)
    assert eng.plan()["wavefronts_per_instance"] == 2
    eng.set_trajectory(n, eng.TRAJ_OBS)
    eng.step(0, n_steps=n, rebalance=1.02)             # a first launch: the checked one starts from a used state
    eng.step(t0, n_steps=n, rebalance=1.02)
    obs = eng.trajectory_obs(n)
    _, st = eng.trajectory(n)
    lanes = np.sort(np.random.default_rng(3).choice(B, 48, replace=False))
    for k in (0, 1, 9, n - 2, n - 1):
        res = check_step(m, tab, off, sc, 1.02, t0 + k, lanes, obs[k].out[lanes], st[k][lanes])
        assert res["ok"] and res["n_converged"] == 48 and res.get("n_iter_mismatch", 0) == 0, (k, res)
    for k in range(n):
        r = obs[k]
        assert (st[k] == 0).all()
        p_bus = np.zeros((B, m.n_sub))
        for sub, val in [(m.line_or_sub, r.p_or), (m.line_ex_sub, r.p_ex), (m.load_sub, r.load_p), (m.gen_sub, -r.gen_p),
                         (m.shunt_sub, r.shunt_p), (m.storage_sub, r.storage_p)]:
            np.add.at(p_bus, (slice(None), sub), val.astype(np.float64))
        assert np.abs(p_bus).max() < 2e-2, k
    last = eng.results(with_bus=False)
    assert np.array_equal(last.out, obs[n - 1].out)
    lay, nl = eng.layout, m.n_load
    inj_d = eng.get_injections()
    T = tab.shape[0]
    for k in lanes[:16]:
        row = tab[(t0 + n - 1 + off[k]) % T]
        assert np.array_equal(inj_d[k, lay.inj_load_p:lay.inj_load_p + nl], (row[:nl] * sc[k, :nl]).astype(np.float64))
    eng.close()

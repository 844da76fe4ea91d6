"""GPU tests of the batched obs.simulate (gpf_simulate_batch): B source lanes x K candidate actions x forecast horizon in ONE
launch, against (1) simulations recorded inside the UNMODIFIED reference environment (tests/golden/simulate_case14.npz, made by
tests/golden/make_simulate_fixtures.py: Observation/baseObservation.py:3365-3670 -> Environment/_obsEnv.py) and (2) the oracle's
restatement of the same path (oracle/env_oracle.py, itself pinned to those recordings on CPU) on other grids / states."""
import numpy as np
import pytest

from oracle.env_oracle import simulate
from oracle.pf_oracle import LaneState

from test_oracle_simulate import sim_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("grid,fixture,min_done", [("l2rpn_case14_sandbox", "simulate_case14.npz", 20),
                                                  ("l2rpn_neurips_2020_track1", "simulate_maintenance_neurips36.npz", 0)])
def test_recorded_reference_simulations_in_one_call_per_horizon(grid, fixture, min_done, load_model, load_npz):
    """second fixture: scheduled maintenance ahead of the observation -- the forecast one step ahead has the line out that the
    maintenance table takes out at the next row (_ObsEnv.init, Environment/_obsEnv.py:361-385), the forecast of horizon 0 has not"""
    from grid2op_amd.engine import PowerFlowEngine
    m = load_model(grid)
    fx = load_npz(fixture)
    cands = sim_cases(fx)
    row0 = int(fx["row0"]) if "row0" in fx else 0
    S, K = fx["row"].shape[0], len(cands)
    eng = PowerFlowEngine(m, n_lanes=S + S * K, device=0)
    eng.upload_chronics(eng.pack_chronics(fx["ch_load_p"], fx["ch_load_q"], fx["ch_prod_p"], fx["ch_prod_v"]))
    eng.upload_forecasts(eng.pack_chronics(fx["fc_load_p"], fx["fc_load_q"], fx["fc_prod_p"], fx["fc_prod_v"]))
    if "maintenance" in fx:
        eng.upload_maintenance(fx["maintenance"])
    eng.set_thermal_limits(fx["thermal_limit"])
    # source lane s = the environment after recorded step s: topology, chronics cursor (row = t_obs + offset), protection counters
    off = np.zeros(S + S * K, np.int32)
    off[:S] = fx["row"] - row0
    eng.set_lane_chronics(lane_offset=off)
    eng.set_topology(fx["topo_vect"].astype(np.int32), lane0=0)
    eng.set_overflow_count(fx["timestep_overflow"], lane0=0)
    kw = dict(cascade=bool(fx["cascade"]) if "cascade" in fx else True, hard_overflow=float(fx["hard_overflow"]), nb_ts_allowed=int(fx["nb_ts_allowed"]))
    n_done = 0
    for ts in (0, 1):
        n = eng.simulate_batch(0, np.arange(S), cands, dst_lane0=S, time_step=ts, last_bus=fx["last_bus"], **kw)
        assert n == S * K
        r = eng.results(S, n)
        rho, _, _ = eng.step_outputs(S, n)
        for s in range(S):
            for k in range(K):
                q = s * K + k
                done = bool(fx[f"sim{ts}_done"][s, k])
                assert (not r.converged[q]) == done, (ts, s, k, r.status[q])
                n_done += done
                if done:
                    continue
                assert np.array_equal(r.topo_vect[q], fx[f"sim{ts}_topo_vect"][s, k]), (ts, s, k)
                assert np.array_equal(r.line_status[q], fx[f"sim{ts}_line_status"][s, k]), (ts, s, k)
                for f, tol in [("p_or", 3e-4), ("q_or", 4e-4), ("p_ex", 3e-4), ("v_or", 3e-4), ("a_or", 2e-3), ("gen_p", 3e-4), ("gen_q", 4e-4),
                               ("load_p", 1e-5), ("load_v", 3e-4)]:
                    assert np.abs(getattr(r, f)[q].astype(np.float64) - fx[f"sim{ts}_{f}"][s, k]).max() < tol, (ts, s, k, f)
                assert np.abs(rho[q] - fx[f"sim{ts}_rho"][s, k]).max() < 3e-5, (ts, s, k)
    assert n_done >= min_done
    if "maintenance" in fx:                     # (the recording has an observation with the line still on whose 1-step forecast has it off)
        assert (fx["line_status"] & ~fx["sim1_line_status"][:, 0]).any()
    # the source lanes were not touched
    t_src, _ = eng.get_topology(0, S)
    assert np.array_equal(t_src, fx["topo_vect"])
    eng.close()


@pytest.mark.parametrize("name,B", [("l2rpn_neurips_2020_track1", 5), ("l2rpn_wcci_2022_dev", 3), ("educ_case14_storage", 6)])
def test_simulate_batch_vs_oracle_restatement(name, B, load_model, load_npz):
    """Other grids (1 wavefront per lane, 2 wavefronts per lane, storage units), source lanes in different states (open lines, a
    split substation, their own chronics rows), 9 candidates each incl. reconnections and bus changes; synthetic forecasts."""
    from grid2op_amd.engine import PowerFlowEngine
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    if "load_p" not in ch:               # no chronics fixture for this grid: 48 jittered copies of the stored state
        rg = np.random.default_rng(99)
        j = lambda v: (np.asarray(v, np.float64)[None, :] * (1 + 0.03 * rg.standard_normal((48, len(v))))).astype(np.float32)  # noqa: E731
        ch.update(load_p=j(m.load_p0), load_q=j(m.load_q0), prod_p=j(m.gen_p0))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    rng = np.random.default_rng(len(name))
    cands = [{}, {"set_line_status": [(1, -1)]}, {"set_line_status": [(2, +1)]}, {"change_line_status": [4]},
             {"lines_or_bus": [(2, 2)]}, {"lines_ex_bus": [(5, -1)]}, {"loads_bus": [(0, 2)]},
             {"change_bus": [int(m.line_or_pos_topo_vect[6]), int(m.gen_pos_topo_vect[1])]}, {"set_line_status": [(0, -1), (3, -1)]}]
    K = len(cands)
    eng = PowerFlowEngine(m, n_lanes=B + B * K, device=0)
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"])[:48]
    fc = np.stack([tab * np.float32(1.01), tab * np.float32(0.98)], axis=1)            # 2 horizons [T, 2, n_chron]
    eng.upload_chronics(tab)
    eng.upload_forecasts(fc[None])
    lim = ch.get("thermal_limits")
    if lim is not None:
        eng.set_thermal_limits(lim)
    off = np.zeros(B + B * K, np.int32)
    off[:B] = rng.integers(0, 40, B)
    eng.set_lane_chronics(lane_offset=off)
    topo = np.tile(m.initial_topo_vect(), (B, 1))
    last = np.ones((B, m.dim_topo), np.int32)
    for b in range(1, B):
        l = int(rng.integers(0, m.n_line))
        topo[b, m.line_or_pos_topo_vect[l]] = topo[b, m.line_ex_pos_topo_vect[l]] = -1
        last[b, m.line_or_pos_topo_vect[l]] = 2
    topo[1, m.line_or_pos_topo_vect[2]] = topo[1, m.line_ex_pos_topo_vect[2]] = -1       # candidate 2 reconnects it (origin end -> busbar 2)
    last[1, m.line_or_pos_topo_vect[2]] = 2
    eng.set_topology(topo, lane0=0)
    ovc = rng.integers(0, 2, (B, m.n_line)).astype(np.int32)
    eng.set_overflow_count(ovc, lane0=0)
    t_obs = 3
    for ts in (0, 2):
        eng.simulate_batch(t_obs, np.arange(B), cands, dst_lane0=B, time_step=ts, last_bus=last, cascade=lim is not None)
        r = eng.results(B, B * K)
        for b in range(B):
            idx = (t_obs + off[b]) % tab.shape[0]
            row = tab[idx] if ts == 0 else fc[idx, ts - 1]
            base = LaneState.from_model(m)
            base.topo = topo[b].copy()
            for k, act in enumerate(cands):
                res, st, _ = simulate(m, base, row, act, lim if lim is not None else np.full(m.n_line, 1e30, np.float32), ovc[b],
                                      last_bus=last[b], cascade=lim is not None)
                q = b * K + k
                assert bool(r.converged[q]) == bool(res.converged), (ts, b, k, r.status[q], res.reason)
                if not res.converged:
                    continue
                assert np.array_equal(r.topo_vect[q], res.topo_vect) and np.array_equal(r.line_status[q], res.line_status.astype(bool)), (ts, b, k)
                for f in ("p_or", "q_or", "a_or", "v_or", "gen_p", "gen_q", "load_v"):
                    ref = getattr(res, f)
                    assert np.all(np.abs(getattr(r, f)[q] - ref) <= 2e-4 + 5e-6 * np.abs(ref)), (ts, b, k, f)
    from grid2op_amd.engine import GridPFError
    with pytest.raises(GridPFError):
        eng.simulate_batch(t_obs, np.arange(B), cands, dst_lane0=B, time_step=3)          # only 2 horizons uploaded
    with pytest.raises(GridPFError):
        eng.simulate_batch(t_obs, [B + 1], cands, dst_lane0=B)                            # source inside the destination range
    with pytest.raises(GridPFError):
        eng.simulate_batch(t_obs, np.arange(B), [{"set_bus": {0: 7}}], dst_lane0=B)       # bus id beyond n_busbar
    eng.close()


def test_a_scratch_lane_cannot_be_the_source_of_another_simulation(load_model, load_npz):
    """Round-4 advisor finding: the chronics cursor of a scratch lane of gpf_simulate_batch is an ABSOLUTE row -- of the forecast tables
    when it simulated a forecast --, not an offset to the time index; using it as the source of another gpf_simulate_batch would read the
    maintenance table (and the chronics) at the wrong row.  The call is refused; after gpf_set_lane_chronics every lane is a source again."""
    from grid2op_amd.engine import GridPFError, PowerFlowEngine
    name = "l2rpn_case14_sandbox"
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    eng = PowerFlowEngine(m, n_lanes=16, device=0)
    prod_v = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch.get("prod_v", prod_v))[:48]
    eng.upload_chronics(tab)
    eng.upload_forecasts(np.stack([tab * np.float32(1.01)], axis=1)[None])
    off = np.zeros(16, np.int32)
    off[:2] = (3, 9)
    eng.set_lane_chronics(lane_offset=off)
    cands = [{}, {"set_line_status": [(1, -1)]}]
    eng.simulate_batch(2, [0, 1], cands, dst_lane0=2, time_step=1)          # lanes 2..5: forecast rows
    eng.simulate_batch(2, [0, 1], cands, dst_lane0=6, time_step=0)          # lanes 6..9: the current chronics row, absolute
    eng.sync()
    assert eng.results(2, 8).converged.all()
    for bad_src, ts in (([2], 1), ([0, 3], 0), ([6], 1)):
        with pytest.raises(GridPFError, match="chained simulate"):
            eng.simulate_batch(2, bad_src, cands, dst_lane0=12, time_step=ts)
    eng.set_lane_chronics(lane_offset=off)                                   # every lane back on the chronics tables
    eng.simulate_batch(2, [2], cands, dst_lane0=8, time_step=1)
    eng.sync()
    assert eng.results(8, 2).converged.all()
    eng.close()


def test_scratch_step_with_two_horizons_reads_no_outage_table_and_keeps_cooldowns(load_model, load_npz):
    """the scratch step of gpf_simulate_batch runs on a row of the FORECAST tables ([table][T][n_horizons][..]); the outage
    tables and their remaining-duration table have [table][T] rows only.  The planned maintenance of the simulated step reaches the candidates
    through the call's own host-side rule (as obs.simulate applies it, _obsEnv.py:321-428); the KERNEL of the scratch step must neither index
    the outage / duration tables with its forecast cursor nor maintain the line cooldowns -- whatever the caller's gpf_step_opts say (a C
    caller's track_cooldown = 1 here; a zero-initialised struct before ABI 323 meant "track").  Two forecast horizons + an uploaded
    maintenance table with line 5 out at every row: the candidates' results equal those of an engine WITHOUT outage tables whose candidates
    open line 5 themselves, and every scratch lane keeps the cooldowns copied from its source."""
    import ctypes as C
    from grid2op_amd._capi import GpfStepOpts, check, ptr
    from grid2op_amd.engine import PowerFlowEngine
    name = "l2rpn_case14_sandbox"
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    prod_v = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    B, K = 3, 3
    outs = []
    for with_tables in (False, True):
        cands = [{}, {"set_line_status": [(1, -1)]}, {"set_line_status": [(3, -1)]}]
        if not with_tables:
            cands = [dict(set_line_status=c.get("set_line_status", []) + [(5, -1)]) for c in cands]
        eng = PowerFlowEngine(m, n_lanes=B + B * K, device=0)
        tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch.get("prod_v", prod_v))[:48]
        eng.upload_chronics(tab)
        eng.upload_forecasts(np.stack([tab * np.float32(1.01), tab * np.float32(0.97)], axis=1)[None])      # fc_h = 2
        off = np.zeros(B + B * K, np.int32)
        off[:B] = (5, 17, 40)
        eng.set_lane_chronics(lane_offset=off)
        cool = np.zeros((B, m.n_line), np.int32)
        cool[0, 3], cool[1, 7], cool[2, 0] = 4, 9, 2
        if with_tables:
            mt = np.zeros((48, m.n_line), np.uint8)
            mt[:, 5] = 1                                       # line 5 in maintenance at every row
            eng.upload_maintenance(mt)
            eng.set_cooldown(cool, lane0=0)
        src = np.arange(B, dtype=np.int32)
        a_off, items = eng.pack_actions(cands)
        o = GpfStepOpts(10, 1e-8, 0.0, 0, 2.0, 1.0, 2, 16, 0, 0, 0, 1, 10)       # track_cooldown = 1, nb_ts_reco = 10: must be ignored here
        for ts in (1, 2):
            check(eng._lib.gpf_simulate_batch(eng._h, 2, ts, B, ptr(src, C.c_int32), K, ptr(a_off, C.c_int32), ptr(items if items.size else None, C.c_int32),
                                              None, B, C.byref(o)), "gpf_simulate_batch")
            r = eng.results(B, B * K)
            assert r.converged.all() and not r.line_status[:, 5].any(), ts
            outs.append((with_tables, ts, r.out.copy(), r.topo_vect.copy()))
            if with_tables:
                got = eng.cooldown(B, B * K).reshape(B, K, m.n_line)
                assert np.array_equal(got, np.repeat(cool[:, None, :], K, axis=1)), ts
        eng.close()
    for (_, ts_a, out_a, tv_a), (_, ts_b, out_b, tv_b) in zip(outs[:2], outs[2:]):
        assert ts_a == ts_b and np.array_equal(out_a, out_b) and np.array_equal(tv_a, tv_b)

"""Pin the CPU oracle (oracle/pf_oracle.py) to the reference solver's own results.

Golden vectors = pandapower outputs shipped INSIDE the reference's fixtures (embedded ``res_*`` tables of
the bundled grid.json files) and the known answers hard-coded in grid2op/tests/BaseBackendTest.py
(:262-287 DC p_or, :289-319 AC p_or, :1584-1607 a_or) -- extracted by tests/golden/make_fixtures.py.
"""
import numpy as np
import pytest

from oracle.pf_oracle import LaneState, solve

FULL_GOLDEN = ["rte_case5_example", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev", "l2rpn_2019", "l2rpn_wcci_2020",
               "l2rpn_icaps_2021"]


@pytest.mark.parametrize("name", FULL_GOLDEN)
def test_ac_solution_matches_embedded_pandapower_results(name, load_model, load_npz):
    m = load_model(name)
    g = load_npz(f"{name}.res.npz")
    r = solve(m, LaneState.from_model(m))
    assert r.converged and r.n_iter <= 5
    nl = m.n_powerline
    tol = 2e-9
    assert np.nanmax(np.abs(r.bus_vm[:m.n_sub] - g["bus_vm_pu"])) < 1e-11
    assert np.nanmax(np.abs(r.bus_va[:m.n_sub] - g["bus_va_degree"])) < 1e-9
    assert np.abs(r.p_or[:nl] - g["line_p_from_mw"]).max() < tol
    assert np.abs(r.q_or[:nl] - g["line_q_from_mvar"]).max() < tol
    assert np.abs(r.p_ex[:nl] - g["line_p_to_mw"]).max() < tol
    assert np.abs(r.q_ex[:nl] - g["line_q_to_mvar"]).max() < tol
    assert np.abs(r.a_or[:nl] - 1000 * g["line_i_from_ka"]).max() < 1e-8
    assert np.abs(r.a_ex[:nl] - 1000 * g["line_i_to_ka"]).max() < 1e-8
    assert np.abs(r.theta_or[:nl] - g["line_va_from_degree"]).max() < 1e-9
    assert np.abs(r.v_or[:nl] / m.sub_vn_kv[m.line_or_sub[:nl]] - g["line_vm_from_pu"]).max() < 1e-11
    if "trafo_p_hv_mw" in g:
        assert np.abs(r.p_or[nl:] - g["trafo_p_hv_mw"]).max() < tol
        assert np.abs(r.q_or[nl:] - g["trafo_q_hv_mvar"]).max() < tol
        assert np.abs(r.p_ex[nl:] - g["trafo_p_lv_mw"]).max() < tol
        assert np.abs(r.q_ex[nl:] - g["trafo_q_lv_mvar"]).max() < tol
        assert np.abs(r.a_or[nl:] - 1000 * g["trafo_i_hv_ka"]).max() < 1e-8
        assert np.abs(r.a_ex[nl:] - 1000 * g["trafo_i_lv_ka"]).max() < 1e-8
    ng = len(g["gen_p_mw"])                      # legacy files: the appended slack generator is not in res_gen
    assert np.abs(r.gen_p[:ng] - g["gen_p_mw"]).max() < tol
    # +-1e9 MVAr default limits make pandapower's own range-split lose ~1e-7 (catastrophic cancellation)
    assert np.abs(r.gen_q[:ng] - g["gen_q_mvar"]).max() < 5e-7
    if "shunt_q_mvar" in g:
        assert np.abs(r.shunt_q - g["shunt_q_mvar"]).max() < tol
        assert np.abs(r.shunt_p - g["shunt_p_mw"]).max() < tol
    if "ext_grid_p_mw" in g:                     # legacy: appended slack gen reports the ext_grid balance
        assert abs(r.gen_p[-1] - g["ext_grid_p_mw"][0]) < tol
        assert abs(r.gen_q[-1] - g["ext_grid_q_mvar"][0]) < tol


@pytest.mark.parametrize("name", ["l2rpn_idf_2023", "rte_case118_example", "l2rpn_neurips_2020_track2_x1"])
def test_branch_level_consistency_when_embedded_results_are_of_another_state(name, load_model, load_npz):
    """l2rpn_idf_2023, rte_case118_example (IEEE 118) and l2rpn_neurips_2020_track2: the embedded results belong to another
    injection state -> only check V -> branch flow consistency (line and transformer models):
    impose the golden bus voltages."""
    m = load_model(name)
    g = load_npz(f"{name}.res.npz")
    V = g["bus_vm_pu"] * np.exp(1j * np.radians(g["bus_va_degree"]))
    f, t = m.line_or_sub, m.line_ex_sub
    Sf = V[f] * np.conj(m.br_yff * V[f] + m.br_yft * V[t]) * m.sn_mva
    St = V[t] * np.conj(m.br_ytf * V[f] + m.br_ytt * V[t]) * m.sn_mva
    nl = m.n_powerline
    assert np.abs(Sf.real[:nl] - g["line_p_from_mw"]).max() < 1e-9
    assert np.abs(Sf.imag[:nl] - g["line_q_from_mvar"]).max() < 1e-9
    assert np.abs(Sf.real[nl:] - g["trafo_p_hv_mw"]).max() < 1e-9
    assert np.abs(St.imag[nl:] - g["trafo_q_lv_mvar"]).max() < 1e-9


def test_known_answers_of_reference_backend_tests(load_model, load_npz):
    """grid2op/tests/BaseBackendTest.py:258-319 (test_runpf_dc / test_runpf) and :1584-1607, on the legacy
    data_test/test_PandaPower/test_case14.json (rows in FILE order, slack generator appended)."""
    m = load_model("test_case14")
    ka = load_npz("known_answers.npz")
    assert m.n_gen == 5 and m.slack_added            # BaseBackendTest.py:105
    st = LaneState.from_model(m)
    r = solve(m, st)
    assert r.converged
    assert np.abs(r.p_or - ka["p_or_ac"]).max() < 1e-6        # reference tolerance: 1e-2 (compare_vect)
    assert np.abs(r.a_or / ka["a_or_init"] - 1).max() < 1e-9
    rdc = solve(m, st, is_dc=True)
    assert rdc.converged
    assert np.abs(rdc.p_or - ka["p_or_dc"]).max() < 1e-7


def test_islanded_grid_diverges_in_ac_and_dc(load_model):
    """aaa_test_backend_interface.py:1095-1164: an islanded grid must give (False, exc) in AC and DC."""
    m = load_model("l2rpn_case14_sandbox")
    st = LaneState.from_model(m)
    # cut every line of substation 13 except none -> isolate the load there
    for l in range(m.n_line):
        if m.line_or_sub[l] == 13 or m.line_ex_sub[l] == 13:
            st.topo[m.line_or_pos_topo_vect[l]] = -1
            st.topo[m.line_ex_pos_topo_vect[l]] = -1
    for dc in (False, True):
        r = solve(m, st, is_dc=dc)
        assert not r.converged
        assert np.all(np.isnan(r.p_or)) and np.all(r.topo_vect == -1) and not r.line_status.any()


def test_observation_recorded_with_pandapower_backend(load_model, load_npz):
    """grid2op/tests/test_Observation.py:307-... (``json_ref``): the complete observation of rte_case14_test after reset,
    recorded with PandaPowerBackend and compared there for exact float32 equality.  The oracle is fed row 0 of
    chronics/0 the way ``_BackendAction`` hands it over (float32 set-points, float32 prod_v / vn_kv) and must reproduce
    the recorded flows / voltages / reactive dispatch; integer vectors bit-exactly."""
    m = load_model("rte_case14_test")
    ch = load_npz("rte_case14_test.chronics.npz")
    ka = load_npz("known_answers.npz")
    st = LaneState.from_model(m)
    st.load_p = ch["load_p"][0].astype(np.float64)
    st.load_q = ch["load_q"][0].astype(np.float64)
    st.gen_p = ch["prod_p"][0].astype(np.float64)
    st.gen_vm = (ch["prod_v"][0] / m.sub_vn_kv[m.gen_sub].astype(np.float32)).astype(np.float64)
    r = solve(m, st)
    assert r.converged
    assert np.array_equal(r.topo_vect, ka["obs14_topo_vect"].astype(np.int32))
    assert np.array_equal(r.line_status.astype(bool), ka["obs14_line_status"].astype(bool))
    for f, tol in [("p_or", 2e-5), ("q_or", 1e-4), ("p_ex", 2e-5), ("q_ex", 1e-4), ("v_or", 2e-5), ("v_ex", 2e-5)]:
        assert np.abs(getattr(r, f) - ka["obs14_" + f]).max() < tol, f
    assert np.abs(r.a_or / ka["obs14_a_or"] - 1).max() < 2e-6
    assert np.abs(r.a_ex / ka["obs14_a_ex"] - 1).max() < 2e-6
    assert np.abs(r.load_p - ka["obs14_load_p"]).max() < 1e-6 and np.abs(r.load_v - ka["obs14_load_v"]).max() < 2e-5
    assert np.abs(r.gen_v - ka["obs14_gen_v"]).max() < 2e-5
    assert np.abs(r.gen_q - ka["obs14_gen_q"]).max() < 1e-4
    ns = ~m.gen_slack
    assert np.abs(r.gen_p[ns] - ka["obs14_gen_p"][ns]).max() < 1e-6
    assert np.abs(r.gen_p[m.gen_slack] - ka["obs14_gen_p"][m.gen_slack]).max() < 1e-4      # slack P = losses balance
    assert np.abs(r.a_or / ch["thermal_limits"] - ka["obs14_rho"]).max() < 2e-6
    for f in ("theta_or", "theta_ex", "load_theta", "gen_theta"):
        if "obs14_" + f in ka:
            d = np.abs(getattr(r, f) - ka["obs14_" + f])
            assert np.minimum(d, 360 - d).max() < 2e-5, f


def test_runner_trajectories_recorded_with_pandapower_backend(load_model, load_npz):
    """grid2op/data_test/runner_data/res_agent_<ver>/{00,01} (loaded by grid2op/tests/test_Runner.py:426,540-585): the
    rte_case5_example RandomAgent episodes recorded with PandaPowerBackend by EVERY grid2op release the reference keeps
    (45 of the 46 version folders -- 1.9.0's vectors do not match its own observation-space file --, 90 episodes, 389
    observations).  Every observation row carries the inputs of its power flow (injections, topo_vect after the random bus
    splits / line switches) and pandapower's results, so each row pins the oracle on a DIFFERENT topology."""
    m = load_model("rte_case5_example")
    rt = load_npz("runner_case5.npz")
    tags = sorted({k[:-len("p_or")] for k in rt if k.endswith("_p_or")})
    assert len(tags) >= 90
    n_rows = n_split = 0
    for tag in tags:
        for t in range(rt[tag + "p_or"].shape[0]):
            topo = rt[tag + "topo_vect"][t].astype(np.int32)
            st = LaneState.from_model(m)
            st.topo = topo.copy()
            st.load_p = rt[tag + "load_p"][t].astype(np.float64)
            st.load_q = rt[tag + "load_q"][t].astype(np.float64)
            st.gen_p = rt[tag + "gen_p"][t].astype(np.float64)
            st.gen_vm = (rt[tag + "gen_v"][t].astype(np.float32) / m.sub_vn_kv[m.gen_sub].astype(np.float32)).astype(np.float64)
            r = solve(m, st)
            assert r.converged, (tag, t)
            assert np.array_equal(r.topo_vect, topo), (tag, t)
            assert np.array_equal(r.line_status.astype(bool), rt[tag + "line_status"][t].astype(bool))
            for f, tol in [("p_or", 5e-5), ("q_or", 2e-4), ("p_ex", 5e-5), ("q_ex", 2e-4), ("v_or", 5e-5), ("v_ex", 5e-5)]:
                assert np.abs(getattr(r, f) - rt[tag + f][t]).max() < tol, (tag, t, f)
            on = rt[tag + "a_or"][t] > 1e-6        # (a line whose ends hang alone on a busbar carries 1e-11 A of rounding noise)
            assert np.abs(r.a_or[on] / rt[tag + "a_or"][t][on] - 1).max() < 5e-6
            assert np.abs(r.a_or[~on]).max(initial=0.0) < 1e-6
            assert np.abs(r.gen_q - rt[tag + "gen_q"][t]).max() < 2e-4
            n_rows += 1
            n_split += int((topo == 2).any())
    assert n_rows >= 380 and n_split >= 250


def stats_episode_tables(fx, e, n_load, n_gen):
    """chronics rows of episode e of tests/golden/stats_case5.npz: [rows, 2 n_load + 2 n_gen]"""
    a, b = int(fx["chronics_start"][e]), int(fx["chronics_start"][e + 1])
    return fx["chronics_rows"][a:b]


def test_do_nothing_episodes_recorded_with_pandapower_backend(load_model, load_npz):
    """grid2op/data/rte_case5_example/_statistics (shipped with the reference; EpisodeStatistics.compute,
    utils/underlying_statistics.py:680-813): the DoNothingAgent through the 20 scenarios of rte_case5_example with the default
    parameters -- overflow protections ON -- and PandaPowerBackend, every observation of every step (7 930 rows; 19 episodes end in a
    game over after overloaded lines tripped).  The headline workload recorded with pandapower itself: replayed here step by step
    with the oracle's environment step (chronics row -> injections -> `next_grid_state`), comparing the line status and the
    protection counters at EVERY step, the flows / voltages / generator results at the sub-sampled rows of the fixture, and the
    step at which the episode ends."""
    from oracle.env_oracle import forecast_state, next_grid_state
    m = load_model("rte_case5_example")
    fx = load_npz("stats_case5.npz")
    start = fx["episode_start"]
    pos = {int(r): i for i, r in enumerate(fx["row_idx"])}
    lim = fx["thermal_limit"]
    n_cmp = n_trip = n_over = 0
    for e in range(len(start) - 1):
        n = int(start[e + 1] - start[e])
        tab = stats_episode_tables(fx, e, m.n_load, m.n_gen)
        st = LaneState.from_model(m)
        ts = np.zeros(m.n_line, np.int64)
        ended = None
        for t in range(n):
            s2 = forecast_state(m, st, tab[t])
            # row 0 is the reset observation: a plain power flow, the protections do not run (Environment.reset)
            res, s3, ts_new = next_grid_state(m, s2, lim, ts, cascade=t > 0)
            if t > 0:
                ts = ts_new
            st.topo = s3.topo
            r = int(start[e]) + t
            if not res.converged:
                ended = t
                break
            assert np.array_equal(res.line_status.astype(bool), fx["line_status_all"][r]), (e, t)
            assert np.array_equal(ts, fx["timestep_overflow_all"][r]), (e, t, ts, fx["timestep_overflow_all"][r])
            if r in pos:
                i = pos[r]
                assert np.array_equal(res.topo_vect, fx["topo_vect"][i]), (e, t)
                for f, tol in [("p_or", 5e-5), ("q_or", 2e-4), ("p_ex", 5e-5), ("q_ex", 2e-4), ("v_or", 5e-5), ("v_ex", 5e-5)]:
                    assert np.abs(getattr(res, f) - fx[f][i]).max() < tol, (e, t, f)
                on = fx["a_or"][i] > 1e-6
                assert np.abs(res.a_or[on] / fx["a_or"][i][on] - 1).max() < 5e-6, (e, t)
                assert np.abs(res.a_or[on] / lim[on] - fx["rho"][i][on]).max() < 5e-6, (e, t)
                assert np.abs(res.gen_p - fx["prod_p"][i]).max() < 5e-5 and np.abs(res.gen_q - fx["prod_q"][i]).max() < 2e-4, (e, t)
                assert np.abs(res.load_p - fx["load_p"][i]).max() < 1e-5 and np.abs(res.load_v - fx["load_v"][i]).max() < 5e-5, (e, t)
                n_cmp += 1
        n_trip += int((~fx["line_status_all"][int(start[e + 1]) - 1]).any())
        if n < 2017:
            # game over: the runner stored the last valid observation once more in the episode's final row; the step that
            # produces it is the one the oracle must fail at (divergence / islanding after the trips)
            assert ended == n - 1, (e, ended, n)
            n_over += 1
        else:
            assert ended is None, (e, ended)
    assert n_cmp >= 2000 and n_over == 19 and n_trip == 19


@pytest.mark.parametrize("name,sizes", [
    ("l2rpn_neurips_2020_track1", (36, 59, 37, 22, 0)), ("l2rpn_icaps_2021", (36, 59, 37, 22, 0)),
    ("l2rpn_neurips_2020_track2_x1", (118, 186, 99, 62, 0)), ("l2rpn_case14_sandbox", (14, 20, 11, 6, 0)),
    ("educ_case14_redisp", (14, 20, 11, 6, 0)), ("educ_case14_storage", (14, 20, 11, 6, 2)),
    ("l2rpn_wcci_2022_dev", (118, 186, 91, 62, 7)), ("l2rpn_idf_2023", (118, 186, 99, 62, 7))])
def test_grid_sizes_asserted_by_the_reference(name, sizes, load_model):
    """grid2op/tests/test_attached_envs.py:30-50, 76-80, 125-129, 169-173, 260-264, 305-309, 350-354, 398-402:
    (n_sub, n_line, n_load, n_gen, n_storage) of the attached environments, exact."""
    m = load_model(name)
    assert (m.n_sub, m.n_line, m.n_load, m.n_gen, m.n_storage) == sizes
    assert m.dim_topo == 2 * m.n_line + m.n_gen + m.n_load + m.n_storage

"""The C oracle (oracle/pf_oracle.c) against its numpy twin (oracle/pf_oracle.py, the one pinned to the reference's
golden vectors) and against the same golden vectors.  Both are test infrastructure."""
import numpy as np
import pytest

from helpers import F32_FIELDS, pack_states, random_states
from oracle.pf_oracle import LaneState, solve
from oracle.pf_oracle_c import COracle


def _row_fields(orc, m):
    from grid2op_amd.engine import _OUT_FIELDS
    sizes = dict(n_line=m.n_line, n_gen=m.n_gen, n_load=m.n_load, n_storage=m.n_storage, n_shunt=m.n_shunt)
    off, sl = 0, {}
    for name, sz in _OUT_FIELDS:
        sl[name] = slice(off, off + sizes[sz])
        off += sizes[sz]
    assert off == orc.n_out
    return sl


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "educ_case14_storage", "test_case14",
                                  "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev", "rte_case118_example", "l2rpn_wcci_2020",
                                  "l2rpn_icaps_2021"])
def test_c_oracle_equals_numpy_oracle(name, load_model):
    m = load_model(name)
    orc = COracle(m)
    sl = _row_fields(orc, m)
    rng = np.random.default_rng(11)
    states = [LaneState.from_model(m)] + random_states(m, 12 if m.n_sub > 100 else 40, rng)
    inj, topo, sb = pack_states(m, states)
    for dc in (False, True):
        r = orc.solve_rows(inj, topo, sb, is_dc=dc)
        for k, s in enumerate(states):
            o = solve(m, s, is_dc=dc)
            assert (r["status"][k, 0] == 0) == o.converged, (k, r["status"][k], o.reason)
            assert np.array_equal(r["topo_vect"][k], o.topo_vect)
            assert np.array_equal(r["line_status"][k], o.line_status)
            if not o.converged:
                assert np.all(np.isnan(r["out"][k]))
                continue
            assert r["status"][k, 1] == o.n_iter
            for f in F32_FIELDS:
                ref = getattr(o, f)
                got = r["out"][k][sl[f]]
                assert np.allclose(got, ref, rtol=1e-9, atol=1e-8), (k, f, np.abs(got - ref).max())
            act = ~np.isnan(o.bus_vm)
            assert np.abs(r["bus_vm"][k][act] - o.bus_vm[act]).max() < 1e-12


def test_c_oracle_golden_wcci(load_model, load_npz):
    m = load_model("l2rpn_wcci_2022_dev")
    g = load_npz("l2rpn_wcci_2022_dev.res.npz")
    orc = COracle(m)
    s = LaneState.from_model(m)
    inj, topo, sb = pack_states(m, [s])
    r = orc.solve_rows(inj, topo, sb)
    nl = m.n_powerline
    assert r["status"][0, 0] == 0
    assert np.abs(r["out"][0][:nl] - g["line_p_from_mw"]).max() < 1e-8
    assert np.abs(r["bus_vm"][0][:m.n_sub] - g["bus_vm_pu"]).max() < 1e-11


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "educ_case14_storage", "l2rpn_neurips_2020_track1",
                                  "l2rpn_wcci_2022_dev", "l2rpn_idf_2023", "rte_case118_example", "l2rpn_icaps_2021"])
def test_sparse_solver_is_pinned_to_the_dense_oracle(name, load_model):
    """The sparse LU path of the C oracle (pfo_set_solver(1): minimum-degree ordering, cached symbolic analysis -- what bench.py times as the
    per-config CPU baseline, because a dense elimination is a straw man on 118 substations) against the dense path that is pinned to the
    reference's golden vectors: same status, same iteration count, same integer outputs, results to 1e-10 -- on the reference topology and on
    random outages / bus splits (islanded and non-converging cases included), AC and DC."""
    from oracle import pf_oracle_c
    m = load_model(name)
    orc = COracle(m)
    rng = np.random.default_rng(23)
    states = [LaneState.from_model(m)] + random_states(m, 40 if m.n_sub > 100 else 80, rng)
    inj, topo, sb = pack_states(m, states)
    n_ok = 0
    try:
        for dc in (False, True):
            pf_oracle_c.set_solver(False)
            a = orc.solve_rows(inj, topo, sb, is_dc=dc)
            pf_oracle_c.set_solver(True)
            b = orc.solve_rows(inj, topo, sb, is_dc=dc)
            b2 = orc.solve_rows(inj[::-1], topo[::-1], None if sb is None else sb[::-1], is_dc=dc)         # other order: the cached symbolic analysis is keyed correctly
            assert np.array_equal(a["status"], b["status"]), np.nonzero((a["status"] != b["status"]).any(axis=1))[0]
            assert np.array_equal(a["topo_vect"], b["topo_vect"]) and np.array_equal(a["line_status"], b["line_status"])
            ok = a["status"][:, 0] == 0
            n_ok += int(ok.sum())
            assert np.isnan(b["out"][~ok]).all()
            assert np.abs(a["out"][ok] - b["out"][ok]).max() < 1e-8 * max(1.0, float(m.sn_mva))          # MW / MVAr / kV / A
            assert np.abs(a["bus_vm"][ok] - b["bus_vm"][ok])[~np.isnan(a["bus_vm"][ok])].max() < 1e-10      # pu
            assert np.abs(a["bus_va"][ok] - b["bus_va"][ok])[~np.isnan(a["bus_va"][ok])].max() < 1e-8       # degrees
            assert np.array_equal(b["status"], b2["status"][::-1]) and np.allclose(b["out"], b2["out"][::-1], rtol=0, atol=1e-9, equal_nan=True)
    finally:
        pf_oracle_c.set_solver(False)
    assert n_ok > len(states)          # (most random states converge: the comparison is not vacuous)

"""GPU tests of gpf_ptdf_build_batch (grid2op_amd/csrc/gridpf_ptdf_batch.hpp): the DC matrices of EVERY distinct topology of a batch
factorised on the device (one workgroup per topology class, blocked Gauss-Jordan on the FP64 matrix cores), PTDF / LODF formed there,
and the flows / N-1 screening kernels evaluating every lane against the tables of its own class.

Reference semantics: every DC power flow of the reference factorises B' of the topology the environment has at that moment
(pp.rundcpp(check_connectivity=True), grid2op/Backend/pandaPowerBackend.py:1090; N1Reward per contingency, grid2op/Reward/n1Reward.py:70-99).
Oracle: oracle/pf_oracle.py `ptdf` / `solve(is_dc=True)` / `dc_n1_worst_loading` and the C oracle's DC power flow, per lane, on that
lane's own topology; islanded topologies must come back as such (NaN flows), never as numbers."""
import numpy as np
import pytest

from oracle.pf_oracle import LaneState, dc_bus_injection, dc_n1_worst_loading, ptdf, solve

from helpers import pack_states

pytestmark = pytest.mark.gpu


def _engine(model, n_lanes):
    from grid2op_amd.engine import PowerFlowEngine
    return PowerFlowEngine(model, n_lanes=n_lanes, device=0)


def _pos_sub(m):
    pos_sub = np.empty(m.dim_topo, dtype=np.int64)
    pos_sub[m.line_or_pos_topo_vect] = m.line_or_sub
    pos_sub[m.line_ex_pos_topo_vect] = m.line_ex_sub
    pos_sub[m.gen_pos_topo_vect] = m.gen_sub
    pos_sub[m.load_pos_topo_vect] = m.load_sub
    if m.n_storage:
        pos_sub[m.storage_pos_topo_vect] = m.storage_sub
    return pos_sub


def random_topologies(m, n_topo, rng, max_out=2, max_split=2):
    """`n_topo` distinct topology rows: up to `max_out` lines out and up to `max_split` substations split over two busbars (every other
    element of the substation to busbar 2).  Islanding combinations are KEPT (they must be reported as such)."""
    pos_sub = _pos_sub(m)
    base = m.initial_topo_vect()
    seen, out = set(), []
    big = [s for s in range(m.n_sub) if (pos_sub == s).sum() >= 4]
    while len(out) < n_topo:
        t = base.copy()
        for l in rng.choice(m.n_line, int(rng.integers(0, max_out + 1)), replace=False):
            t[m.line_or_pos_topo_vect[l]] = -1
            t[m.line_ex_pos_topo_vect[l]] = -1
        for s in rng.choice(big, int(rng.integers(0, max_split + 1)), replace=False):
            pos = np.nonzero(pos_sub == s)[0]
            off = int(rng.integers(0, 2))
            sel = pos[off::2]
            t[sel] = np.where(t[sel] >= 1, 2, t[sel])
        key = t.tobytes()
        if key not in seen:
            seen.add(key)
            out.append(t)
    return out


def _states(m, topos, lane_topo, rng):
    base = LaneState.from_model(m)
    states = []
    for k, ti in enumerate(lane_topo):
        st = LaneState.from_model(m)
        st.topo = topos[ti].copy()
        st.load_p = base.load_p * (1 + 0.2 * rng.standard_normal(m.n_load))
        st.gen_p = base.gen_p * (1 + 0.2 * rng.standard_normal(m.n_gen))
        if m.n_storage:
            st.storage_p = rng.uniform(-2, 2, m.n_storage)
        states.append(st)
    return states


@pytest.mark.parametrize("name,n_lanes,n_topo,global_kernel", [("l2rpn_case14_sandbox", 70, 24, False), ("l2rpn_neurips_2020_track1", 90, 40, False),
                                                               ("l2rpn_idf_2023", 96, 48, False), ("educ_case14_storage", 37, 12, False),
                                                               ("l2rpn_idf_2023", 40, 20, True), ("l2rpn_neurips_2020_track1", 33, 16, True)])
def test_tables_and_flows_of_every_class_match_the_oracle(name, n_lanes, n_topo, global_kernel, load_model, monkeypatch):
    """`global_kernel`: the builder for topologies of more than 128 non-reference buses (matrix in global memory, panels in LDS:
    `ptdf_build_kernel`) forced on the small grids too -- the default on these grids is the register-resident kernel."""
    if global_kernel:
        monkeypatch.setenv("GRIDPF_PTDFB_GLOBAL", "1")
    m = load_model(name)
    rng = np.random.default_rng(21)
    topos = random_topologies(m, n_topo, rng)
    lane_topo = np.concatenate([np.arange(n_topo), rng.integers(0, n_topo, n_lanes - n_topo)])   # every topology used, ragged class sizes
    rng.shuffle(lane_topo)
    states = _states(m, topos, lane_topo, rng)
    eng = _engine(m, n_lanes)
    inj, topo, sb = pack_states(m, states)
    eng.set_injections(inj)
    eng.set_topology(topo, sb)
    info = eng.ptdf_build_batch(with_lodf=True)
    assert info["n_classes"] == n_topo and info["kernel_ms"] > 0
    # lanes with the same topology row share a class, different rows never do
    lc = info["lane_class"]
    for a in range(n_lanes):
        same = lane_topo == lane_topo[a]
        assert np.array_equal(lc == lc[a], same), a
    flows = eng.ptdf_flows()
    ref_dc = [solve(m, st, is_dc=True) for st in states]
    n_ok = 0
    for c in range(n_topo):
        k = int(np.nonzero(lc == c)[0][0])
        st = states[k]
        if not ref_dc[k].converged:                                     # islanded / no slack: no tables, NaN flows for every lane of the class
            assert info["class_status"][c] in (2, 3), (c, info["class_status"][c])
            assert np.isnan(flows[lc == c]).all()
            continue
        assert info["class_status"][c] == 0, (c, info["class_status"][c])
        T, L = eng.ptdf_class(c, lodf=True)
        T_ref = ptdf(m, st)
        assert np.abs(T - T_ref).max() < 1e-9, (c, np.abs(T - T_ref).max())
        n_ok += 1
        for kk in np.nonzero(lc == c)[0]:
            ref = T_ref @ dc_bus_injection(m, states[kk])
            tol = 2e-4 + 5e-6 * np.abs(ref)
            assert np.all(np.abs(flows[kk] - ref) <= tol), (c, kk, np.abs(flows[kk] - ref).max())
            assert np.all(np.abs(flows[kk] - ref_dc[kk].p_or) <= 2 * tol), (c, kk)      # = the DC power flow of that lane on its own topology
    assert n_ok >= n_topo // 2                                           # most random topologies are connected; the islanded ones were checked above
    # the device's own per-lane DC solve (kernel S refactorises B' per lane): same flows, same islanding verdicts
    eng.runpf(is_dc=True)
    r = eng.results()
    assert np.array_equal(r.converged, ~np.isnan(flows).any(axis=1))
    ok = r.converged
    assert np.all(np.abs(flows[ok] - r.p_or[ok]) <= 4e-4 + 1e-5 * np.abs(r.p_or[ok]))
    # N-1 screening of every lane against the LODF of its own class vs the brute-force DC N-1 of the oracle (sample)
    w = eng.lodf_screen()
    for kk in rng.choice(np.nonzero(ok)[0], min(10, int(ok.sum())), replace=False):
        ref = dc_n1_worst_loading(m, states[kk], None)
        assert np.array_equal(np.isinf(w[kk]), np.isinf(ref)), kk
        fin = np.isfinite(ref)
        assert np.allclose(w[kk][fin], ref[fin], rtol=2e-5, atol=2e-4), (kk, np.abs(w[kk][fin] - ref[fin]).max())
    assert np.isnan(w[~ok]).all()
    # the single-topology tables come back with gpf_ptdf_build, and equal the class tables of that topology
    k0 = int(np.nonzero(ok)[0][0])
    eng.ptdf_build(lane=k0)
    assert np.abs(eng.ptdf() - eng_class_table(eng, m, states[k0])).max() < 1e-9
    eng.close()


def eng_class_table(eng, m, st):
    return ptdf(m, st)


def test_bench_shape_2048_lanes_256_topologies_on_118_substations(load_model, load_npz):
    """BASELINE.json configs[4] with per-lane topologies: l2rpn_idf_2023, 2 048 lanes, 256 distinct topologies (outages + bus splits), the
    lanes stepped on their chronics rows: (1) PTDF * P_bus of the injection rows the lanes hold vs the C oracle's DC power flow of the
    same rows on each lane's own topology (80 lanes), (2) gpf_ptdf_flows_rows (chronics gather + GEMM per class) = gpf_ptdf_flows of
    the same row, (3) topologies changed by the DEVICE (a line tripped by gpf_disconnect_line after the build) need a rebuild: the
    rebuilt tables follow them."""
    from oracle.pf_oracle_c import COracle
    name = "l2rpn_idf_2023"
    m = load_model(name)
    ch = load_npz(f"{name}.chronics.npz")
    rng = np.random.default_rng(3)
    B, n_topo = 2048, 256
    topos = random_topologies(m, n_topo, rng)
    lane_topo = np.concatenate([np.arange(n_topo), rng.integers(0, n_topo, B - n_topo)])
    rng.shuffle(lane_topo)
    eng = _engine(m, B)
    prod_v = ch["prod_v"] if "prod_v" in ch else np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], prod_v)
    eng.upload_chronics(tab)
    Tn = tab.shape[0]
    off = (7 * np.arange(B)) % Tn
    sc = (1.0 + 0.05 * rng.standard_normal((B, 2 * m.n_load))).astype(np.float32)
    eng.set_lane_chronics(lane_offset=off.astype(np.int32), lane_scale=sc)
    topo = np.stack([topos[i] for i in lane_topo]).astype(np.int32)
    eng.set_topology(topo)
    eng.step(5, n_steps=1, rebalance=1.02)                  # leaves the injection rows of chronics row 5 (jittered, rebalanced) in the lanes
    info = eng.ptdf_build_batch(with_lodf=True)
    assert info["n_classes"] == n_topo
    flows = eng.ptdf_flows()
    inj = eng.get_injections()
    sample = np.sort(rng.choice(B, 80, replace=False))
    tp, sbv = eng.get_topology()
    ref = COracle(m).solve_rows(inj[sample], tp[sample], sbv[sample] if m.n_shunt else None, is_dc=True)
    lay = eng.layout
    p_ref = ref["out"][:, lay.out_p_or:lay.out_p_or + m.n_line]
    st_ref = ref["status"][:, 0]
    for j, k in enumerate(sample):
        if st_ref[j] != 0:
            assert np.isnan(flows[k]).all(), k
            assert info["class_status"][info["lane_class"][k]] != 0
        else:
            assert np.all(np.abs(flows[k] - p_ref[j]) <= 2e-4 + 5e-6 * np.abs(p_ref[j])), (k, np.abs(flows[k] - p_ref[j]).max())
    assert (st_ref == 0).sum() >= 40
    # (2) rows kernel: row 0 of a 3-row launch at t0 = 5 is the row the lanes hold
    rows = eng.ptdf_flows_rows(5, 3, rebalance=1.02)
    assert rows.shape == (3, B, m.n_line)
    okl = ~np.isnan(flows).any(axis=1)
    assert np.array_equal(np.isnan(rows[0]).any(axis=1), ~okl)
    assert np.all(np.abs(rows[0][okl] - flows[okl]) <= 1e-3 + 1e-5 * np.abs(flows[okl]))     # (f32 gather arithmetic vs the stored f64 rows)
    assert not np.array_equal(rows[1][okl], rows[0][okl])
    # (3) a device-side topology change after the build
    k = int(np.nonzero(okl)[0][0])
    on = np.nonzero(tp[k][m.line_or_pos_topo_vect] >= 1)[0]
    eng.disconnect_line(k, int(on[3]))
    info2 = eng.ptdf_build_batch(with_lodf=False)
    assert info2["n_classes"] in (n_topo, n_topo + 1)
    f2 = eng.ptdf_flows()
    if not np.isnan(f2[k]).any():
        assert abs(f2[k][on[3]]) == 0.0 and not np.allclose(f2[k], flows[k], atol=1e-3)
    others = np.setdiff1d(np.nonzero(okl)[0], [k])
    assert np.array_equal(f2[others], flows[others])
    eng.close()


def test_rebuilds_reuse_cached_descriptors_and_follow_changed_topologies(load_model, monkeypatch):
    """The host keeps the descriptor of every topology row it has built (hash + the row, compared on a hit): a rebuild after SOME lanes
    changed -- to topologies seen before, to new ones, and back -- must give, class by class and bit for bit, the tables a build
    without the cache gives (`GRIDPF_PTDFB_NO_CACHE`), with the classes numbered by first appearance in the lane range."""
    name = "l2rpn_neurips_2020_track1"
    m = load_model(name)
    rng = np.random.default_rng(21)
    B, n_topo = 96, 30
    topos = random_topologies(m, n_topo + 12, rng)
    eng = _engine(m, B)

    def tables(lane_topo):
        eng.set_topology(np.stack([topos[i] for i in lane_topo]).astype(np.int32))
        info = eng.ptdf_build_batch(with_lodf=True)
        tabs = [eng.ptdf_class(c, lodf=True) if info["class_status"][c] == 0 else None for c in range(info["n_classes"])]
        return info, tabs

    def same(a, b):
        (ia, ta), (ib, tb) = a, b
        assert ia["n_classes"] == ib["n_classes"] and np.array_equal(ia["lane_class"], ib["lane_class"])
        assert np.array_equal(ia["class_status"], ib["class_status"]) and np.array_equal(ia["class_n"], ib["class_n"])
        for x, y in zip(ta, tb):
            assert (x is None) == (y is None)
            if x is not None:
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1], equal_nan=True)

    lt0 = rng.integers(0, n_topo, B)
    lt1 = lt0.copy()
    lt1[rng.choice(B, 20, replace=False)] = rng.integers(n_topo, n_topo + 12, 20)      # 20 lanes move to topologies never built
    lt1[rng.choice(B, 10, replace=False)] = rng.integers(0, n_topo, 10)                # 10 to ones seen before
    runs = {}
    for label, env in (("cached", None), ("uncached", "1")):
        if env:
            monkeypatch.setenv("GRIDPF_PTDFB_NO_CACHE", env)
        runs[label] = [tables(lt0), tables(lt1), tables(lt0)]
    for a, b in zip(runs["cached"], runs["uncached"]):
        same(a, b)
    same(runs["cached"][0], runs["cached"][2])                                       # back to the first assignment: the first tables
    assert runs["cached"][1][0]["n_classes"] != runs["cached"][0][0]["n_classes"] or not np.array_equal(runs["cached"][1][0]["lane_class"], runs["cached"][0][0]["lane_class"])
    eng.close()


@pytest.mark.parametrize("name,B,n_topo", [("l2rpn_neurips_2020_track1", 96, 30), ("l2rpn_wcci_2022_dev", 300, 40), ("l2rpn_case14_sandbox", 17, 6)])
def test_device_grouping_equals_the_host_path(name, B, n_topo, load_model, monkeypatch):
    """Round 6: which lanes share a topology, and the descriptor of every distinct topology (live buses, compact numbering, connectivity
    verdict, rows of B'), are computed ON THE DEVICE (gridpf_ptdf_group.hpp: hash, one-workgroup sort, verification, one workgroup per
    class) instead of on the host from rows read back over PCIe.  Same partition of the lanes (class numbers are in hash order there, in
    order of first appearance on the host), and per class the same status, dimension, PTDF and LODF bit for bit -- the factorisation
    kernel gets identical descriptors; flows of every lane equal."""
    m = load_model(name)
    rng = np.random.default_rng(5)
    topos = random_topologies(m, n_topo, rng)
    lane_topo = np.concatenate([np.arange(n_topo), rng.integers(0, n_topo, B - n_topo)])
    rng.shuffle(lane_topo)
    eng = _engine(m, B)
    eng.set_topology(np.stack([topos[i] for i in lane_topo]).astype(np.int32))
    got = {}
    for label in ("device", "host"):
        if label == "host":
            monkeypatch.setenv("GRIDPF_PTDFB_HOST", "1")
        info = eng.ptdf_build_batch(with_lodf=True)
        tabs = [eng.ptdf_class(c, lodf=True) if info["class_status"][c] == 0 else None for c in range(info["n_classes"])]
        got[label] = (info, tabs, eng.ptdf_flows())
    monkeypatch.delenv("GRIDPF_PTDFB_HOST")
    (i_d, t_d, f_d), (i_h, t_h, f_h) = got["device"], got["host"]
    assert i_d["n_classes"] == i_h["n_classes"] == len(set(lane_topo.tolist()))
    # same partition: the map device class -> host class is a bijection consistent on every lane
    d2h = {}
    for cd, chh in zip(i_d["lane_class"].tolist(), i_h["lane_class"].tolist()):
        assert d2h.setdefault(cd, chh) == chh
    assert len(set(d2h.values())) == len(d2h) == i_d["n_classes"]
    for cd, chh in d2h.items():
        assert i_d["class_status"][cd] == i_h["class_status"][chh] and i_d["class_n"][cd] == i_h["class_n"][chh]
        assert (t_d[cd] is None) == (t_h[chh] is None)
        if t_d[cd] is not None:
            assert np.array_equal(t_d[cd][0], t_h[chh][0]) and np.array_equal(t_d[cd][1], t_h[chh][1], equal_nan=True)
    assert np.array_equal(f_d, f_h, equal_nan=True)
    eng.close()

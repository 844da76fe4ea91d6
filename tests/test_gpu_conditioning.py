"""GPU tests of the numerically hard corners: the block-sparse LU pivots on the bus blocks only
(no pivoting between blocks, `fast_rcp` reciprocals) whereas the oracle -- like pandapower's sparse LU -- uses partial
pivoting, so lanes near voltage collapse, lines with ~zero reactance, generators behind tiny reactances and heavily loaded
radial feeders are where the two could part.  Required: the same convergence verdict and the same Newton iteration count as the
oracle on every case, and the same voltages within what the conditioning allows.  Plus bitwise run-to-run reproducibility of
the 118-substation kernel, whose 2 wavefronts per instance accumulate with cross-wave LDS atomics."""
import dataclasses

import numpy as np
import pytest

from oracle.pf_oracle import LaneState, solve

pytestmark = pytest.mark.gpu

from helpers import pack_states  # noqa: E402


def _engine(model, n_lanes):
    from grid2op_amd.engine import PowerFlowEngine
    return PowerFlowEngine(model, n_lanes=n_lanes, device=0)


def _run(m, states, tol_mva=1e-8):
    eng = _engine(m, len(states))
    inj, topo, sb = pack_states(m, states)
    eng.set_injections(inj)
    eng.set_topology(topo, sb)
    eng.runpf(tol_mva=tol_mva)
    r = eng.results()
    eng.close()
    return r


def _check(m, r, states, v_tol, noise_limited_tol_mva=None):
    """verdict and iteration count bit-exact; voltages within v_tol (pu).

    noise_limited_tol_mva: the case sits where the rounding noise of the mismatch (|Y| * eps) reaches the 1e-8 MVA tolerance, so
    whether the last iterate passes the test is decided by the summation order of either implementation.  A lane on which the
    two verdicts differ is then re-run by BOTH at this looser tolerance, where both must converge to the same voltages."""
    n_conv = 0
    relaxed = None
    for k, s in enumerate(states):
        o = solve(m, s)
        if noise_limited_tol_mva is not None and bool(r.converged[k]) != bool(o.converged):
            if relaxed is None:
                relaxed = _run(m, states, tol_mva=noise_limited_tol_mva)
            o2 = solve(m, s, tol_mva=noise_limited_tol_mva)
            assert o2.converged and relaxed.converged[k], (k, o2.reason, relaxed.status[k])
            act = ~np.isnan(o2.bus_vm)
            assert np.abs(relaxed.bus_vm[k][act] - o2.bus_vm[act]).max() < v_tol, k
            if r.converged[k]:                                   # the strict GPU answer is that solution too
                assert np.abs(r.bus_vm[k][act] - o2.bus_vm[act]).max() < v_tol, k
            continue
        assert bool(r.converged[k]) == bool(o.converged), (k, r.status[k], o.reason, o.n_iter)
        if not o.converged:
            assert np.isnan(r.out[k]).all()
            continue
        n_conv += 1
        assert r.n_iter[k] == o.n_iter, (k, r.n_iter[k], o.n_iter)
        act = ~np.isnan(o.bus_vm)
        assert np.abs(r.bus_vm[k][act] - o.bus_vm[act]).max() < v_tol, (k, np.abs(r.bus_vm[k][act] - o.bus_vm[act]).max())
        assert np.allclose(r.p_or[k], o.p_or, rtol=1e-5, atol=2e-3), k
    return n_conv


@pytest.mark.parametrize("name", ["l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev"])
def test_loading_ramp_up_to_voltage_collapse(name, load_model):
    """Loads and generation scaled from 1.0 until (well past) the point where Newton-Raphson stops converging: the lanes
    right below the nose of the PV curve have a nearly singular Jacobian."""
    m = load_model(name)
    states = []
    for f in np.concatenate([np.linspace(1.0, 2.0, 6), np.linspace(2.05, 4.5, 50), [5.0, 6.0, 8.0]]):
        s = LaneState.from_model(m)
        s.load_p, s.load_q, s.gen_p = s.load_p * f, s.load_q * f, s.gen_p * f
        states.append(s)
    r = _run(m, states)
    n_conv = _check(m, r, states, v_tol=1e-7)
    assert n_conv > 3
    if name != "l2rpn_neurips_2020_track1":    # (the 36-substation grid still converges at 8x its stored loading)
        assert n_conv < len(states)            # the ramp crosses the collapse point
    # the most loaded converging lanes needed more iterations than the base case (they are close to the nose)
    assert r.n_iter[r.converged].max() > r.n_iter[0]


@pytest.mark.parametrize("name", ["l2rpn_case14_sandbox", "l2rpn_wcci_2022_dev"])
@pytest.mark.parametrize("x_pu", [1e-4, 1e-5, 1e-6])   # (at 1e-8 pu the mismatch tolerance sits below the rounding noise of
                                                       # the 1e8 pu admittances: the ORACLE itself then stagnates for 9 iterations)
def test_lines_with_almost_zero_reactance(name, x_pu, load_model):
    """A few lines become (almost) ideal connections: their 2x2 blocks are ~1/x larger than the others, which is what breaks
    an LU without pivoting between blocks if the elimination order is unlucky."""
    m0 = load_model(name)
    rng = np.random.default_rng(int(-np.log10(x_pu)))
    states, models = [], []
    lines = rng.choice(m0.n_powerline, size=6, replace=False)
    yff, yft, ytf, ytt, bdc = (a.copy() for a in (m0.br_yff, m0.br_yft, m0.br_ytf, m0.br_ytt, m0.br_bdc))
    for l in lines[:3]:                                    # lossless short lines; one also gets a generator end (PV behind ~0)
        y = 1.0 / (1j * x_pu)
        yff[l], ytt[l], yft[l], ytf[l], bdc[l] = y, y, -y, -y, 1.0 / x_pu
    m = dataclasses.replace(m0, br_yff=yff, br_yft=yft, br_ytf=ytf, br_ytt=ytt, br_bdc=bdc)
    gen_subs = set(m.gen_sub.tolist())
    for l in range(m.n_powerline):                         # plus: a line that ends at a generator substation
        if (int(m.line_or_sub[l]) in gen_subs) != (int(m.line_ex_sub[l]) in gen_subs) and l not in lines[:3]:
            y = 1.0 / (1j * x_pu)
            yff[l], ytt[l], yft[l], ytf[l], bdc[l] = y, y, -y, -y, 1.0 / x_pu
            break
    m = dataclasses.replace(m0, br_yff=yff, br_yft=yft, br_ytf=ytf, br_ytt=ytt, br_bdc=bdc)
    for f in (0.6, 1.0, 1.3, 1.6):
        s = LaneState.from_model(m)
        s.load_p, s.load_q, s.gen_p = s.load_p * f, s.load_q * f, s.gen_p * f
        states.append(s)
    s = LaneState.from_model(m)                             # and with another line open next to a short one
    s.topo[m.line_or_pos_topo_vect[lines[3]]] = -1
    s.topo[m.line_ex_pos_topo_vect[lines[3]]] = -1
    states.append(s)
    r = _run(m, states)
    # voltages across a 1e-8 pu line are determined to ~1e-16 / 1e-8 relative: the tolerance follows the conditioning
    # 1e6 pu admittances: |Y| * eps = 2e-10 pu is above the 1e-10 pu tolerance -> the verdict of a lane may be a matter of rounding
    _check(m, r, states, v_tol=max(1e-9, 1e-14 / x_pu), noise_limited_tol_mva=1e-6 if x_pu <= 1e-6 else None)


def test_heavily_loaded_radial_feeder(load_model):
    """case14 opened into a tree (every loop broken) and loaded up: long radial paths, low voltages at the ends."""
    m = load_model("l2rpn_case14_sandbox")
    # spanning tree by union-find over the lines in file order; every other line is opened
    parent = list(range(m.n_sub))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    off = []
    for l in range(m.n_line):
        a, b = find(int(m.line_or_sub[l])), find(int(m.line_ex_sub[l]))
        if a == b:
            off.append(l)
        else:
            parent[a] = b
    assert len(off) == m.n_line - (m.n_sub - 1)
    states = []
    for f in np.linspace(0.3, 1.6, 27):
        s = LaneState.from_model(m)
        for l in off:
            s.topo[m.line_or_pos_topo_vect[l]] = -1
            s.topo[m.line_ex_pos_topo_vect[l]] = -1
        s.load_p, s.load_q, s.gen_p = s.load_p * f, s.load_q * f, s.gen_p * f
        states.append(s)
    r = _run(m, states)
    n_conv = _check(m, r, states, v_tol=1e-7)
    assert n_conv >= 3


@pytest.mark.parametrize("name,B,det", [("l2rpn_wcci_2022_dev", 192, True), ("l2rpn_wcci_2022_dev", 192, False),
                                        ("l2rpn_case14_sandbox", 515, False), ("l2rpn_neurips_2020_track1", 130, False)])
def test_bitwise_run_to_run_reproducibility(name, B, det, load_model, load_npz):
    """Repeated launches of the same batch -- and the same lanes at another position in the batch -- must give bit-identical
    results: grid2op's contract is "same seeds -> same episode" (a borderline overflow must not flip a cascade from run to run).
    Single-wavefront instances do because the LDS applies the atomics of ONE wavefront in a fixed order; the 2-wavefront kernel of
    the 118-substation grids does since round 3 by construction: accumulation loops on wavefront 0, wave-closed LU passes, per-wave
    partial sums of S added in a fixed order (gridpf_sparse.hpp).  `det` additionally forces one wavefront per lane."""
    from grid2op_amd.sharding import synthetic_lane_inputs
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    eng = _engine(m, B)
    if det:
        eng.set_deterministic(True)
    bitwise = True
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"])
    eng.upload_chronics(tab)
    off, sc = synthetic_lane_inputs(m.n_load, tab.shape[0], np.arange(B))
    eng.set_lane_chronics(lane_offset=off, lane_scale=sc)

    def same(a, b, f32=False):
        if bitwise:
            return np.array_equal(a, b, equal_nan=True)
        tol = 3e-7 if f32 else 1e-12           # (float32 outputs: a last-bit difference of the float64 value can flip one ulp)
        return np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a, b, rtol=tol, atol=tol, equal_nan=True)
    ref = None
    for rep in range(6):
        eng.step(4, rebalance=1.02)
        r = eng.results()
        assert r.converged.all()
        cur = (r.bus_vm.copy(), r.bus_va.copy(), r.out.astype(np.float64))
        if ref is None:
            ref = cur
        else:
            for q, (a, b) in enumerate(zip(ref, cur)):
                assert same(a, b, f32=q == 2), rep
    # same lanes, reversed order in the batch
    eng.set_lane_chronics(lane_offset=off[::-1].copy(), lane_scale=sc[::-1].copy())
    eng.step(4, rebalance=1.02)
    r = eng.results()
    assert same(r.bus_vm[::-1], ref[0]) and same(r.out[::-1].astype(np.float64), ref[2], f32=True)
    # a multi-step launch (topology tables kept on chip) against single steps
    eng.step(2, rebalance=1.02, n_steps=3)
    r = eng.results()
    assert np.allclose(r.bus_vm[::-1], ref[0], rtol=0, atol=1e-11, equal_nan=True)
    eng.close()

"""Spot check of engine results against the C oracle.  TEST INFRASTRUCTURE (like everything under oracle/): used by the GPU
parity tests and by bench.py's *checker* leg AFTER a timed workload -- never inside a timed region, never by the product.

`check_lanes` takes the inputs the lanes really hold on the device (injection rows as the last step left them, topology rows
incl. tripped lines, shunt buses) and re-solves them with oracle/pf_oracle.c; `check_step` re-derives the injections of the
synthetic DoNothing workload from the chronics table itself (chronics row -> jitter -> rebalancing, float32 as the reference does
it: pandaPowerBackend.py:927) and therefore also covers the device-side chronics gather (K9)."""
from __future__ import annotations

import numpy as np

from .pf_oracle_c import COracle

ABS_TOL = 2e-4          # MW, MVAr, kV, A, deg: the bar on the float32 API outputs (grid2op's dt_float; 1 ulp of a 500 MW flow is 3e-5 MW)
REL_TOL = 5e-6
PU_BAR = 1e-4           # north star: max line-flow error < 1e-4 pu vs the reference -- on the grid's OWN base sn_mva (1e-2 MW on the 100 MVA
                        # grids, 1e-4 MW on l2rpn_neurips / wcci / idf whose sn_mva is 1): checked on the PRE-CAST float64 state (below),
                        # which the float32 API outputs cannot resolve on the sn_mva = 1 grids


def tolerance_text(m):
    """the parity bar of a grid, in its own units (bench.py / test docstrings)"""
    return (f"float32 API outputs: 2e-4 + 5e-6 |x| (MW, MVAr, kV, A, deg) = {ABS_TOL / m.sn_mva:.0e} pu + rel. on this grid's base sn_mva = {m.sn_mva:g} MVA; "
            f"float64 pre-cast line flows (from the engine's bus_vm / bus_va): < {PU_BAR:g} pu = {PU_BAR * m.sn_mva:g} MW; status, n_iter, topo_vect, line_status bit-exact")


def flows_from_bus_voltages(m, bus_vm, bus_va_deg, topo_vect, n_busbar=2):
    """(p_or, q_or, p_ex, q_ex) in MW / MVAr, float64, of one lane from its per-bus voltages (pu, degrees; bus = sub + (local - 1) n_sub)
    and its topo_vect: S_or = V_f conj(yff V_f + yft V_t) sn_mva -- the engine's state BEFORE the cast to float32."""
    V = np.where(np.isnan(bus_vm), 0.0, bus_vm) * np.exp(1j * np.deg2rad(np.where(np.isnan(bus_va_deg), 0.0, bus_va_deg)))
    lo, le = topo_vect[m.line_or_pos_topo_vect], topo_vect[m.line_ex_pos_topo_vect]
    on = (lo >= 1) & (le >= 1)
    bf = m.line_or_sub + (np.maximum(lo, 1) - 1) * m.n_sub
    bt = m.line_ex_sub + (np.maximum(le, 1) - 1) * m.n_sub
    vf, vt = V[bf], V[bt]
    s_or = vf * np.conj(m.br_yff * vf + m.br_yft * vt) * m.sn_mva
    s_ex = vt * np.conj(m.br_ytf * vf + m.br_ytt * vt) * m.sn_mva
    z = np.where(on, 1.0, 0.0)
    return s_or.real * z, s_or.imag * z, s_ex.real * z, s_ex.imag * z


def check_flows_pu(eng, lanes, ref, slices):
    """float64 line flows recomputed from the engine's pre-cast bus voltages vs the oracle's float64 rows `ref` of the same lanes:
    the largest |error| in pu of the grid's base (converged AC lanes only)."""
    m = eng.model
    worst = 0.0
    for i, k in enumerate(np.asarray(lanes)):
        if ref["status"][i, 0] != 0:
            continue
        r = eng.results(int(k), 1, with_bus=True)
        if r.status[0, 0] != 0:
            continue
        got = flows_from_bus_voltages(m, r.bus_vm[0], r.bus_va[0], r.topo_vect[0])
        for name, g_ in zip(("p_or", "q_or", "p_ex", "q_ex"), got):
            worst = max(worst, float(np.abs(g_ - ref["out"][i, slices[name]]).max()) / m.sn_mva)
    return worst


def compare_rows(out, ref_out, status, ref_status, what="lanes"):
    """float32 result rows vs float64 oracle rows: {n, n_converged, status_mismatch, n_iter_mismatch, max_abs_err, max_excess}
    (max_excess <= 0 <=> every entry within ABS_TOL + REL_TOL * |ref|; non-converged lanes must be all-NaN)."""
    out = np.asarray(out, dtype=np.float64)
    ref_out = np.asarray(ref_out, dtype=np.float64)
    conv = ref_status[:, 0] == 0
    res = {"what": what, "n": int(out.shape[0]), "n_converged": int(conv.sum()),
           "status_mismatch": int((status[:, 0] != ref_status[:, 0]).sum())}
    if status.shape[1] > 1 and (status[:, 1] >= 0).all():
        res["n_iter_mismatch"] = int((status[conv, 1] != ref_status[conv, 1]).sum())
    both = conv & (status[:, 0] == 0)
    if both.any():
        err = np.abs(out[both] - ref_out[both])
        res["max_abs_err"] = float(np.nanmax(err))
        res["max_excess"] = float(np.nanmax(err - (ABS_TOL + REL_TOL * np.abs(ref_out[both]))))
        res["nan_in_converged"] = int(np.isnan(out[both]).sum())
    else:
        res["max_abs_err"], res["max_excess"], res["nan_in_converged"] = 0.0, 0.0, 0
    bad = status[:, 0] != 0
    res["non_nan_in_failed"] = int((~np.isnan(out[bad])).sum()) if bad.any() else 0
    res["ok"] = bool(res["status_mismatch"] == 0 and res.get("n_iter_mismatch", 0) == 0 and res["max_excess"] <= 0.0 and
                     res["nan_in_converged"] == 0 and res["non_nan_in_failed"] == 0)
    return res


def check_lanes(eng, lanes, is_dc=False, results=None, n_busbar=2, max_iter=10, tol_mva=1e-8, pu_flows=True):
    """Re-solve the inputs lanes `lanes` hold on the device with the C oracle and compare with the engine's result rows."""
    m = eng.model
    lanes = np.asarray(lanes, dtype=np.int64)
    lo, hi = int(lanes.min()), int(lanes.max()) + 1
    inj = eng.get_injections(lo, hi - lo)[lanes - lo]
    topo, sb = eng.get_topology(lo, hi - lo)
    r = results if results is not None else eng.results(lo, hi - lo, with_bus=False)
    off = 0 if results is not None else lo
    orc = COracle(m, n_busbar)
    ref = orc.solve_rows(inj, topo[lanes - lo], sb[lanes - lo] if m.n_shunt else None, is_dc=is_dc, max_iter=max_iter, tol_mva=tol_mva)
    res = compare_rows(r.out[lanes - off], ref["out"], r.status[lanes - off], ref["status"])
    res["topo_vect_mismatch"] = int((r.topo_vect[lanes - off] != ref["topo_vect"]).any(axis=1).sum())
    res["line_status_mismatch"] = int((r.line_status[lanes - off] != ref["line_status"]).any(axis=1).sum())
    res["ok"] = bool(res["ok"] and res["topo_vect_mismatch"] == 0 and res["line_status_mismatch"] == 0)
    if not is_dc and pu_flows and hasattr(eng, "out_slices"):
        # the north star's bar in pu of THIS grid's base, on the float64 state (the lanes' own rows must be those compared: last step of a launch)
        idx = np.arange(lanes.size)[:: max(1, lanes.size // 32)][:32]
        sub = lanes[idx]
        ref_sub = {"out": ref["out"][idx], "status": ref["status"][idx]}
        res["max_flow_err_pu_f64"] = check_flows_pu(eng, sub, ref_sub, eng.out_slices)
        res["pu_bar"] = PU_BAR
        res["ok"] = bool(res["ok"] and res["max_flow_err_pu_f64"] < PU_BAR)
    return res


def check_step(m, tab, offsets, scale, rebalance, t, lanes, out_rows, status_rows):
    """The synthetic DoNothing step `t` of lanes `lanes` (chronics table `tab`, per-lane offsets / jitter) re-computed by the C
    oracle from the chronics table, vs the engine's result rows `out_rows[len(lanes)]`."""
    orc = COracle(m)
    lanes = np.asarray(lanes, dtype=np.int64)
    ref = np.empty((lanes.size, orc.n_out))
    st = np.empty((lanes.size, 4), np.int32)
    for i, k in enumerate(lanes):
        _, o, s = orc.step_batch(tab, offsets, scale, rebalance, int(t), int(k), 1, want_out=True)
        ref[i], st[i] = o[0], s[0]
    status_rows = np.asarray(status_rows, dtype=np.int32).reshape(lanes.size, -1)       # [n, 4] status rows, or [n, 1] GPF_ST_* only
    return compare_rows(out_rows, ref, status_rows, st[:, :status_rows.shape[1]])

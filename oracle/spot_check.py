"""Spot check of engine results against the C oracle.  TEST INFRASTRUCTURE (like everything under oracle/): used by the GPU
parity tests and by bench.py's *checker* leg AFTER a timed workload -- never inside a timed region, never by the product.

`check_lanes` takes the inputs the lanes really hold on the device (injection rows as the last step left them, topology rows
incl. tripped lines, shunt buses) and re-solves them with oracle/pf_oracle.c; `check_step` re-derives the injections of the
synthetic DoNothing workload from the chronics table itself (chronics row -> jitter -> rebalancing, float32 as the reference does
it: pandaPowerBackend.py:927) and therefore also covers the device-side chronics gather (K9)."""
from __future__ import annotations

import numpy as np

from .pf_oracle_c import COracle

ABS_TOL = 2e-4          # MW, MVAr, kV, A, deg (tests/test_gpu_parity.py: two orders of magnitude inside the 1e-2 MW north-star bar)
REL_TOL = 5e-6


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


def check_lanes(eng, lanes, is_dc=False, results=None, n_busbar=2, max_iter=10, tol_mva=1e-8):
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

#!/usr/bin/env python
"""TEST INFRASTRUCTURE (GPU box): randomised parity sweep, larger than the other tests' samples -- imported by
tests/test_gpu_fuzz_parity.py for a small sweep, run as a script for the large ones (profiles/r05_fuzz_parity.json).

Per grid: B lanes with random topologies (0-3 lines out, 0-2 substations split over two busbars, incl. combinations that island the
grid), random chronics rows and load jitter +-20 %; one AC step and one DC solve (or, `cascade`: three AC steps with the protections on
and the thermal limits x 0.8); EVERY lane of a random sample is re-solved by the C oracle from the inputs the lane holds on the device
(oracle/spot_check.check_lanes: status, n_iter, topo_vect, line_status bit-exact, float32 outputs within 2e-4 + 5e-6 |x|, float64
pre-cast flows in pu of the grid's base).

usage: python tests/fuzz_parity.py [lanes per grid] [checked per grid] [seed] [cascade]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
GOLD = os.path.join(ROOT, "tests", "golden")
GRIDS = ("l2rpn_case14_sandbox", "educ_case14_storage", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev", "l2rpn_idf_2023", "rte_case118_example")
KEEP = ("n", "n_converged", "status_mismatch", "n_iter_mismatch", "nan_in_converged", "non_nan_in_failed", "topo_vect_mismatch",
        "line_status_mismatch", "max_excess", "max_flow_err_pu_f64", "ok")


def fuzz_grid(env, n_lanes, n_check, seed, cascade=False):
    """one grid: {"ac": verdict, "dc": verdict (plain runs only), "distinct_topologies", "seconds"}"""
    from grid2op_amd.engine import PowerFlowEngine
    from grid2op_amd.grid_model import GridModel
    from oracle.spot_check import check_lanes
    from test_gpu_ptdf_batch import random_topologies
    m = GridModel.load_npz(os.path.join(GOLD, f"{env}.grid.npz"))
    chp = os.path.join(GOLD, f"{env}.chronics.npz")
    ch = dict(np.load(chp)) if os.path.exists(chp) else {}
    if "prod_p" not in ch:                                  # no recorded chronics: 8 rows around the grid file's own operating point
        f = (1.0 + 0.1 * np.random.default_rng(99).uniform(-1, 1, (8, 1))).astype(np.float32)    # (and the engine's default limits: the
        ch = {"load_p": f * m.load_p0.astype(np.float32), "load_q": f * m.load_q0.astype(np.float32),     #  environment's belong to its chronics)
              "prod_p": f * m.gen_p0.astype(np.float32)}
    rng = np.random.default_rng(seed)
    eng = PowerFlowEngine(m, n_lanes=n_lanes)
    pv = ch.get("prod_v", np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1)))
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], pv)
    eng.upload_chronics(tab)
    T = tab.shape[0]
    n_topo = min(n_lanes, 512)
    topos = random_topologies(m, n_topo, rng, max_out=3, max_split=2)
    topo = np.stack([topos[i] for i in rng.integers(0, n_topo, n_lanes)]).astype(np.int32)
    topo[: n_lanes // 8] = m.initial_topo_vect()                  # an eighth of the lanes on the reference topology
    eng.set_topology(topo)
    sc = (1.0 + 0.2 * rng.uniform(-1.0, 1.0, (n_lanes, 2 * m.n_load))).astype(np.float32)
    eng.set_lane_chronics(lane_offset=rng.integers(0, T, n_lanes).astype(np.int32), lane_scale=sc)
    t0 = time.time()
    if cascade and "thermal_limits" in ch:
        eng.set_thermal_limits(np.asarray(ch["thermal_limits"]) * 0.8)
    eng.step(int(rng.integers(0, T)), n_steps=3 if cascade else 1, rebalance=1.02, cascade=cascade)
    lanes = np.sort(rng.choice(n_lanes, min(n_check, n_lanes), replace=False))
    ac = check_lanes(eng, lanes)
    res = {"ac": {k: ac.get(k) for k in KEEP}, "distinct_topologies": n_topo}
    if not cascade:
        eng.runpf(0, n_lanes, is_dc=True)
        dc = check_lanes(eng, lanes, is_dc=True, pu_flows=False)
        res["dc"] = {k: dc.get(k) for k in KEEP}
    res["seconds"] = round(time.time() - t0, 1)
    eng.close()
    return res


def _random_engine(env, n_lanes, seed, limits_scale=None):
    """(model, engine, T) with random topologies / chronics rows / load jitter, as fuzz_grid sets them up"""
    from grid2op_amd.engine import PowerFlowEngine
    from grid2op_amd.grid_model import GridModel
    from test_gpu_ptdf_batch import random_topologies
    m = GridModel.load_npz(os.path.join(GOLD, f"{env}.grid.npz"))
    chp = os.path.join(GOLD, f"{env}.chronics.npz")
    ch = dict(np.load(chp)) if os.path.exists(chp) else {}
    if "prod_p" not in ch:
        f = (1.0 + 0.1 * np.random.default_rng(99).uniform(-1, 1, (8, 1))).astype(np.float32)
        ch = {"load_p": f * m.load_p0.astype(np.float32), "load_q": f * m.load_q0.astype(np.float32), "prod_p": f * m.gen_p0.astype(np.float32)}
    rng = np.random.default_rng(seed)
    eng = PowerFlowEngine(m, n_lanes=n_lanes)
    pv = ch.get("prod_v", np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1)))
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], pv)
    eng.upload_chronics(tab)
    n_topo = min(n_lanes, 256)
    topos = random_topologies(m, n_topo, rng, max_out=2, max_split=2)
    topo = np.stack([topos[i] for i in rng.integers(0, n_topo, n_lanes)]).astype(np.int32)
    topo[: n_lanes // 4] = m.initial_topo_vect()
    eng.set_topology(topo)
    sc = (1.0 + 0.2 * rng.uniform(-1.0, 1.0, (n_lanes, 2 * m.n_load))).astype(np.float32)
    eng.set_lane_chronics(lane_offset=rng.integers(0, tab.shape[0], n_lanes).astype(np.int32), lane_scale=sc)
    if limits_scale is not None and "thermal_limits" in ch:
        eng.set_thermal_limits(np.asarray(ch["thermal_limits"]) * limits_scale)
    return m, eng, tab.shape[0]


def fuzz_multistep(env, n_lanes, seed, n_steps=6, auto_reset=False):
    """ONE launch of `n_steps` env steps == `n_steps` single-step launches (integers bit for bit, floats to float32 rounding) on random topologies with the protections on
    (thermal limits x 0.85: lines trip, lanes black out, with `auto_reset` they restart): the result rows, topo_vect, line status, status,
    protection counters, line cooldowns and episode counters of every lane after the last step, and the status of every step."""
    _, a, T = _random_engine(env, n_lanes, seed, limits_scale=0.85)
    _, b, _ = _random_engine(env, n_lanes, seed, limits_scale=0.85)
    kw = dict(rebalance=1.02, cascade=True, auto_reset=auto_reset)
    t0 = int(np.random.default_rng(seed + 1).integers(0, T))
    a.set_trajectory(n_steps, a.TRAJ_RHO)
    a.step(t0, n_steps=n_steps, **kw)
    _, st_a = a.trajectory(n_steps)
    st_b = []
    for j in range(n_steps):
        b.step(t0 + j, n_steps=1, **kw)
        st_b.append(b.results(with_bus=False).status[:, 0].astype(np.int8))
    ra, rb = a.results(), b.results()
    # Integers (status, iteration counts, topology, line status, protection counters, cooldowns, episode counters) bit for bit.  The float
    # rows to float32 rounding: steps >= 2 of a launch start Newton from DC angles obtained with the factors / the static inverse the
    # launch keeps, a single-step launch factorises afresh -- the float64 solutions agree to ~1e-16 and a float32 cast can round the
    # other way (seen on 14 substations in the 1 024-lane sweeps: a handful of entries, 1 ulp).
    so_a, so_b = a.step_outputs(), b.step_outputs()
    same = {"out": bool(np.array_equal(np.isnan(ra.out), np.isnan(rb.out)) and np.allclose(ra.out, rb.out, rtol=2e-6, atol=2e-5, equal_nan=True)),
            "status": np.array_equal(ra.status, rb.status),
            "topo_vect": np.array_equal(ra.topo_vect, rb.topo_vect), "line_status": np.array_equal(ra.line_status, rb.line_status),
            "step_status": np.array_equal(st_a, np.stack(st_b)),
            "step_outputs": bool(np.allclose(so_a[0], so_b[0], rtol=2e-6, atol=1e-6, equal_nan=True) and np.array_equal(so_a[1], so_b[1]) and np.array_equal(so_a[2], so_b[2])),
            "cooldown": np.array_equal(a.cooldown(), b.cooldown()), "episode": all(np.array_equal(x, y) for x, y in zip(a.episode(), b.episode())),
            # (float64 state behind the float32 rows: a multi-step launch starts Newton from the DC factors it keeps across its steps,
            #  a single-step launch factorises afresh -- same solution, last bits of the float64 voltages may differ)
            "bus_vm": bool(np.allclose(ra.bus_vm, rb.bus_vm, rtol=0.0, atol=1e-12, equal_nan=True)),
            "bus_va": bool(np.allclose(ra.bus_va, rb.bus_va, rtol=0.0, atol=1e-10, equal_nan=True))}
    res = {"same": same, "ok": all(same.values()), "converged_last": int(ra.converged.sum()), "tripped_lanes": int((a.step_outputs()[2] >= 0).any(axis=1).sum()),
           "failed_some_step": int((st_a != 0).any(axis=0).sum())}
    a.close(); b.close()
    return res


def fuzz_specialised(env, n_lanes, seed, cache_dir, n_steps=4):
    """The grid-specialised step kernels (gpf_jit_enable) == the shipped ones, bit for bit, on random topologies (topology-class kernels
    of the split lanes included) with the protections on: everything a launch leaves behind, the float64 bus voltages too."""
    _, a, T = _random_engine(env, n_lanes, seed, limits_scale=0.85)
    _, b, _ = _random_engine(env, n_lanes, seed, limits_scale=0.85)
    b.specialize(True, cache_dir=cache_dir, verify=False)
    kw = dict(rebalance=1.02, cascade=True, auto_reset=True)
    t0 = int(np.random.default_rng(seed + 1).integers(0, T))
    for e in (a, b):
        e.set_trajectory(n_steps, e.TRAJ_RHO)
        e.step(t0, n_steps=n_steps, **kw)
        e.step(t0 + n_steps, n_steps=1, **kw)
    ra, rb = a.results(), b.results()
    info = b.specialization()
    same = {"out": np.array_equal(ra.out, rb.out, equal_nan=True), "status": np.array_equal(ra.status, rb.status),
            "topo_vect": np.array_equal(ra.topo_vect, rb.topo_vect), "line_status": np.array_equal(ra.line_status, rb.line_status),
            "step_outputs": all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a.step_outputs(), b.step_outputs())),
            "cooldown": np.array_equal(a.cooldown(), b.cooldown()), "episode": all(np.array_equal(x, y) for x, y in zip(a.episode(), b.episode())),
            "bus_vm": np.array_equal(ra.bus_vm, rb.bus_vm, equal_nan=True), "bus_va": np.array_equal(ra.bus_va, rb.bus_va, equal_nan=True)}
    res = {"same": same, "ok": all(same.values()) and info["failed"] == 0 and info["launches"] > 0 and a.specialization()["launches"] == 0,
           "specialization": info, "converged_last": int(ra.converged.sum())}
    a.close(); b.close()
    return res


def main(argv):
    n_lanes = int(argv[1]) if len(argv) > 1 else 4096
    n_check = int(argv[2]) if len(argv) > 2 else 1500
    seed = int(argv[3]) if len(argv) > 3 else 1
    cascade = len(argv) > 4 and argv[4] == "cascade"
    report = {"lanes_per_grid": n_lanes, "checked_per_grid": n_check, "seed": seed, "cascade": cascade, "grids": {}}
    for k, env in enumerate(GRIDS):
        report["grids"][env] = fuzz_grid(env, n_lanes, n_check, seed + k, cascade)
        print(env, json.dumps(report["grids"][env]), flush=True)
    report["all_ok"] = all(v["ok"] for g in report["grids"].values() for key, v in g.items() if key in ("ac", "dc"))
    print(json.dumps(report))
    return 0 if report["all_ok"] else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))

#!/usr/bin/env python
"""Developer tool (GPU box): randomised parity sweep, larger than the tests' samples.  Per grid: B lanes with random topologies (0-3 lines
out, 0-2 substations split over two busbars, incl. combinations that island the grid), random chronics rows and load jitter +-20 %;
one AC step and one DC solve; EVERY lane of a random sample re-solved by the C oracle from the inputs the lane holds on the device
(oracle/spot_check.check_lanes: status, n_iter, topo_vect, line_status bit-exact, float32 outputs within 2e-4 + 5e-6 |x|, float64
pre-cast flows in pu).  usage: python tools/fuzz_parity.py [lanes per grid] [checked per grid] [seed] [cascade]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
from grid2op_amd.grid_model import GridModel  # noqa: E402
from grid2op_amd.engine import PowerFlowEngine  # noqa: E402
from oracle.spot_check import check_lanes  # noqa: E402  (checker: developer tool, like the tests)
from test_gpu_ptdf_batch import random_topologies  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N_CHECK = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 1
CASCADE = len(sys.argv) > 4 and sys.argv[4] == "cascade"      # thermal limits x 0.8, protections on: the lanes' FINAL state is checked
GOLD = os.path.join(ROOT, "tests", "golden")
report = {"lanes_per_grid": B, "checked_per_grid": N_CHECK, "seed": SEED, "cascade": CASCADE, "grids": {}}
for env in ("l2rpn_case14_sandbox", "educ_case14_storage", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev", "l2rpn_idf_2023", "rte_case118_example"):
    m = GridModel.load_npz(os.path.join(GOLD, f"{env}.grid.npz"))
    chp = os.path.join(GOLD, f"{env}.chronics.npz")
    ch = dict(np.load(chp)) if os.path.exists(chp) else {}
    if "prod_p" not in ch:                                  # no recorded chronics: 8 rows around the grid file's own operating point
        r0 = np.random.default_rng(99)
        f = (1.0 + 0.1 * r0.uniform(-1, 1, (8, 1))).astype(np.float32)
        ch = {"load_p": f * m.load_p0.astype(np.float32), "load_q": f * m.load_q0.astype(np.float32), "prod_p": f * m.gen_p0.astype(np.float32)}
    rng = np.random.default_rng(SEED + len(report["grids"]))
    eng = PowerFlowEngine(m, n_lanes=B)
    pv = ch.get("prod_v", np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1)))
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], pv)
    eng.upload_chronics(tab)
    T = tab.shape[0]
    n_topo = min(B, 512)
    topos = random_topologies(m, n_topo, rng, max_out=3, max_split=2)
    lane_topo = rng.integers(0, n_topo, B)
    topo = np.stack([topos[i] for i in lane_topo]).astype(np.int32)
    topo[: B // 8] = m.initial_topo_vect()                         # an eighth of the lanes on the reference topology
    eng.set_topology(topo)
    sc = (1.0 + 0.2 * rng.uniform(-1.0, 1.0, (B, 2 * m.n_load))).astype(np.float32)
    eng.set_lane_chronics(lane_offset=rng.integers(0, T, B).astype(np.int32), lane_scale=sc)
    t0 = time.time()
    if CASCADE and "thermal_limits" in ch:
        eng.set_thermal_limits(np.asarray(ch["thermal_limits"]) * 0.8)
    eng.step(int(rng.integers(0, T)), n_steps=3 if CASCADE else 1, rebalance=1.02, cascade=CASCADE)
    lanes = np.sort(rng.choice(B, min(N_CHECK, B), replace=False))
    ac = check_lanes(eng, lanes)
    eng.runpf(0, B, is_dc=True)
    dc = check_lanes(eng, lanes, is_dc=True, pu_flows=False)
    keep = ("n", "n_converged", "status_mismatch", "n_iter_mismatch", "nan_in_converged", "non_nan_in_failed", "topo_vect_mismatch",
            "line_status_mismatch", "max_abs_err_vs_oracle", "max_excess", "max_flow_err_pu_f64", "ok")
    report["grids"][env] = {"ac": {k: ac.get(k) for k in keep}, "dc": {k: dc.get(k) for k in keep}, "distinct_topologies": n_topo,
                            "seconds": round(time.time() - t0, 1)}
    print(env, json.dumps(report["grids"][env]), flush=True)
    eng.close()
report["all_ok"] = all(g["ac"]["ok"] and g["dc"]["ok"] for g in report["grids"].values())
print(json.dumps(report))

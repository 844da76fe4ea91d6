"""Developer tool (GPU box): the DC sensitivity path vs batch size.  usage: python tools/ptdf_bench.py [env] [B1,B2,...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid2op_amd.grid_model import GridModel
from grid2op_amd.engine import PowerFlowEngine
env = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_idf_2023"
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2048, 16384, 131072]
m = GridModel.load_npz(f"tests/golden/{env}.grid.npz")
for B in sizes:
    eng = PowerFlowEngine(m, n_lanes=B)
    inj = np.tile(eng.get_injections(0, 1), (B, 1))
    inj *= 1 + 0.02 * np.random.default_rng(0).standard_normal(inj.shape)
    eng.set_injections(inj)
    eng.ptdf_build(0)
    for _ in range(5): eng.ptdf_flows(fetch=False)
    eng.sync()
    reps = max(10, 200 * 2048 // B)
    eng.set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(reps): eng.ptdf_flows(fetch=False)
    eng.sync(); el = (time.perf_counter() - t0) / reps
    k_ms, n_l = eng.kernel_time(); eng.set_profiling(0)
    nb = int(np.isfinite(eng.ptdf()).any(axis=0).sum())
    nb_pad, lp = (m.n_sub + 3) // 4 * 4, (m.n_line + 15) // 16 * 16
    us = k_ms / n_l * 1e3
    t0 = time.perf_counter(); w = eng.lodf_screen(); lod = time.perf_counter() - t0
    t0 = time.perf_counter(); w = eng.lodf_screen(); lod = min(lod, time.perf_counter() - t0)
    print(json.dumps({"env": env, "lanes": B, "us_per_batch": us, "host_us": el * 1e6, "dc_flows_per_s": B / (us * 1e-6), "tflops": 2.0 * B * nb_pad * lp / (us * 1e-6) / 1e12,
                      "hbm_gbs": (8.0 * B * eng.n_inj + 4.0 * B * lp) / (us * 1e-6) / 1e9, "lodf_ms_incl_copy": lod * 1e3}), flush=True)
    eng.close()

"""Developer tool (GPU box): throughput of the batched env step vs lane count.  usage: python tools/batch_sweep.py [env] [B1,B2,...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid2op_amd.grid_model import GridModel
from grid2op_amd.engine import PowerFlowEngine
env = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_case14_sandbox"
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 2048, 4096, 8192, 16384, 32768, 65536]
m = GridModel.load_npz(f"tests/golden/{env}.grid.npz")
ch = dict(np.load(f"tests/golden/{env}.chronics.npz"))
if "prod_v" not in ch:
    ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
out = []
for B in sizes:
    eng = PowerFlowEngine(m, n_lanes=B)
    eng.upload_chronics(eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]))
    T = ch["load_p"].shape[0]
    rng = np.random.default_rng(0)
    eng.set_lane_chronics(lane_offset=(7 * np.arange(B)) % T, lane_scale=(1 + 0.05 * rng.standard_normal((B, 2 * m.n_load))).astype(np.float32))
    for t in range(300): eng.step(t, rebalance=1.02)
    eng.sync()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        N = 200
        for t in range(N): eng.step(t, rebalance=1.02)
        eng.sync(); best = min(best, (time.perf_counter() - t0) / N)
    r = eng.results(0, min(B, 1024))
    out.append({"env": env, "lanes": B, "us_per_step": best * 1e6, "steps_per_s": B / best, "frac_converged": float(r.converged.mean())})
    print(json.dumps(out[-1]), flush=True)
    eng.close()

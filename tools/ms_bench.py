"""Developer tool (GPU box): env steps/s vs steps per launch.  usage: python tools/ms_bench.py [env] [lanes] [n1,n2,...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid2op_amd.grid_model import GridModel
from grid2op_amd.engine import PowerFlowEngine
env = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_case14_sandbox"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ns = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16, 32, 64]
kw = dict(rebalance=1.02, cascade=bool(int(os.environ.get("CASCADE", "0"))), auto_reset=bool(int(os.environ.get("AUTO_RESET", "0"))))
m = GridModel.load_npz(f"tests/golden/{env}.grid.npz")
ch = dict(np.load(f"tests/golden/{env}.chronics.npz"))
if "prod_v" not in ch:
    ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
eng = PowerFlowEngine(m, n_lanes=B)
eng.upload_chronics(eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]))
T = ch["load_p"].shape[0]
rng = np.random.default_rng(0)
eng.set_lane_chronics(lane_offset=(7 * np.arange(B)) % T, lane_scale=(1 + 0.05 * rng.standard_normal((B, 2 * m.n_load))).astype(np.float32))
if "thermal_limits" in ch: eng.set_thermal_limits(ch["thermal_limits"] * float(os.environ.get("LIMIT_SCALE", "1")))
for t in range(300): eng.step(t, **kw)
eng.sync()
for n in ns:
    K = max(1, 1024 // n)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(K): eng.step(k * n, n_steps=n, **kw)
        eng.sync(); best = min(best, (time.perf_counter() - t0) / (K * n))
    r = eng.results(0, min(B, 512))
    print(json.dumps({"env": env, "lanes": B, "steps_per_launch": n, "us_per_step": best * 1e6, "steps_per_s": B / best, "conv": float(r.converged.mean())}), flush=True)

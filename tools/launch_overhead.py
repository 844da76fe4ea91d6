"""Developer tool: host enqueue time and wall time per batched step with event timing off / window / off (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from grid2op_amd.grid_model import GridModel
from grid2op_amd.engine import PowerFlowEngine
env = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_case14_sandbox"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = GridModel.load_npz(f"tests/golden/{env}.grid.npz")
ch = dict(np.load(f"tests/golden/{env}.chronics.npz"))
eng = PowerFlowEngine(m, n_lanes=B)
eng.upload_chronics(eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]))
eng.set_lane_chronics(lane_offset=(7 * np.arange(B)) % ch["load_p"].shape[0])
for prof in (False, True, False):
    eng.set_profiling(prof)
    for t in range(20): eng.step(t, rebalance=1.02)
    eng.sync()
    t0 = time.perf_counter()
    N = 300
    for t in range(N): eng.step(t, rebalance=1.02)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print(f"profiling={prof}: host enqueue {1e6*(t1-t0)/N:.2f} us/step, wall {1e6*(t2-t0)/N:.2f} us/step", eng.kernel_time() if prof else "")

"""Developer tool (GPU box): latency of ONE `HipBackend.runpf` (the drop-in single-environment path: push the lane state, solve,
read everything back) on top of the grid2op stand-in.  usage: python tools/facade_latency.py [grid] [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests", "grid2op_stub"), os.path.join(ROOT, "tests"), ROOT]
from grid2op_amd.backend import HipBackend
grid = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_case14_sandbox"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
bk = HipBackend()
bk.set_env_name("lat")
bk.load_grid(os.path.join(ROOT, "tests", "golden", f"{grid}.grid.npz"))
bk.assert_grid_correct()
for _ in range(50):
    ok, exc = bk.runpf()
assert ok
t0 = time.perf_counter()
for _ in range(n):
    bk.runpf()
dt = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    bk.runpf(is_dc=True)
dt_dc = (time.perf_counter() - t0) / n
cp = bk.copy()
t0 = time.perf_counter()
for _ in range(200):
    c2 = bk.copy(); c2.close()
dt_copy = (time.perf_counter() - t0) / 200
print(f"{grid}: HipBackend.runpf AC {dt*1e6:.1f} us, DC {dt_dc*1e6:.1f} us, copy()+close() {dt_copy*1e6:.1f} us, a_or[0] = {bk.lines_or_info()[3][0]:.3f}")
bk.close(); cp.close()

#!/usr/bin/env python
"""Developer tool: per-phase cycle counts of the small-grid step kernel (needs a -DGPF_TIMING build of the library).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGPF_TIMING grid2op_amd/csrc/gridpf_capi.hip -o /tmp/libgridpf_timing.so
    GRIDPF_LIB=/tmp/libgridpf_timing.so python tools/phase_timing.py [batch]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grid2op_amd.grid_model import GridModel  # noqa: E402
from grid2op_amd.engine import PowerFlowEngine  # noqa: E402
from grid2op_amd import _capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = sys.argv[2] if len(sys.argv) > 2 else "l2rpn_case14_sandbox"
gold = os.path.join(ROOT, "tests", "golden")
m = GridModel.load_npz(os.path.join(gold, f"{env}.grid.npz"))
ch = dict(np.load(os.path.join(gold, f"{env}.chronics.npz")))
eng = PowerFlowEngine(m, n_lanes=B)
eng.upload_chronics(eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch.get("prod_v", np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1)))))
eng.set_lane_chronics(lane_offset=(7 * np.arange(B)) % ch["load_p"].shape[0])
if os.environ.get("PT_JIT"):      # the grid-specialised kernels with the stamps: GRIDPF_JIT_FLAGS=-DGPF_TIMING PT_JIT=1 (the library must be the timing build too)
    print("specialised:", eng.specialize(verify=False))
NS = int(os.environ.get("NSTEPS", "1"))
for t in range(5):
    eng.step(t * NS, rebalance=1.02, n_steps=NS)
eng.sync()
L = _capi.lib()
ROW = 88
buf = np.zeros((B, ROW))
L.gpf_debug_read_work.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64]
assert L.gpf_debug_read_work(eng._h, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size) == 0
SPARSE = True
if SPARSE:
    names_s = {8: "kernel start", 0: "K9 chronics gather", 1: "K1 + bus types (atomics)", 2: "connectivity", 3: "Ybus + DC assembly",
               4: "DC block-LU solve", 5: "Newton loop total", 6: "results", 9: "cascade check / back in step kernel", 15: "rho + counters"}
    order_s = [8, 0, 1, 2, 3, 4, 5, 6, 9, 15]
    med = np.median(buf, axis=0)
    print(f"[kernel S] {env} batch {B}, {NS} step(s) per launch: median cycle counts per phase (LAST step of the launch)")
    print(f"  launch prologue (carve, static staging, lane constants) {med[19] - med[8]:10.0f}" if NS == 1 else
          f"  whole launch {med[15] - med[8]:.0f} cycles = {(med[15] - med[8]) / NS:.0f} per step")
    m8 = float(med[8])
    med[8] = med[19]
    names_s[9] = "cascade check"
    names_s[7] = "rho + counters + episode"
    order_s = [8, 0, 1, 2, 3, 4, 5, 6, 9, 7]
    prev = med[8]
    for k in order_s[1:]:
        print(f"  {names_s[k]:45s} {med[k] - prev:10.0f}")
        prev = med[k]
    print(f"  {'TOTAL (one step)':45s} {med[7] - med[8]:10.0f}")
    if med[39] > 0 and NS == 1:
        print(f"  prologue detail: before the kept-state block {med[34] - m8:.0f}, loads + key comparison {med[36] - med[34]:.0f}, state -> LDS {med[37] - med[36]:.0f}, "
              f"block verdict {med[38] - med[37]:.0f}, barrier {med[39] - med[38]:.0f}, rest {med[19] - med[39]:.0f}")
    print(f"  DC block-LU: elimination levels + scaling {med[20]:.0f}, back substitution {med[21]:.0f} cycles")
    print(f"  first Newton iteration: initial sincos {med[10] - med[4]:.0f}, Jacobian blocks + S {med[11] - med[10]:.0f}, "
          f"diag + mismatch + test {med[12] - med[11]:.0f}, block LU {med[13] - med[12]:.0f}, update + sincos {med[14] - med[13]:.0f}")
    print(f"  K9 detail: topo row + first loads {med[16] - med[8]:.0f}, load rows + sums {med[17] - med[16]:.0f}, reductions {med[18] - med[17]:.0f}, "
          f"gens + stores {med[0] - med[18]:.0f}")
    print(f"  start of the Newton loop (register-resident path): constants of the solve {med[20] - med[4]:.0f}, fused DC start {med[21] - med[20]:.0f}, "
          f"sincos + first stores {med[10] - med[21]:.0f}")
    print(f"  K9 tail: generators {med[29] - med[18]:.0f}, storage / shunt sums + boundary {med[30] - med[29]:.0f}, cascade-loop set-up {med[31] - med[30]:.0f}, "
          f"entry of the solve {med[0] - med[31]:.0f}")
    print(f"  K1 detail: topo row -> LDS {med[27] - med[0]:.0f}, element loops (atomics) {med[28] - med[27]:.0f}, types / counts {med[1] - med[28]:.0f}")
    print(f"  results detail: line flows {med[22] - med[5]:.0f}, loads/storages/shunts {med[23] - med[22]:.0f}, gen accumulate {med[24] - med[23]:.0f}, "
          f"gen write {med[25] - med[24]:.0f}, topo_out {med[26] - med[25]:.0f}, bus V {med[6] - med[26]:.0f}")
    pt = buf[:, 40:]
    ok = pt[:, 0] > 0
    if ok.any():
        d = np.diff(pt[ok], axis=1)
        d = np.where((pt[ok][:, 1:] > 0) & (d > 0), d, np.nan)
        print("  first factorisation, cycles per pass (forward then back; median / min / max over lanes):")
        for k in range(min(d.shape[1], 20)):
            if np.isfinite(d[:, k]).any():
                print(f"    pass {k:2d}: {np.nanmedian(d[:, k]):7.0f} {np.nanmin(d[:, k]):7.0f} {np.nanmax(d[:, k]):7.0f}")
    print(f"  LU of iteration 1: forward {med[32]:.0f}, back {med[33]:.0f}")
    sys.exit(0)

"""Developer tool (round 5): where the host-side microseconds of one timed bench window go (the wall clock of a K = 20-step window is one 0.4 ms
launch + this).  python tools/host_overhead.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    stub_engine = False; share_device = False; no_jit = False; profile = False; steps_per_launch = 20; last_obs_only = False
    dist_backend = "nccl"; no_oracle_check = True; dump_aot = None


def main():
    import torch
    ctx = bench.Ctx(A())
    m, ch = bench.load_env("l2rpn_case14_sandbox")
    eng, T, _ = bench.setup_engine(ctx, m, ch, 4096)
    eng.set_trajectory(20, eng.TRAJ_OBS)
    kw = dict(rebalance=1.02, cascade=False)
    for _ in range(20):
        eng.step(0, n_steps=20, **kw)
    eng.sync()
    t = {k: [] for k in ("step_call", "prof3", "eng_sync", "torch_sync", "total", "prof1", "barrier_free_total")}
    for rep in range(200):
        eng.sync(); torch.cuda.synchronize()
        a = time.perf_counter(); eng.set_profiling(1); b0 = time.perf_counter()
        w0 = time.perf_counter()
        eng.step(rep, n_steps=20, **kw)
        c = time.perf_counter()
        eng.set_profiling(3)
        d = time.perf_counter()
        eng.sync()
        e = time.perf_counter()
        torch.cuda.synchronize()
        f = time.perf_counter()
        ms, n = eng.kernel_time()
        eng.set_profiling(0)
        t["prof1"].append(b0 - a); t["step_call"].append(c - w0); t["prof3"].append(d - c); t["eng_sync"].append(e - d); t["torch_sync"].append(f - e)
        t["total"].append(f - w0); t["barrier_free_total"].append((f - w0) - ms * 1e-3)
    for k, v in t.items():
        v = np.array(v[20:]) * 1e6
        print(f"{k:>20}: median {np.median(v):8.1f} us   min {v.min():8.1f}   p90 {np.percentile(v, 90):8.1f}")
    eng.close()


main()

#!/usr/bin/env python
"""Developer tool (GPU box): gpf_ptdf_flows_rows at the BASELINE configs[4] size, for A/B of kernel variants (GRIDPF_PTDF_MT)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench

class A: no_jit=True; profile=False; stub_engine=False; share_device=False; dist_backend="nccl"; no_oracle_check=False; steps_per_launch=16; last_obs_only=False
ctx = bench.Ctx(A())
m, ch = bench.load_env("l2rpn_idf_2023")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng, T, l0 = bench.setup_engine(ctx, m, ch, B)
eng.step(0, n_steps=1, rebalance=1.02)
eng.ptdf_build(0)
for n_rows in (1, 4, 16, 64):
    rec = bench.workload_ptdf_rows(ctx, eng, m, B, 5, n_rows=n_rows, reps=30)
    print(n_rows, "rows: %.1f us/launch  %.1f TFLOP/s  frac %.3f  ok %s" % (rec["us_per_launch"], rec["roofline"]["achieved"], rec["roofline"]["frac"], rec["oracle_check"]))

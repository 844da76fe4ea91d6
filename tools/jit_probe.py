"""Developer tool (GPU box): grid-specialised (run-time compiled) step kernels vs the shipped ones -- bit-identity of a multi-step
launch with the observation trajectory, compile time, and the generated headers of the bench grids (gpurun_out/jit/).
usage: python tools/jit_probe.py [grid ...]"""
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from grid2op_amd.engine import PowerFlowEngine  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "jit")
os.makedirs(OUT, exist_ok=True)


class A:
    share_device = False
    dist_backend = "gloo"
    stub_engine = False
    no_jit = True


def run(name, B, jit):
    ctx = bench.Ctx(A())
    m, ch = bench.load_env(name)
    eng, T, _ = bench.setup_engine(ctx, m, ch, B)
    info = None
    if jit:
        eng.specialize(True, cache_dir=os.path.join(OUT, "cache"), verify=False)
    eng.set_trajectory(16, eng.TRAJ_OBS)
    kw = dict(rebalance=1.02, cascade=False, auto_reset=True)
    t0 = time.time()
    eng.step(0, n_steps=16, **kw)
    eng.sync()
    first = time.time() - t0
    for k in range(4):
        eng.step(16 * (k + 1), n_steps=16, **kw)
    eng.sync()
    t1 = time.time()
    for k in range(100):
        eng.step(16 * (k + 5), n_steps=16, **kw)
    eng.sync()
    dt = (time.time() - t1) / 100
    res = eng.results()
    tr = eng.trajectory_obs(16)
    if jit:
        info = eng.specialization()
        open(os.path.join(OUT, f"{name}.h"), "w").write(eng.specialization_header())
    out = {"out": res.out, "topo": res.topo_vect, "ls": res.line_status, "st": res.status, "vm": res.bus_vm, "va": res.bus_va}
    for k, r in enumerate(tr):
        out[f"traj_out{k}"] = r.out
        out[f"traj_st{k}"] = r.status
        out[f"traj_ls{k}"] = r.line_status
    return out, dt, first, info, eng.plan()


grids = sys.argv[1:] or ["l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev"]
for name, B in [(g, int(os.environ.get("JIT_PROBE_LANES", 4096 if "case14" in g else 1024))) for g in grids]:
    a, dta, fa, _, plan = run(name, B, False)
    for flags in os.environ.get("JIT_PROBE_FLAGS", "|-fno-unroll-loops").split("|"):
        os.environ["GRIDPF_JIT_FLAGS"] = flags
        b, dtb, fb, info, _ = run(name, B, True)
        same = all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)
        print(json.dumps({"grid": name, "lanes": B, "flags": flags, "bit_identical": bool(same), "aot_us_per_16_steps": dta * 1e6, "jit_us_per_16_steps": dtb * 1e6,
                          "first_launch_s_jit": fb, "jit": info, "plan": plan}))
        if not same:
            bad = [k for k in a if not np.array_equal(a[k], b[k], equal_nan=True)]
            print("  differs:", bad[:8])
            k = "traj_st1" if "traj_st1" in bad else bad[0]
            rows = np.where((a[k] != b[k]).reshape(a[k].shape[0], -1).any(axis=1))[0]
            print("  lanes differing in", k, len(rows), rows[:10], "aot", a[k][rows[:3]].tolist(), "jit", b[k][rows[:3]].tolist())

"""Developer experiment (round 5): launch-time distribution of the 118-substation step kernel vs batch size / observation trajectory.
Usage: python tools/exp_variance.py [env] -- prints per-launch kernel times (HIP event pair per launch)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    stub_engine = False; share_device = False; no_jit = False; profile = True; steps_per_launch = 20; last_obs_only = False
    dist_backend = "nccl"; no_oracle_check = True; dump_aot = None


def main():
    env = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_idf_2023"
    ctx = bench.Ctx(A())
    m, ch = bench.load_env(env)
    for B in [int(x) for x in os.environ.get("EXP_B", "1024,2048,3072,4096").split(",")]:
        for traj in (1, 0):
            eng, T, _ = bench.setup_engine(ctx, m, ch, B)
            spl = 20
            if traj:
                eng.set_trajectory(spl, eng.TRAJ_OBS)
            for _ in range(6):
                eng.step(0, n_steps=spl, rebalance=1.02)
            eng.sync()
            ts = []
            t = 0
            for rep in range(24):
                eng.set_profiling(2)
                eng.step(t, n_steps=spl, rebalance=1.02)
                eng.sync()
                ms, n = eng.kernel_time()
                eng.set_profiling(0)
                ts.append(ms)
                t += spl
            # back-to-back pairs (as the bench windows)
            tw = []
            for rep in range(8):
                eng.set_profiling(1)
                eng.step(t, n_steps=spl, rebalance=1.02); t += spl
                eng.step(t, n_steps=spl, rebalance=1.02); t += spl
                eng.set_profiling(3)
                eng.sync()
                ms, n = eng.kernel_time()
                eng.set_profiling(0)
                tw.append(ms / 2)
            ts = np.array(ts); tw = np.array(tw)
            print(f"{env} B={B} traj={traj} plan={eng.plan()['lds_bytes']}B single launches ms: min {ts.min():.3f} med {np.median(ts):.3f} max {ts.max():.3f} | "
                  f"{np.round(ts, 3).tolist()} | pairs/2: {np.round(tw, 3).tolist()} -> {B * spl / np.median(ts) / 1e3:.2f} M steps/s", flush=True)
            eng.close()


main()

#!/usr/bin/env python
"""bench.py -- batched DoNothing env.step throughput of the MI355X power-flow engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env NAME] [--batch B] [--cascade]

One "step" = one pass of the hot path over one batch: for every lane (independent environment copy)
chronics row -> injections -> AC Newton-Raphson power flow (init="dc", <=10 iterations, 1e-8 MVA) ->
result extraction -> overflow bookkeeping, i.e. what ``Environment.step`` asks of the Backend for a
DoNothing agent with ``NO_OVERFLOW_DISCONNECTION=True`` (the setting of the reference's own DoNothing
profiler, _profiling/profiler_do_nothing.py:42-65).  Inputs (chronics tables, lane state) are resident
in HBM before the timed region; outputs stay in HBM.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) cfg 2): ``l2rpn_case14_sandbox``, batch = 4096 lanes
per GPU, lane k reads chronics row (t + 7k) mod 576 with loads scaled by 1 + 0.05 N(0,1)
(``default_rng(k)``) and prod_p rescaled to 1.02 * sum(load).

Multi-GPU: the lanes are independent, so the batch is sharded statically, 4096 lanes per rank (weak
scaling), one process per GPU, NO collective on the data path; torch.distributed (RCCL) is only used for
the barrier and the max-over-ranks timing the contract asks for.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
F64_PEAK_TFLOPS = 78.6         # MI355X FP64 vector == FP64 matrix peak (spec)


def cpu_baseline(m, ch, T, budget_s=12.0):
    """Time the CPU oracle (a port of the reference's pandapower arithmetic, see oracle/) on a bounded
    sample of the SAME workload, single host thread.  Reported baseline, not the target."""
    try:
        from oracle import pf_oracle_c
        have_c = pf_oracle_c.available()
    except Exception:
        have_c = False
    from oracle.pf_oracle import LaneState, solve
    n_done = 0
    t0 = time.perf_counter()
    if have_c:
        n_done, elapsed = pf_oracle_c.time_steps(m, ch, T, budget_s)
        impl = "oracle/pf_oracle.c (gcc -O2, float64 dense NR)"
    else:
        vn = m.sub_vn_kv[m.gen_sub].astype(np.float32)
        while time.perf_counter() - t0 < budget_s:
            k = n_done
            row = (7 * k) % T
            rng = np.random.default_rng(k)
            sc = (1.0 + 0.05 * rng.standard_normal(2 * m.n_load)).astype(np.float32)
            st = LaneState.from_model(m)
            st.load_p = (ch["load_p"][row] * sc[:m.n_load]).astype(np.float64)
            st.load_q = (ch["load_q"][row] * sc[m.n_load:]).astype(np.float64)
            st.gen_p = ch["prod_p"][row].astype(np.float64)
            st.gen_vm = (ch["prod_v"][row] / vn).astype(np.float64)
            solve(m, st)
            n_done += 1
        elapsed = time.perf_counter() - t0
        impl = "oracle/pf_oracle.py (numpy, float64 dense NR)"
    return {"value": n_done / elapsed, "unit": "env steps/sec", "cores": 1, "kind": "port",
            "sample": f"{n_done} lane-steps of the same synthetic workload in {elapsed:.1f} s, 1 thread, {impl}",
            "host_cpus": os.cpu_count()}


def measured_traffic_bytes():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC run (profiles/r01_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes of this same command, KiB -> bytes, FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def run_workload(eng, steps, warmup, step_kw, sync_all):
    t = 0
    # untimed pre-roll before the W warm-up steps: an idle MI355X needs tens of milliseconds of work to reach its clocks
    # (measured: 2.7x slower steps right after a 12 s host-only phase with a 12-step warm-up)
    for _ in range(max(0, 400 - warmup)):
        eng.step(t, **step_kw)
        t += 1
    for _ in range(warmup):
        eng.step(t, **step_kw)
        t += 1
    eng.sync()
    eng.kernel_time()
    sync_all()
    eng.set_profiling(1)        # ONE HIP event pair on the engine's stream around the K timed launches (no per-launch events)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(t, **step_kw)
        t += 1
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms, n_launch = eng.kernel_time()
    eng.set_profiling(False)
    return elapsed, kern_ms, n_launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--env", default="l2rpn_case14_sandbox")
    ap.add_argument("--batch", type=int, default=4096, help="lanes per GPU")
    ap.add_argument("--cascade", action="store_true", help="enable overflow disconnections (cascade loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 118-substation secondary workload")
    ap.add_argument("--n1", action="store_true",
                    help="BASELINE.json configs[2]: every env copy is stepped together with its N-1 contingencies "
                         "(one extra lane per line, that line forced off), all fused into the same launch")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    try:
        import torch  # noqa: F811  (device plumbing + RCCL barrier only)
    except Exception:
        torch = None
    # launched by torch.distributed.run (RANK / MASTER_ADDR set): always go through RCCL, even with one rank,
    # so that the N>1 code path (init, barrier, max-reduce) is the one exercised on a single-GPU box too
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist  # noqa: F811
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from grid2op_amd.grid_model import GridModel
    from grid2op_amd.engine import PowerFlowEngine
    from grid2op_amd.sharding import lane_range, max_over_ranks, synthetic_lane_inputs

    m = GridModel.load_npz(os.path.join(GOLD, f"{args.env}.grid.npz"))
    ch = dict(np.load(os.path.join(GOLD, f"{args.env}.chronics.npz")))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    B = args.batch
    n_envs = B
    fan = 1
    if args.n1:
        fan = 1 + m.n_line
        B = n_envs * fan
    eng = PowerFlowEngine(m, n_lanes=B, device=local_rank)
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"])
    T = tab.shape[0]
    eng.upload_chronics(tab)
    lane0, n_mine = lane_range(world * n_envs, world, rank)     # contiguous block of GLOBAL env ids (weak scaling)
    assert n_mine == n_envs
    offsets, scale = synthetic_lane_inputs(m.n_load, T, lane0 + np.arange(n_envs))
    if args.n1:
        # lane (k, c): env k, contingency c (c = 0: intact grid, c >= 1: line c-1 forced off); the contingencies of one
        # env sit on the same GPU and share its chronics row / jitter
        offsets = np.repeat(offsets, fan)
        scale = np.repeat(scale, fan, axis=0)
        topo = np.tile(m.initial_topo_vect(), (B, 1))
        for c in range(1, fan):
            topo[c::fan, m.line_or_pos_topo_vect[c - 1]] = -1
            topo[c::fan, m.line_ex_pos_topo_vect[c - 1]] = -1
        eng.set_topology(topo)
    eng.set_lane_chronics(lane_offset=offsets, lane_scale=scale)
    if "thermal_limits" in ch:
        eng.set_thermal_limits(ch["thermal_limits"])
    step_kw = dict(rebalance=1.02, cascade=args.cascade)

    def sync_all():
        eng.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    elapsed, kern_ms, n_launch = run_workload(eng, args.steps, args.warmup, step_kw, sync_all)
    r = eng.results()
    frac_conv = float(r.converged.mean())
    mean_iter = float(r.n_iter[r.converged].mean()) if r.converged.any() else float("nan")

    elapsed = max_over_ranks(elapsed, dist, device="cuda" if dist is not None else None)
    total_steps = world * n_envs * args.steps
    value = total_steps / elapsed

    if rank == 0:
        bytes_step = eng.algorithmic_bytes_per_step()
        avg_launch_s = (kern_ms / max(n_launch, 1)) * 1e-3
        achieved_gbs = bytes_step * B / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # algorithmic flops per AC power flow: iters*(2/3 J^3 + 2 J^2), dense, J = NR unknowns
        nb = int(eng.results(0, 1).status[0, 2])
        npv = len(set(m.gen_sub[~m.gen_slack].tolist()) - set(m.gen_sub[m.gen_slack].tolist()))
        J = 2 * (nb - 1) - npv
        flops_pf = (mean_iter if mean_iter == mean_iter else 0) * (2.0 / 3.0 * J ** 3 + 2.0 * J ** 2)
        res = {
            "metric": "env steps/sec (batched DoNothing)",
            "value": value,
            "unit": "env steps/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.env} AC Newton-Raphson DoNothing env.step, batch={B} lanes per GPU "
                                   f"(row (t+7k) mod {T}, loads x (1+0.05 N(0,1)), prod_p rebalanced to 1.02 sum(load))",
                       "env": args.env, "lanes_per_gpu": B, "envs_per_gpu": n_envs, "n1_fanout": fan, "cascade": bool(args.cascade), "max_iter": 10,
                       "tol_mva": 1e-8, "parallelism": f"independent lanes, static shard x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": (measured_traffic_bytes() or {}).get("hbm_bytes_per_launch"),
                         "kernel": (measured_traffic_bytes() or {}).get("kernel"), "avg_launch_us": avg_launch_s * 1e6, "launches": int(n_launch),
                         "algorithmic_bytes_per_step": bytes_step,
                         "lds_pipe_busy_frac": (measured_traffic_bytes() or {}).get("lds_pipe_busy_frac"),
                         "note": "chains of small FP64 block factorisations in LDS dominate: the kernel is LDS-pipe / issue / latency "
                                 "bound, not HBM bound (SURVEY.md 8(d), DESIGN.md 3); lds_pipe_busy_frac is the PMC figure of the "
                                 "committed profile, f64 the dense-equivalent flop rate",
                         "f64": {"achieved_tflops": flops_pf * B / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0,
                                 "peak_tflops": F64_PEAK_TFLOPS, "algorithmic_flops_per_step": flops_pf, "J": J}},
            "frac_converged": frac_conv,
            "mean_nr_iterations": mean_iter,
        }
        res["cpu_baseline"] = None
    # ---- same workload with topology actions: 10 % of the lanes with a split substation (topology classes) --------
    split = None
    if not args.no_secondary and args.env == "l2rpn_case14_sandbox" and not args.n1:
        topo = np.tile(m.initial_topo_vect(), (B, 1))
        sub = int(np.argmax(m.sub_info))
        start = int(np.concatenate(([0], np.cumsum(m.sub_info)))[sub])
        ends = set(m.line_or_pos_topo_vect.tolist()) | set(m.line_ex_pos_topo_vect.tolist())
        line_pos = [q for q in range(start, start + int(m.sub_info[sub])) if q in ends]
        pick = np.random.default_rng(lane0).random(B) < 0.10
        for q in line_pos[::2][:max(1, len(line_pos) // 2 - 1)]:
            topo[pick, q] = 2
        eng.set_topology(topo)
        el_s, k_s, n_s = run_workload(eng, max(20, args.steps // 4), max(2, args.warmup // 4), step_kw, sync_all)
        el_s = max_over_ranks(el_s, dist, device="cuda" if dist is not None else None)
        conv_s = float(eng.results().converged.mean())
        if rank == 0:
            split = {"workload": f"{args.env}, batch={B}: substation {sub} split (lines alternating between its two busbars) in "
                                 f"{100.0 * pick.mean():.0f} % of the lanes",
                     "value": world * B * max(20, args.steps // 4) / el_s, "unit": "env steps/sec",
                     "us_per_step": k_s / max(n_s, 1) * 1e3, "frac_converged": conv_s}
    eng.close()

    # ---- secondary workload: the 118-substation grid of BASELINE.json configs[3] (1024 lanes per GPU) -------------------
    sec = None
    if not args.no_secondary and args.env == "l2rpn_case14_sandbox":
        env2, B2 = "l2rpn_wcci_2022_dev", 1024
        m2 = GridModel.load_npz(os.path.join(GOLD, f"{env2}.grid.npz"))
        ch2 = dict(np.load(os.path.join(GOLD, f"{env2}.chronics.npz")))
        if "prod_v" not in ch2:
            ch2["prod_v"] = np.tile((m2.gen_vm0 * m2.sub_vn_kv[m2.gen_sub]).astype(np.float32), (ch2["prod_p"].shape[0], 1))
        eng2 = PowerFlowEngine(m2, n_lanes=B2, device=local_rank)
        tab2 = eng2.pack_chronics(ch2["load_p"], ch2["load_q"], ch2["prod_p"], ch2["prod_v"])
        T2 = tab2.shape[0]
        eng2.upload_chronics(tab2)
        l0, _ = lane_range(world * B2, world, rank)
        off2, sc2 = synthetic_lane_inputs(m2.n_load, T2, l0 + np.arange(B2))
        eng2.set_lane_chronics(lane_offset=off2, lane_scale=sc2)
        if m2.n_storage:                                      # BASELINE.json configs[3]: storage actions, U(-2, 2) MW per unit
            inj2 = eng2.get_injections()
            lay2 = eng2.layout
            for k in range(B2):
                inj2[k, lay2.inj_storage_p:lay2.inj_storage_p + m2.n_storage] = np.random.default_rng(l0 + k).uniform(-2.0, 2.0, m2.n_storage)
            eng2.set_injections(inj2)
        steps2 = max(10, args.steps // 4)

        def sync2():
            eng2.sync()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()
        el2, k2, n2 = run_workload(eng2, steps2, max(2, args.warmup // 4), dict(rebalance=1.02, cascade=args.cascade), sync2)
        el2 = max_over_ranks(el2, dist, device="cuda" if dist is not None else None)
        r2 = eng2.results()
        if rank == 0:
            b2 = eng2.algorithmic_bytes_per_step()
            sec = {"workload": f"{env2} (118 substations) AC NR env.step with storage set-points U(-2,2) MW, batch={B2} lanes per GPU", "value": world * B2 * steps2 / el2,
                   "unit": "env steps/sec", "ms_per_step": el2 / steps2 * 1e3, "steps": steps2,
                   "avg_launch_us": k2 / max(n2, 1) * 1e3, "algorithmic_bytes_per_step": b2,
                   "hbm_gbs": b2 * B2 / (k2 / max(n2, 1) * 1e-3) / 1e9 if k2 > 0 else 0.0,
                   "frac_converged": float(r2.converged.mean()), "mean_nr_iterations": float(r2.n_iter[r2.converged].mean())}
        eng2.close()
    # ---- DC sensitivity path of BASELINE.json configs[4]: l2rpn_idf_2023, 2048 lanes, PTDF GEMM next to the AC solve ------------
    ptdf = None
    if not args.no_secondary and args.env == "l2rpn_case14_sandbox":
        env3, B3 = "l2rpn_idf_2023", 2048
        m3 = GridModel.load_npz(os.path.join(GOLD, f"{env3}.grid.npz"))
        eng3 = PowerFlowEngine(m3, n_lanes=B3, device=local_rank)
        l0, _ = lane_range(world * B3, world, rank)
        inj3 = np.tile(eng3.get_injections(0, 1), (B3, 1))
        lay = eng3.layout
        for k in range(B3):                                   # +-5 % load jitter per lane, generators follow
            f = 1.0 + 0.05 * np.random.default_rng(l0 + k).standard_normal(m3.n_load)
            lp = inj3[k, lay.inj_load_p:lay.inj_load_p + m3.n_load]
            gp = inj3[k, lay.inj_gen_p:lay.inj_gen_p + m3.n_gen]
            gp *= (lp * f).sum() / lp.sum()
            lp *= f
        eng3.set_injections(inj3)
        eng3.ptdf_build(0)
        reps = max(20, args.steps)
        for _ in range(3):
            eng3.ptdf_flows(fetch=False)
        eng3.sync()
        if dist is not None:
            dist.barrier()
        eng3.set_profiling(1)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng3.ptdf_flows(fetch=False)
        eng3.sync()
        el3 = time.perf_counter() - t0
        k3, n3 = eng3.kernel_time()
        eng3.set_profiling(0)
        el3 = max_over_ranks(el3, dist, device="cuda" if dist is not None else None)
        flows = eng3.ptdf_flows()
        eng3.lodf_screen(0, 8)
        t0 = time.perf_counter()
        for _ in range(5):
            worst = eng3.lodf_screen()                          # synchronous: includes the copy of [B3, n_line] floats to the host
        lodf_s = (time.perf_counter() - t0) / 5
        eng3.runpf(is_dc=True)
        eng3.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            eng3.runpf(is_dc=True)
        eng3.sync()
        dc_solve_s = (time.perf_counter() - t0) / 5
        r3 = eng3.results()
        eng3.runpf()
        eng3.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            eng3.runpf()
        eng3.sync()
        ac_solve_s = (time.perf_counter() - t0) / 5
        r3ac = eng3.results()
        if rank == 0:
            nb_act = int(r3.status[0, 2])                     # active buses of the topology = K of the GEMM
            nb_pad, line_pad = (nb_act + 3) // 4 * 4, (m3.n_line + 15) // 16 * 16
            us = k3 / max(n3, 1) * 1e3
            ptdf = {"workload": f"{env3} (118 substations) batch={B3} lanes per GPU: DC line flows of every lane as ONE FP64 MFMA GEMM "
                                f"(flows = P_bus[{B3}x{nb_pad}] . PTDF^T[{nb_pad}x{line_pad}]) for a fixed topology",
                    "value": world * B3 * reps / el3, "unit": "DC power flows/sec", "us_per_batch": us,
                    "roofline": {"bound": "mfma", "achieved": 2.0 * B3 * nb_pad * line_pad / (us * 1e-6) / 1e12 if us > 0 else 0.0,
                                 "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": (2.0 * B3 * nb_pad * line_pad / (us * 1e-6) / 1e12 / F64_PEAK_TFLOPS) if us > 0 else 0.0,
                                 "note": "90 MFLOP per batch: launch/latency bound, two kernels (bus injections + GEMM)"},
                    "per_lane_dc_solve_value": world * B3 / dc_solve_s, "per_lane_dc_solve_unit": "DC power flows/sec (kernel S, B' refactorised per lane)",
                    "max_abs_diff_vs_per_lane_dc_solve_mw": float(np.abs(flows - r3.p_or).max()),
                    "lodf_n1_value": world * B3 * m3.n_line / lodf_s, "lodf_n1_unit": "DC contingency cases/sec (every single-line outage of "
                    "every lane: worst post-outage flow via LODF, result copied to the host)",
                    "lodf_n1_frac_islanding": float(np.isinf(worst).mean()),
                    "ac_runpf_value": world * B3 / ac_solve_s, "ac_runpf_unit": "AC power flows/sec (same lanes, gpf_runpf)",
                    "ac_frac_converged": float(r3ac.converged.mean()),
                    "max_abs_dc_vs_ac_p_or_mw": float(np.abs(flows - r3ac.p_or)[r3ac.converged].max())}
        eng3.close()
    if rank == 0:
        # the CPU baseline is timed LAST (rank 0 of the 1-GPU run only): 12 s of host-only work in the middle of the run would
        # let the GPU clocks drop before the secondary workloads
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(m, ch, T)
        res["secondary"] = sec
        res["dc_ptdf"] = ptdf
        res["split_topologies"] = split
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

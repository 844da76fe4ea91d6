#!/usr/bin/env python
"""bench.py -- batched DoNothing env.step throughput of the MI355X power-flow engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env NAME] [--batch B] [--cascade]

One "step" = one pass of the hot path over one batch: for every lane (independent environment copy)
chronics row -> injections -> AC Newton-Raphson power flow (init="dc", <=10 iterations, 1e-8 MVA) ->
result extraction -> overflow bookkeeping, i.e. what ``Environment.step`` asks of the Backend for a
DoNothing agent.  The headline uses ``NO_OVERFLOW_DISCONNECTION=True`` (the setting of the reference's own
DoNothing profiler, _profiling/profiler_do_nothing.py:42-65); the same workload with the reference's DEFAULT
parameters (overflow disconnections / cascade on) is reported beside it (``cascade_on``).  Inputs (chronics
tables, lane state) are resident in HBM before the timed region; outputs stay in HBM.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) cfg 2): ``l2rpn_case14_sandbox``, batch = 4096 lanes
per GPU, lane k reads chronics row (t + 7k) mod 576 with loads scaled by 1 + 0.05 N(0,1)
(``default_rng(k)``) and prod_p rescaled to 1.02 * sum(load).

Multi-GPU: the lanes are independent, so the batch is sharded statically, 4096 lanes per rank (weak
scaling), ONE PROCESS PER GPU, no collective on the data path; torch.distributed (RCCL) is only used for the
barrier and the max-over-ranks timing.  ``python bench.py --gpus N`` without a torchrun environment launches
the N ranks itself (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``)
and FAILS when fewer than N devices are visible; under torchrun ``--gpus`` must equal WORLD_SIZE.

Observations: the reference returns one observation per ``env.step``.  The headline therefore runs its multi-step launches
(``--steps-per-launch``, default 16) with the per-step observation trajectory ON (``gpf_set_trajectory(.., GPF_TRAJ_OBS)``):
every one of the 16 steps of a launch writes its complete backend observation (results row, topo_vect, shunt buses, line
status, rho, status) to its own rows in HBM.  The cheaper contracts are labelled secondaries: ``rollout_last_observation_only``
(16 steps per launch, each step overwrites the lane's single row: only the last observation of a launch exists) and
``one_launch_per_step``.  After each timed workload a sample of lanes is re-solved by the C oracle (checker only, outside
every timed region): ``max_abs_err_vs_oracle``.

Timing: W warm-up steps, then ``--windows`` (default 5) timed windows of EXACTLY K steps each, every window
bracketed by barrier + synchronize on both sides.  ``value`` / ``ms_per_step`` are the WALL CLOCK between those brackets,
MAX-reduced over the ranks, of the MEDIAN window (min / median / max in ``windows``).  The HIP-event window on the engine's
stream around the same K steps (closed right behind the last launch) is reported beside it (``value_hip_event_window``) and
is the clock of ``roofline`` (average launch duration of the dominant kernel).  The BASELINE-config secondaries are kernel
measurements: >= 5 HIP-event windows of >= 2 launches each, min / median / max reported.

Output: ONE stdout line (rank 0) -- a compact JSON record (< 4 KB: the contract's keys, ``roofline``, ``cpu_baseline``, one number +
roofline fraction per BASELINE config, the oracle verdict).  Everything else the run measured goes to ``bench_full.json`` next to
this file (and to ``gpurun_out/`` when that directory exists).

Kernels: by default every engine switches its solver launches to kernels compiled at run time for the workload's grid
(``PowerFlowEngine.specialize`` = gpf_jit_enable: the grid's sizes / table offsets as literals, results bit-identical to the
shipped kernels, self-tested against them on a twin engine first; compilation happens in the untimed warm-up and is cached on
disk; ``specialization`` in the JSON says what was loaded).  The same headline on the shipped kernels is reported as
``shipped_kernels``; ``--no-jit`` runs everything on the shipped kernels (also what happens when no hipcc is available).

"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
F64_PEAK_TFLOPS = 78.6         # MI355X FP64 vector == FP64 matrix peak (spec)
TRAFFIC_PROFILE = os.path.join("profiles", "r06_traffic.json")
TRAFFIC_N1 = os.path.join("profiles", "r06_traffic_n1_neurips36.json")
TRAFFIC_N1_118 = os.path.join("profiles", "r06_traffic_n1_wcci118.json")
TRAFFIC_WCCI = os.path.join("profiles", "r06_traffic_wcci118.json")
TRAFFIC_IDF = os.path.join("profiles", "r06_traffic_idf118.json")
TRAFFIC_1PL = os.path.join("profiles", "r06_traffic_case14_1perlaunch.json")
TRAFFIC_PTDF = os.path.join("profiles", "r05_traffic_ptdf.json")
CASCADE_LIMIT_SCALE = 0.85     # `cascade_tripping` secondary: thermal limits x 0.85 -> ~20 % of the lane-steps overflow softly


# ---------------------------------------------------------------------------------------------------------------------
# launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv) -> int:
    """``--gpus N`` (N > 1) outside torchrun: start one rank per GPU and pass its single JSON line through."""
    if not args.stub_engine and not args.share_device:
        from grid2op_amd.sharding import visible_devices
        n_dev = visible_devices()
        if n_dev < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {n_dev} HIP device(s) are visible; refusing to "
                             f"measure fewer GPUs than asked for\n")
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = a port of the reference's pandapower arithmetic; reported baseline, not the target)
def cpu_baseline(env_name, m, ch, T, budget_s=10.0, with_dense=True):
    """The C port of the reference's pandapower arithmetic (oracle/pf_oracle.c) on the host cores, on a bounded sample of the SAME synthetic
    workload (chronics row -> injections -> AC Newton-Raphson -> result row, per lane).  `value` is the SPARSE path (LU on a minimum-degree
    ordering, symbolic analysis kept across solves: what pandapower's scipy spsolve / lightsim2grid's KLU do -- pinned to the dense path
    by tests/test_oracle_c.py); the dense elimination, a straw man beyond a few dozen buses, is reported beside it."""
    from oracle import pf_oracle_c
    res = {"unit": "env steps/sec", "kind": "port", "solver": "sparse LU (min-degree, symbolic kept)", "host_cpus": os.cpu_count(),
           "pandapower": "unavailable (PandaPowerBackend needs pandapower>=3.1.1: not installed in this image, no network)",
           "lightsim2grid": "unavailable (LightSimBackend: package not installed in this image, no network)"}
    n_done, elapsed = pf_oracle_c.time_steps(m, ch, T, 0.6 * budget_s, sparse=True)
    impl = "oracle/pf_oracle.c (gcc -O2, float64 Newton-Raphson, sparse LU)"
    res.update({"value": n_done / elapsed, "cores": 1,
                "sample": f"{n_done} lane-steps of the same synthetic workload in {elapsed:.1f} s, 1 thread, {impl}"})
    if with_dense:
        nd_, ed_ = pf_oracle_c.time_steps(m, ch, T, 0.3 * budget_s, sparse=False)
        res["dense_1core"] = {"value": nd_ / ed_, "sample": f"{nd_} lane-steps in {ed_:.1f} s, dense Gaussian elimination (the path pinned to the golden vectors)"}
    ncores, how = usable_cores()
    res["usable_cores"] = f"{ncores} ({how})"
    if ncores > 1:
        t0 = time.perf_counter()
        tot, wall = pf_oracle_c.time_steps_all_cores(os.path.join(GOLD, f"{env_name}.grid.npz"),
                                                     os.path.join(GOLD, f"{env_name}.chronics.npz"), ncores, 0.6 * budget_s, sparse=True)
        res["all_cores"] = {"value": tot / wall, "unit": "env steps/sec", "cores": ncores,
                            "sample": f"{tot} lane-steps by {ncores} processes (one per host core, 256 lanes each, stepped through t = 0, 1, ...) in "
                                      f"{wall:.1f} s of wall time after their common start; total {time.perf_counter() - t0:.1f} s"}
    return res


_CPU_CFG_CACHE = {}


def cpu_baseline_config(env_name, budget_s=4.0):
    """Per-config CPU row: the sparse port on THAT config's grid -- lane power flows (= env steps of one lane) per second on
    1 core and on every usable core.  An N-1 fan-out lane is the same arithmetic with one line out, so its figure is the grid's."""
    if env_name not in _CPU_CFG_CACHE:
        m, ch = load_env(env_name)
        T = ch["load_p"].shape[0]
        r = cpu_baseline(env_name, m, ch, T, budget_s=budget_s, with_dense=False)
        ac = r.get("all_cores") or {}
        _CPU_CFG_CACHE[env_name] = {"value": r["value"], "cores": 1, "all_cores": {"value": ac.get("value"), "cores": ac.get("cores")} if ac else None,
                                    "kind": "sparse port", "unit": "lane power flows (env steps of one lane) per second", "grid": env_name,
                                    "sample": r["sample"]}
    return _CPU_CFG_CACHE[env_name]


def reference_step_loop():
    """The reference's own DoNothing profiler loop (_profiling/profiler_do_nothing.py:42-61: Environment.step, DoNothingAgent,
    NO_OVERFLOW_DISCONNECTION, AlwaysLegal) with BOTH engines under the unmodified framework: HipBackend + libgridpf.so and the facade
    over the CPU oracle.  grid2op cannot be part of this repository or of the GPU box's image: when tools/gpurun_staged.sh has staged
    the unmodified reference package for this call (`_stage/`), the loop is MEASURED HERE (tests/reference_on_hip.py timing, a
    subprocess); otherwise the committed figures of the staged MI355X run of this round are quoted and labelled as such."""
    stage = os.path.join(ROOT, "_stage")
    if os.path.isdir(os.path.join(stage, "grid2op")):
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "REFERENCE_ON_HIP_DRYRUN")}
            env["GRID2OP_REFERENCE"] = stage
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "reference_on_hip.py"), "timing", "500"], capture_output=True,
                               text=True, timeout=600, env=env, cwd=ROOT)
            line = [l for l in p.stdout.splitlines() if l.startswith("TIMING ")][-1]
            return dict(json.loads(line[len("TIMING "):]), source="measured in this run (unmodified reference package staged for this call)")
        except Exception as exc:
            err = repr(exc)[:200]
    else:
        err = None
    import glob
    staged = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_step_loop_mi355x.json")))      # the newest round's record
    rel_staged = os.path.relpath(staged[-1], ROOT) if staged else os.path.join("profiles", "r05_reference_step_loop_mi355x.json")
    for rel, src in ((rel_staged,
                      f"committed {rel_staged}: tests/reference_on_hip.py timing on an MI355X box with the unmodified "
                      "reference package staged for that call (tools/gpurun_staged.sh); NOT measured in this run (the reference cannot be staged here)"),
                     (os.path.join("profiles", "r02_framework_loop_cpu.json"), "committed profiles/r02_framework_loop_cpu.json (build container CPU, round 2)")):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                d = dict(json.load(f), source=src)
            if err:
                d["staged_run_error"] = err
            return d
        except Exception:
            continue
    return None


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256 logical CPUs
    behind a quota of 16: 256 busy processes there are SLOWER than 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    how = "affinity mask"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(float(quota) / float(period) + 0.5))
            if q < n:
                n, how = q, f"cgroup cpu.max quota of {q} CPUs, {os.cpu_count()} logical CPUs visible"
    except Exception:
        pass
    return n, how


def traffic_profile(rel=None):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC run (FETCH_SIZE and WRITE_SIZE collected
    in separate --pmc passes of this same command, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950).  Not
    measured in this run: the JSON says so (``traffic_source``)."""
    for rel in (rel or TRAFFIC_PROFILE,):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                d = json.load(f)
            d["_file"] = rel
            return d
        except Exception:
            continue
    return {}


# ---------------------------------------------------------------------------------------------------------------------
class Ctx:
    """rank / device plumbing shared by the workloads"""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = 0 if args.share_device else self.local_rank      # --share-device: every rank drives device 0
        self.dist = None
        self.torch = None
        self.red_dev = None
        self.engines, self.jit_verified, self.jit_errors, self.jit_closed = [], set(), [], []
        self._cuda = None
        try:
            import torch
            self.torch = torch
        except Exception:
            pass
        # launched by torch.distributed.run (RANK / MASTER_ADDR set): always go through the process group, even with one
        # rank, so that the N>1 code path (init, barrier, max-reduce) is the one exercised on a single-GPU box too
        if self.world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
            import torch.distributed as dist
            self.dist = dist
            if args.dist_backend == "nccl":
                self.torch.cuda.set_device(self.device)
                dist.init_process_group(backend="nccl", device_id=self.torch.device("cuda", self.device))
                self.red_dev = "cuda"
            else:
                dist.init_process_group(backend=args.dist_backend)

    def cuda_ok(self):
        if self._cuda is None:                 # (asked once: torch.cuda.is_available() costs microseconds, this runs inside the timed brackets)
            self._cuda = bool(self.torch is not None and not self.args.stub_engine and self.torch.cuda.is_available())
        return self._cuda

    def cuda_sync(self):
        if self.cuda_ok():
            self.torch.cuda.synchronize()

    def sync_all(self, *engines):
        for e in engines:
            e.sync()
        self.cuda_sync()
        if self.dist is not None:
            self.dist.barrier()
            self.cuda_sync()

    def max(self, v):
        from grid2op_amd.sharding import max_over_ranks
        return max_over_ranks(v, self.dist, device=self.red_dev)

    def make_engine(self, m, n_lanes):
        if self.args.stub_engine:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from stub_engine import StubEngine           # launcher test hook (CPU): no arithmetic, see tests/stub_engine.py
            return StubEngine(m, n_lanes=n_lanes, device=self.device)
        from grid2op_amd.engine import PowerFlowEngine
        eng = PowerFlowEngine(m, n_lanes=n_lanes, device=self.device)
        if not self.args.no_jit:
            # step kernels compiled for THIS grid (gpf_jit_enable): the first engine of each grid runs the library's self-test of the
            # specialised kernels against the shipped ones (64-lane twin, bit-for-bit); compilation happens at the first launch of each
            # kernel variant, i.e. in the untimed warm-up, and is cached on disk
            key = (m.n_sub, m.n_line, m.n_gen, m.n_load, getattr(m, "name", None))
            try:
                eng.specialize(True, verify=key not in self.jit_verified and not self.args.profile)   # (--profile: no twin-engine dispatches in the trace)
                self.jit_verified.add(key)
            except Exception as exc:                 # no hipcc on this host / self-test failed: shipped kernels, said so in the JSON
                self.jit_errors.append(str(exc)[:300])
        self.engines.append(eng)
        orig_close = eng.close

        def close_and_remember():               # the workloads close their engines: keep what they launched for the record
            try:
                self.jit_closed.append(eng.specialization())
                if getattr(self.args, "dump_aot", None):
                    self.dump_aot(eng, getattr(m, "name", None) or f"grid{m.n_sub}")
            except Exception:
                pass
            if eng in self.engines:
                self.engines.remove(eng)
            orig_close()
        eng.close = close_and_remember
        return eng

    def dump_aot(self, eng, label):
        """developer (--dump-aot DIR): the generated header of every engine's grid + the kernel variants it launched, merged into
        DIR/manifest.json -- the input of __graft_entry__.build()'s ahead-of-time specialisation (grid2op_amd/aot/)"""
        import hashlib
        d = self.args.dump_aot
        os.makedirs(d, exist_ok=True)
        hdr = eng.specialization_header()
        hid = hashlib.sha1(hdr.encode()).hexdigest()[:12]
        mp = os.path.join(d, "manifest.json")
        man = json.load(open(mp)) if os.path.exists(mp) else {}
        ent = man.setdefault(hid, {"header": f"{hid}.h", "grids": [], "variants": []})
        with open(os.path.join(d, ent["header"]), "w") as f:
            f.write(hdr)
        if label not in ent["grids"]:
            ent["grids"].append(label)
        for v in eng.specialization()["variants"].split(" | ")[0].split():
            kname = "runpf" if v.startswith("runpf<") else "step"
            var = v[v.index("<") + 1:v.index(">")]
            if [kname, var] not in ent["variants"]:
                ent["variants"].append([kname, var])
        with open(mp, "w") as f:
            json.dump(man, f, indent=1, sort_keys=True)

    def specialization(self):
        """what the engines of this run launched: shipped kernels or kernels specialised at run time (summed over the engines)"""
        tot = {"enabled": False, "aot": 0, "compiled": 0, "cached": 0, "failed": 0, "launches": 0, "seconds": 0.0, "variants": []}
        infos = list(self.jit_closed)
        for e in self.engines:
            if hasattr(e, "specialization"):
                try:
                    infos.append(e.specialization())
                except Exception:
                    pass
        for i in infos:
            tot["enabled"] = tot["enabled"] or i["enabled"]
            for k in ("aot", "compiled", "cached", "failed", "launches"):
                tot[k] += i.get(k, 0)
            tot["seconds"] += i["seconds"]
            for v in i["variants"].split(" | ")[0].split():
                if v not in tot["variants"]:
                    tot["variants"].append(v)
        tot["errors"] = self.jit_errors or None
        return tot


def load_env(name):
    from grid2op_amd.grid_model import GridModel
    m = GridModel.load_npz(os.path.join(GOLD, f"{name}.grid.npz"))
    try:
        m.name = name
    except Exception:
        pass
    ch = dict(np.load(os.path.join(GOLD, f"{name}.chronics.npz")))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    return m, ch


def setup_engine(ctx, m, ch, n_envs, fan=1):
    """Engine with the synthetic workload of SURVEY.md 8(d): ``n_envs`` env copies per rank (x ``fan`` lanes each)."""
    from grid2op_amd.sharding import lane_range, synthetic_lane_inputs
    B = n_envs * fan
    eng = ctx.make_engine(m, B)
    tab = eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"])
    T = tab.shape[0]
    eng.upload_chronics(tab)
    lane0, n_mine = lane_range(ctx.world * n_envs, ctx.world, ctx.rank)     # contiguous block of GLOBAL env ids (weak scaling)
    assert n_mine == n_envs
    offsets, scale = synthetic_lane_inputs(m.n_load, T, lane0 + np.arange(n_envs))
    if fan > 1:
        offsets = np.repeat(offsets, fan)
        scale = np.repeat(scale, fan, axis=0)
    eng.set_lane_chronics(lane_offset=offsets, lane_scale=scale)
    if "thermal_limits" in ch:
        eng.set_thermal_limits(ch["thermal_limits"])
    eng.bench_inputs = (tab, offsets, scale)            # for the oracle spot check after the timed region
    return eng, T, lane0


def oracle_spot_check(ctx, eng, n_lanes=32, t_last=None, rebalance=1.02, is_dc=False, seed=0, alive_only=False):
    """CHECKER leg (never timed, never the product): re-solve a sample of lanes with the C oracle and compare with what the engine
    holds after the timed workload.  `t_last` given (DoNothing workloads without trips): the oracle recomputes the step from the
    chronics table (covers the device-side chronics gather); otherwise from the injection / topology rows the lanes hold."""
    if ctx.args.stub_engine or ctx.args.no_oracle_check:
        return None
    try:
        from oracle import spot_check
        pool = np.arange(eng.n_lanes)
        if alive_only:       # auto_reset workloads: a lane whose LAST step failed already holds its reset topology (results NaN): not comparable
            pool = np.nonzero(eng.results(with_bus=False).status[:, 0] == 0)[0]
        lanes = np.sort(np.random.default_rng(seed).choice(pool, min(n_lanes, pool.size), replace=False))
        if t_last is not None:
            tab, off, sc = eng.bench_inputs
            r = eng.results(with_bus=False)
            res = spot_check.check_step(eng.model, tab, off, sc, rebalance, t_last, lanes, r.out[lanes], r.status[lanes])
            res["against"] = "oracle/pf_oracle.c, step recomputed from the chronics table"
        else:
            res = spot_check.check_lanes(eng, lanes, is_dc=is_dc)
            res["against"] = "oracle/pf_oracle.c, re-solve of the injection / topology rows the lanes hold"
        res["max_abs_err_vs_oracle"] = res.pop("max_abs_err")
        res["tolerance"] = spot_check.tolerance_text(eng.model)
        return res
    except Exception as exc:                      # the checker must never take the measurement down
        return {"error": repr(exc)[:300]}


def run_steps(eng, t, n, step_kw, spl):
    """``n`` env steps starting at time index ``t``, ``spl`` steps per launch (the last launch takes the remainder)."""
    n_launch = -(-n // spl)                       # balanced: 20 steps at 16 per launch = 10 + 10, not 16 + 4
    done = 0
    for i in range(n_launch):
        k = (n - done) // (n_launch - i)
        eng.step(t + done, n_steps=k, **step_kw)
        done += k
    return t + n


def preroll(eng, step_kw, n, spl=1):
    """untimed: an idle MI355X needs tens of milliseconds of work to reach its clocks (measured: 2.7x slower steps right
    after a 12 s host-only phase with a 12-step warm-up)"""
    run_steps(eng, 0, n, step_kw, spl)
    eng.sync()


def timed_windows(ctx, eng, steps, warmup, step_kw, n_windows, t0=0, preroll_steps=400, spl=None):
    """W warm-up steps, then ``n_windows`` windows of exactly ``steps`` steps (``spl`` env steps per launch); returns the
    per-window (elapsed seconds MAX-reduced over the ranks, kernel ms, launches) and the next time index."""
    spl = ctx.args.steps_per_launch if spl is None else spl
    t = t0
    preroll(eng, step_kw, max(0, preroll_steps - warmup), spl)
    t = run_steps(eng, t, warmup, step_kw, spl)
    out = []
    for _ in range(n_windows):
        eng.sync()
        eng.kernel_time()
        ctx.sync_all(eng)
        eng.set_profiling(1)     # ONE HIP event pair on the engine's stream around the K timed launches (no per-launch events)
        w0 = time.perf_counter()
        t = run_steps(eng, t, steps, step_kw, spl)
        eng.set_profiling(3)     # the event window ends right behind the last launch (not when it is read, after the barrier)
        ctx.sync_all(eng)
        el = time.perf_counter() - w0
        k_ms, n_l = eng.kernel_time()
        eng.set_profiling(0)
        # the timed quantity: this rank's HIP-event window around its K steps (first launch begins ... last launch ends, on the
        # engine's stream), MAX-reduced over the ranks AFTER the window -- the closing barrier (tens of microseconds of RCCL / gloo)
        # is outside it.  The wall clock between the two barrier + synchronize brackets is kept beside it (`wall`).
        wall = ctx.max(el)
        dev = ctx.max(k_ms * 1e-3) if k_ms > 0 else wall
        out.append((dev, k_ms, n_l, wall))
    return out, t


def median_window(wins, key=0):
    """the median window by its HIP-event time (key 0) or by its barrier-to-barrier wall clock (key 3)"""
    order = sorted(range(len(wins)), key=lambda i: wins[i][key])
    return wins[order[len(order) // 2]]


def summarize(wins, total_steps_per_window, key=0):
    el = [w[key] for w in wins]
    med = median_window(wins, key)[key]
    out = {"n": len(wins), "value_median": total_steps_per_window / med, "value_min": total_steps_per_window / max(el),
           "value_max": total_steps_per_window / min(el), "elapsed_ms": [round(e * 1e3, 4) for e in el]}
    if len(wins[0]) > 3:
        out["wall_clock_ms"] = [round(w[3] * 1e3, 4) for w in wins]
        out["value_wall_clock_median"] = total_steps_per_window / sorted(w[3] for w in wins)[len(wins) // 2]
    return out


def measure_modes(ctx, eng, k, w, step_kw, n_win, preroll_steps, spl=None, last_obs_too=True):
    """One workload in the two observation contracts: (1) EVERY env step of a launch writes its complete backend observation to its
    own rows in HBM (gpf_set_trajectory GPF_TRAJ_OBS: the reference returns one observation per env.step) -- the figure every
    config reports --, (2) the labelled sibling: each step overwrites the lane's row, only the last observation of a launch exists.
    Returns (windows with observations, windows last-observation-only or None, next time index)."""
    spl = ctx.args.steps_per_launch if spl is None else spl
    traj = spl > 1 and not ctx.args.stub_engine and not ctx.args.last_obs_only
    if getattr(ctx.args, "profile", False):
        preroll_steps, last_obs_too = 0, False
    if traj:
        eng.set_trajectory(spl, eng.TRAJ_OBS)
    w_obs, t = timed_windows(ctx, eng, k, w, step_kw, n_win, preroll_steps=preroll_steps, spl=spl)
    w_last = None
    if traj and last_obs_too:
        eng.set_trajectory(0)
        w_last, t = timed_windows(ctx, eng, k, 0, step_kw, max(2, n_win - 1), t0=t, preroll_steps=0, spl=spl)
        eng.set_trajectory(spl, eng.TRAJ_OBS)        # the engine keeps the observation contract for whatever follows (spot checks read it)
    return w_obs, w_last, t


DEFAULT_STEPS_PER_LAUNCH = 16
DRIVER_STEPS = 20             # the driver's command: bench.py --gpus 1 --steps 20 --warmup W (BENCH_rNN.json)


def launch_length(steps: int, spl: int = DEFAULT_STEPS_PER_LAUNCH) -> int:
    """Env steps per kernel launch of the headline for a timed window of ``steps`` steps: a window of K <= 2 * spl steps is ONE launch of K
    steps, not two half-size ones (reported in config; tests/test_gpu_bench_parity.py checks the headline at exactly these lengths)."""
    return steps if (spl > 1 and spl < steps <= 2 * spl) else spl


N1_118_SPL = int(os.environ.get("GRIDPF_BENCH_N1_118_SPL", "16"))   # env steps per launch of the 118-substation N-1 fan-out (its observation trajectory is 2.7 GB per step: 43 GB of the 288 GB at 16)
N_WIN_CFG = 5           # timed windows of every BASELINE-config secondary (min / median / max reported)


def cfg_steps(ctx, k_sec):
    """steps per timed window of a BASELINE-config secondary: at least two full launches"""
    return max(k_sec, 2 * ctx.args.steps_per_launch)


OBS_EVERY = ("every env step: each step of a launch writes its complete backend observation (results row, topo_vect, shunt buses, line "
             "status, rho, status) to its own rows in HBM (gpf_set_trajectory GPF_TRAJ_OBS)")
OBS_LAST = "LAST step of each launch only (each step overwrites the lane's result row) -- NOT the reference's env.step contract"


def ptdf_traffic(prefix):
    """HBM bytes per launch of a PTDF-path kernel at a bench shape, from the committed PMC passes (profiles/r05_traffic_ptdf.json; entry whose
    name starts with `prefix`); (bytes, source text) or (None, None)"""
    tp = traffic_profile(TRAFFIC_PTDF)
    for key, v in tp.items():
        if isinstance(v, dict) and key.startswith(prefix):
            return v.get("hbm_bytes_per_launch"), f"committed profile {TRAFFIC_PTDF}: {key} (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE; NOT measured in this run)"
    return None, None


def roofline_block(eng, wins, B, k, profile, note=None):
    """`roofline` of one workload: ALGORITHMIC bytes per launch (SURVEY.md 8(d) bytes per env step x lanes x steps per launch) / the
    average launch duration of the median HIP-event window; `traffic` = PMC bytes per launch of the committed profile of the same
    command (profiles/<profile>, rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE)."""
    med = median_window(wins)
    n_l = max(int(med[2]), 1)
    # a batch of a few residency rounds goes out as one kernel dispatch per round (gridpf_launch_step.hip): the block is stated per DISPATCH --
    # what rocprofv3 lists and the committed PMC profile counts --, i.e. lanes / dispatches-per-launch lanes in launch-duration / dispatches each
    ratio = 1.0
    if hasattr(eng, "counters"):
        try:
            cn = eng.counters()
            ratio = max(1.0, round(cn["kernel_dispatches"] / max(cn["step_launches"], 1)))
        except Exception:
            pass
    avg_s = ((med[1] * 1e-3) / n_l if med[1] > 0 else med[0] / n_l) / ratio
    b_step = eng.algorithmic_bytes_per_step()
    spl_eff = k / n_l
    gbs = b_step * (B / ratio) * spl_eff / avg_s / 1e9 if avg_s > 0 else 0.0
    tp = traffic_profile(profile)
    blk = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "traffic": tp.get("hbm_bytes_per_launch"), "traffic_over_algorithmic": tp.get("traffic_over_algorithmic"),
           "traffic_source": (f"committed profile {tp.get('_file')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                              f"{tp.get('env_steps_per_launch')} env steps per dispatch, observation trajectory on; NOT measured in this run)") if tp else None,
           "kernel": tp.get("kernel"), "avg_launch_us": avg_s * 1e6, "launches": int(n_l * ratio), "env_steps_per_launch": spl_eff,
           "lanes_per_dispatch": int(B / ratio), "kernel_dispatches_per_step_launch": int(ratio),
           "algorithmic_bytes_per_step": b_step, "lds_pipe_busy_frac": tp.get("lds_pipe_busy_frac"),
           "lds_bank_conflict_frac_of_lds_cycles": tp.get("lds_bank_conflict_frac_of_lds_cycles"), "valu_busy_frac": tp.get("valu_busy_frac")}
    if note:
        blk["note"] = note
    return blk



# ---------------------------------------------------------------------------------------------------------------------
# the ONE stdout line: a compact record (< 4 KB; the driver keeps 8 KB of stdout); everything else goes to bench_full.json
COMPACT_LIMIT = 4096


def _r(x, nd=4):
    """round floats to `nd` significant digits (JSON size), pass everything else through"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd}g}")
    return x


def _win3(w):
    """[min, median, max] of a `summarize` block"""
    return None if not w else [_r(w.get("value_min"), 5), _r(w.get("value_median"), 5), _r(w.get("value_max"), 5)]


def _roof(rf):
    if not rf:
        return None
    out = {k: _r(rf.get(k), 5) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_us")}
    k = rf.get("kernel")
    if k:
        out["kernel"] = k.replace("void gpf::", "").replace(" ", "")
    for extra in ("lds_bank_conflict_frac_of_lds_cycles", "lds_pipe_busy_frac"):
        if rf.get(extra) is not None:
            out[{"lds_bank_conflict_frac_of_lds_cycles": "lds_conflict_frac", "lds_pipe_busy_frac": "lds_busy_frac"}[extra]] = _r(rf[extra], 3)
    return out


def _cfg_line(rec, value_key="value", extra=()):
    """one BASELINE config: value, unit, [min, median, max] over its windows, roofline fraction, oracle verdict"""
    if not rec:
        return None
    out = {"value": _r(rec.get(value_key), 5)}
    unit = (rec.get("unit") or "").split(" (")[0].split(",")[0]
    if unit != "env steps/sec":                               # (the headline's unit goes without saying: the line has a size target)
        out["unit"] = unit
    if rec.get("windows"):
        out["min_med_max"] = _win3(rec["windows"])          # over N_WIN_CFG = 5 timed windows
    rf = rec.get("roofline")
    if rf:
        out["bound"], out["frac"] = rf.get("bound"), _r(rf.get("frac"), 4)
        if rf.get("traffic_over_algorithmic") is not None:
            out["traffic_over_algorithmic"] = _r(rf["traffic_over_algorithmic"], 3)
    chk = rec.get("oracle_check")
    if chk:
        out["oracle_ok"] = chk.get("ok")
    for k in extra:
        if rec.get(k) is not None:
            out[k] = _r(rec[k], 5)
    cb = rec.get("cpu_baseline")
    if cb:                                                    # [1 core, all usable cores, their number]: the sparse port on this config's grid (lane power flows / s)
        ac = cb.get("all_cores") or {}
        out["cpu"] = [_r(cb.get("value"), 4), _r(ac.get("value"), 4), ac.get("cores")]
    return out


def _all_checks(node, acc):
    if isinstance(node, dict):
        if "max_abs_err_vs_oracle" in node and "ok" in node:
            acc.append(node)
        for v in node.values():
            _all_checks(v, acc)
    elif isinstance(node, list):
        for v in node:
            _all_checks(v, acc)


def compact_record(res, full_path=None):
    """The LAST (and only) stdout line of bench.py: the contract's keys, `roofline`, `cpu_baseline`, one number + roofline fraction
    per BASELINE config, the verdict of the oracle spot checks.  Shrinks itself below COMPACT_LIMIT bytes (drops optional detail)."""
    cfg = res.get("config", {})
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    obs = cfg.get("observations") or ""
    out["config"] = {"workload": (cfg.get("workload") or "").split(" (row")[0], "env": cfg.get("env"), "lanes_per_gpu": cfg.get("lanes_per_gpu"),
                     "total_lanes": cfg.get("total_lanes"), "env_steps_per_launch": cfg.get("env_steps_per_launch"),
                     "observations": "every env step -> HBM" if obs.startswith("every env step") else "last step of a launch only",
                     "cascade": cfg.get("cascade"), "kernels": cfg.get("kernels"), "parallelism": f"static lane shards x{res.get('n_gpus')}, no collective"}
    w = res.get("windows") or {}
    out["windows"] = {"n": w.get("n"), "steps_each": w.get("steps_each"), "min_med_max": _win3(w), "timed": "wall clock of K steps, barrier+sync brackets, MAX over ranks, median"}
    if res.get("value_hip_event_window") is not None:        # (`value` itself is the wall-clock figure)
        out["value_hip_event_window"] = _r(res["value_hip_event_window"], 6)
    out["roofline"] = _roof(res.get("roofline"))
    cb = res.get("cpu_baseline")
    if cb:
        ac = cb.get("all_cores") or {}
        out["cpu_baseline"] = {"value": _r(cb.get("value"), 5), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "solver": "sparse LU", "sample": (cb.get("sample") or "")[:40],
                               "all_cores": {"value": _r(ac.get("value"), 5), "cores": ac.get("cores")} if ac else None,
                               "dense_1core": _r((cb.get("dense_1core") or {}).get("value"), 4),
                               "per_config": "configs[*].cpu = [1 core, all cores, n] lane PF/s",
                               "pandapower": "unavailable (not installed)", "lightsim2grid": "unavailable (not installed)"}
    else:
        out["cpu_baseline"] = None
    dc = res.get("dc_ptdf") or {}
    rows = dc.get("chronics_rows_per_launch") or {}
    configs = {
        "n1_fanout_36sub": _cfg_line(res.get("n1_fanout"), extra=("lane_power_flows_per_sec",)),
        "n1_fanout_118sub": _cfg_line(res.get("n1_fanout_118"), extra=("lane_power_flows_per_sec",)),
        "wcci_118sub": _cfg_line(res.get("secondary")),
        "wcci_env_dynamics": _cfg_line(res.get("secondary_env_dynamics")),
        "idf_ac_118sub": _cfg_line(dc.get("ac_env_steps")),
        "ptdf_rows": _cfg_line(rows, extra=("rows_per_launch",)),
        "ptdf_1row": _cfg_line(dc) if dc else None,
        "ptdf_build_batch": _cfg_line(res.get("ptdf_build_batch"), extra=("classes", "call_value", "call_value_unseen_topologies")),
    }
    act = (res.get("secondary_env_dynamics") or {}).get("acting_every_step")
    if act and configs["wcci_env_dynamics"]:                # agents acting at EVERY step: actions from the host / written on the device
        configs["wcci_env_dynamics"]["acting_every_step"] = {k: _r(act[k]["value_median"], 4) for k in ("host_actions", "device_actions") if k in act}
    out["configs"] = {k: v for k, v in configs.items() if v}
    side = {}
    for k in ("shipped_kernels", "one_launch_per_step", "rollout_last_observation_only", "cascade_on", "cascade_tripping", "split_topologies"):
        if res.get(k):
            side[k] = _r(res[k].get("value_median"), 5)
    if (res.get("one_launch_per_step") or {}).get("roofline"):
        side["one_launch_per_step_traffic_over_algorithmic"] = _r(res["one_launch_per_step"]["roofline"].get("traffic_over_algorithmic"), 3)
    if res.get("simulate_batch"):
        side["simulate_pairs_per_sec"] = _r(res["simulate_batch"].get("value"), 5)
    if res.get("single_env_runpf"):
        side["single_env_runpf_us"] = _r(res["single_env_runpf"].get("us_per_call_ac"), 4)
    if side:
        out["same_workload_variants"] = side
    checks = []
    _all_checks(res, checks)
    if checks:
        errs = [c["max_abs_err_vs_oracle"] for c in checks if c.get("max_abs_err_vs_oracle") is not None]
        out["parity"] = {"oracle_checks": len(checks), "all_ok": all(bool(c.get("ok")) for c in checks), "max_abs_err_vs_oracle": _r(max(errs), 4) if errs else None,
                         "max_flow_err_pu_f64": _r(max([c["max_flow_err_pu_f64"] for c in checks if c.get("max_flow_err_pu_f64") is not None], default=None), 3),
                         "tolerance": "f32 outputs 2e-4+5e-6|x|; f64 flows <1e-4 pu; status, n_iter, topo_vect bit-exact"}
    out["frac_converged"] = _r(res.get("frac_converged"), 6)
    out["mean_nr_iterations"] = _r(res.get("mean_nr_iterations"), 4)
    sp = res.get("specialization") or {}
    if isinstance(sp, dict) and sp:
        out["specialization"] = {k: _r(sp.get(k), 3) for k in ("enabled", "aot", "compiled", "cached", "failed", "seconds") if sp.get(k) is not None}
    out["full_record"] = full_path
    # shrink until it fits (it does with room to spare; this is the guard the CPU test exercises with inflated records)
    for drop in ("same_workload_variants", "specialization", "parity"):
        if len(json.dumps(out)) < COMPACT_LIMIT:
            break
        out.pop(drop, None)
    if len(json.dumps(out)) >= COMPACT_LIMIT:
        for v in out.get("configs", {}).values():
            for k in ("min_med_max", "traffic_over_algorithmic"):
                v.pop(k, None)
    return out


def write_full_record(res, path=None):
    """Everything the run measured -> bench_full.json (next to bench.py, or $GRIDPF_BENCH_FULL); returns the path written or None."""
    path = path or os.environ.get("GRIDPF_BENCH_FULL") or os.path.join(ROOT, "bench_full.json")
    try:
        with open(path, "w") as f:
            json.dump(res, f)
        scratch = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(scratch) and os.path.dirname(os.path.abspath(path)) != scratch:
            with open(os.path.join(scratch, "bench_full.json"), "w") as f:
                json.dump(res, f)
        return os.path.relpath(path, ROOT)
    except OSError as exc:
        sys.stderr.write(f"bench.py: could not write the full record to {path}: {exc}\n")
        return None


def emit(res):
    """full record -> file, compact record -> the single stdout line"""
    line = json.dumps(compact_record(res, write_full_record(res)))
    print(line, flush=True)
    return line


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each (the median one is reported)")
    ap.add_argument("--steps-per-launch", type=int, default=DEFAULT_STEPS_PER_LAUNCH,
                    help="env steps per kernel launch (gpf_step_n): every step does the full work and writes its results; between "
                         "the steps of a launch the lane state stays on chip (1 = one launch per step, also reported)")
    ap.add_argument("--env", default="l2rpn_case14_sandbox")
    ap.add_argument("--batch", type=int, default=4096, help="lanes per GPU")
    ap.add_argument("--cascade", action="store_true", help="headline with overflow disconnections (cascade loop) enabled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--n1", action="store_true",
                    help="headline = BASELINE.json configs[2] shape: every env copy is stepped together with its N-1 contingencies "
                         "(one extra lane per line, that line forced off), all fused into the same launch")
    ap.add_argument("--last-obs-only", action="store_true",
                    help="headline WITHOUT the per-step observation trajectory: each step of a launch overwrites the lane's result "
                         "row, only the last observation of a launch reaches HBM (the round-2 headline; now a labelled secondary)")
    ap.add_argument("--no-oracle-check", action="store_true", help="skip the oracle spot checks after the timed workloads")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank drives HIP device 0 (with --dist-backend gloo: the REAL multi-rank path -- init, per-rank lane "
                         "blocks, barrier, max-reduce, JSON aggregation -- on a box with one GPU; not a scaling measurement)")
    ap.add_argument("--dump-results", default=None, help="rank r writes its lanes' final result rows to <path>.rank<r>.npz")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default) or gloo")
    ap.add_argument("--only-env-dynamics", action="store_true",
                    help="developer (profiling): run only the 118-substation workload with the injection dynamics on and print its record")
    ap.add_argument("--only", default=None, choices=["n1_fanout", "n1_fanout_118", "secondary", "dc_ptdf", "secondary_env_dynamics", "ptdf_build_batch", "simulate"],
                    help="developer (profiling): run only that BASELINE config's workload exactly as the default run does and print its record")
    ap.add_argument("--profile", action="store_true",
                    help="developer (rocprofv3 runs): no pre-roll launches of odd sizes and no last-observation-only sibling windows, so that "
                         "every dispatch of the step kernel is a full --steps-per-launch launch with the observation trajectory on")
    ap.add_argument("--no-jit", action="store_true",
                    help="run on the shipped (ahead-of-time) kernels only.  Default: the engines switch their step launches to kernels compiled at run "
                         "time for the workload's grid (gpf_jit_enable: sizes / offsets as literals, bit-identical results, self-tested against the "
                         "shipped kernels); the shipped-kernel headline is then reported beside it as `shipped_kernels`")
    ap.add_argument("--dump-aot", default=None, help="developer: write the generated headers + launched kernel variants of every engine to this directory")
    ap.add_argument("--stub-engine", action="store_true", help=argparse.SUPPRESS)   # CPU launcher test: no arithmetic
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    ctx = Ctx(args)
    args.steps_per_launch = launch_length(args.steps, args.steps_per_launch)
    if ctx.world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ctx.world}: launch with --nproc-per-node {args.gpus} "
                         f"(or without torchrun: bench.py starts the ranks itself)\n")
        sys.exit(2)
    world, rank = ctx.world, ctx.rank

    if args.only_env_dynamics:
        dyn = workload_wcci_dynamics(ctx, "l2rpn_wcci_2022_dev", 1024, max(16, args.steps), max(16, args.warmup))
        if rank == 0:
            print(json.dumps(dyn))
        return
    if args.only:
        k_o, w_o = max(16, args.steps), max(2, args.warmup)
        rec = {"n1_fanout": lambda: workload_n1(ctx, "l2rpn_neurips_2020_track1", 1024, k_sec=k_o),
               "secondary": lambda: workload_wcci(ctx, "l2rpn_wcci_2022_dev", 1024, k_o, w_o, args.cascade),
               "secondary_env_dynamics": lambda: workload_wcci_dynamics(ctx, "l2rpn_wcci_2022_dev", 1024, k_o, w_o),
               "dc_ptdf": lambda: workload_ptdf(ctx, "l2rpn_idf_2023", 2048, 50, k_sec=k_o, w_sec=w_o),
               "n1_fanout_118": lambda: workload_n1(ctx, "l2rpn_wcci_2022_dev", 1024, k_sec=2 * N1_118_SPL, profile=TRAFFIC_N1_118, spl=N1_118_SPL),
               "ptdf_build_batch": lambda: workload_ptdf_build_batch(ctx, "l2rpn_idf_2023", 2048, 256),
               "simulate": lambda: workload_simulate(ctx, args.env, 256, 16)}[args.only]()
        if rank == 0:
            rec["specialization"] = ctx.specialization()
            print(json.dumps(rec))
        return
    m, ch = load_env(args.env)
    n_envs = args.batch
    fan = 1 + m.n_line if args.n1 else 1
    eng, T, lane0 = setup_engine(ctx, m, ch, n_envs, fan)
    B = n_envs * fan
    if args.n1:
        # lane (k, c): env k, contingency c (c = 0: intact grid, c >= 1: line c-1 forced off); the contingencies of one
        # env sit on the same GPU and share its chronics row / jitter
        topo = np.tile(m.initial_topo_vect(), (B, 1))
        for c in range(1, fan):
            topo[c::fan, m.line_or_pos_topo_vect[c - 1]] = -1
            topo[c::fan, m.line_ex_pos_topo_vect[c - 1]] = -1
        eng.set_topology(topo)
    step_kw = dict(rebalance=1.02, cascade=args.cascade)
    obs_every_step = args.steps_per_launch > 1 and not args.last_obs_only and not args.stub_engine
    if obs_every_step:
        eng.set_trajectory(args.steps_per_launch, eng.TRAJ_OBS)     # every step of a launch writes its observation to HBM

    wins, t_next = timed_windows(ctx, eng, args.steps, args.warmup, step_kw, args.windows, preroll_steps=0 if args.profile else 400)
    # `value`: the contract's clock -- K steps between two barrier + synchronize brackets, MAX over the ranks, median window.  The HIP-event
    # window of the same steps (first launch begins ... last launch ends on the engine's stream) feeds `roofline` (average launch
    # duration of the kernel) and is reported beside it as `value_hip_event_window`.
    elapsed, kern_ms, n_launch, _ = median_window(wins)
    wall_el = median_window(wins, key=3)[3]
    r = eng.results()
    check = oracle_spot_check(ctx, eng, 32, t_last=None if (args.cascade or args.n1) else t_next - 1) if rank == 0 else None
    if args.dump_results:
        np.savez(f"{args.dump_results}.rank{rank}.npz", out=r.out, status=r.status, topo_vect=r.topo_vect, lane0=lane0, t_last=t_next - 1)
    frac_conv = float(r.converged.mean())
    mean_iter = float(r.n_iter[r.converged].mean()) if r.converged.any() else float("nan")
    total_steps = world * n_envs * args.steps
    value = total_steps / wall_el

    res = None
    if rank == 0:
        bytes_step = eng.algorithmic_bytes_per_step()
        avg_launch_s = (kern_ms / max(n_launch, 1)) * 1e-3
        steps_per_launch = args.steps / max(n_launch, 1)
        achieved_gbs = bytes_step * B * steps_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # algorithmic flops per AC power flow: iters*(2/3 J^3 + 2 J^2), dense, J = NR unknowns
        nb = int(eng.results(0, 1).status[0, 2])
        npv = len(set(m.gen_sub[~m.gen_slack].tolist()) - set(m.gen_sub[m.gen_slack].tolist()))
        J = 2 * (nb - 1) - npv
        flops_pf = (mean_iter if mean_iter == mean_iter else 0) * (2.0 / 3.0 * J ** 3 + 2.0 * J ** 2)
        tp = traffic_profile()
        res = {
            "metric": "env steps/sec (batched DoNothing)",
            "value": value,
            "unit": "env steps/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not args.stub_engine else "STUB ENGINE (launcher test, not a measurement)",
            "config": {"workload": f"{args.env} AC Newton-Raphson DoNothing env.step, batch={B} lanes per GPU "
                                   f"(row (t+7k) mod {T}, loads x (1+0.05 N(0,1)), prod_p rebalanced to 1.02 sum(load))",
                       "env": args.env, "lanes_per_gpu": B, "envs_per_gpu": n_envs, "total_lanes": world * B, "n1_fanout": fan,
                       "cascade": bool(args.cascade), "max_iter": 10, "tol_mva": 1e-8, "env_steps_per_launch": args.steps_per_launch,
                       "observations": OBS_EVERY if obs_every_step else
                                       ("every env step (one launch per step)" if args.steps_per_launch == 1 else OBS_LAST),
                       "share_device": bool(args.share_device), "dist_backend": args.dist_backend if ctx.dist is not None else None,
                       "parallelism": f"independent lanes, static shard x{world} (one process per GPU), no collective"},
            "timed_quantity": "wall clock of K steps between barrier + synchronize brackets, MAX over ranks, median window",
            "windows": dict(summarize(wins, total_steps, key=3), steps_each=args.steps,
                            hip_event_windows=summarize(wins, total_steps),
                            note="value / ms_per_step are those of the median window by WALL CLOCK: K steps bracketed by barrier + synchronize on both "
                                 "sides, MAX-reduced over the ranks (the contract's clock; rounds 1-3 and 5 -- round 4 quoted the HIP-event window).  "
                                 "hip_event_windows: the HIP-event window on the engine's stream around the same K steps (first launch begins ... last "
                                 "launch ends; the closing synchronize / barrier is outside it) = value_hip_event_window, the clock of `roofline`"),
            "value_wall_clock": total_steps / wall_el, "value_hip_event_window": total_steps / elapsed,
            "ms_per_step_hip_event_window": elapsed / args.steps * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": tp.get("hbm_bytes_per_launch"),
                         "achieved_is": "ALGORITHMIC bytes per launch (SURVEY.md 8(d): inputs + outputs of one env step at API dtype, x "
                                        "lanes x steps per launch) / average launch duration" +
                                        ("; with the observation trajectory on every step's output row IS written to HBM" if obs_every_step
                                         else "; NOTE only the last step's rows of a launch reach HBM in this mode"),
                         "counter_gbs": (tp.get("hbm_bytes_per_launch") / (tp.get("avg_launch_us") * 1e-6) / 1e9)
                                        if tp.get("hbm_bytes_per_launch") and tp.get("avg_launch_us") else None,
                         "traffic_over_algorithmic": tp.get("traffic_over_algorithmic"),
                         "traffic_source": (f"committed profile {tp.get('_file')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                            f"command; NOT measured in this run)") if tp else None,
                         "kernel": tp.get("kernel"), "avg_launch_us": avg_launch_s * 1e6, "launches": int(n_launch),
                         "env_steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_step": bytes_step,
                         "lds_pipe_busy_frac": tp.get("lds_pipe_busy_frac"),
                         "note": "chains of small FP64 block factorisations in LDS dominate: the kernel is LDS-pipe / issue / latency "
                                 "bound, not HBM bound (SURVEY.md 8(d), DESIGN.md 3); lds_pipe_busy_frac is the PMC figure of the "
                                 "committed profile, f64 the dense-equivalent flop rate",
                         "f64": {"achieved_tflops": flops_pf * B * steps_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0,
                                 "peak_tflops": F64_PEAK_TFLOPS, "algorithmic_flops_per_step": flops_pf, "J": J}},
            "frac_converged": frac_conv,
            "mean_nr_iterations": mean_iter,
            "oracle_check": check,
            "max_abs_err_vs_oracle": check.get("max_abs_err_vs_oracle") if check else None,
            "cpu_baseline": None,
        }

    secondary = not args.no_secondary and args.env == "l2rpn_case14_sandbox" and not args.n1 and not args.stub_engine
    k_sec = max(20, args.steps // 4)
    w_sec = max(2, args.warmup // 4)

    def sib(w_last, n_lanes_total, k):
        return None if w_last is None else dict(summarize(w_last, n_lanes_total * k), unit="env steps/sec", observations=OBS_LAST)

    # ---- the same workload, same engine, same state on the SHIPPED (ahead-of-time) kernels ------------------------------------------
    jit_on = (not args.stub_engine) and hasattr(eng, "specialization") and eng.specialization()["enabled"]
    if jit_on and not args.profile:
        eng.specialize(False)
        w, _ = timed_windows(ctx, eng, k_sec, w_sec, step_kw, 3, preroll_steps=0)
        eng.specialize(True, verify=False)
        if rank == 0:
            res["shipped_kernels"] = dict(summarize(w, world * B * k_sec), unit="env steps/sec", steps_each=k_sec, observations=res["config"]["observations"],
                                          us_per_step=median_window(w)[0] / k_sec * 1e6,
                                          note="the headline workload on the kernels libgridpf.so ships (grid sizes / offsets read from the launch "
                                               "parameter block); `value` runs the same source compiled at run time with this grid's numbers as literals")

    # ---- the same workload WITHOUT the observation trajectory: only the last observation of each launch exists in HBM --------
    if secondary and obs_every_step:
        eng.set_trajectory(0)
        w, _ = timed_windows(ctx, eng, k_sec, w_sec, step_kw, 3, preroll_steps=0)
        eng.set_trajectory(args.steps_per_launch, eng.TRAJ_OBS)
        if rank == 0:
            res["rollout_last_observation_only"] = dict(
                summarize(w, world * B * k_sec), unit="env steps/sec", steps_each=k_sec, us_per_step=median_window(w)[0] / k_sec * 1e6,
                note=f"{args.steps_per_launch} env steps per launch, every step overwrites the lane's result row: a rollout that only "
                     "consumes rho / status (or nothing) between launches; NOT the reference's env.step contract, never `value`")

    # ---- the same workload, ONE launch per env step (an agent that acts between any two steps) -----------------------------
    if secondary and args.steps_per_launch != 1:
        eng.set_trajectory(0)
        w, t_l = timed_windows(ctx, eng, k_sec, w_sec, step_kw, 3, preroll_steps=0, spl=1)
        if rank == 0:
            res["one_launch_per_step"] = dict(summarize(w, world * B * k_sec), unit="env steps/sec", steps_each=k_sec,
                                              observations="every env step (one launch per step: the lane's own rows)",
                                              kept_state="lanes on the reference topology load its derived state (element -> bus maps, Ybus blocks, DC factors) "
                                                         "from one shared blob instead of rebuilding it in every launch (gpf::KeepArgs; GRIDPF_KEEP=0 turns "
                                                         "it off; results bit-identical either way, tests/test_gpu_keep.py)",
                                              us_per_step=median_window(w)[0] / k_sec * 1e6,
                                              roofline=roofline_block(eng, w, B, k_sec, TRAFFIC_1PL,
                                                                      note="one env step per launch: the float64 bus voltages and the injection row go "
                                                                           "back to HBM every step (2.3 x the algorithmic bytes)"),
                                              oracle_check=oracle_spot_check(ctx, eng, 32, t_last=t_l - 1, seed=1))

    # ---- OPT-IN, NOT the reference's algorithm (never the headline): Newton warm-started from the previous step -----------------
    if secondary and args.steps_per_launch != 1:
        eng.reset()
        w, w_l, _ = measure_modes(ctx, eng, k_sec, w_sec, dict(step_kw, warm_start=True), 3, 0, last_obs_too=False)
        rw = eng.results()
        if rank == 0:
            res["warm_start_opt_in"] = dict(summarize(w, world * B * k_sec), unit="env steps/sec", steps_each=k_sec, observations=OBS_EVERY,
                                            note="gpf_step_opts.warm_start=1: steps 2..n of a launch start Newton from the previous "
                                                 "step's voltages instead of pandapower's per-call DC initialisation; same solution "
                                                 "within tol_mva, n_iter differs from the reference's; NOT used for `value`",
                                            mean_nr_iterations=float(rw.n_iter[rw.converged].mean()) if rw.converged.any() else None,
                                            frac_converged=float(rw.converged.mean()))
        eng.reset()

    # ---- the same workload with the reference's DEFAULT parameters: overflow disconnections (cascade) on ---------------------
    if secondary and not args.cascade:
        eng.reset()
        kw_c = dict(rebalance=1.02, cascade=True, auto_reset=True)
        w, w_l, _ = measure_modes(ctx, eng, k_sec, w_sec, kw_c, 3, 0)
        rc = eng.results()
        _, _, n_resets = eng.episode()
        if rank == 0:
            res["cascade_on"] = dict(summarize(w, world * B * k_sec), unit="env steps/sec", steps_each=k_sec, observations=OBS_EVERY,
                                     last_observation_only=sib(w_l, world * B, k_sec),
                                     workload="same lanes, Parameters.NO_OVERFLOW_DISCONNECTION=False (hard_overflow 2.0, "
                                              "NB_TIMESTEP_OVERFLOW_ALLOWED 2): lines trip and the power flow is re-run inside the step; "
                                              "a lane whose step fails (game over) restarts at the next step like env.reset() "
                                              "(auto_reset: without it the dead lanes re-diverge at every step and, the batch being "
                                              "exactly one residency round, their blocks set the duration of every launch)",
                                     frac_converged=float(rc.converged.mean()),
                                     frac_lanes_with_a_tripped_line=float((~rc.line_status).any(axis=1).mean()),
                                     lane_resets=int(np.asarray(n_resets).sum()))
        eng.reset()
        # ... and with thermal limits x CASCADE_LIMIT_SCALE, so that lines really trip and the power flow is re-run inside the step
        if "thermal_limits" in ch:
            eng.set_thermal_limits(ch["thermal_limits"] * np.float32(CASCADE_LIMIT_SCALE))
            w, w_l, _ = measure_modes(ctx, eng, k_sec, w_sec, kw_c, 3, 0, last_obs_too=False)
            rc = eng.results()
            _, _, n_resets = eng.episode()
            rounds = rc.status[:, 3]
            if rank == 0:
                res["cascade_tripping"] = dict(
                    summarize(w, world * B * k_sec), unit="env steps/sec", steps_each=k_sec, observations=OBS_EVERY,
                    workload=f"as cascade_on with thermal limits x {CASCADE_LIMIT_SCALE}: soft overflows accumulate, lines trip, the "
                             "power flow is re-solved inside the step, lanes that end islanded / diverged restart (auto_reset)",
                    frac_converged=float(rc.converged.mean()),
                    frac_lanes_with_a_tripped_line=float((~rc.line_status).any(axis=1).mean()),
                    frac_lanes_resolved_in_last_step=float((rounds > 0).mean()), lane_resets=int(np.asarray(n_resets).sum()),
                    oracle_check=oracle_spot_check(ctx, eng, 32, seed=2, alive_only=True))
            eng.set_thermal_limits(ch["thermal_limits"])
        eng.reset()

    # ---- same workload with topology actions: 10 % of the lanes with a split substation (topology classes) ------
    if secondary:
        topo = np.tile(m.initial_topo_vect(), (B, 1))
        sub = int(np.argmax(m.sub_info))
        start = int(np.concatenate(([0], np.cumsum(m.sub_info)))[sub])
        ends = set(m.line_or_pos_topo_vect.tolist()) | set(m.line_ex_pos_topo_vect.tolist())
        line_pos = [q for q in range(start, start + int(m.sub_info[sub])) if q in ends]
        pick = np.random.default_rng(lane0).random(B) < 0.10
        for q in line_pos[::2][:max(1, len(line_pos) // 2 - 1)]:
            topo[pick, q] = 2
        eng.set_topology(topo)
        w, w_l, _ = measure_modes(ctx, eng, k_sec, w_sec, step_kw, 3, 0)
        conv_s = float(eng.results().converged.mean())
        if rank == 0:
            res["split_topologies"] = dict(summarize(w, world * B * k_sec), unit="env steps/sec", steps_each=k_sec, observations=OBS_EVERY,
                                           last_observation_only=sib(w_l, world * B, k_sec),
                                           oracle_check=oracle_spot_check(ctx, eng, 32, seed=3),
                                           workload=f"{args.env}, batch={B}: substation {sub} split (lines alternating between its two "
                                                    f"busbars) in {100.0 * pick.mean():.0f} % of the lanes", frac_converged=conv_s)
    eng.close()

    # ---- batch sweep (throughput vs lanes per GPU), observation per env step at every point -----------------------------------
    if secondary and world == 1:
        sweep = []
        for Bs in (1024, 2048, 8192, 16384, 65536):
            e_s, _, _ = setup_engine(ctx, m, ch, Bs)
            k_s = max(10, min(k_sec, 200 * 4096 // Bs))
            w, w_l, _ = measure_modes(ctx, e_s, k_s, 2, step_kw, 3, 20)
            sweep.append({"lanes": Bs, "value": Bs * k_s / median_window(w)[0], "us_per_step": median_window(w)[0] / k_s * 1e6,
                          "value_last_observation_only": Bs * k_s / median_window(w_l)[0] if w_l else None})
            e_s.close()
        sweep.append({"lanes": B, "value": res["value_hip_event_window"], "us_per_step": res["ms_per_step_hip_event_window"] * 1e3,
                      "value_last_observation_only": (res.get("rollout_last_observation_only") or {}).get("value_median")})
        res["batch_sweep"] = {"unit": "env steps/sec", "workload": f"{args.env}, same synthetic inputs, 1 GPU", "observations": OBS_EVERY,
                              "points": sorted(sweep, key=lambda d: d["lanes"])}

    # ---- BASELINE.json configs[2]: 36-substation grid, 1024 envs x (1 + 59 N-1 outages) fused into one batch ------------------
    if secondary and world == 1:
        res["n1_fanout"] = workload_n1(ctx, "l2rpn_neurips_2020_track1", 1024, k_sec=max(10, args.steps // 40))

    # ---- configs[2] on a REAL 118-substation grid (BASELINE says "IEEE 118-bus"; the bundled l2rpn_neurips_2020_track1 is a 36-substation
    #      sub-area): 1024 envs x (1 intact + 186 single-line outages) = 191 488 lanes, observation per step ------------------
    if secondary and world == 1:
        res["n1_fanout_118"] = workload_n1(ctx, "l2rpn_wcci_2022_dev", 1024, k_sec=2 * N1_118_SPL, profile=TRAFFIC_N1_118, spl=N1_118_SPL)

    # ---- BASELINE.json configs[3]: 118-substation grid, storage set-points + zero-sum redispatch, 1024 lanes per GPU -----------
    if secondary:
        sec = workload_wcci(ctx, "l2rpn_wcci_2022_dev", 1024, k_sec, w_sec, args.cascade)
        if rank == 0:
            res["secondary"] = sec

    # ---- configs[3] with the environment's injection dynamics evolving INSIDE the launch (storage state of charge, ramp-limited
    #      redispatch re-solved at every step) -- what "storage + redispatch actions" means for agents that act
    if secondary:
        dyn = workload_wcci_dynamics(ctx, "l2rpn_wcci_2022_dev", 1024, k_sec, w_sec)
        if rank == 0:
            res["secondary_env_dynamics"] = dyn

    # ---- batched obs.simulate: B environments x K candidate actions on the 1-step-ahead forecast, one call --------------------
    if secondary and world == 1:
        res["simulate_batch"] = workload_simulate(ctx, args.env, 256, 16)

    # ---- the drop-in boundary at batch 1: what ONE environment's Backend.runpf costs (HipBackend.runpf = one gpf_solve_lane call) ----
    if secondary and world == 1:
        res["single_env_runpf"] = workload_single_env(ctx, args.env, 2000)

    # ---- DC sensitivity path of BASELINE.json configs[4]: l2rpn_idf_2023, 2048 lanes, PTDF GEMM next to the AC solve ------------
    if secondary and world == 1:
        res["dc_ptdf"] = workload_ptdf(ctx, "l2rpn_idf_2023", 2048, max(20, args.steps // 2), k_sec=max(16, k_sec // 4), w_sec=w_sec)

    # ---- configs[4], per-lane topologies: the DC matrices of 256 distinct topologies factorised on the matrix cores in one launch ----
    if secondary and world == 1:
        res["ptdf_build_batch"] = workload_ptdf_build_batch(ctx, "l2rpn_idf_2023", 2048, 256)

    if rank == 0:
        # the CPU baseline is timed LAST (rank 0 of the 1-GPU run only): host-only work in the middle of the run would
        # let the GPU clocks drop before the secondary workloads
        if not args.no_cpu_baseline and world == 1 and not args.stub_engine:
            res["cpu_baseline"] = cpu_baseline(args.env, m, ch, T)
            res["cpu_baseline"]["reference_environment_step_loop"] = reference_step_loop()
            # an honest CPU row for every BASELINE config (the same sparse port on that config's grid)
            for key, env_c in (("n1_fanout", "l2rpn_neurips_2020_track1"), ("n1_fanout_118", "l2rpn_wcci_2022_dev"), ("secondary", "l2rpn_wcci_2022_dev"),
                               ("secondary_env_dynamics", "l2rpn_wcci_2022_dev")):
                if isinstance(res.get(key), dict):
                    res[key]["cpu_baseline"] = cpu_baseline_config(env_c)
            if isinstance((res.get("dc_ptdf") or {}).get("ac_env_steps"), dict):
                res["dc_ptdf"]["ac_env_steps"]["cpu_baseline"] = cpu_baseline_config("l2rpn_idf_2023")
        res["specialization"] = dict(ctx.specialization(),
                                     what="step kernels compiled at run time for each workload's grid (gpf_jit_enable: hipcc --genco of the unchanged "
                                          "kernel source with the grid's sizes / table offsets as literals; results bit-identical to the shipped "
                                          "kernels, self-tested at enable time; compile + load seconds are outside every timed region)"
                                     if not args.no_jit else "off (--no-jit): shipped kernels")
        res["config"]["kernels"] = ("grid-specialised at run time (gpf_jit_enable)" if res["specialization"].get("launches") else "shipped (ahead-of-time)")
        emit(res)
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


def workload_n1(ctx, env, n_envs, k_sec, profile=None, spl=None):
    """`spl`: env steps per launch of this workload (default: --steps-per-launch); the 118-substation fan-out (191 488 lanes) runs
    N1_118_SPL = 16 steps per launch -- its observation trajectory is 2.7 GB per step, 43 GB of the 288 GB (4 per launch: -12 %)"""
    m, ch = load_env(env)
    fan = 1 + m.n_line
    eng, T, _ = setup_engine(ctx, m, ch, n_envs, fan)
    B = n_envs * fan
    topo = np.tile(m.initial_topo_vect(), (B, 1))
    for c in range(1, fan):
        topo[c::fan, m.line_or_pos_topo_vect[c - 1]] = -1
        topo[c::fan, m.line_ex_pos_topo_vect[c - 1]] = -1
    eng.set_topology(topo)
    k_sec = cfg_steps(ctx, k_sec) if spl is None else max(k_sec, 2 * spl)
    w, w_l, _ = measure_modes(ctx, eng, k_sec, 2, dict(rebalance=1.02), N_WIN_CFG, 10 if spl is None else 2 * spl, spl=spl)
    r = eng.results()
    med = median_window(w)[0]
    out = {"workload": f"{env} ({m.n_sub} substations): {n_envs} envs x (1 intact + {m.n_line} single-line outages) = {B} lanes per GPU, the "
                       f"obs.simulate / N1Reward fan-out fused into the stepped batch (BASELINE.json configs[2])",
           "observations": OBS_EVERY if not ctx.args.last_obs_only else OBS_LAST,
           "value": ctx.world * n_envs * k_sec / med, "unit": "env steps/sec (each with its full N-1 screening)",
           "lane_power_flows_per_sec": ctx.world * B * k_sec / med, "ms_per_step": med / k_sec * 1e3, "steps_each": k_sec,
           "windows": summarize(w, ctx.world * n_envs * k_sec),
           "last_observation_only": None if w_l is None else dict(summarize(w_l, ctx.world * n_envs * k_sec), observations=OBS_LAST,
                                                                  lane_power_flows_per_sec=ctx.world * B * k_sec / median_window(w_l)[0]),
           "roofline": roofline_block(eng, w, B, k_sec, profile or TRAFFIC_N1),
           "plan": eng.plan() if hasattr(eng, "plan") else None,
           "frac_converged": float(r.converged.mean()),
           "frac_contingencies_diverged_or_islanding": float(1.0 - r.converged.reshape(n_envs, fan)[:, 1:].mean()),
           "oracle_check": oracle_spot_check(ctx, eng, 64, seed=4)}
    eng.close()
    return out


def workload_wcci(ctx, env, n_envs, k_sec, w_sec, cascade):
    from grid2op_amd.sharding import lane_range
    m, ch = load_env(env)
    eng, T, l0 = setup_engine(ctx, m, ch, n_envs)
    B = n_envs
    note = ""
    if m.n_storage:                                      # storage actions, U(-2, 2) MW per unit
        inj = eng.get_injections()
        lay = eng.layout
        for k in range(B):
            inj[k, lay.inj_storage_p:lay.inj_storage_p + m.n_storage] = np.random.default_rng(l0 + k).uniform(-2.0, 2.0, m.n_storage)
        eng.set_injections(inj)
        note += " with storage set-points U(-2,2) MW"
    if hasattr(eng, "set_lane_redispatch"):                # zero-sum +-1 MW redispatch on 2 random dispatchable generators
        disp = np.nonzero(~m.gen_slack)[0]
        delta = np.zeros((B, m.n_gen), dtype=np.float32)
        for k in range(B):
            a, b = np.random.default_rng(10_000_000 + l0 + k).choice(disp, size=2, replace=False)
            delta[k, a], delta[k, b] = 1.0, -1.0
        eng.set_lane_redispatch(delta)
        note += " and a zero-sum +-1 MW redispatch on 2 random generators per lane"
    k_sec = cfg_steps(ctx, k_sec)
    w, w_l, _ = measure_modes(ctx, eng, k_sec, w_sec, dict(rebalance=1.02, cascade=cascade), N_WIN_CFG, 200)
    med = median_window(w)[0]
    r = eng.results()
    out = None
    if ctx.rank == 0:
        out = {"workload": f"{env} (118 substations) AC NR env.step{note}, batch={B} lanes per GPU (BASELINE.json configs[3])",
               "observations": OBS_EVERY if not ctx.args.last_obs_only else OBS_LAST,
               "value": ctx.world * B * k_sec / med, "unit": "env steps/sec", "ms_per_step": med / k_sec * 1e3, "steps_each": k_sec,
               "windows": summarize(w, ctx.world * B * k_sec),
               "last_observation_only": None if w_l is None else dict(summarize(w_l, ctx.world * B * k_sec), observations=OBS_LAST),
               "roofline": roofline_block(eng, w, B, k_sec, TRAFFIC_WCCI),
               "plan": eng.plan() if hasattr(eng, "plan") else None,
               "frac_converged": float(r.converged.mean()), "mean_nr_iterations": float(r.n_iter[r.converged].mean()),
               "oracle_check": oracle_spot_check(ctx, eng, 32, seed=5)}
    eng.close()
    return out


def acting_every_step(ctx, eng, m, red, sto, t, kw, n=200, n_win=3):
    """Agents that act at EVERY step (one single-step launch per env step, a new redispatch +-1 MW and storage action per lane and
    step): (a) actions handed over from the host (`set_lane_actions`: PCIe upload + synchronisation per step), (b) written on the
    device into the engine's action buffers by torch ops on the engine's stream (`lane_actions_on_device`: nothing crosses PCIe)."""
    torch = ctx.torch
    if torch is None or not ctx.cuda_ok() or not hasattr(eng, "lane_actions_on_device"):
        return None
    B = red.shape[0]
    v = eng.device_views()
    dev = v["act_redispatch"].device
    red_pm = [red, -red]
    red_dev = [torch.from_numpy(a).to(dev) for a in red_pm]
    sto_dev = torch.from_numpy(sto).to(dev)

    def run(on_device, t0, k_steps):
        for k in range(k_steps):
            if on_device:
                with torch.cuda.stream(v["stream"]):
                    v["act_redispatch"].copy_(red_dev[k % 2])
                    v["act_storage"].copy_(sto_dev)
                eng.lane_actions_on_device(redispatch=True, storage_power=True)
            else:
                eng.set_lane_actions(red_pm[k % 2], sto)
            eng.step(t0 + k, n_steps=1, **kw)
        return t0 + k_steps
    res = {}
    for label, on_device in (("host_actions", False), ("device_actions", True)):
        t = run(on_device, t, 20)
        wins = []
        for _ in range(n_win):
            ctx.sync_all(eng)
            w0 = time.perf_counter()
            t = run(on_device, t, n)
            ctx.sync_all(eng)
            wins.append((ctx.max(time.perf_counter() - w0), 0.0, 0))
        res[label] = dict(summarize(wins, ctx.world * B * n), us_per_step=median_window(wins)[0] / n * 1e6)
    res["what"] = ("one single-step launch per env step, a new redispatch + storage action per lane and step; host_actions: set_lane_actions "
                   "(PCIe + synchronisation per step), device_actions: torch copies into device_views()['act_*'] on the engine's stream + "
                   "lane_actions_on_device (no PCIe, no synchronisation); the launches load the reference topology's derived state from the "
                   "shared blob (gpf::KeepArgs, GRIDPF_KEEP=0: off)")
    return res


def workload_wcci_dynamics(ctx, env, n_envs, k_sec, w_sec):
    """configs[3] with gpf_set_env_dynamics: every lane holds a storage action U(-2, 2) MW per unit over the launch and starts it with a
    zero-sum +-1 MW redispatch on two generators; the kernel evolves the state of charge and re-solves the ramp-limited dispatch
    (BaseEnv._compute_dispatch_vect) at every step.  Generator / storage characteristics: tests/golden/envdyn_<env>.npz (recorded from
    the reference environment).  Every step of a launch writes its observation to HBM (trajectory on)."""
    m, ch = load_env(env)
    fxp = os.path.join(GOLD, f"envdyn_{env}.npz")
    if not os.path.exists(fxp) or ctx.args.stub_engine:
        return None
    fx = dict(np.load(fxp))
    eng, T, l0 = setup_engine(ctx, m, ch, n_envs)
    B = n_envs
    eng.set_gen_limits(fx["pmin"], fx["pmax"], fx["ramp_up"], fx["ramp_down"], fx["redispatchable"], eps_poly=float(fx["eps_poly"]))
    eng.set_storage_params(fx["storage_Emax"], fx["storage_Emin"], fx["storage_loss"], fx["storage_charging_efficiency"],
                           fx["storage_discharging_efficiency"], fx["storage_charge0"], float(fx["delta_time_seconds"]),
                           bool(fx["activate_storage_loss"]))
    eng.set_env_dynamics(True, tol_poly=float(fx["tol_poly"]))
    disp = np.nonzero(fx["redispatchable"] & ~m.gen_slack)[0]
    red = np.zeros((B, m.n_gen), np.float32)
    sto = np.zeros((B, m.n_storage), np.float32)
    for k in range(B):
        rg = np.random.default_rng(20_000_000 + l0 + k)
        a, b = rg.choice(disp, size=2, replace=False)
        red[k, a], red[k, b] = 1.0, -1.0
        sto[k] = rg.uniform(-2.0, 2.0, m.n_storage)
    spl = ctx.args.steps_per_launch
    kw = dict(rebalance=1.02, auto_reset=True)
    traj = spl > 1 and not ctx.args.last_obs_only
    if traj:
        eng.set_trajectory(spl, eng.TRAJ_OBS)

    def run(t, n):
        done = 0
        while done < n:
            k = min(spl, n - done)
            eng.set_lane_actions(red, sto, hold_storage=True)      # the agents act at every launch boundary
            eng.step(t + done, n_steps=k, **kw)
            done += k
        return t + n
    k_sec = cfg_steps(ctx, k_sec)
    t = run(0, max(w_sec, 10 * spl))          # (pre-roll: the clocks of an idle MI355X settle within ~10 ms of work)
    wins = []
    for _ in range(N_WIN_CFG):
        ctx.sync_all(eng)
        w0 = time.perf_counter()
        t = run(t, k_sec)
        ctx.sync_all(eng)
        wins.append((ctx.max(time.perf_counter() - w0), 0.0, 0))
    med = median_window(wins)[0]
    r = eng.results()
    st = eng.env_state()
    acting = acting_every_step(ctx, eng, m, red, sto, t, kw)
    out = None
    if ctx.rank == 0:
        out = {"workload": f"{env} (118 substations), batch={B} lanes per GPU, environment injection dynamics ON (gpf_set_env_dynamics): per lane a "
                           f"held storage action U(-2,2) MW per unit and a zero-sum +-1 MW redispatch at every launch boundary ({spl} steps); the "
                           "state of charge and the ramp-limited dispatch (BaseEnv._compute_dispatch_vect, exact QP) evolve at every step "
                           "inside the launch (BASELINE.json configs[3]: storage + redispatch actions); actions uploaded from the host per launch",
               "observations": OBS_EVERY if traj else OBS_LAST, "timing": "wall clock incl. the per-launch upload of the agents' actions",
               "value": ctx.world * B * k_sec / med, "unit": "env steps/sec", "ms_per_step": med / k_sec * 1e3, "steps_each": k_sec,
               "windows": summarize(wins, ctx.world * B * k_sec), "frac_converged": float(r.converged.mean()),
               "frac_infeasible_redispatch": float((r.status[:, 0] == 6).mean()),
               "mean_abs_actual_dispatch_mw": float(np.abs(st["actual"]).mean()), "mean_state_of_charge_mwh": float(st["charge"].mean()),
               "acting_every_step": acting,
               "oracle_check": oracle_spot_check(ctx, eng, 32, seed=7, alive_only=True)}
    eng.close()
    return out


def workload_ptdf_rows(ctx, eng, m, B, t0, n_rows=16, reps=20):
    """configs[4], the DC path at the CONFIGURED batch: the flows of `n_rows` consecutive chronics rows of all B lanes as ONE FP64-MFMA
    GEMM with M = B x n_rows (gpf_ptdf_flows_rows: the injections are gathered from the device-resident chronics table in the
    launch's prologue, as the step kernel's K9 does) -- the next `n_rows` DoNothing env steps of the batch in the DC approximation."""
    lay = eng.layout
    for _ in range(3):
        eng.ptdf_flows_rows(t0, n_rows, rebalance=1.02, fetch=False)
    eng.sync()
    eng.set_profiling(1)
    t1 = time.perf_counter()
    for _ in range(reps):
        eng.ptdf_flows_rows(t0, n_rows, rebalance=1.02, fetch=False)
    eng.sync()
    el = time.perf_counter() - t1
    k_ms, n_l = eng.kernel_time()
    eng.set_profiling(0)
    flows = eng.ptdf_flows_rows(t0, n_rows, rebalance=1.02)
    r_dc = eng.results(0, 1)
    chk = None
    if not ctx.args.no_oracle_check and not ctx.args.stub_engine:
        try:                       # CHECKER leg: 64 (lane, row) pairs vs the C oracle's DC power flow of the K9 injections of that row
            from oracle.pf_oracle_c import COracle
            tab, off, sc = eng.bench_inputs
            rg = np.random.default_rng(16)
            ls_, rs_ = rg.choice(B, 64, replace=False), rg.integers(0, n_rows, 64)
            nl, ng = m.n_load, m.n_gen
            inj0 = eng.get_injections()
            rows = []
            for k, j in zip(ls_, rs_):
                row = tab[(t0 + j + off[k]) % tab.shape[0]]
                lp = row[:nl] * sc[k, :nl]
                pp = row[2 * nl:2 * nl + ng].copy()
                ns = ~m.gen_slack
                pp[ns] = pp[ns] * np.float32(1.02 * lp.astype(np.float64).sum() / row[2 * nl:2 * nl + ng][ns].astype(np.float64).sum())
                x = inj0[k].copy()
                x[lay.inj_load_p:lay.inj_load_p + nl] = lp
                x[lay.inj_gen_p:lay.inj_gen_p + ng] = pp
                rows.append(x)
            topo_, sb_ = eng.get_topology(0, 1)
            ref = COracle(m).solve_rows(np.asarray(rows), np.tile(topo_, (64, 1)), np.tile(sb_, (64, 1)) if m.n_shunt else None, is_dc=True)
            p_ref = ref["out"][:, lay.out_p_or:lay.out_p_or + m.n_line]
            got = flows[rs_, ls_]
            chk = {"n": 64, "max_abs_err_vs_oracle": float(np.abs(got - p_ref).max()), "ok": bool(np.all(np.abs(got - p_ref) <= 2e-4 + 5e-6 * np.abs(p_ref))),
                   "against": "oracle/pf_oracle.c DC power flow of the chronics row's injections, 64 (lane, row) pairs"}
        except Exception as exc:
            chk = {"error": repr(exc)[:300]}
    nb_act = int(np.count_nonzero(~np.isnan(eng.results(0, 1).bus_vm[0])))
    nb_pad, line_pad = (nb_act + 3) // 4 * 4, (m.n_line + 15) // 16 * 16
    us = k_ms / max(n_l, 1) * 1e3
    M = B * n_rows
    tr_b, tr_src = ptdf_traffic(f"ptdf_rows_kernel<2>, {n_rows} chronics rows x {B:,}".replace(",", " "))
    tf = 2.0 * M * nb_pad * line_pad / (us * 1e-6) / 1e12 if us > 0 else 0.0
    hbm = 4.0 * M * line_pad + 4.0 * M * (m.n_load + m.n_gen)            # flows out + chronics values in (the table itself is L2 resident)
    return {"workload": f"{n_rows} consecutive chronics rows of all {B} lanes per launch: flows = P_bus[{M}x{nb_pad}] . PTDF^T[{nb_pad}x{line_pad}], "
                        "injections gathered from the device-resident chronics table in the launch (gpf_ptdf_flows_rows)",
            "value": M * reps / el, "unit": "DC power flows/sec (one per lane and chronics row)", "us_per_launch": us, "rows_per_launch": n_rows,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_PEAK_TFLOPS,
                         "hbm_gbs": hbm / (us * 1e-6) / 1e9 if us > 0 else 0.0, "traffic": tr_b, "traffic_source": tr_src,
                         "flops_per_launch": 2.0 * M * nb_pad * line_pad, "avg_launch_us": us, "kernel": "ptdf_rows_kernel<2>"},
            "oracle_check": chk}


def oracle_dc_check(ctx, eng, flows, lanes):
    """CHECKER leg (never timed): the lanes' DC flows vs the C oracle's DC power flow of the same injection rows on each lane's OWN topology;
    topologies the oracle reports as islanded must come back as NaN rows."""
    if ctx.args.no_oracle_check or ctx.args.stub_engine:
        return None
    try:
        from oracle.pf_oracle_c import COracle
        m = eng.model
        inj = eng.get_injections()
        tp, sbv = eng.get_topology()
        ref = COracle(m).solve_rows(inj[lanes], tp[lanes], sbv[lanes] if m.n_shunt else None, is_dc=True)
        lay = eng.layout
        p_ref = ref["out"][:, lay.out_p_or:lay.out_p_or + m.n_line]
        bad_ref = ref["status"][:, 0] != 0
        got = flows[lanes]
        isl_ok = bool(np.array_equal(np.isnan(got).any(axis=1), bad_ref))
        g_, r_ = got[~bad_ref], p_ref[~bad_ref]
        return {"n": int(len(lanes)), "n_islanded_in_sample": int(bad_ref.sum()), "islanding_verdicts_equal": isl_ok,
                "max_abs_err_vs_oracle": float(np.abs(g_ - r_).max()), "ok": bool(isl_ok and np.all(np.abs(g_ - r_) <= 2e-4 + 5e-6 * np.abs(r_))),
                "against": "oracle/pf_oracle.c DC power flow of the same injection rows on each lane's own topology"}
    except Exception as exc:
        return {"error": repr(exc)[:300]}


def synthetic_topologies(m, n_topo, rng, max_out=2, max_split=2):
    """`n_topo` distinct topology rows of an agent population: up to `max_out` lines out, up to `max_split` substations split over two busbars
    (every other element to busbar 2).  Combinations that island the grid are kept: the engine must report them, not solve them."""
    pos_sub = np.empty(m.dim_topo, dtype=np.int64)
    for pos, sub in ((m.line_or_pos_topo_vect, m.line_or_sub), (m.line_ex_pos_topo_vect, m.line_ex_sub), (m.gen_pos_topo_vect, m.gen_sub),
                     (m.load_pos_topo_vect, m.load_sub)) + (((m.storage_pos_topo_vect, m.storage_sub),) if m.n_storage else ()):
        pos_sub[pos] = sub
    big = [s_ for s_ in range(m.n_sub) if (pos_sub == s_).sum() >= 4]
    base = m.initial_topo_vect()
    seen, out = set(), []
    while len(out) < n_topo:
        t = base.copy()
        for l in rng.choice(m.n_line, int(rng.integers(0, max_out + 1)), replace=False):
            t[m.line_or_pos_topo_vect[l]] = -1
            t[m.line_ex_pos_topo_vect[l]] = -1
        for s_ in rng.choice(big, int(rng.integers(0, max_split + 1)), replace=False):
            sel = np.nonzero(pos_sub == s_)[0][int(rng.integers(0, 2))::2]
            t[sel] = np.where(t[sel] >= 1, 2, t[sel])
        if t.tobytes() not in seen:
            seen.add(t.tobytes())
            out.append(t)
    return out


def workload_ptdf_build_batch(ctx, env, B, n_topo, reps=5):
    """BASELINE.json configs[4] ("MFMA batched dense solve") with PER-LANE topologies: `B` lanes of the 118-substation grid hold `n_topo`
    distinct topologies (line outages + bus splits); gpf_ptdf_build_batch factorises the DC matrix of every distinct topology ON THE
    DEVICE (one workgroup per class, blocked Gauss-Jordan on the FP64 matrix cores, PTDF / LODF formed there), then every lane's DC flows
    are PTDF_class(lane) x P_bus.  Timed: the build kernel (HIP events) and the whole call (host grouping + upload + kernel); beside it the
    round-3 host path (gpf_ptdf_build: one topology per call, Gauss-Jordan on a host core)."""
    if ctx.args.stub_engine:
        return None
    m, ch = load_env(env)
    eng, T, l0 = setup_engine(ctx, m, ch, B)
    rng = np.random.default_rng(77)
    topos = synthetic_topologies(m, n_topo, rng)
    lane_topo = np.concatenate([np.arange(n_topo), rng.integers(0, n_topo, B - n_topo)])
    rng.shuffle(lane_topo)
    eng.set_topology(np.stack([topos[i] for i in lane_topo]).astype(np.int32))
    eng.step(3, n_steps=1, rebalance=1.02)                       # the lanes hold the injections of their chronics row
    info = eng.ptdf_build_batch(with_lodf=True)                   # warm-up (allocations, LDS attribute)
    # THE CALL A USER MAKES, end to end: host grouping + descriptors + upload + kernel + class status back (ptdf_build_batch returns the
    # status: it waits for the kernel); beside it the ENQUEUE time (info=False: the call returns once the kernel is queued -- the flows /
    # screening calls that consume the tables queue behind it on the same stream)
    eng.sync()
    k_ms, w_ms, q_ms = [], [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        info = eng.ptdf_build_batch(with_lodf=True)
        w_ms.append((time.perf_counter() - t0) * 1e3)
        k_ms.append(info["kernel_ms"])
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.ptdf_build_batch(with_lodf=True, info=False)
        q_ms.append((time.perf_counter() - t0) * 1e3)
        eng.sync()
    k_med, w_med = float(np.median(k_ms)), float(np.median(w_ms))
    os.environ["GRIDPF_PTDFB_NO_CACHE"] = "1"                   # the same call when NONE of the topologies has been seen before
    w_new, q_new = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.ptdf_build_batch(with_lodf=True)
        w_new.append((time.perf_counter() - t0) * 1e3)
    for _ in range(5):
        t0 = time.perf_counter()
        eng.ptdf_build_batch(with_lodf=True, info=False)
        q_new.append((time.perf_counter() - t0) * 1e3)
        eng.sync()
    del os.environ["GRIDPF_PTDFB_NO_CACHE"]
    info = eng.ptdf_build_batch(with_lodf=True)
    ok_cls = info["class_status"] == 0
    npad = (np.maximum(info["class_n"], 1) + 15) // 16 * 16
    flops_padded = float((2.0 * npad[ok_cls].astype(np.float64) ** 3).sum())
    flops = float((2.0 * info["class_n"][ok_cls].astype(np.float64) ** 3).sum())      # ALGORITHMIC: 2 n^3, n = the reduced dimension itself (117 -> 128 in tiles is x 1.31)
    tf = flops / (k_med * 1e-3) / 1e12
    # the host path beside it: gpf_ptdf_build of 8 of the same topologies (one lane each)
    reps_h = [int(np.nonzero(info["lane_class"] == c)[0][0]) for c in np.nonzero(ok_cls)[0][:8]]
    t0 = time.perf_counter()
    for k in reps_h:
        eng.ptdf_build(k)
    host_s = (time.perf_counter() - t0) / max(len(reps_h), 1)
    info = eng.ptdf_build_batch(with_lodf=True)                   # back to the per-lane tables
    for _ in range(3):
        eng.ptdf_flows(fetch=False)
    eng.sync()
    eng.set_profiling(1)
    for _ in range(20):
        eng.ptdf_flows(fetch=False)
    eng.sync()
    f_ms, f_n = eng.kernel_time()
    eng.set_profiling(0)
    flows = eng.ptdf_flows()
    t0 = time.perf_counter()
    for _ in range(3):
        worst = eng.lodf_screen()
    lodf_s = (time.perf_counter() - t0) / 3
    chk = oracle_dc_check(ctx, eng, flows, np.sort(np.random.default_rng(78).choice(B, 64, replace=False)))
    out = {"workload": f"{env} (118 substations), {B} lanes holding {n_topo} distinct topologies (0-2 line outages + 0-2 split substations each): "
                       "gpf_ptdf_build_batch = B' of every distinct topology assembled, inverted (blocked Gauss-Jordan, FP64 MFMA 16x16x4 tiles) and turned "
                       "into PTDF^T + LODF on the device, one workgroup per topology (BASELINE.json configs[4]: MFMA batched dense solve)",
           "classes": int(info["n_classes"]), "classes_ok": int(ok_cls.sum()), "classes_islanded": int((info["class_status"] == 2).sum()),
           "reduced_dimension": {"min": int(info["class_n"].min()), "max": int(info["class_n"].max())},
           "value": float(ok_cls.sum()) / (k_med * 1e-3), "unit": "topologies factorised/sec (build kernel)", "kernel_ms": k_med, "kernel_ms_all": [round(x, 4) for x in k_ms],
           "call_ms": w_med, "call_value": float(ok_cls.sum()) / (w_med * 1e-3), "call_is": "the whole call a user makes: grouping of the lanes' topology rows + class descriptors ON THE DEVICE "
           "(gridpf_ptdf_group.hpp: hash, one-workgroup sort, verification, one workgroup per class; six integers read back) + factorisation kernel + class status / lane -> class map back",
           "call_ms_unseen_topologies": float(np.median(w_new)), "call_value_unseen_topologies": float(ok_cls.sum()) / (float(np.median(w_new)) * 1e-3),
           "enqueue_ms": float(np.median(q_ms)), "enqueue_ms_unseen_topologies": float(np.median(q_new)),
           "enqueue_is": "the same call with info=False: it returns once the kernel is queued (host grouping + descriptors + upload); consumers queue behind it on the stream",
           "host_builds_per_sec": 1.0 / host_s, "host_is": "gpf_ptdf_build: the same matrices by Gauss-Jordan on ONE host core (round-3 path), per topology",
           "speedup_vs_host_path": (float(ok_cls.sum()) / (w_med * 1e-3)) * host_s,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_PEAK_TFLOPS,
                        "traffic": ptdf_traffic("ptdf_build_lds_kernel")[0], "traffic_source": ptdf_traffic("ptdf_build_lds_kernel")[1],
                        "hbm_gbs": (ptdf_traffic("ptdf_build_lds_kernel")[0] or 0.0) / (k_med * 1e-3) / 1e9,
                        "flops_per_launch": flops, "flops_are": "ALGORITHMIC: 2 n^3 per class, n = the reduced dimension (unpadded)", "flops_padded_to_tiles": flops_padded,
                        "frac_on_padded_flops": flops_padded / (k_med * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                        "avg_launch_us": k_med * 1e3, "kernel": "ptdf_build_lds_kernel (reduced dimension <= 128: matrix in LDS)"},
           "flows_per_lane_topology": {"value": B / (f_ms / max(f_n, 1) * 1e-3), "unit": "DC power flows/sec (each lane x its own PTDF)", "us_per_batch": f_ms / max(f_n, 1) * 1e3},
           "lodf_n1_value": B * m.n_line / lodf_s, "lodf_n1_unit": "DC contingency cases/sec, every lane screened against the LODF of ITS topology (result copied to the host)",
           "oracle_check": chk}
    eng.close()
    return out


def workload_simulate(ctx, env, n_envs, n_act):
    """Batched obs.simulate (gpf_simulate_batch): `n_envs` environments x `n_act` candidate actions (do nothing + single-line
    disconnections) on the 1-step-ahead forecast, ONE call = host topology bookkeeping + device copy + one launch.  Synthetic forecast
    tables: the next chronics row."""
    if ctx.args.stub_engine:
        return None
    m, ch = load_env(env)
    eng, T, l0 = setup_engine(ctx, m, ch, n_envs * (1 + n_act))          # lanes [0, n_envs) = environments, the rest scratch
    tab = eng.bench_inputs[0]
    eng.upload_forecasts(np.roll(tab, -1, axis=0)[None])
    cands = [{}] + [{"set_line_status": [(l, -1)]} for l in range(min(n_act - 1, m.n_line))]
    eng.step(0, n_steps=4, rebalance=1.02)
    src = np.arange(n_envs)
    kw = dict(rebalance=1.02, cascade=True)
    for _ in range(3):
        eng.simulate_batch(3, src, cands, dst_lane0=n_envs, time_step=1, **kw)
    eng.sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.simulate_batch(3, src, cands, dst_lane0=n_envs, time_step=1, **kw)
    eng.sync()
    el = time.perf_counter() - t0
    r = eng.results(n_envs, n_envs * len(cands), with_bus=False)
    out = {"workload": f"{env}: {n_envs} environments x {len(cands)} candidate actions (do nothing + single-line disconnections), forecast "
                       "1 step ahead, overflow cascade on: one gpf_simulate_batch call per batch (obs.simulate, "
                       "Observation/baseObservation.py:3365-3670)",
           "value": n_envs * len(cands) * reps / el, "unit": "simulated (environment, action) pairs/sec, wall clock incl. the host-side "
           "topology bookkeeping of every call", "ms_per_call": el / reps * 1e3, "frac_converged": float(r.converged.mean()),
           "oracle_check": oracle_spot_check(ctx, eng, 32, seed=8)}
    eng.close()
    return out


def workload_single_env(ctx, env, reps):
    """ONE environment behind the reference's Backend interface: `HipBackend.runpf` (grid2op_amd/backend.py) is ONE `gpf_solve_lane`
    call = injections + topology to the device, one launch of the power flow, results back, a single synchronisation.  Wall clock per
    call on a 1-lane engine (the latency a user who only swaps the backend of a single `grid2op.make` environment gets; the batched
    engine is the product, this is its floor)."""
    if ctx.args.stub_engine:
        return None
    m, ch = load_env(env)
    eng = ctx.make_engine(m, 1)
    inj0 = eng.get_injections(0, 1)[0]
    topo = np.ones(m.dim_topo, np.int32)
    sb = np.ones(m.n_shunt, np.int32) if m.n_shunt else None
    lay = eng.layout
    rng = np.random.default_rng(5)
    injs = np.tile(inj0, (16, 1))
    for k in range(16):                                   # 16 different operating points (loads +- 3 %), cycled
        f = 1.0 + 0.03 * rng.standard_normal(m.n_load)
        injs[k, lay.inj_load_p:lay.inj_load_p + m.n_load] *= f
        injs[k, lay.inj_load_q:lay.inj_load_q + m.n_load] *= f
    for k in range(20):
        r = eng.solve_lane(0, injs[k % 16], topo, sb)
    t0 = time.perf_counter()
    ok = 0
    for k in range(reps):
        r = eng.solve_lane(0, injs[k % 16], topo, sb)
        ok += int(r.converged[0])
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    for k in range(reps):
        r = eng.solve_lane(0, injs[k % 16], topo, sb, is_dc=True)
    el_dc = time.perf_counter() - t1
    eng.close()
    return {"workload": f"{env}: 1 lane, {reps} sequential gpf_solve_lane calls (= HipBackend.runpf: set injections + topology, one AC "
                        "power flow launch, results to the host, one synchronisation) on 16 cycled operating points",
            "us_per_call_ac": el / reps * 1e6, "us_per_call_dc": el_dc / reps * 1e6, "value": reps / el, "unit": "runpf calls/sec (one environment)",
            "frac_converged": ok / reps}


def workload_ptdf(ctx, env, B, reps, k_sec=64, w_sec=16):
    """BASELINE.json configs[4] as SURVEY.md 8(d) specifies it: l2rpn_idf_2023 (118 substations), chronics 2035-01-15_0 (576 rows x
    {99 load_p, 99 load_q, 62 prod_p}; tests/golden/l2rpn_idf_2023.chronics.npz), lane k on row (t + 7k) mod 576 with the jitter of
    default_rng(k), batch = 2048 lanes: (1) AC Newton-Raphson env steps, one observation per step; (2) the DC sensitivity path next to
    it ON THE SAME CHRONICS ROWS -- the injection rows the last AC step of every lane left on the device are evaluated as ONE FP64-MFMA
    PTDF GEMM."""
    m, ch = load_env(env)
    eng, T, l0 = setup_engine(ctx, m, ch, B)
    lay = eng.layout
    kw = dict(rebalance=1.02)
    k_sec = cfg_steps(ctx, k_sec)
    w, w_l, t_next = measure_modes(ctx, eng, k_sec, w_sec, kw, N_WIN_CFG, 200)
    med = median_window(w)[0]
    r_ac = eng.results()
    ac = {"workload": f"{env} (118 substations) AC NR DoNothing env.step on chronics 2035-01-15_0 (row (t+7k) mod {T}, loads x (1+0.05 N(0,1)), "
                      f"prod_p rebalanced), batch={B} lanes per GPU", "observations": OBS_EVERY if not ctx.args.last_obs_only else OBS_LAST,
          "value": ctx.world * B * k_sec / med, "unit": "env steps/sec", "ms_per_step": med / k_sec * 1e3, "steps_each": k_sec,
          "windows": summarize(w, ctx.world * B * k_sec),
          "last_observation_only": None if w_l is None else dict(summarize(w_l, ctx.world * B * k_sec), observations=OBS_LAST),
          "roofline": roofline_block(eng, w, B, k_sec, TRAFFIC_IDF), "plan": eng.plan() if hasattr(eng, "plan") else None,
          "frac_converged": float(r_ac.converged.mean()), "mean_nr_iterations": float(r_ac.n_iter[r_ac.converged].mean()),
          "oracle_check": oracle_spot_check(ctx, eng, 32, t_last=t_next - 1, seed=9)}
    # ---- the DC sensitivity path on the rows the lanes hold now (= chronics rows of step t_next - 1, jittered and rebalanced) ------
    inj = eng.get_injections()
    eng.ptdf_build(0)
    for _ in range(3):
        eng.ptdf_flows(fetch=False)
    eng.sync()
    eng.set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.ptdf_flows(fetch=False)
    eng.sync()
    el = time.perf_counter() - t0
    k_ms, n_l = eng.kernel_time()
    eng.set_profiling(0)
    flows = eng.ptdf_flows()
    ptdf_check = None
    if not ctx.args.no_oracle_check and not ctx.args.stub_engine:
        try:                                              # CHECKER leg: 64 lanes vs the C oracle's DC power flow (pp.rundcpp restated)
            from oracle.pf_oracle_c import COracle
            ls_ = np.sort(np.random.default_rng(6).choice(B, 64, replace=False))
            topo_, sb_ = eng.get_topology(0, 1)
            ref = COracle(m).solve_rows(inj[ls_], np.tile(topo_, (64, 1)), np.tile(sb_, (64, 1)) if m.n_shunt else None, is_dc=True)
            p_ref = ref["out"][:, lay.out_p_or:lay.out_p_or + m.n_line]
            ptdf_check = {"n": 64, "max_abs_err_vs_oracle": float(np.abs(flows[ls_] - p_ref).max()),
                          "ok": bool(np.all(np.abs(flows[ls_] - p_ref) <= 2e-4 + 5e-6 * np.abs(p_ref))),
                          "against": "oracle/pf_oracle.c DC power flow of the same injection rows (64 lanes)"}
        except Exception as exc:
            ptdf_check = {"error": repr(exc)[:300]}
    rows_rec = None
    if hasattr(eng, "ptdf_flows_rows"):
        rows_rec = workload_ptdf_rows(ctx, eng, m, B, t_next, n_rows=ctx.args.steps_per_launch if ctx.args.steps_per_launch > 1 else 16)
        r64 = workload_ptdf_rows(ctx, eng, m, B, t_next, n_rows=64, reps=10)
        rows_rec["with_64_rows_per_launch"] = {k: r64[k] for k in ("value", "us_per_launch", "roofline", "oracle_check")}
    eng.lodf_screen(0, 8)
    t0 = time.perf_counter()
    for _ in range(5):
        worst = eng.lodf_screen()                          # synchronous: includes the copy of [B, n_line] floats to the host
    lodf_s = (time.perf_counter() - t0) / 5

    def timed_runpf(is_dc):
        eng.runpf(is_dc=is_dc)
        eng.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.runpf(is_dc=is_dc)
        eng.sync()
        return (time.perf_counter() - t1) / 5, eng.results()
    dc_s, r_dc = timed_runpf(True)
    nb_act = int(r_dc.status[0, 2])                     # active buses of the topology = K of the GEMM
    nb_pad, line_pad = (nb_act + 3) // 4 * 4, (m.n_line + 15) // 16 * 16
    us = k_ms / max(n_l, 1) * 1e3
    tf = 2.0 * B * nb_pad * line_pad / (us * 1e-6) / 1e12 if us > 0 else 0.0
    hbm_bytes = 8.0 * B * lay.n_inj + 4.0 * B * line_pad + 8.0 * nb_pad * line_pad
    # the same GEMM at a batch that leaves the launch floor (the 2048-lane batch is 90 MFLOP: a few microseconds)
    big = None
    try:
        Bb = 32768
        eb = ctx.make_engine(m, Bb)
        eb.set_injections(np.tile(inj, (Bb // B, 1)))
        eb.ptdf_build(0)
        for _ in range(3):
            eb.ptdf_flows(fetch=False)
        eb.sync()
        eb.set_profiling(1)
        for _ in range(20):
            eb.ptdf_flows(fetch=False)
        eb.sync()
        kb, nb_l = eb.kernel_time()
        eb.set_profiling(0)
        eb.close()
        usb = kb / max(nb_l, 1) * 1e3
        big = {"lanes": Bb, "us_per_batch": usb, "value": Bb / (usb * 1e-6), "unit": "DC power flows/sec",
               "tflops": 2.0 * Bb * nb_pad * line_pad / (usb * 1e-6) / 1e12, "frac_of_fp64_mfma_peak": 2.0 * Bb * nb_pad * line_pad / (usb * 1e-6) / 1e12 / F64_PEAK_TFLOPS,
               "hbm_gbs": (8.0 * Bb * lay.n_inj + 4.0 * Bb * line_pad) / (usb * 1e-6) / 1e9}
    except Exception as exc:          # (memory on a shared box): the headline does not depend on it
        big = {"error": str(exc)[:200]}
    out = {"workload": f"{env} (118 substations) batch={B} lanes per GPU, chronics 2035-01-15_0: AC env steps (`ac_env_steps`) and, alongside, "
                       f"the DC line flows of every lane's current chronics row as ONE FP64 MFMA GEMM (flows = P_bus[{B}x{nb_pad}] . "
                       f"PTDF^T[{nb_pad}x{line_pad}], P_bus built in LDS from the injection rows in the same launch) for the fixed topology "
                       f"(BASELINE.json configs[4])",
           "ac_env_steps": ac, "chronics_rows_per_launch": rows_rec,
           "large_batch": big, "oracle_check": ptdf_check,
           "value": B * reps / el, "unit": "DC power flows/sec", "us_per_batch": us, "launches_per_batch": n_l / max(reps, 1),
           "roofline": {"bound": "mfma", "achieved": tf, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_PEAK_TFLOPS,
                        "hbm_gbs": hbm_bytes / (us * 1e-6) / 1e9 if us > 0 else 0.0, "traffic": ptdf_traffic(f"ptdf_flows_kernel, 1 row x {B:,}".replace(",", " "))[0],
                        "traffic_source": ptdf_traffic(f"ptdf_flows_kernel, 1 row x {B:,}".replace(",", " "))[1], "avg_launch_us": us, "kernel": "ptdf_flows_kernel",
                        "note": f"{2.0 * B * nb_pad * line_pad / 1e6:.0f} MFLOP and {hbm_bytes / 1e6:.1f} MB per launch of ONE chronics row per lane: "
                                f"launch / latency bound at this size (see chronics_rows_per_launch and large_batch)"},
           "per_lane_dc_solve_value": B / dc_s, "per_lane_dc_solve_unit": "DC power flows/sec (kernel S, B' refactorised per lane)",
           "max_abs_diff_vs_per_lane_dc_solve_mw": float(np.abs(flows - r_dc.p_or).max()),
           "lodf_n1_value": B * m.n_line / lodf_s, "lodf_n1_unit": "DC contingency cases/sec (every single-line outage of "
           "every lane: worst post-outage flow via LODF, result copied to the host)",
           "lodf_n1_frac_islanding": float(np.isinf(worst).mean()),
           "max_abs_dc_vs_ac_p_or_mw": float(np.abs(flows - r_ac.p_or)[r_ac.converged].max())}
    eng.close()
    return out


if __name__ == "__main__":
    main()

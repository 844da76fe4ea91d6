"""Static sharding of the lane batch over the GPUs of one node.

Lanes (environment copies / N-1 contingencies) never communicate, so the batch is cut into contiguous blocks, one
per rank (one process per GPU); there is NO collective on the data path.  ``torch.distributed`` (RCCL on the GPU
box, gloo in the CPU tests) is only used for the barrier and the max-over-ranks timing that ``bench.py`` reports.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

__all__ = ["lane_range", "synthetic_lane_inputs", "max_over_ranks", "sum_over_ranks", "visible_devices", "ShardedEngine"]


def lane_range(total_lanes: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block ``[lane0, lane0+n)`` of global lane ids owned by ``rank`` (sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total_lanes), int(world))
    n = base + (1 if rank < rem else 0)
    lane0 = rank * base + min(rank, rem)
    return lane0, n


def synthetic_lane_inputs(n_load: int, T: int, lane_ids) -> Tuple[np.ndarray, np.ndarray]:
    """The synthetic workload of SURVEY.md 8(d) cfg 2, a pure function of the GLOBAL lane id (so that any
    sharding reproduces the same global batch): lane k reads chronics row ``(t + 7k) mod T`` and scales its loads by
    ``1 + 0.05 N(0,1)`` drawn from ``numpy.random.default_rng(k)``."""
    lane_ids = np.asarray(lane_ids, dtype=np.int64)
    offsets = ((7 * lane_ids) % T).astype(np.int32)
    scale = np.empty((lane_ids.size, 2 * n_load), dtype=np.float32)
    for i, k in enumerate(lane_ids):
        scale[i] = 1.0 + 0.05 * np.random.default_rng(int(k)).standard_normal(2 * n_load)
    return offsets, scale


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX-reduce a host scalar over the ranks (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def visible_devices() -> int:
    """Number of HIP devices this process sees (``gpf_device_count``; raises when the library is not built)."""
    import ctypes as C
    from . import _capi
    n = C.c_int32(0)
    _capi.check(_capi.lib().gpf_device_count(C.byref(n)), "gpf_device_count")
    return int(n.value)


class ShardedEngine:
    """Single-process multi-GPU engine: one `PowerFlowEngine` (own HIP stream) per device, the global lane batch cut
    into the contiguous blocks of `lane_range`.  Lanes never communicate, so there is no cross-device operation: every
    call is forwarded to the engines that own the addressed lanes (asynchronous calls stay asynchronous, i.e. one host
    thread keeps all the devices busy) and results are concatenated in global lane order.

    Counterpart in the reference: its only parallelism is one process per environment -- ``Runner._run_parrallel``
    (grid2op/Runner/runner.py:1071-1253, a ``multiprocessing.Pool`` over episodes) and ``BaseMultiProcessEnvironment``
    (grid2op/Environment/baseMultiProcessEnv.py:22, 293, one worker process per environment copy).  ``bench.py``
    uses the other form (one PROCESS per GPU, `lane_range` of the global batch per rank)."""

    def __init__(self, model, n_lanes: int, devices=None, n_busbar: int = 2, engine_factory=None):
        if devices is None:
            devices = list(range(visible_devices()))
        devices = [int(d) for d in devices]
        if not devices:
            raise RuntimeError("ShardedEngine: no HIP device visible (there is no CPU fallback)")
        if n_lanes < len(devices):
            raise ValueError("ShardedEngine: fewer lanes than devices")
        if engine_factory is None:
            from .engine import PowerFlowEngine
            engine_factory = lambda m, n, dev, nbb: PowerFlowEngine(m, n_lanes=n, device=dev, n_busbar=nbb)  # noqa: E731
        self.model = model
        self.n_lanes = int(n_lanes)
        self.devices = devices
        self.blocks = [lane_range(self.n_lanes, len(devices), r) for r in range(len(devices))]
        self.engines = [engine_factory(model, n, dev, n_busbar) for dev, (_, n) in zip(devices, self.blocks)]
        e0 = self.engines[0]
        for k in ("n_inj", "n_out", "n_chron", "nb_total", "layout", "out_slices", "inj_slices", "init_inj"):
            if hasattr(e0, k):
                setattr(self, k, getattr(e0, k))

    # ---- routing ------------------------------------------------------------------------------------------------------
    def _parts(self, lane0: int = 0, n=None):
        """(engine, local lane0, count, offset into the caller's rows) of the shards that intersect [lane0, lane0+n)."""
        if n is None:
            n = self.n_lanes - lane0
        if lane0 < 0 or n < 0 or lane0 + n > self.n_lanes:
            raise ValueError("ShardedEngine: lane range out of bounds")
        out = []
        for eng, (b0, bn) in zip(self.engines, self.blocks):
            a, b = max(lane0, b0), min(lane0 + n, b0 + bn)
            if a < b:
                out.append((eng, a - b0, b - a, a - lane0))
        return out

    def owner(self, lane: int):
        (eng, l0, _, _), = self._parts(lane, 1)
        return eng, l0

    # ---- state ----------------------------------------------------------------------------------------------------------
    def pack_injections(self, n: int = 1, **fields):
        return self.engines[0].pack_injections(n, **fields)

    def pack_chronics(self, *a):
        return self.engines[0].pack_chronics(*a)

    def set_injections(self, inj, lane0: int = 0):
        inj = np.asarray(inj).reshape(-1, self.n_inj)
        for eng, l0, n, off in self._parts(lane0, inj.shape[0]):
            eng.set_injections(inj[off:off + n], lane0=l0)

    def get_injections(self, lane0: int = 0, n=None):
        return np.concatenate([eng.get_injections(l0, k) for eng, l0, k, _ in self._parts(lane0, n)])

    def set_topology(self, topo, shunt_bus=None, lane0: int = 0):
        topo = np.asarray(topo).reshape(-1, self.model.dim_topo)
        sb = None if shunt_bus is None or not self.model.n_shunt else np.asarray(shunt_bus).reshape(-1, self.model.n_shunt)
        for eng, l0, n, off in self._parts(lane0, topo.shape[0]):
            eng.set_topology(topo[off:off + n], None if sb is None else sb[off:off + n], lane0=l0)

    def reset(self, lane0: int = 0, n=None):
        for eng, l0, k, _ in self._parts(lane0, n):
            eng.reset(l0, k)

    def disconnect_line(self, lane: int, line_id: int):
        eng, l0 = self.owner(lane)
        eng.disconnect_line(l0, line_id)

    def upload_chronics(self, tables):
        for eng in self.engines:                     # every device holds its own copy of the (small) tables
            eng.upload_chronics(tables)
        self.chron_T = self.engines[0].chron_T

    def set_lane_chronics(self, lane_table=None, lane_offset=None, lane_scale=None):
        for eng, (b0, bn) in zip(self.engines, self.blocks):
            cut = lambda a: None if a is None else np.asarray(a)[b0:b0 + bn]  # noqa: E731
            eng.set_lane_chronics(cut(lane_table), cut(lane_offset), cut(lane_scale))

    def set_thermal_limits(self, limit_a):
        for eng in self.engines:
            eng.set_thermal_limits(limit_a)

    # ---- solve (asynchronous: queued on every device's stream) ----------------------------------------------------------------
    # One host thread per device: a launch call is planning + parameter upload + the launch itself (~10 us of host time on an
    # MI355X box) and the ctypes call releases the GIL, so the launches of the devices are issued CONCURRENTLY -- issued from one
    # thread, 8 devices x ~10 us serialised per step would be a third of a 26 us step.  The threads are created lazily and only with
    # more than one device; HIP handles are thread-compatible (one engine is only ever driven by one thread at a time: `_fan` joins
    # before it returns).
    def _fan(self, calls):
        if len(calls) <= 1:
            for fn in calls:
                fn()
            return
        if getattr(self, "_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self.engines), thread_name_prefix="gridpf-dev")
        for f in [self._pool.submit(fn) for fn in calls]:
            f.result()                                   # re-raises a GridPFError of any device

    def runpf(self, lane0: int = 0, n=None, **kw):
        self._fan([(lambda e=eng, a=l0, b=k: e.runpf(a, b, **kw)) for eng, l0, k, _ in self._parts(lane0, n)])

    def step(self, t: int, **kw):
        self._fan([(lambda e=eng: e.step(t, **kw)) for eng in self.engines])

    def set_lane_redispatch(self, delta_mw):
        for eng, (b0, bn) in zip(self.engines, self.blocks):
            eng.set_lane_redispatch(None if delta_mw is None else np.asarray(delta_mw)[b0:b0 + bn])

    def episode(self, lane0: int = 0, n=None):
        parts = [eng.episode(l0, k) for eng, l0, k, _ in self._parts(lane0, n)]
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))

    def sync(self):
        for eng in self.engines:
            eng.sync()

    def results(self, lane0: int = 0, n=None, with_bus: bool = True, pinned: bool = False):
        """(``pinned``: every device's rows come over by DMA into its engine's pinned block; the concatenation below is the one host copy)"""
        from .engine import LaneResults
        rs = [eng.results(l0, k, with_bus=with_bus, **({"pinned": True} if pinned else {})) for eng, l0, k, _ in self._parts(lane0, n)]
        cat = lambda f: None if getattr(rs[0], f) is None else np.concatenate([getattr(r, f) for r in rs])  # noqa: E731
        return LaneResults(out=cat("out"), topo_vect=cat("topo_vect"), shunt_bus=cat("shunt_bus"), line_status=cat("line_status"),
                           status=cat("status"), bus_vm=cat("bus_vm"), bus_va=cat("bus_va"), _slices=rs[0]._slices)

    def cooldown(self, lane0: int = 0, n=None):
        return np.concatenate([eng.cooldown(l0, k) for eng, l0, k, _ in self._parts(lane0, n)])

    def set_cooldown(self, line_cooldown, lane0: int = 0):
        c = np.asarray(line_cooldown).reshape(-1, self.model.n_line)
        for eng, l0, k, off in self._parts(lane0, c.shape[0]):
            eng.set_cooldown(c[off:off + k], lane0=l0)

    def step_outputs(self, lane0: int = 0, n=None):
        parts = [eng.step_outputs(l0, k) for eng, l0, k, _ in self._parts(lane0, n)]
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))

    # ---- the rest of the batched API: per-table calls go to every device, per-lane calls are cut by block ---------------------
    TRAJ_RHO, TRAJ_OBS = 1, 2

    def upload_maintenance(self, maintenance):
        for eng in self.engines:
            eng.upload_maintenance(maintenance)

    def upload_outage_durations(self, durations):
        for e in self.engines:
            e.upload_outage_durations(durations)

    def upload_hazards(self, hazards):
        for eng in self.engines:
            eng.upload_hazards(hazards)

    def set_deterministic(self, flag: bool = True):
        for eng in self.engines:
            eng.set_deterministic(flag)

    def set_gen_limits(self, *a, **kw):
        for eng in self.engines:
            eng.set_gen_limits(*a, **kw)

    def redispatch(self, new_p, prev_p, actual, target, modified, rhs, lane0: int = 0, apply: bool = False):
        ng = self.model.n_gen
        rows = [np.asarray(a).reshape(-1, ng) for a in (new_p, prev_p, actual, target, modified)]
        n = rows[0].shape[0]
        rhs = np.broadcast_to(np.asarray(rhs, dtype=np.float64), (n,))
        oks, afters = [], []
        for eng, l0, k, off in self._parts(lane0, n):
            ok, after = eng.redispatch(*[r[off:off + k] for r in rows], rhs[off:off + k], lane0=l0, apply=apply)
            oks.append(ok)
            afters.append(after)
        return np.concatenate(oks), np.concatenate(afters)

    def set_trajectory(self, n_steps_cap: int, what: int = 1):
        for eng in self.engines:
            eng.set_trajectory(n_steps_cap, what)

    def trajectory(self, n_steps: int, step0: int = 0, lane0: int = 0, n=None):
        parts = [eng.trajectory(n_steps, step0, l0, k) for eng, l0, k, _ in self._parts(lane0, n)]
        return np.concatenate([p[0] for p in parts], axis=1), np.concatenate([p[1] for p in parts], axis=1)

    def trajectory_obs(self, n_steps: int, step0: int = 0, lane0: int = 0, n=None):
        from .engine import LaneResults
        parts = [eng.trajectory_obs(n_steps, step0, l0, k) for eng, l0, k, _ in self._parts(lane0, n)]
        out = []
        for s in range(n_steps):
            rs = [p[s] for p in parts]
            cat = lambda f: np.concatenate([getattr(r, f) for r in rs])  # noqa: E731
            out.append(LaneResults(out=cat("out"), topo_vect=cat("topo_vect"), shunt_bus=cat("shunt_bus"), line_status=cat("line_status"),
                                   status=cat("status"), bus_vm=None, bus_va=None, _slices=rs[0]._slices))
        return out

    def trajectory_cooldown(self, n_steps: int, step0: int = 0, lane0: int = 0, n=None):
        return np.concatenate([eng.trajectory_cooldown(n_steps, step0, l0, k) for eng, l0, k, _ in self._parts(lane0, n)], axis=1)

    def counters(self) -> dict:
        """Step launches / kernel dispatches summed over the devices."""
        cs = [eng.counters() for eng in self.engines]
        return {k: sum(c[k] for c in cs) for k in cs[0]}

    # DC sensitivities of per-lane topologies: every device builds the tables of the topologies ITS lanes hold; class ids are per shard
    # (a class of shard s is reported as ``class_offset[s] + local id``), so that `ptdf_class` finds the owner again
    def ptdf_build_batch(self, lane0: int = 0, n=None, with_lodf: bool = True) -> dict:
        if n is None:
            n = self.n_lanes - lane0
        lc = np.empty(n, np.int32)
        st, cn, ms, self._ptdfb_owner = [], [], 0.0, []
        for eng, l0, k, off in self._parts(lane0, n):
            r = eng.ptdf_build_batch(l0, k, with_lodf=with_lodf)
            base = len(st)
            lc[off:off + k] = np.where(r["lane_class"] >= 0, r["lane_class"] + base, -1)
            st.extend(r["class_status"].tolist()); cn.extend(r["class_n"].tolist())
            self._ptdfb_owner.extend((eng, c) for c in range(r["n_classes"]))
            ms = max(ms, r["kernel_ms"])                      # the devices run concurrently
        return {"n_classes": len(st), "lane_class": lc, "class_status": np.asarray(st, np.int32), "class_n": np.asarray(cn, np.int32), "kernel_ms": ms}

    def ptdf_class(self, cls: int, lodf: bool = False):
        eng, c = self._ptdfb_owner[int(cls)]
        return eng.ptdf_class(c, lodf=lodf)

    def copy_lanes(self, src: int, dst: int, n: int = 1):
        """Device-side copy inside one shard; a copy that crosses devices goes through the host (inputs, protection counters and --
        when the injection dynamics are on -- the dispatch / storage / curtailment state: the results of the destination lanes are
        those of their next solve)."""
        ps, pd = self._parts(src, n), self._parts(dst, n)
        if len(ps) == 1 and len(pd) == 1 and ps[0][0] is pd[0][0]:
            ps[0][0].copy_lanes(ps[0][1], pd[0][1], n)
            return
        topo, sb = self.get_topology(src, n)
        self.set_injections(self.get_injections(src, n), lane0=dst)
        self.set_topology(topo, sb, lane0=dst)
        _, ovc, _ = self.step_outputs(src, n)
        self.set_overflow_count(ovc, lane0=dst)
        self.set_cooldown(self.cooldown(src, n), lane0=dst)
        if all(getattr(e, "env_dynamics_on", False) for e in self.engines):
            self.set_env_state(dst, **self.env_state(src, n))

    def get_topology(self, lane0: int = 0, n=None):
        parts = [eng.get_topology(l0, k) for eng, l0, k, _ in self._parts(lane0, n)]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def fanout_n1(self, src_lane: int, dst_lane0: int, out_lines):
        """N-1 fan-out of one source lane; source and destinations must live on the same device (the contingencies of an
        environment sit next to it: SURVEY.md 8(e))."""
        ol = np.asarray(out_lines, dtype=np.int32)
        (es, ls, _, _), = self._parts(src_lane, 1)
        pd = self._parts(dst_lane0, ol.size)
        if len(pd) != 1 or pd[0][0] is not es:
            raise ValueError("ShardedEngine.fanout_n1: the source lane and its contingency lanes must be on the same device")
        es.fanout_n1(ls, pd[0][1], ol)

    def simulate_candidates(self, src_lane: int, dst_lane0: int, actions=None, topologies=None, **kw):
        (es, ls, _, _), = self._parts(src_lane, 1)
        n = len(actions) if topologies is None else np.asarray(topologies).reshape(-1, self.model.dim_topo).shape[0]
        pd = self._parts(dst_lane0, n)
        if len(pd) != 1 or pd[0][0] is not es:
            raise ValueError("ShardedEngine.simulate_candidates: the source lane and its candidate lanes must be on the same device")
        return es.simulate_candidates(ls, pd[0][1], actions=actions, topologies=topologies, **kw)

    def set_storage_params(self, *a, **kw):
        for eng in self.engines:
            eng.set_storage_params(*a, **kw)

    def set_env_dynamics(self, *a, **kw):
        for eng in self.engines:
            eng.set_env_dynamics(*a, **kw)

    def set_lane_actions(self, redispatch=None, storage_power=None, hold_storage: bool = False):
        for eng, (b0, bn) in zip(self.engines, self.blocks):
            cut = lambda a: None if a is None else np.asarray(a)[b0:b0 + bn]  # noqa: E731
            eng.set_lane_actions(cut(redispatch), cut(storage_power), hold_storage)

    def lane_actions_on_device(self, *a, **kw):
        """every device's action buffers (`device_views()[k]["act_*"]`) hold the next launch's actions of its own lanes"""
        for eng in self.engines:
            eng.lane_actions_on_device(*a, **kw)

    def set_gen_renewable(self, renewable):
        for eng in self.engines:
            eng.set_gen_renewable(renewable)

    def set_lane_curtailment(self, limit):
        for eng, (b0, bn) in zip(self.engines, self.blocks):
            eng.set_lane_curtailment(None if limit is None else np.asarray(limit)[b0:b0 + bn])

    def env_state(self, lane0: int = 0, n=None) -> dict:
        parts = [eng.env_state(l0, k) for eng, l0, k, _ in self._parts(lane0, n)]
        return {key: np.concatenate([p[key] for p in parts]) for key in parts[0]}

    def set_env_state(self, lane0: int = 0, **fields):
        """`PowerFlowEngine.set_env_state` on the global lane range starting at ``lane0``: every array is cut by device block."""
        arrs = {k: np.asarray(v) for k, v in fields.items() if v is not None}
        if not arrs:
            return
        n = next(iter(arrs.values())).shape[0]
        for eng, l0, k, off in self._parts(lane0, n):
            eng.set_env_state(l0, **{key: a[off:off + k] for key, a in arrs.items()})

    def set_overflow_count(self, counts, lane0: int = 0):
        c = np.asarray(counts).reshape(-1, self.model.n_line)
        for eng, l0, n, off in self._parts(lane0, c.shape[0]):
            eng.set_overflow_count(c[off:off + n], lane0=l0)

    def upload_forecasts(self, tables):
        for eng in self.engines:
            eng.upload_forecasts(tables)

    def simulate_batch(self, t_obs: int, src_lanes, actions, dst_lane0: int, **kw):
        """`PowerFlowEngine.simulate_batch`; the source lanes and their candidate lanes must live on ONE device."""
        src = np.asarray(src_lanes, dtype=np.int64).reshape(-1)
        owners = {id(self.owner(int(k))[0]) for k in src}
        pd = self._parts(dst_lane0, src.size * len(actions))
        if len(owners) != 1 or len(pd) != 1 or id(pd[0][0]) not in owners:
            raise ValueError("ShardedEngine.simulate_batch: the source lanes and their candidate lanes must be on the same device")
        eng = pd[0][0]
        b0 = self.blocks[self.engines.index(eng)][0]
        return eng.simulate_batch(t_obs, src - b0, actions, pd[0][1], **kw)

    def device_views(self):
        """One dict of zero-copy torch views per device (global lane order = concatenation over the list)."""
        return [eng.device_views() for eng in self.engines]

    def plan(self):
        return [eng.plan() for eng in self.engines]

    def specialize(self, enable: bool = True, cache_dir=None, verify: bool = True):
        """`PowerFlowEngine.specialize` on every device's engine (same grid: the self-test runs once, the code objects are shared
        through the on-disk cache); returns one `specialization()` dict per device."""
        out = []
        for k, eng in enumerate(self.engines):
            out.append(eng.specialize(enable, cache_dir=cache_dir, verify=verify and k == 0) if hasattr(eng, "specialize") else None)
        return out

    def specialization(self):
        return [eng.specialization() if hasattr(eng, "specialization") else None for eng in self.engines]

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for eng in self.engines:
            eng.close()
        self.engines = []

"""`PowerFlowEngine`: the batched MI355X power-flow engine (Python over the C ABI, no grid2op needed).

One engine = one grid (`GridModel`) x ``n_lanes`` independent grid instances ("lanes": environment
copies or N-1 contingencies) resident in the HBM of one GPU.  All arithmetic happens in
``libgridpf.so`` (hand-written HIP for gfx950); this module only marshals numpy arrays.

The single-environment drop-in `grid2op_amd.backend.HipBackend` is a one-lane view on such an engine;
the batched stepping API (`upload_chronics` / `step`) is what ``bench.py`` measures.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from . import _capi
from ._capi import GpfGridDesc, GpfLayout, GpfStepOpts, GridPFError, check, ptr
from .grid_model import GridModel

__all__ = ["PowerFlowEngine", "LaneResults", "GridPFError", "ST_CONVERGED", "STATUS_TEXT"]

ST_CONVERGED = 0
STATUS_TEXT = {
    0: "converged",
    1: "Newton-Raphson did not converge within max_iter iterations",
    2: "islanded grid (an active bus is not connected to the slack bus)",
    3: "no in-service slack generator",
    4: "singular matrix",
    5: "engine capacity exceeded",
    6: "infeasible redispatching (ImpossibleRedispatching: game over)",
    -1: "power flow not run",
}

_OUT_FIELDS = [
    ("p_or", "n_line"), ("q_or", "n_line"), ("v_or", "n_line"), ("a_or", "n_line"), ("theta_or", "n_line"),
    ("p_ex", "n_line"), ("q_ex", "n_line"), ("v_ex", "n_line"), ("a_ex", "n_line"), ("theta_ex", "n_line"),
    ("gen_p", "n_gen"), ("gen_q", "n_gen"), ("gen_v", "n_gen"), ("gen_theta", "n_gen"),
    ("load_p", "n_load"), ("load_q", "n_load"), ("load_v", "n_load"), ("load_theta", "n_load"),
    ("storage_p", "n_storage"), ("storage_q", "n_storage"), ("storage_v", "n_storage"), ("storage_theta", "n_storage"),
    ("shunt_p", "n_shunt"), ("shunt_q", "n_shunt"), ("shunt_v", "n_shunt"),
]
_INJ_FIELDS = [("gen_p", "n_gen"), ("gen_vm", "n_gen"), ("load_p", "n_load"), ("load_q", "n_load"),
               ("storage_p", "n_storage"), ("storage_q", "n_storage"), ("shunt_p", "n_shunt"), ("shunt_q", "n_shunt")]


@dataclass
class LaneResults:
    """Results of ``n`` lanes; every float field is a float32 ``[n, n_el]`` view of one ``out`` block."""
    out: np.ndarray
    topo_vect: np.ndarray
    shunt_bus: np.ndarray
    line_status: np.ndarray
    status: np.ndarray           # [n, 4] {status, n_iter, n_active_bus, n_cascade_rounds}
    bus_vm: np.ndarray           # float64 [n, nb_total] (pu), NaN for inactive buses
    bus_va: np.ndarray           # float64 [n, nb_total] (deg)
    _slices: Dict[str, slice]

    def __getattr__(self, name):
        sl = self.__dict__.get("_slices", {}).get(name)
        if sl is None:
            raise AttributeError(name)
        return self.out[:, sl]

    @property
    def converged(self) -> np.ndarray:
        return self.status[:, 0] == ST_CONVERGED

    @property
    def n_iter(self) -> np.ndarray:
        return self.status[:, 1]


class PowerFlowEngine:
    def __init__(self, model: GridModel, n_lanes: int = 1, device: int = 0, n_busbar: int = 2, deterministic: bool = False):
        self.model = model
        self.n_busbar = int(n_busbar)
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        m = model
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        br_y = np.empty((m.n_line, 8), dtype=np.float64)
        for k, y in enumerate((m.br_yff, m.br_yft, m.br_ytf, m.br_ytt)):
            br_y[:, 2 * k] = y.real
            br_y[:, 2 * k + 1] = y.imag
        init_inj = np.concatenate([f64(m.gen_p0), f64(m.gen_vm0), f64(m.load_p0), f64(m.load_q0), f64(m.storage_p0),
                                   f64(m.storage_q0), f64(m.shunt_p0), f64(m.shunt_q0)])
        keep = dict(
            sub_vn_kv=f64(m.sub_vn_kv), line_or_sub=i32(m.line_or_sub), line_ex_sub=i32(m.line_ex_sub),
            line_or_pos=i32(m.line_or_pos_topo_vect), line_ex_pos=i32(m.line_ex_pos_topo_vect), br_y=br_y,
            br_bdc=f64(m.br_bdc), gen_sub=i32(m.gen_sub), gen_pos=i32(m.gen_pos_topo_vect), gen_min_q=f64(m.gen_min_q),
            gen_max_q=f64(m.gen_max_q), gen_slack=np.ascontiguousarray(m.gen_slack, dtype=np.uint8),
            load_sub=i32(m.load_sub), load_pos=i32(m.load_pos_topo_vect), sto_sub=i32(m.storage_sub),
            sto_pos=i32(m.storage_pos_topo_vect), shunt_sub=i32(m.shunt_sub), shunt_fact=f64(m.shunt_fact),
            init_inj=init_inj, init_topo=i32(m.initial_topo_vect()), init_shunt_bus=i32(m.initial_shunt_bus()))
        d = GpfGridDesc()
        d.n_sub, d.n_busbar = m.n_sub, self.n_busbar
        d.n_line, d.n_gen, d.n_load, d.n_storage, d.n_shunt, d.dim_topo = (m.n_line, m.n_gen, m.n_load, m.n_storage,
                                                                              m.n_shunt, m.dim_topo)
        d.sn_mva = m.sn_mva
        d.sub_vn_kv = ptr(keep["sub_vn_kv"], C.c_double)
        d.line_or_sub = ptr(keep["line_or_sub"], C.c_int32)
        d.line_ex_sub = ptr(keep["line_ex_sub"], C.c_int32)
        d.line_or_pos_topo_vect = ptr(keep["line_or_pos"], C.c_int32)
        d.line_ex_pos_topo_vect = ptr(keep["line_ex_pos"], C.c_int32)
        d.br_y = ptr(keep["br_y"], C.c_double)
        d.br_bdc = ptr(keep["br_bdc"], C.c_double)
        d.gen_sub = ptr(keep["gen_sub"], C.c_int32)
        d.gen_pos_topo_vect = ptr(keep["gen_pos"], C.c_int32)
        d.gen_min_q = ptr(keep["gen_min_q"], C.c_double)
        d.gen_max_q = ptr(keep["gen_max_q"], C.c_double)
        d.gen_slack = ptr(keep["gen_slack"], C.c_uint8)
        d.load_sub = ptr(keep["load_sub"], C.c_int32)
        d.load_pos_topo_vect = ptr(keep["load_pos"], C.c_int32)
        d.storage_sub = ptr(keep["sto_sub"], C.c_int32)
        d.storage_pos_topo_vect = ptr(keep["sto_pos"], C.c_int32)
        d.shunt_sub = ptr(keep["shunt_sub"], C.c_int32)
        d.shunt_fact = ptr(keep["shunt_fact"], C.c_double)
        d.init_inj = ptr(keep["init_inj"], C.c_double)
        d.init_topo = ptr(keep["init_topo"], C.c_int32)
        d.init_shunt_bus = ptr(keep["init_shunt_bus"], C.c_int32)
        check(self._lib.gpf_create(C.byref(d), int(n_lanes), int(device), C.byref(self._h)), "gpf_create")
        self._pid = os.getpid()          # HIP handles belong to the creating process (a forked child must not destroy them)
        if deterministic:
            self.set_deterministic(True)
        self.n_lanes = int(n_lanes)
        self.device = int(device)
        lay = GpfLayout()
        check(self._lib.gpf_get_layout(self._h, C.byref(lay)), "gpf_get_layout")
        self.layout = lay
        self.n_inj, self.n_out, self.n_chron, self.nb_total = lay.n_inj, lay.n_out, lay.n_chron, lay.nb_total
        sizes = dict(n_line=m.n_line, n_gen=m.n_gen, n_load=m.n_load, n_storage=m.n_storage, n_shunt=m.n_shunt)
        self.out_slices = {}
        for name, sz in _OUT_FIELDS:
            key = "out_" + name
            off = getattr(lay, key)
            self.out_slices[name] = slice(off, off + sizes[sz])
        self.inj_slices = {}
        for name, sz in _INJ_FIELDS:
            off = getattr(lay, "inj_" + name)
            self.inj_slices[name] = slice(off, off + sizes[sz])
        self.init_inj = init_inj.copy()
        self.env_dynamics_on = False

    # ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if getattr(self, "_pid", None) == os.getpid():
                self._lib.gpf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_deterministic(self, flag: bool = True):
        """Results are bit-identical from run to run on every grid by default; ``True`` additionally forces one wavefront per lane
        on grids with >= 64 substations (cross-check variant, ~20 % slower there)."""
        check(self._lib.gpf_set_deterministic(self._h, int(bool(flag))), "gpf_set_deterministic")

    def _range(self, lane0, n):
        if n is None:
            n = self.n_lanes - lane0
        return int(lane0), int(n)

    # ---- state --------------------------------------------------------------------------------------
    def pack_injections(self, n: int = 1, **fields) -> np.ndarray:
        """``[n, n_inj]`` float64 rows starting from the pristine injections; override by field name."""
        inj = np.tile(self.init_inj, (n, 1))
        for k, v in fields.items():
            inj[:, self.inj_slices[k]] = v
        return inj

    def set_injections(self, inj: np.ndarray, lane0: int = 0):
        inj = np.ascontiguousarray(inj, dtype=np.float64).reshape(-1, self.n_inj)
        check(self._lib.gpf_set_injections(self._h, lane0, inj.shape[0], ptr(inj, C.c_double)), "gpf_set_injections")

    def get_injections(self, lane0: int = 0, n: Optional[int] = None) -> np.ndarray:
        lane0, n = self._range(lane0, n)
        inj = np.empty((n, self.n_inj), dtype=np.float64)
        check(self._lib.gpf_get_injections(self._h, lane0, n, ptr(inj, C.c_double)), "gpf_get_injections")
        return inj

    def set_topology(self, topo: np.ndarray, shunt_bus: Optional[np.ndarray] = None, lane0: int = 0):
        topo = np.ascontiguousarray(topo, dtype=np.int32).reshape(-1, self.model.dim_topo)
        sb = None
        if shunt_bus is not None and self.model.n_shunt:
            sb = np.ascontiguousarray(shunt_bus, dtype=np.int32).reshape(-1, self.model.n_shunt)
            assert sb.shape[0] == topo.shape[0]
        check(self._lib.gpf_set_topology(self._h, lane0, topo.shape[0], ptr(topo, C.c_int32), ptr(sb, C.c_int32)),
              "gpf_set_topology")

    def get_topology(self, lane0: int = 0, n: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        lane0, n = self._range(lane0, n)
        topo = np.empty((n, self.model.dim_topo), dtype=np.int32)
        sb = np.empty((n, self.model.n_shunt), dtype=np.int32)
        check(self._lib.gpf_get_topology(self._h, lane0, n, ptr(topo, C.c_int32), ptr(sb, C.c_int32)), "gpf_get_topology")
        return topo, sb

    def disconnect_line(self, lane: int, line_id: int):
        check(self._lib.gpf_disconnect_line(self._h, lane, line_id), "gpf_disconnect_line")

    def reset(self, lane0: int = 0, n: Optional[int] = None):
        lane0, n = self._range(lane0, n)
        check(self._lib.gpf_reset_lanes(self._h, lane0, n), "gpf_reset_lanes")

    def copy_lanes(self, src: int, dst: int, n: int = 1):
        check(self._lib.gpf_copy_lanes(self._h, src, dst, n), "gpf_copy_lanes")

    def fanout_n1(self, src_lane: int, dst_lane0: int, out_lines):
        ol = np.ascontiguousarray(out_lines, dtype=np.int32)
        check(self._lib.gpf_fanout_n1(self._h, src_lane, dst_lane0, ol.size, ptr(ol, C.c_int32)), "gpf_fanout_n1")

    def candidate_topologies(self, base_topo, actions, last_bus=None) -> np.ndarray:
        """``[len(actions), dim_topo]`` topology rows = ``base_topo`` modified by each candidate action, described as grid2op
        describes them (Action/baseAction.py ``set_bus`` / ``set_line_status`` / ``change_bus``): a dict with any of
        ``"set_bus": {topo_vect position: bus}``, ``"lines_or_bus" / "lines_ex_bus" / "loads_bus" / "gens_bus" / "storages_bus":
        [(element id, bus)]``, ``"set_line_status": [(line id, +1 | -1)]``, ``"change_bus": [positions]`` (1 <-> 2; refused with
        more than 2 busbars per substation, as grid2op refuses it).  A reconnection without an explicit bus puts each end back on
        its LAST KNOWN busbar (``last_bus``: the ``[dim_topo]`` row _BackendAction keeps in ``last_topo_registered``,
        Action/_backendAction.py; busbar 1 where it is unknown or not given).  An empty dict is the do-nothing candidate."""
        m = self.model
        base = np.asarray(base_topo, dtype=np.int32).reshape(m.dim_topo)
        last = None if last_bus is None else np.asarray(last_bus, dtype=np.int32).reshape(m.dim_topo)
        pos_of = {"lines_or_bus": m.line_or_pos_topo_vect, "lines_ex_bus": m.line_ex_pos_topo_vect, "loads_bus": m.load_pos_topo_vect,
                  "gens_bus": m.gen_pos_topo_vect, "storages_bus": m.storage_pos_topo_vect}
        out = np.tile(base, (len(actions), 1))
        for k, act in enumerate(actions):
            row = out[k]
            for line, st in act.get("set_line_status", ()):
                po, pe = m.line_or_pos_topo_vect[line], m.line_ex_pos_topo_vect[line]
                if st < 0:
                    row[po] = row[pe] = -1
                elif st > 0 and (row[po] < 1 or row[pe] < 1):
                    for p_ in (po, pe):
                        if row[p_] < 1:
                            row[p_] = last[p_] if (last is not None and last[p_] >= 1) else 1
            for key, pos in pos_of.items():
                for el, bus in act.get(key, ()):
                    row[pos[el]] = bus
            for p_, bus in dict(act.get("set_bus", {})).items():
                row[p_] = bus
            if act.get("change_bus", ()) and self.n_busbar > 2:
                raise ValueError("change_bus is only defined for 2 busbars per substation (grid2op refuses it otherwise)")
            for p_ in act.get("change_bus", ()):
                if row[p_] >= 1:
                    row[p_] = 2 if row[p_] == 1 else 1
        return out

    def simulate_candidates(self, src_lane: int, dst_lane0: int, actions=None, topologies=None, is_dc: bool = False,
                            max_iter: int = 10, tol_mva: float = 1e-8) -> int:
        """Batched ``obs.simulate`` (Observation/_obsEnv.py:321-503, Reward/n1Reward.py:70-99): lanes ``dst_lane0 ...`` become
        copies of ``src_lane`` (injections, shunts) with one candidate topology each -- built from ``actions``
        (`candidate_topologies`) or given as rows -- and are solved in ONE launch (asynchronous; read them with `results`).
        Returns the number of candidate lanes."""
        if topologies is None:
            base, _ = self.get_topology(src_lane, 1)
            topologies = self.candidate_topologies(base[0], actions)
        topologies = np.ascontiguousarray(topologies, dtype=np.int32).reshape(-1, self.model.dim_topo)
        n = topologies.shape[0]
        self.fanout_n1(src_lane, dst_lane0, np.full(n, -1, dtype=np.int32))      # device-side copy of the source lane's state
        self.set_topology(topologies, lane0=dst_lane0)
        self.runpf(dst_lane0, n, is_dc=is_dc, max_iter=max_iter, tol_mva=tol_mva)
        return n

    ACT_SET_BUS, ACT_SET_LINE_STATUS, ACT_CHANGE_BUS, ACT_CHANGE_LINE_STATUS, ACT_SET_SHUNT_BUS = 0, 1, 2, 3, 4

    def upload_forecasts(self, tables):
        """The ``*_forecasted`` tables of the uploaded chronics (`grid2op_amd.chronics` loads them with ``forecasts=True``;
        `pack_chronics` packs them like the chronics): ``[n_tables, T, n_chron]`` (one horizon, the reference's default) or
        ``[n_tables, T, n_horizons, n_chron]``; None removes them.  Row ``r`` holds the forecast made at chronics row ``r``
        (Chronics/gridStateFromFileWithForecasts.py:311-353)."""
        if tables is None:
            check(self._lib.gpf_upload_forecasts(self._h, 0, 0, 0, None), "gpf_upload_forecasts")
            return
        t = np.ascontiguousarray(tables, dtype=np.float32)
        if t.ndim == 2:
            t = t[None]
        if t.ndim == 3:
            t = t[:, :, None, :]
        assert t.ndim == 4 and t.shape[3] == self.n_chron, t.shape
        check(self._lib.gpf_upload_forecasts(self._h, t.shape[0], t.shape[1], t.shape[2], ptr(t, C.c_float)), "gpf_upload_forecasts")

    def pack_actions(self, actions):
        """Candidate actions (dicts as `candidate_topologies` takes them, plus ``"change_line_status": [line ids]`` and
        ``"shunts_bus": [(shunt id, bus)]``) -> (offsets ``[n + 1]``, items ``[n_items, 3]``) of `gpf_simulate_batch`."""
        m = self.model
        pos_of = {"lines_or_bus": m.line_or_pos_topo_vect, "lines_ex_bus": m.line_ex_pos_topo_vect, "loads_bus": m.load_pos_topo_vect,
                  "gens_bus": m.gen_pos_topo_vect, "storages_bus": m.storage_pos_topo_vect}
        off, items = [0], []
        for act in actions:
            for line, st in act.get("set_line_status", ()):
                items.append((self.ACT_SET_LINE_STATUS, int(line), int(st)))
            for line in act.get("change_line_status", ()):
                items.append((self.ACT_CHANGE_LINE_STATUS, int(line), 0))
            for key, pos in pos_of.items():
                for el, bus in act.get(key, ()):
                    items.append((self.ACT_SET_BUS, int(pos[el]), int(bus)))
            for p_, bus in dict(act.get("set_bus", {})).items():
                items.append((self.ACT_SET_BUS, int(p_), int(bus)))
            for p_ in act.get("change_bus", ()):
                items.append((self.ACT_CHANGE_BUS, int(p_), 0))
            for sh, bus in act.get("shunts_bus", ()):
                items.append((self.ACT_SET_SHUNT_BUS, int(sh), int(bus)))
            off.append(len(items))
        return np.asarray(off, dtype=np.int32), np.asarray(items, dtype=np.int32).reshape(-1, 3)

    def simulate_batch(self, t_obs: int, src_lanes, actions, dst_lane0: int, time_step: int = 1, last_bus=None, max_iter: int = 10,
                       tol_mva: float = 1e-8, rebalance: float = 0.0, cascade: bool = False, hard_overflow: float = 2.0,
                       soft_overflow: float = 1.0, nb_ts_allowed: int = 2, max_rounds: int = 16, is_dc: bool = False) -> int:
        """Batched ``obs.simulate(action, time_step)`` (Observation/baseObservation.py:3365-3670): for every source lane ``b``
        (an environment whose current observation is the step at time index ``t_obs``) and every candidate action ``k``, lane
        ``dst_lane0 + b * len(actions) + k`` becomes a copy of lane ``b`` with the action applied as ``_BackendAction`` applies it
        and is stepped ONCE on the injections forecast ``time_step`` steps ahead (0: the observation's own injections) -- all
        ``len(src_lanes) * len(actions)`` candidates in one launch (asynchronous).  Read them with `results` / `step_outputs`
        on the destination range.  ``last_bus``: ``[len(src_lanes), dim_topo]`` last known busbars (reconnections), default 1.
        Returns the number of destination lanes."""
        src = np.ascontiguousarray(src_lanes, dtype=np.int32).reshape(-1)
        off, items = self.pack_actions(actions)
        lb = None if last_bus is None else np.ascontiguousarray(last_bus, dtype=np.int32).reshape(src.size, self.model.dim_topo)
        o = GpfStepOpts(int(max_iter), float(tol_mva), float(rebalance), int(bool(cascade)), float(hard_overflow), float(soft_overflow),
                        int(nb_ts_allowed), int(max_rounds), int(bool(is_dc)), 0, 0, 0, 0)
        check(self._lib.gpf_simulate_batch(self._h, int(t_obs), int(time_step), src.size, ptr(src, C.c_int32), len(actions),
                                           ptr(off, C.c_int32), ptr(items if items.size else None, C.c_int32), ptr(lb, C.c_int32),
                                           int(dst_lane0), C.byref(o)), "gpf_simulate_batch")
        return src.size * len(actions)

    # ---- solve --------------------------------------------------------------------------------------
    def runpf(self, lane0: int = 0, n: Optional[int] = None, is_dc: bool = False, max_iter: int = 10,
              tol_mva: float = 1e-8):
        lane0, n = self._range(lane0, n)
        check(self._lib.gpf_runpf(self._h, lane0, n, int(bool(is_dc)), int(max_iter), float(tol_mva)), "gpf_runpf")

    def solve_lane(self, lane: int, inj: np.ndarray, topo: np.ndarray, shunt_bus: Optional[np.ndarray] = None, is_dc: bool = False,
                   max_iter: int = 10, tol_mva: float = 1e-8) -> LaneResults:
        """`set_injections` + `set_topology` + `runpf` + `results` of ONE lane in one C call with a single synchronisation: the
        whole of a Backend's ``runpf`` (what `HipBackend` uses)."""
        m = self.model
        inj = np.ascontiguousarray(inj, dtype=np.float64).reshape(self.n_inj)
        topo = np.ascontiguousarray(topo, dtype=np.int32).reshape(m.dim_topo)
        sb_in = None if not m.n_shunt else np.ascontiguousarray(shunt_bus, dtype=np.int32).reshape(m.n_shunt)
        out = np.empty((1, self.n_out), dtype=np.float32)
        tv = np.empty((1, m.dim_topo), dtype=np.int32)
        sb = np.empty((1, m.n_shunt), dtype=np.int32)
        ls = np.empty((1, m.n_line), dtype=np.uint8)
        st = np.empty((1, 4), dtype=np.int32)
        bvm = np.empty((1, self.nb_total), dtype=np.float64)
        bva = np.empty((1, self.nb_total), dtype=np.float64)
        check(self._lib.gpf_solve_lane(self._h, int(lane), ptr(inj, C.c_double), ptr(topo, C.c_int32), ptr(sb_in, C.c_int32),
                                       int(bool(is_dc)), int(max_iter), float(tol_mva), ptr(out, C.c_float), ptr(tv, C.c_int32),
                                       ptr(sb, C.c_int32), ptr(ls, C.c_uint8), ptr(st, C.c_int32), ptr(bvm, C.c_double),
                                       ptr(bva, C.c_double)), "gpf_solve_lane")
        return LaneResults(out=out, topo_vect=tv, shunt_bus=sb, line_status=ls.astype(bool), status=st, bus_vm=bvm, bus_va=bva,
                           _slices=self.out_slices)

    def results(self, lane0: int = 0, n: Optional[int] = None, with_bus: bool = True, pinned: bool = False) -> LaneResults:
        """The result rows of lanes ``[lane0, lane0 + n)`` on the host.  ``pinned=True``: the arrays ALIAS a pinned block the engine owns
        (gpf_get_results_pinned: DMA at the PCIe rate, no second host copy) and are only valid until the next ``results(pinned=True)``
        call -- for a host agent that reads every lane at every step; copy what must outlive the step."""
        lane0, n = self._range(lane0, n)
        m = self.model
        if pinned:
            ptrs = (C.c_void_p * 8)()
            what = 0b0011111 | (0b1100000 if with_bus else 0)
            check(self._lib.gpf_get_results_pinned(self._h, lane0, n, what, ptrs), "gpf_get_results_pinned")

            def arr(k, cols, ctype, dtype):
                if not ptrs[k] or cols == 0:
                    return np.empty((n, cols), dtype=dtype)
                return np.ctypeslib.as_array(C.cast(ptrs[k], C.POINTER(ctype)), shape=(n, cols))
            return LaneResults(out=arr(0, self.n_out, C.c_float, np.float32), topo_vect=arr(1, m.dim_topo, C.c_int32, np.int32),
                               shunt_bus=arr(2, m.n_shunt, C.c_int32, np.int32), line_status=arr(3, m.n_line, C.c_uint8, np.uint8).view(np.bool_),
                               status=arr(4, 4, C.c_int32, np.int32), bus_vm=arr(5, self.nb_total, C.c_double, np.float64) if with_bus else None,
                               bus_va=arr(6, self.nb_total, C.c_double, np.float64) if with_bus else None, _slices=self.out_slices)
        out = np.empty((n, self.n_out), dtype=np.float32)
        tv = np.empty((n, m.dim_topo), dtype=np.int32)
        sb = np.empty((n, m.n_shunt), dtype=np.int32)
        ls = np.empty((n, m.n_line), dtype=np.uint8)
        st = np.empty((n, 4), dtype=np.int32)
        bvm = np.empty((n, self.nb_total), dtype=np.float64) if with_bus else None
        bva = np.empty((n, self.nb_total), dtype=np.float64) if with_bus else None
        check(self._lib.gpf_get_results(self._h, lane0, n, ptr(out, C.c_float), ptr(tv, C.c_int32), ptr(sb, C.c_int32),
                                        ptr(ls, C.c_uint8), ptr(st, C.c_int32), ptr(bvm, C.c_double), ptr(bva, C.c_double)),
              "gpf_get_results")
        return LaneResults(out=out, topo_vect=tv, shunt_bus=sb, line_status=ls.astype(bool), status=st, bus_vm=bvm,
                           bus_va=bva, _slices=self.out_slices)

    # ---- batched stepping -----------------------------------------------------------------------------
    def pack_chronics(self, load_p, load_q, prod_p, prod_v) -> np.ndarray:
        """``[..., T, n_chron]`` float32 table from the four ``[..., T, n]`` chronics arrays."""
        return np.ascontiguousarray(np.concatenate([load_p, load_q, prod_p, prod_v], axis=-1), dtype=np.float32)

    def upload_chronics(self, tables: np.ndarray):
        tables = np.ascontiguousarray(tables, dtype=np.float32)
        if tables.ndim == 2:
            tables = tables[None]
        assert tables.shape[2] == self.n_chron, (tables.shape, self.n_chron)
        check(self._lib.gpf_upload_chronics(self._h, tables.shape[0], tables.shape[1], ptr(tables, C.c_float)),
              "gpf_upload_chronics")
        self.chron_T = tables.shape[1]

    def upload_maintenance(self, maintenance):
        """Scheduled maintenance of the uploaded tables: ``[n_tables, T, n_line]`` (or ``[T, n_line]``) 0/1; None removes it."""
        self._has_maint = maintenance is not None
        self._has_outages = self._has_maint or getattr(self, "_has_hazard", False)
        if maintenance is None:
            check(self._lib.gpf_upload_maintenance(self._h, 0, 0, None), "gpf_upload_maintenance")
            return
        mt = np.ascontiguousarray(maintenance, dtype=np.uint8)
        if mt.ndim == 2:
            mt = mt[None]
        assert mt.shape[2] == self.model.n_line
        check(self._lib.gpf_upload_maintenance(self._h, mt.shape[0], mt.shape[1], ptr(mt, C.c_uint8)), "gpf_upload_maintenance")

    def upload_outage_durations(self, durations):
        """Remaining duration of the maintenance / hazard under way at every row, ``[n_tables, T, n_line]`` (or ``[T, n_line]``): what the line
        cooldowns are held at during an outage.  Only needed when the uploaded tables are a window of longer chronics (the library derives
        the durations from the outage tables otherwise); None: derived again."""
        if durations is None:
            check(self._lib.gpf_upload_outage_durations(self._h, 0, 0, None), "gpf_upload_outage_durations")
            return
        d = np.ascontiguousarray(np.minimum(np.asarray(durations), 65535), dtype=np.uint16)
        if d.ndim == 2:
            d = d[None]
        assert d.shape[2] == self.model.n_line
        check(self._lib.gpf_upload_outage_durations(self._h, d.shape[0], d.shape[1], d.ctypes.data_as(C.POINTER(C.c_uint16))), "gpf_upload_outage_durations")

    def upload_hazards(self, hazards):
        """Hazards of the uploaded tables (``hazards.csv``: unplanned outages): ``[n_tables, T, n_line]`` (or ``[T, n_line]``) 0/1; None
        removes them.  Independent of the maintenance table; a line is out of service where either flags it."""
        self._has_hazard = hazards is not None
        self._has_outages = self._has_hazard or getattr(self, "_has_maint", False)
        if hazards is None:
            check(self._lib.gpf_upload_hazards(self._h, 0, 0, None), "gpf_upload_hazards")
            return
        hz = np.ascontiguousarray(hazards, dtype=np.uint8)
        if hz.ndim == 2:
            hz = hz[None]
        assert hz.shape[2] == self.model.n_line
        check(self._lib.gpf_upload_hazards(self._h, hz.shape[0], hz.shape[1], ptr(hz, C.c_uint8)), "gpf_upload_hazards")

    def set_lane_chronics(self, lane_table=None, lane_offset=None, lane_scale=None):
        lt = None if lane_table is None else np.ascontiguousarray(lane_table, dtype=np.int32)
        lo = None if lane_offset is None else np.ascontiguousarray(lane_offset, dtype=np.int32)
        ls = None if lane_scale is None else np.ascontiguousarray(lane_scale, dtype=np.float32)
        if lt is not None:
            assert lt.size == self.n_lanes
        if lo is not None:
            assert lo.size == self.n_lanes
        if ls is not None:
            assert ls.shape == (self.n_lanes, 2 * self.model.n_load)
        check(self._lib.gpf_set_lane_chronics(self._h, ptr(lt, C.c_int32), ptr(lo, C.c_int32), ptr(ls, C.c_float)),
              "gpf_set_lane_chronics")

    def set_thermal_limits(self, limit_a):
        lim = np.ascontiguousarray(limit_a, dtype=np.float32)
        assert lim.size == self.model.n_line
        check(self._lib.gpf_set_thermal_limits(self._h, ptr(lim, C.c_float)), "gpf_set_thermal_limits")

    def step(self, t: int, max_iter: int = 10, tol_mva: float = 1e-8, rebalance: float = 0.0, cascade: bool = False,
             hard_overflow: float = 2.0, soft_overflow: float = 1.0, nb_ts_allowed: int = 2, max_rounds: int = 16,
             is_dc: bool = False, n_steps: int = 1, auto_reset: bool = False, warm_start: bool = False, nb_ts_reco: Optional[int] = None):
        """``n_steps`` consecutive DoNothing ``env.step`` (t, t+1, ...) for every lane in ONE launch (asynchronous).  Every step
        writes its results; the getters return the last one, `trajectory` the rho / status of each when requested.
        ``warm_start`` (opt-in, NOT the reference's algorithm) starts Newton of steps 2..n from the previous step's voltages
        while a lane's topology stands: same solution within ``tol_mva``, fewer iterations, ``n_iter`` differs.
        ``nb_ts_reco`` (Parameters.NB_TIMESTEP_RECONNECTION): the environment's line cooldowns (`cooldown`, obs.time_before_cooldown_line) are
        maintained at every step -- default: 10 (the reference's default) when lines can go out by themselves (``cascade`` or uploaded
        maintenance / hazard tables), else not tracked; -1 switches the tracking off."""
        if nb_ts_reco is None:
            nb_ts_reco = 10 if (cascade or getattr(self, "_has_outages", False)) else -1
        o = GpfStepOpts(int(max_iter), float(tol_mva), float(rebalance), int(bool(cascade)), float(hard_overflow), float(soft_overflow),
                        int(nb_ts_allowed), int(max_rounds), int(bool(is_dc)), int(bool(auto_reset)), int(bool(warm_start)), int(nb_ts_reco >= 0), max(int(nb_ts_reco), 0))
        check(self._lib.gpf_step_n(self._h, int(t), int(n_steps), C.byref(o)), "gpf_step_n")

    def set_lane_redispatch(self, delta_mw):
        """Per-lane additive generator set-point delta (MW, ``[n_lanes, n_gen]``; None switches it off): the redispatch the
        environment adds to the chronics' prod_p."""
        d = None if delta_mw is None else np.ascontiguousarray(delta_mw, dtype=np.float32)
        if d is not None:
            assert d.shape == (self.n_lanes, self.model.n_gen)
        check(self._lib.gpf_set_lane_redispatch(self._h, ptr(d, C.c_float)), "gpf_set_lane_redispatch")

    def set_gen_limits(self, pmin, pmax, ramp_up, ramp_down, redispatchable, eps_poly: float = 1e-4):
        """Generator characteristics (``prods_charac.csv``: Pmin, Pmax, max_ramp_up, max_ramp_down, redispatchable) for
        `redispatch`."""
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        a = [f64(pmin), f64(pmax), f64(ramp_up), f64(ramp_down)]
        r = np.ascontiguousarray(redispatchable, dtype=np.uint8)
        assert all(x.size == self.model.n_gen for x in a) and r.size == self.model.n_gen
        check(self._lib.gpf_set_gen_limits(self._h, *[ptr(x, C.c_double) for x in a], ptr(r, C.c_uint8), float(eps_poly)),
              "gpf_set_gen_limits")

    def redispatch(self, new_p, prev_p, actual, target, modified, rhs, lane0: int = 0, apply: bool = False):
        """The environment's redispatching automaton for a batch of lanes (``BaseEnv._compute_dispatch_vect``): rows
        ``[n, n_gen]`` of new_p / prev_p / actual dispatch / target dispatch / modified mask, ``rhs[n]`` = storage - curtailment
        + detached MW.  Returns ``(ok[n] bool, actual_dispatch_after[n, n_gen] float32)``; ``apply`` also installs the result as
        the lanes' redispatch delta for the next `step`."""
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64).reshape(-1, self.model.n_gen)  # noqa: E731
        new_p, prev_p, actual, target = f64(new_p), f64(prev_p), f64(actual), f64(target)
        n = new_p.shape[0]
        mod = np.ascontiguousarray(modified, dtype=np.uint8).reshape(n, self.model.n_gen)
        rhs = np.ascontiguousarray(np.broadcast_to(np.asarray(rhs, dtype=np.float64), (n,)))
        ok = np.empty(n, dtype=np.uint8)
        after = np.empty((n, self.model.n_gen), dtype=np.float32)
        check(self._lib.gpf_redispatch(self._h, int(lane0), n, ptr(new_p, C.c_double), ptr(prev_p, C.c_double), ptr(actual, C.c_double),
                                       ptr(target, C.c_double), ptr(mod, C.c_uint8), ptr(rhs, C.c_double), int(bool(apply)),
                                       ptr(ok, C.c_uint8), ptr(after, C.c_float)), "gpf_redispatch")
        return ok.astype(bool), after

    TRAJ_RHO, TRAJ_OBS = 1, 2

    # ---- injection dynamics of the environment (storage state of charge, redispatch projection) inside the stepped batch ------
    def set_storage_params(self, emax, emin, loss, eff_charge, eff_discharge, charge0, delta_time_seconds: float = 300.0,
                           activate_loss: bool = True):
        """Storage characteristics (``storage_Emax / Emin / loss / charging_efficiency / discharging_efficiency``, initial charge
        in MWh), the step length and ``Parameters.ACTIVATE_STORAGE_LOSS``."""
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64).reshape(self.model.n_storage)  # noqa: E731
        a = [f64(emax), f64(emin), f64(loss), f64(eff_charge), f64(eff_discharge)]
        c0 = np.ascontiguousarray(charge0, dtype=np.float32).reshape(self.model.n_storage)
        check(self._lib.gpf_set_storage_params(self._h, *[ptr(x, C.c_double) for x in a], ptr(c0, C.c_float), float(delta_time_seconds),
                                               int(bool(activate_loss))), "gpf_set_storage_params")

    def set_env_dynamics(self, on: bool = True, tol_poly: float = 1e-2):
        """Switch the environment's injection dynamics on: every step of `step` then evolves the storage state of charge and
        re-solves the ramp-limited redispatch (BaseEnv.step's path between the chronics and the backend); needs `set_gen_limits`
        (and `set_storage_params` on a grid with storage units).  Resets the dynamics of every lane."""
        check(self._lib.gpf_set_env_dynamics(self._h, int(bool(on)), float(tol_poly)), "gpf_set_env_dynamics")
        self.env_dynamics_on = bool(on)

    def set_lane_actions(self, redispatch=None, storage_power=None, hold_storage: bool = False):
        """The agents' actions of the NEXT launch: redispatch ``[n_lanes, n_gen]`` MW (consumed by its first step), storage power
        ``[n_lanes, n_storage]`` MW (first step only, or every step until replaced with ``hold_storage``)."""
        r = None if redispatch is None else np.ascontiguousarray(redispatch, dtype=np.float32).reshape(self.n_lanes, self.model.n_gen)
        s_ = None if storage_power is None or not self.model.n_storage else \
            np.ascontiguousarray(storage_power, dtype=np.float32).reshape(self.n_lanes, self.model.n_storage)
        check(self._lib.gpf_set_lane_actions(self._h, ptr(r, C.c_float), ptr(s_, C.c_float), int(bool(hold_storage))), "gpf_set_lane_actions")

    def lane_actions_on_device(self, redispatch: bool = False, storage_power: bool = False, curtailment: bool = False,
                               hold_storage: bool = False):
        """The actions of the NEXT launch were written ON THE DEVICE, into ``device_views()["act_redispatch" / "act_storage" /
        "act_curtail"]`` (on ``views["stream"]`` or ordered before the launch): say which buffers hold an action.  Same semantics as
        `set_lane_actions` / `set_lane_curtailment`, no PCIe transfer, no synchronisation -- the hand-over of an agent that lives next
        to the engine (curtailment ratios are NOT range-checked on this path)."""
        check(self._lib.gpf_lane_actions_on_device(self._h, int(bool(redispatch)), int(bool(storage_power)), int(bool(curtailment)),
                                                   int(bool(hold_storage))), "gpf_lane_actions_on_device")

    def set_gen_renewable(self, renewable):
        """``gen_renewable`` mask (curtailment only acts on these generators); None switches curtailment off."""
        r = None if renewable is None else np.ascontiguousarray(renewable, dtype=np.uint8).reshape(self.model.n_gen)
        check(self._lib.gpf_set_gen_renewable(self._h, ptr(r, C.c_uint8)), "gpf_set_gen_renewable")

    def set_lane_curtailment(self, limit):
        """Curtailment action of the NEXT launch: ``[n_lanes, n_gen]`` ratios of pmax in [0, 1], -1 = no change (consumed by the
        launch's first step; the limits then stay in the lanes' state until changed)."""
        a = None if limit is None else np.ascontiguousarray(limit, dtype=np.float32).reshape(self.n_lanes, self.model.n_gen)
        check(self._lib.gpf_set_lane_curtailment(self._h, ptr(a, C.c_float)), "gpf_set_lane_curtailment")

    def env_state(self, lane0: int = 0, n: Optional[int] = None) -> dict:
        """``target`` / ``actual`` dispatch, ``prev_p``, ``already_modified`` ``[n, n_gen]``, ``charge`` ``[n, n_storage]``,
        ``amount_prev`` ``[n]`` of the lanes' environment dynamics; ``illegal`` ``[n]``: actions cancelled as illegal redispatch since the reset."""
        lane0, n = self._range(lane0, n)
        ng, ns = self.model.n_gen, self.model.n_storage
        d = dict(target=np.empty((n, ng), np.float32), actual=np.empty((n, ng), np.float32), prev_p=np.empty((n, ng), np.float32),
                 already_modified=np.empty((n, ng), np.uint8), charge=np.empty((n, ns), np.float32), amount_prev=np.empty(n, np.float32),
                 curtail_limit=np.empty((n, ng), np.float32), curtail_prev=np.empty(n, np.float32))
        check(self._lib.gpf_get_env_state(self._h, lane0, n, ptr(d["target"], C.c_float), ptr(d["actual"], C.c_float),
                                          ptr(d["prev_p"], C.c_float), ptr(d["already_modified"], C.c_uint8),
                                          ptr(d["charge"] if ns else None, C.c_float), ptr(d["amount_prev"], C.c_float),
                                          ptr(d["curtail_limit"], C.c_float), ptr(d["curtail_prev"], C.c_float)), "gpf_get_env_state")
        d["already_modified"] = d["already_modified"].astype(bool)
        d["illegal"] = np.empty(n, np.int32)             # steps whose action BaseEnv.step would have cancelled as an illegal redispatch
        check(self._lib.gpf_get_env_illegal(self._h, lane0, n, ptr(d["illegal"], C.c_int32)), "gpf_get_env_illegal")
        return d

    def set_env_state(self, lane0: int = 0, target=None, actual=None, prev_p=None, already_modified=None, charge=None, amount_prev=None,
                      curtail_limit=None, curtail_prev=None, illegal=None):
        """Overwrite (parts of) the lanes' environment dynamics, e.g. to restore them from an observation.  Accepts every key
        `env_state` returns (``eng.set_env_state(l0, **eng.env_state(l1, n))`` moves the complete state of n lanes)."""
        f = lambda a, w: None if a is None else np.ascontiguousarray(a, dtype=np.float32).reshape(-1, w)  # noqa: E731
        ng, ns = self.model.n_gen, max(self.model.n_storage, 1)
        arrs = [f(target, ng), f(actual, ng), f(prev_p, ng)]
        am = None if already_modified is None else np.ascontiguousarray(already_modified, dtype=np.uint8).reshape(-1, ng)
        ch = None if charge is None or not self.model.n_storage else f(charge, ns)
        ap = None if amount_prev is None else np.ascontiguousarray(amount_prev, dtype=np.float32).reshape(-1)
        cl = f(curtail_limit, ng)
        cp = None if curtail_prev is None else np.ascontiguousarray(curtail_prev, dtype=np.float32).reshape(-1)
        il = None if illegal is None else np.ascontiguousarray(illegal, dtype=np.int32).reshape(-1)
        n = next((x.shape[0] for x in arrs + [am, ch, ap, cl, cp] if x is not None), None)
        if il is not None:
            check(self._lib.gpf_set_env_illegal(self._h, int(lane0), il.shape[0], ptr(il, C.c_int32)), "gpf_set_env_illegal")
        if n is None:
            return
        check(self._lib.gpf_set_env_state(self._h, int(lane0), n, ptr(arrs[0], C.c_float), ptr(arrs[1], C.c_float), ptr(arrs[2], C.c_float),
                                          ptr(am, C.c_uint8), ptr(ch, C.c_float), ptr(ap, C.c_float), ptr(cl, C.c_float), ptr(cp, C.c_float)),
              "gpf_set_env_state")

    def set_trajectory(self, n_steps_cap: int, what: int = 1):
        """Trajectory buffers of multi-step launches: ``what`` = `TRAJ_RHO` (rho + status of every step) or `TRAJ_OBS` (in
        addition the complete backend observation of every step: results row, topo_vect, shunt buses, line status).
        ``n_steps_cap = 0`` releases them."""
        check(self._lib.gpf_set_trajectory(self._h, int(n_steps_cap), int(what)), "gpf_set_trajectory")
        self._traj_cap = int(n_steps_cap) if what else 0

    def trajectory(self, n_steps: int, step0: int = 0, lane0: int = 0, n: Optional[int] = None):
        """(rho ``[n_steps, n, n_line]`` float32, status ``[n_steps, n]`` int8) of the steps of the last multi-step launch."""
        lane0, n = self._range(lane0, n)
        rho = np.empty((n_steps, n, self.model.n_line), dtype=np.float32)
        st = np.empty((n_steps, n), dtype=np.int8)
        check(self._lib.gpf_get_trajectory(self._h, int(step0), int(n_steps), lane0, n, ptr(rho, C.c_float), ptr(st, C.c_int8)),
              "gpf_get_trajectory")
        return rho, st

    def trajectory_obs(self, n_steps: int, step0: int = 0, lane0: int = 0, n: Optional[int] = None):
        """The backend observation of every step of the last multi-step launch (`set_trajectory(cap, TRAJ_OBS)`): a list of
        `LaneResults` (one per step; ``status`` column 0 from the status trajectory, bus voltages are not part of it)."""
        lane0, n = self._range(lane0, n)
        m = self.model
        out = np.empty((n_steps, n, self.n_out), dtype=np.float32)
        tv = np.empty((n_steps, n, m.dim_topo), dtype=np.int32)
        sb = np.empty((n_steps, n, m.n_shunt), dtype=np.int32)
        ls = np.empty((n_steps, n, m.n_line), dtype=np.uint8)
        check(self._lib.gpf_get_trajectory_obs(self._h, int(step0), int(n_steps), lane0, n, ptr(out, C.c_float), ptr(tv, C.c_int32),
                                               ptr(sb, C.c_int32), ptr(ls, C.c_uint8)), "gpf_get_trajectory_obs")
        _, st = self.trajectory(n_steps, step0, lane0, n)
        res = []
        for k in range(n_steps):
            st4 = np.full((n, 4), -1, dtype=np.int32)
            st4[:, 0] = st[k]
            res.append(LaneResults(out=out[k], topo_vect=tv[k], shunt_bus=sb[k], line_status=ls[k].astype(bool), status=st4,
                                   bus_vm=None, bus_va=None, _slices=self.out_slices))
        return res

    def episode(self, lane0: int = 0, n: Optional[int] = None):
        """(done ``[n]`` bool, steps survived since the last reset ``[n]``, auto-resets ``[n]``)."""
        lane0, n = self._range(lane0, n)
        done = np.empty(n, dtype=np.uint8)
        sr = np.empty((n, 2), dtype=np.int32)
        check(self._lib.gpf_get_episode(self._h, lane0, n, ptr(done, C.c_uint8), ptr(sr, C.c_int32)), "gpf_get_episode")
        return done.astype(bool), sr[:, 0].copy(), sr[:, 1].copy()

    # ---- zero-copy device views ------------------------------------------------------------------------------------------
    def device_views(self):
        """The engine's result buffers as torch tensors that ALIAS the device memory (no copy, no PCIe): ``out`` float32
        ``[n_lanes, n_out]`` (columns: `out_slices`), ``rho`` ``[n_lanes, n_line]``, ``status`` int32 ``[n_lanes, 4]``,
        ``topo_vect``, ``line_status`` uint8, ``overflow_count``, ``done`` uint8, ``episode`` int32 ``[n_lanes, 2]``, ``inj``
        float64, ``bus_vm`` / ``bus_va`` float64; with the environment dynamics on also the action buffers ``act_redispatch`` /
        ``act_curtail`` ``[n_lanes, n_gen]``, ``act_storage`` ``[n_lanes, n_storage]`` (`lane_actions_on_device`) and
        ``target_dispatch`` / ``actual_dispatch`` / ``storage_charge`` float32 (obs.target_dispatch, ...).  The engine works on its own HIP stream: call `sync` (or make the consumer's
        stream wait on ``views["stream"]``, a ``torch.cuda.ExternalStream``) before reading."""
        import torch
        ptrs = (C.c_void_p * 28)()
        stream = C.c_void_p()
        check(self._lib.gpf_device_pointers_n(self._h, ptrs, 28, C.byref(stream)), "gpf_device_pointers_n")
        cap = self._lib.gpf_lane_capacity(self._h)
        m = self.model
        dev = torch.device("cuda", self.device)

        class _Arr:                          # __cuda_array_interface__ v2: torch.as_tensor wraps it without a copy
            def __init__(self, p, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(p), False), "version": 2,
                                                 "strides": None}

        def view(idx, cols, typestr):
            if cols == 0 or not ptrs[idx]:
                return None
            t = torch.as_tensor(_Arr(ptrs[idx], (cap, cols), typestr), device=dev)
            return t[:self.n_lanes]
        v = {"inj": view(0, self.n_inj, "<f8"), "topo": view(1, m.dim_topo, "<i4"), "out": view(3, self.n_out, "<f4"),
             "topo_vect": view(4, m.dim_topo, "<i4"), "line_status": view(5, m.n_line, "|u1"), "status": view(6, 4, "<i4"),
             "rho": view(8, m.n_line, "<f4"), "overflow_count": view(9, m.n_line, "<i4"), "done": view(10, 1, "|u1"),
             "episode": view(11, 2, "<i4"), "bus_vm": view(12, self.nb_total, "<f8"), "bus_va": view(13, self.nb_total, "<f8"),
             "disc_round": view(15, m.n_line, "<i4"),
             "act_redispatch": view(22, m.n_gen, "<f4"), "act_storage": view(23, m.n_storage, "<f4"), "act_curtail": view(24, m.n_gen, "<f4"),
             "target_dispatch": view(25, m.n_gen, "<f4"), "actual_dispatch": view(26, m.n_gen, "<f4"),
             "storage_charge": view(27, m.n_storage, "<f4")}

        def tview(idx, cols, typestr):       # trajectory buffers: [cap_steps][cap][cols]
            if cols == 0 or not ptrs[idx] or not getattr(self, "_traj_cap", 0):
                return None
            t = torch.as_tensor(_Arr(ptrs[idx], (self._traj_cap, cap, cols), typestr), device=dev)
            return t[:, :self.n_lanes]
        v.update({"traj_rho": tview(16, m.n_line, "<f4"), "traj_status": tview(17, 1, "|i1"), "traj_out": tview(18, self.n_out, "<f4"),
                  "traj_topo_vect": tview(19, m.dim_topo, "<i4"), "traj_shunt_bus": tview(20, m.n_shunt, "<i4"),
                  "traj_line_status": tview(21, m.n_line, "|u1")})
        v["stream"] = torch.cuda.ExternalStream(stream.value, device=dev)
        return v

    def set_overflow_count(self, counts, lane0: int = 0):
        """The protection counters (consecutive steps above the thermal limit, ``obs.timestep_overflow``) of lanes
        ``lane0 ...``: ``[n, n_line]`` ints."""
        c = np.ascontiguousarray(counts, dtype=np.int32).reshape(-1, self.model.n_line)
        check(self._lib.gpf_set_overflow_count(self._h, int(lane0), c.shape[0], ptr(c, C.c_int32)), "gpf_set_overflow_count")

    def cooldown(self, lane0: int = 0, n: Optional[int] = None) -> np.ndarray:
        """The environment's line cooldowns of the lanes (obs.time_before_cooldown_line; `step(nb_ts_reco=...)`), int32 ``[n, n_line]``."""
        lane0, n = self._range(lane0, n)
        out = np.empty((n, self.model.n_line), dtype=np.int32)
        check(self._lib.gpf_get_cooldown(self._h, lane0, n, ptr(out, C.c_int32)), "gpf_get_cooldown")
        return out

    def set_cooldown(self, line_cooldown, lane0: int = 0):
        """Restore the line cooldowns (an environment restored from an observation hands over obs.time_before_cooldown_line)."""
        c = np.ascontiguousarray(line_cooldown, dtype=np.int32).reshape(-1, self.model.n_line)
        check(self._lib.gpf_set_cooldown(self._h, int(lane0), c.shape[0], ptr(c, C.c_int32)), "gpf_set_cooldown")

    def trajectory_cooldown(self, n_steps: int, step0: int = 0, lane0: int = 0, n: Optional[int] = None) -> np.ndarray:
        """Line cooldowns after every step of the last multi-step launch, int16 ``[n_steps, n, n_line]`` (needs `set_trajectory`)."""
        lane0, n = self._range(lane0, n)
        out = np.empty((n_steps, n, self.model.n_line), dtype=np.int16)
        check(self._lib.gpf_get_trajectory_cooldown(self._h, int(step0), int(n_steps), lane0, n, out.ctypes.data_as(C.POINTER(C.c_int16))), "gpf_get_trajectory_cooldown")
        return out

    def step_outputs(self, lane0: int = 0, n: Optional[int] = None):
        lane0, n = self._range(lane0, n)
        nl = self.model.n_line
        rho = np.empty((n, nl), dtype=np.float32)
        oc = np.empty((n, nl), dtype=np.int32)
        dr = np.empty((n, nl), dtype=np.int32)
        check(self._lib.gpf_get_step_outputs(self._h, lane0, n, ptr(rho, C.c_float), ptr(oc, C.c_int32), ptr(dr, C.c_int32)),
              "gpf_get_step_outputs")
        return rho, oc, dr

    # ---- measurement -------------------------------------------------------------------------------------
    # ---- DC sensitivity (PTDF) path: fixed topology, flows = PTDF * P_bus as one FP64 MFMA GEMM ----------------------
    def ptdf_build(self, lane: int = 0):
        """Factorise the DC system of the topology currently held by ``lane`` (once per topology)."""
        check(self._lib.gpf_ptdf_build(self._h, int(lane)), "gpf_ptdf_build")

    def ptdf_build_batch(self, lane0: int = 0, n: Optional[int] = None, with_lodf: bool = True, info: bool = True) -> dict:
        """PTDF (and LODF) tables of EVERY distinct topology the lanes ``[lane0, lane0 + n)`` hold right now, built on the device in
        one launch (gpf_ptdf_build_batch: one workgroup per topology class, blocked Gauss-Jordan on the FP64 matrix cores).  Afterwards
        `ptdf_flows` / `ptdf_flows_rows` / `lodf_screen` evaluate every lane against the tables of its own class.  Returns
        ``lane_class`` [n], ``class_status`` [n_classes] (0 ok, 1 singular, 2 islanded, 3 no slack: the lanes of such a class get NaN
        flows), ``class_n`` (dimension of each reduced B'), ``kernel_ms``."""
        lane0, n = self._range(lane0, n)
        nc = C.c_int32(0)
        check(self._lib.gpf_ptdf_build_batch(self._h, lane0, n, 1 if with_lodf else 0, C.byref(nc)), "gpf_ptdf_build_batch")
        if not info:                                        # asynchronous: the build kernel is queued, nobody waits (`ptdf_batch_info()` later)
            return {"n_classes": int(nc.value)}
        return self.ptdf_batch_info(n, nc.value)

    def ptdf_batch_info(self, n: Optional[int] = None, n_classes: Optional[int] = None) -> dict:
        """Lane -> class map, class status / dimension and the kernel time of the last `ptdf_build_batch` (waits for its kernel)."""
        if n is None or n_classes is None:
            raise ValueError("ptdf_batch_info: pass the lane count and the class count of the build call")
        nc = C.c_int32(int(n_classes))
        lc, st, cn = np.empty(n, np.int32), np.empty(nc.value, np.int32), np.empty(nc.value, np.int32)
        ms = C.c_double(0.0)
        check(self._lib.gpf_ptdf_batch_info(self._h, ptr(lc, C.c_int32), ptr(st, C.c_int32), ptr(cn, C.c_int32), C.byref(ms)), "gpf_ptdf_batch_info")
        return {"n_classes": int(nc.value), "lane_class": lc, "class_status": st, "class_n": cn, "kernel_ms": float(ms.value)}

    def ptdf_class(self, cls: int, lodf: bool = False):
        """PTDF [n_line, n_sub * n_busbar] of topology class ``cls`` of the last `ptdf_build_batch` (and its LODF [n_line, n_line])."""
        out = np.empty((self.model.n_line, self.nb_total), dtype=np.float64)
        lo = np.empty((self.model.n_line, self.model.n_line), dtype=np.float64) if lodf else None
        check(self._lib.gpf_ptdf_batch_get(self._h, int(cls), ptr(out, C.c_double), ptr(lo, C.c_double)), "gpf_ptdf_batch_get")
        return (out, lo) if lodf else out

    def ptdf(self) -> np.ndarray:
        """PTDF [n_line, n_sub * n_busbar] (MW of origin-side flow per MW injected at the bus, slack-referenced)."""
        out = np.empty((self.model.n_line, self.nb_total), dtype=np.float64)
        check(self._lib.gpf_ptdf_get(self._h, ptr(out, C.c_double)), "gpf_ptdf_get")
        return out

    def ptdf_flows(self, lane0: int = 0, n: Optional[int] = None, fetch: bool = True) -> Optional[np.ndarray]:
        """DC active-power flows (MW, float32 [n, n_line]) of the lanes' current injection rows (asynchronous launch;
        ``fetch=False`` leaves the result on the device)."""
        lane0, n = self._range(lane0, n)
        check(self._lib.gpf_ptdf_flows(self._h, lane0, n), "gpf_ptdf_flows")
        if not fetch:
            return None
        out = np.empty((n, self.model.n_line), dtype=np.float32)
        check(self._lib.gpf_get_ptdf_flows(self._h, lane0, n, ptr(out, C.c_float)), "gpf_get_ptdf_flows")
        return out

    def ptdf_flows_rows(self, t0: int, n_rows: int, rebalance: float = 0.0, fetch: bool = True, lane0: int = 0, n: Optional[int] = None):
        """DC active-power flows of ``n_rows`` consecutive chronics rows of EVERY lane in ONE launch (the GEMM has
        ``n_lanes * n_rows`` rows): row ``j`` of lane ``k`` is chronics row ``(t0 + j + lane_offset[k]) mod T`` turned into injections
        as `step` does.  Returns float32 ``[n_rows, n, n_line]`` (MW at the origin side) of lanes ``[lane0, lane0 + n)``, or None with
        ``fetch=False`` (asynchronous)."""
        check(self._lib.gpf_ptdf_flows_rows(self._h, int(t0), int(n_rows), float(rebalance)), "gpf_ptdf_flows_rows")
        if not fetch:
            return None
        lane0, n = self._range(lane0, n)
        out = np.empty((n_rows, n, self.model.n_line), dtype=np.float32)
        check(self._lib.gpf_get_ptdf_flows_rows(self._h, 0, int(n_rows), lane0, n, ptr(out, C.c_float)), "gpf_get_ptdf_flows_rows")
        return out

    def lodf_screen(self, lane0: int = 0, n: Optional[int] = None, cap_mw: Optional[np.ndarray] = None) -> np.ndarray:
        """DC N-1 screening from the flows of the last ``ptdf_flows``: [n, n_line] largest post-outage loading
        max_l |f_l + LODF[l, k] f_k| / cap_mw[l] for every single-line outage k (MW if ``cap_mw`` is None; inf: the
        outage islands the grid)."""
        lane0, n = self._range(lane0, n)
        out = np.empty((n, self.model.n_line), dtype=np.float32)
        cap = None if cap_mw is None else np.ascontiguousarray(cap_mw, dtype=np.float32)
        check(self._lib.gpf_lodf_screen(self._h, lane0, n, ptr(cap, C.c_float), ptr(out, C.c_float)), "gpf_lodf_screen")
        return out

    def sync(self):
        check(self._lib.gpf_sync(self._h), "gpf_sync")

    def set_profiling(self, mode):
        """0/False: off; 1/True: one HIP event pair around the window of launches up to the next ``kernel_time()``;
        2: an event pair per launch (exact per-kernel durations, costs ~7 us of stream time per launch); 3 (inside a window): the
        window ends at this point of the stream (recorded asynchronously behind the launches issued so far)."""
        check(self._lib.gpf_set_profiling(self._h, int(mode)), "gpf_set_profiling")

    def kernel_time(self) -> Tuple[float, int]:
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        check(self._lib.gpf_get_kernel_time(self._h, C.byref(ms), C.byref(n)), "gpf_get_kernel_time")
        return ms.value, n.value

    def specialize(self, enable: bool = True, cache_dir: str = None, verify: bool = True) -> dict:
        """Switch the engine's solver launches (`step`, `simulate_batch`, `runpf`, `solve_lane`) to kernels compiled AT RUN TIME FOR THIS GRID
        (gpf_jit_enable): every size and table offset of the grid is a literal in them instead of a value read from the
        launch parameter block.  The first launch of each kernel variant compiles it (hipcc, about a second; cached on disk per
        grid in ``cache_dir`` / $GRIDPF_JIT_CACHE / ``grid2op_amd/_jit_cache``); results are bit-identical to the shipped kernels.
        ``verify`` (default): before this engine is switched, a 64-lane twin engine of the same grid runs two 8-step launches
        (synthetic chronics around the grid's own injections, load jitter, generation rebalancing) and an AC + a DC `runpf` with the
        shipped and with the specialised kernels, and every result must agree bit for bit -- a specialised kernel is a new binary, and a binary
        is only trusted after it reproduced the validated one; on a mismatch the engine keeps the shipped kernels and
        `GridPFError` is raised.  Raises `GridPFError` too when no hipcc is available (shipped kernels stay)."""
        if not enable:
            check(self._lib.gpf_jit_disable(self._h), "gpf_jit_disable")
            return self.specialization()
        if verify:
            self._verify_specialization(cache_dir)
        # (the kernel sources beside this package; $GRIDPF_JIT_SRC: those of another build -- same-box A/B runs of a library selected with $GRIDPF_LIB)
        src = (os.environ.get("GRIDPF_JIT_SRC") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")).encode()
        check(self._lib.gpf_jit_enable(self._h, src, cache_dir.encode() if cache_dir else None), "gpf_jit_enable")
        return self.specialization()

    def _verify_specialization(self, cache_dir=None, n_lanes: int = 64, n_steps: int = 8, T: int = 24):
        twin = PowerFlowEngine(self.model, n_lanes=n_lanes, device=self.device, n_busbar=self.n_busbar)
        try:
            sl, inj = twin.inj_slices, twin.init_inj
            m = self.model
            w = 1.0 + 0.04 * np.sin(0.7 * np.arange(T))[:, None]
            load_p = inj[sl["load_p"]][None, :] * w
            load_q = inj[sl["load_q"]][None, :] * w
            prod_p = inj[sl["gen_p"]][None, :] * w
            prod_v = np.tile((inj[sl["gen_vm"]] * m.sub_vn_kv[m.gen_sub])[None, :], (T, 1))
            twin.upload_chronics(twin.pack_chronics(load_p, load_q, prod_p, prod_v))
            rng = np.random.default_rng(0)
            twin.set_lane_chronics(lane_offset=(5 * np.arange(n_lanes) % T).astype(np.int32),
                                   lane_scale=(1.0 + 0.03 * rng.standard_normal((n_lanes, 2 * m.n_load))).astype(np.float32))

            def run():
                twin.reset()
                twin.set_trajectory(n_steps, twin.TRAJ_OBS)
                got = []
                for k in range(2):
                    twin.step(k * n_steps, n_steps=n_steps, rebalance=1.02, auto_reset=True)
                    r = twin.results()
                    got += [r.out, r.topo_vect, r.status, r.bus_vm, r.bus_va] + [x.out for x in twin.trajectory_obs(n_steps)]
                twin.runpf()                                        # the one-power-flow-per-lane kernels (runpf / solve_lane), AC and DC
                r = twin.results()
                got += [r.out, r.status, r.bus_vm, r.bus_va]
                twin.runpf(is_dc=True)
                got += [twin.results().out]
                return got

            ref = run()
            src = (os.environ.get("GRIDPF_JIT_SRC") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")).encode()
            check(twin._lib.gpf_jit_enable(twin._h, src, cache_dir.encode() if cache_dir else None), "gpf_jit_enable")
            got = run()
            info = twin.specialization()
            if info["launches"] == 0:
                raise GridPFError(f"specialised kernels could not be built for this grid ({info['variants']}): the shipped kernels stay")
            bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not np.array_equal(a, b, equal_nan=True)]
            if bad:
                raise GridPFError(f"specialised kernels {info['variants']} do NOT reproduce the shipped kernels on the {n_lanes}-lane self-test "
                                  f"({len(bad)} of {len(ref)} result arrays differ): the shipped kernels stay")
        finally:
            twin.close()

    def specialization(self) -> dict:
        """State of the run-time specialised kernels (gpf_jit_info): enabled, variants compiled / taken from the cache / failed,
        launches that went through them, seconds spent compiling + loading, the variants' template arguments."""
        counts = (C.c_int64 * 6)()
        sec = C.c_double()
        text = C.create_string_buffer(2048)
        check(self._lib.gpf_jit_info(self._h, counts, C.byref(sec), text, len(text)), "gpf_jit_info")
        return {"enabled": bool(counts[0]), "compiled": int(counts[1]), "cached": int(counts[2]), "failed": int(counts[3]),
                "launches": int(counts[4]), "aot": int(counts[5]), "seconds": float(sec.value), "variants": text.value.decode()}

    def specialization_header(self) -> str:
        """The generated header the specialised kernels are compiled with (gpf_jit_source): this grid's numbers as C literals."""
        need = C.c_size_t()
        check(self._lib.gpf_jit_source(self._h, None, 0, C.byref(need)), "gpf_jit_source")
        buf = C.create_string_buffer(need.value + 1)
        check(self._lib.gpf_jit_source(self._h, buf, len(buf), None), "gpf_jit_source")
        return buf.value.decode()

    def counters(self) -> dict:
        """Step launches issued since the engine was created and the kernel dispatches they took (a batch of a few residency rounds goes
        out as one dispatch per round)."""
        out = (C.c_int64 * 2)()
        check(self._lib.gpf_get_counters(self._h, out), "gpf_get_counters")
        return {"step_launches": int(out[0]), "kernel_dispatches": int(out[1])}

    def plan(self) -> dict:
        """Diagnostics: the kernel configuration a launch over all lanes would use right now (gpf_get_plan)."""
        out = (C.c_int32 * 8)()
        check(self._lib.gpf_get_plan(self._h, out), "gpf_get_plan")
        keys = ("busbars_per_block", "instances_per_wavefront", "wavefronts_per_instance", "staging_tier", "ybus_in_registers",
                "dc_factors_kept", "lds_bytes", "topology_classes")
        return dict(zip(keys, [int(v) for v in out]))

    def algorithmic_bytes_per_step(self) -> int:
        """inputs at API dtype + outputs at API dtype, topology unchanged."""
        m = self.model
        bytes_in = 4 * (2 * m.n_load + 2 * m.n_gen + m.n_storage)
        bytes_out = 4 * (8 * m.n_line + 3 * m.n_gen + 3 * m.n_load + 3 * m.n_storage
                         + (2 * m.n_line + m.n_load + m.n_gen + m.n_storage) + 4 * m.n_shunt) + 4 * m.dim_topo + m.n_line
        return bytes_in + bytes_out

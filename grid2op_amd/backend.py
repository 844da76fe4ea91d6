"""`HipBackend`: drop-in ``grid2op.Backend.Backend`` whose power flow runs on an MI355X.

This is the host-side mirror of ``grid2op/Backend/pandaPowerBackend.py`` (class ``PandaPowerBackend``): the
same public surface (``load_grid``, ``apply_action``, ``runpf``, ``get_topo_vect``, ``generators_info``,
``loads_info``, ``lines_or_info``, ``lines_ex_info``, ``storages_info``, ``shunt_info``, ``get_theta``,
``get_line_status``, ``get_line_flow``, ``_disconnect_line``, ``reset``, ``copy``, ``close``), the same argument
meaning, units, dtypes and error behaviour, so that the *unmodified* reference ``Environment.step()``,
``Runner`` and ``Observation.simulate()`` consume it.  The pandas tables of the reference are replaced by a few
numpy vectors (one lane of a `grid2op_amd.engine.PowerFlowEngine`); all arithmetic is done by the HIP kernels
behind ``libgridpf.so``.  There is no CPU fallback: without the HIP library / a GPU, ``load_grid`` raises.

Usage (exactly as any other grid2op backend)::

    import grid2op
    from grid2op_amd.backend import HipBackend
    env = grid2op.make("l2rpn_case14_sandbox", backend=HipBackend())
"""
from __future__ import annotations

import copy
import os
import time
import warnings
import weakref
from typing import Optional, Tuple, Union

import numpy as np

from grid2op.Backend.backend import Backend
from grid2op.dtypes import dt_bool, dt_float, dt_int
from grid2op.Exceptions import BackendError

from .grid_model import GridModel, load_grid_model

__all__ = ["HipBackend"]


_MODEL_CACHE = {}      # (real path, mtime) -> GridModel: the parsed grid file is immutable, re-loads reuse it (and its engine)
MAX_BUSBAR_PER_SUB = 64  # include/gridpf.h GPF_MAX_BUSBAR (split substations run on the bus-level graph of their topology class)


class _LanePool:
    """Backends of the same grid / device / busbar count share ONE engine and take one lane each.

    ``Backend.copy`` is hammered by the reference (``ObservationSpace._create_backend_obs`` observationSpace.py:250-254,
    ``N1Reward`` n1Reward.py:71, ``Simulator``): with the pool a copy costs no device allocation -- the façade pushes
    its complete lane state before every power flow, so a copy only needs a free lane.  Engines are per process: a
    forked child starts with an empty pool and never touches the parent's handles (`PowerFlowEngine.close` is a no-op in a
    process that did not create the engine).  ROCm cannot be used in a child forked AFTER the parent initialised HIP: run
    ``Runner(..., nb_process > 1)`` / multi-process environments with the ``spawn`` start method."""
    LANES = 32
    _pools = {}
    _pid = None

    @classmethod
    def acquire(cls, key, factory):
        if cls._pid != os.getpid():
            cls._pools = {}
            cls._pid = os.getpid()
        for ent in cls._pools.setdefault(key, []):
            if ent["free"]:
                return ent["engine"], ent["free"].pop()
        eng = factory(cls.LANES)
        n = int(getattr(eng, "n_lanes", 1))
        ent = {"engine": eng, "free": list(range(n - 1, 0, -1))}
        cls._pools[key].append(ent)
        return eng, 0

    @classmethod
    def release(cls, key, engine, lane):
        if cls._pid != os.getpid():
            return
        ents = cls._pools.get(key, [])
        for ent in ents:
            if ent["engine"] is engine and lane not in ent["free"]:
                ent["free"].append(lane)
                if len(ent["free"]) == int(getattr(engine, "n_lanes", 1)):   # last user gone: free the device memory
                    ents.remove(ent)
                    engine.close()
                    if not ents:
                        cls._pools.pop(key, None)
                return


class HipBackend(Backend):
    """See module docstring.  Keyword arguments mirror ``PandaPowerBackend.__init__``
    (pandaPowerBackend.py:119-143) so that ``Runner`` can re-instantiate the class from ``_my_kwargs``
    (Runner/runner.py:739-756); ``lightsim2grid`` and ``with_numba`` select pandapower solver variants with the same results and are
    accepted and ignored.  ``dist_slack=True`` is REFUSED (`BackendError`): pandapower then spreads the slack power over the generators by
    ``gen.slack_weight`` (pandaPowerBackend.py:1097-1105, ``distributed_slack=self._dist_slack``), the engine solves the single-slack power
    flow only -- accepting the flag would return results that silently differ from PandaPowerBackend's."""

    shunts_data_available = True

    def __init__(self,
                 detailed_infos_for_cascading_failures: bool = False,
                 lightsim2grid: bool = False,
                 dist_slack: bool = False,
                 max_iter: int = 10,
                 can_be_copied: bool = True,
                 with_numba: bool = False,
                 device: int = 0,
                 tol_mva: float = 1e-8,
                 specialize: bool = False):
        """``specialize`` (beyond PandaPowerBackend's arguments): the engine's kernels are compiled at run time for the loaded grid
        (`PowerFlowEngine.specialize`: bit-identical results, ~12 % less time per ``runpf``; needs hipcc on the host -- without it the
        shipped kernels stay and a warning says so)."""
        if dist_slack:
            raise BackendError("HipBackend: dist_slack=True (pandapower's distributed slack, pandaPowerBackend.py:1097-1105) is not implemented -- "
                               "the engine solves the single-slack power flow; refusing rather than returning results that differ silently")
        Backend.__init__(self,
                         detailed_infos_for_cascading_failures=detailed_infos_for_cascading_failures,
                         can_be_copied=can_be_copied,
                         lightsim2grid=lightsim2grid,
                         dist_slack=dist_slack,
                         max_iter=max_iter,
                         with_numba=with_numba,
                         device=device,
                         tol_mva=tol_mva,
                         specialize=specialize)
        self._specialize = bool(specialize)
        self._needs_active_bus = False      # the engine derives the active buses from the topology itself
        self._max_iter = int(max_iter)
        self._tol_mva = float(tol_mva)
        self._device = int(device)
        self.can_output_theta = True
        self._m: Optional[GridModel] = None
        self._engine = None
        self._lane = 0
        self._pool_key = None
        self._lane_finalizer = None
        self.div_exception = None
        self.tol = 1e-5                     # storage "produces / absorbs anything" threshold (:864)
        self._topo_vect = None
        self.line_status = None
        self.cst_1 = dt_float(1.0)

    # ------------------------------------------------------------------------------------------------------------
    def _make_engine(self, model: GridModel, n_busbar: int, n_lanes: int = 1):
        """Hook: the one place where the compute engine is created (tests swap in the CPU oracle here)."""
        from .engine import PowerFlowEngine
        eng = PowerFlowEngine(model, n_lanes=n_lanes, device=self._device, n_busbar=n_busbar)
        if getattr(self, "_specialize", False):
            try:
                eng.specialize(True)
            except Exception as exc:           # no compiler at run time / self-test refused: the shipped kernels are the product path
                warnings.warn(f"HipBackend(specialize=True): {exc}")
        return eng

    def _acquire_lane(self):
        self._release_lane()
        m, nbb = self._m, self.n_busbar_per_sub
        # backends of the same grid FILE share an engine: copies, and the fresh instances ``Runner`` loads per episode
        self._pool_key = (type(self)._make_engine, getattr(m, "_source_key", None) or id(m), self._device, nbb, getattr(self, "_specialize", False))
        self._engine, self._lane = _LanePool.acquire(self._pool_key, lambda n: self._make_engine(m, nbb, n))
        # a backend that is simply dropped (the reference never closes the copies ``next_grid_state`` appends to its infos,
        # backend.py:1490-1492, nor do users close ``env.copy()``) must give its lane back like a closed one
        self._lane_finalizer = weakref.finalize(self, _LanePool.release, self._pool_key, self._engine, self._lane)

    def _release_lane(self):
        fin = getattr(self, "_lane_finalizer", None)
        if fin is not None:
            fin()                      # runs _LanePool.release at most once
        self._lane_finalizer = None
        self._engine = None

    # ---- load_grid (pandaPowerBackend.py:356-617 + 670-874) ------------------------------------------------------------
    def load_grid(self, path: Union[os.PathLike, str], filename: Optional[Union[os.PathLike, str]] = None) -> None:
        self.can_handle_more_than_2_busbar()
        self.can_handle_detachment()
        full_path = self.make_complete_path(path, filename)
        src = (os.path.realpath(str(full_path)), os.path.getmtime(str(full_path)))
        m = _MODEL_CACHE.get(src)
        if m is None:
            if str(full_path).endswith(".npz"):      # a grid description saved by ``save_file`` / tests/golden/*.grid.npz
                m = GridModel.load_npz(str(full_path))
            else:
                m = load_grid_model(full_path)
            m._source_key = src
            if len(_MODEL_CACHE) >= 16:
                _MODEL_CACHE.pop(next(iter(_MODEL_CACHE)))
            _MODEL_CACHE[src] = m
        self._m = m
        if self.n_busbar_per_sub > MAX_BUSBAR_PER_SUB:
            # PandaPowerBackend takes any n_busbar_per_sub (pandaPowerBackend.py:562-577 duplicates the buses n times); the engine
            # takes up to include/gridpf.h GPF_MAX_BUSBAR (a split substation is solved on the bus-level graph of its topology
            # class, whatever the busbar count): refuse beyond that at load time, with the reason
            raise BackendError(f"HipBackend supports at most {MAX_BUSBAR_PER_SUB} busbars per substation "
                               f"(n_busbar_per_sub={self.n_busbar_per_sub} requested)")
        self._init_from_model(m)
        self._acquire_lane()

    def _init_from_model(self, m: GridModel) -> None:
        self.n_line = m.n_line
        self.n_gen = m.n_gen
        self.n_load = m.n_load
        self.n_sub = m.n_sub
        self.name_line = np.array([str(x) for x in m.name_line])
        self.name_gen = np.array([str(x) for x in m.name_gen])
        self.name_load = np.array([str(x) for x in m.name_load])
        self.name_sub = np.array([str(x) for x in m.name_sub])
        self.n_storage = m.n_storage
        if m.n_storage == 0:
            self.set_no_storage()
        else:
            self.name_storage = np.array([str(x) for x in m.name_storage])
        self.n_shunt = m.n_shunt if type(self).shunts_data_available else None
        self._number_true_line = m.n_powerline
        self.sub_info = m.sub_info.astype(dt_int)
        self.load_to_subid = m.load_sub.astype(dt_int)
        self.gen_to_subid = m.gen_sub.astype(dt_int)
        self.line_or_to_subid = m.line_or_sub.astype(dt_int)
        self.line_ex_to_subid = m.line_ex_sub.astype(dt_int)
        self.load_to_sub_pos = m.load_to_sub_pos.astype(dt_int)
        self.gen_to_sub_pos = m.gen_to_sub_pos.astype(dt_int)
        self.line_or_to_sub_pos = m.line_or_to_sub_pos.astype(dt_int)
        self.line_ex_to_sub_pos = m.line_ex_to_sub_pos.astype(dt_int)
        if m.n_storage > 0:
            self.storage_to_subid = m.storage_sub.astype(dt_int)
            self.storage_to_sub_pos = m.storage_to_sub_pos.astype(dt_int)
        self.dim_topo = int(m.dim_topo)
        if type(self).shunts_data_available:
            self.shunt_to_subid = m.shunt_sub.astype(dt_int)
            self.name_shunt = np.array([str(x) for x in m.name_shunt]).astype(str)
            self._sh_vnkv = m.shunt_vn_kv.astype(np.float64)
        self._compute_pos_big_topo()

        self.load_pu_to_kv = m.sub_vn_kv[m.load_sub].astype(dt_float)
        self.prod_pu_to_kv = m.sub_vn_kv[m.gen_sub].astype(dt_float)
        self.lines_or_pu_to_kv = m.sub_vn_kv[m.line_or_sub].astype(dt_float)
        self.lines_ex_pu_to_kv = m.sub_vn_kv[m.line_ex_sub].astype(dt_float)
        self.storage_pu_to_kv = m.sub_vn_kv[m.storage_sub].astype(dt_float)
        self.thermal_limit_a = m.thermal_limit_a.astype(dt_float)

        # --- dynamic state (what the pandas tables hold in the reference) ------------------------------------------
        self._bus_gen = np.ones(m.n_gen, dtype=dt_int)
        self._bus_load = np.ones(m.n_load, dtype=dt_int)
        self._bus_sto = np.ones(m.n_storage, dtype=dt_int)
        self._bus_lor = np.ones(m.n_line, dtype=dt_int)
        self._bus_lex = np.ones(m.n_line, dtype=dt_int)
        self._bus_shunt = np.ones(m.n_shunt, dtype=dt_int)
        self._act_gen = m.gen_status0.copy()
        self._act_load = m.load_status0.copy()
        self._act_sto = m.storage_status0.copy()
        self._act_line = m.line_status0.copy()
        self._act_shunt = m.shunt_status0.copy()
        self._gen_p = m.gen_p0.astype(np.float64).copy()
        self._gen_vm = m.gen_vm0.astype(np.float64).copy()
        self._load_p = m.load_p0.astype(np.float64).copy()
        self._load_q = m.load_q0.astype(np.float64).copy()
        self._sto_p = m.storage_p0.astype(np.float64).copy()
        self._sto_q = m.storage_q0.astype(np.float64).copy()
        self._sh_p = m.shunt_p0.astype(np.float64).copy()
        self._sh_q = m.shunt_q0.astype(np.float64).copy()
        self._pristine = self._snapshot_state()

        # topo_vect position -> (kind, element id); kinds: 0 load, 1 gen, 2 line or, 3 line ex, 4 storage
        self._pos_kind = np.full(self.dim_topo, -1, dtype=np.int8)
        self._pos_id = np.zeros(self.dim_topo, dtype=dt_int)
        for kind, pos in ((0, m.load_pos_topo_vect), (1, m.gen_pos_topo_vect), (2, m.line_or_pos_topo_vect),
                          (3, m.line_ex_pos_topo_vect), (4, m.storage_pos_topo_vect)):
            self._pos_kind[pos] = kind
            self._pos_id[pos] = np.arange(len(pos))

        nanv = lambda n: np.full(n, np.nan, dtype=dt_float)
        for nm in ("p_or", "q_or", "v_or", "a_or", "p_ex", "q_ex", "v_ex", "a_ex", "theta_or", "theta_ex"):
            setattr(self, nm, nanv(m.n_line))
        for nm in ("load_p", "load_q", "load_v", "load_theta"):
            setattr(self, nm, nanv(m.n_load))
        for nm in ("prod_p", "prod_q", "prod_v", "gen_theta"):
            setattr(self, nm, nanv(m.n_gen))
        for nm in ("storage_p", "storage_q", "storage_v", "storage_theta"):
            setattr(self, nm, nanv(m.n_storage))
        self._shunt_res = (nanv(m.n_shunt), nanv(m.n_shunt), nanv(m.n_shunt), np.full(m.n_shunt, -1, dtype=dt_int))
        self._topo_vect = np.full(self.dim_topo, -1, dtype=dt_int)
        self.line_status = np.zeros(m.n_line, dtype=dt_bool)
        self._refresh_status_and_topo()
        self.comp_time = 0.0

    _STATE_FIELDS = ("_bus_gen", "_bus_load", "_bus_sto", "_bus_lor", "_bus_lex", "_bus_shunt", "_act_gen", "_act_load",
                     "_act_sto", "_act_line", "_act_shunt", "_gen_p", "_gen_vm", "_load_p", "_load_q", "_sto_p", "_sto_q",
                     "_sh_p", "_sh_q")
    _RESULT_FIELDS = ("p_or", "q_or", "v_or", "a_or", "p_ex", "q_ex", "v_ex", "a_ex", "theta_or", "theta_ex", "load_p",
                      "load_q", "load_v", "load_theta", "prod_p", "prod_q", "prod_v", "gen_theta", "storage_p", "storage_q",
                      "storage_v", "storage_theta")

    def _snapshot_state(self):
        return {k: getattr(self, k).copy() for k in self._STATE_FIELDS}

    def _restore_state(self, snap):
        for k, v in snap.items():
            setattr(self, k, v.copy())

    # ---- apply_action (pandaPowerBackend.py:902-975) ------------------------------------------------------------------
    def apply_action(self, backend_action) -> None:
        if backend_action is None:
            return
        cls = type(self)
        (_active_bus, (prod_p, prod_v, load_p, load_q, storage), topo__, shunts__) = backend_action()

        # generators (:925-931).  prod_v arrives in kV; the solver set-point is prod_v / vn_kv computed in float32
        ch = prod_p.changed
        self._gen_p[ch] = prod_p.values[ch]
        ch = prod_v.changed
        self._gen_vm[ch] = prod_v.values[ch] / self.prod_pu_to_kv[ch]
        # loads (:934-935)
        ch = load_p.changed
        self._load_p[ch] = load_p.values[ch]
        ch = load_q.changed
        self._load_q[ch] = load_q.values[ch]
        # storage (:937-951)
        if cls.n_storage > 0:
            ch = storage.changed
            self._sto_p[ch] = storage.values[ch]
            stor_bus = backend_action.get_storages_bus()
            chb = stor_bus.changed
            if chb.any():
                newb = stor_bus.values
                deact = chb & (newb <= -1)
                react = chb & (newb >= 1)
                self._act_sto[deact] = False
                self._bus_sto[deact] = 1
                self._act_sto[react] = True
                self._bus_sto[react] = newb[react]
        # shunts (:953-969)
        if cls.shunts_data_available and shunts__ is not None:
            shunt_p, shunt_q, shunt_bus = shunts__
            ch = shunt_p.changed
            self._sh_p[ch] = shunt_p.values[ch]
            ch = shunt_q.changed
            self._sh_q[ch] = shunt_q.values[ch]
            chb = shunt_bus.changed
            if chb.any():
                in_serv = shunt_bus.values != -1
                self._act_shunt[chb] = in_serv[chb]
                mv = chb & in_serv
                self._bus_shunt[mv] = shunt_bus.values[mv]
        # topology (:971-975, 977-1067): new_bus >= 1 -> in service on that bus, else out of service (bus kept)
        chg = np.nonzero(topo__.changed)[0]
        if chg.size:
            vals = topo__.values
            for pos in chg:
                kind = self._pos_kind[pos]
                idx = self._pos_id[pos]
                nb = int(vals[pos])
                if kind == 0:
                    self._set_bus(self._bus_load, self._act_load, idx, nb)
                elif kind == 1:
                    self._set_bus(self._bus_gen, self._act_gen, idx, nb)
                elif kind == 2:
                    self._set_bus(self._bus_lor, self._act_line, idx, nb)
                elif kind == 3:
                    self._set_bus(self._bus_lex, self._act_line, idx, nb)
                # kind 4 (storage) is handled above, as in the reference (:973-975)

    @staticmethod
    def _set_bus(bus_arr, act_arr, idx, new_bus):
        if new_bus >= 1:
            act_arr[idx] = True
            bus_arr[idx] = new_bus
        else:
            act_arr[idx] = False

    # ---- status / topology vectors (pandaPowerBackend.py:1450-1459, 1489-1524) ------------------------------------------
    def _refresh_status_and_topo(self):
        cls = type(self)
        m = self._m
        self.line_status.flags.writeable = True
        self.line_status[:] = self._act_line
        self.line_status.flags.writeable = False
        tv = self._topo_vect
        tv.flags.writeable = True
        tv[m.line_or_pos_topo_vect] = np.where(self._act_line, self._bus_lor, -1)
        tv[m.line_ex_pos_topo_vect] = np.where(self._act_line, self._bus_lex, -1)
        tv[m.load_pos_topo_vect] = np.where(self._act_load, self._bus_load, -1)
        tv[m.gen_pos_topo_vect] = np.where(self._act_gen, self._bus_gen, -1)
        if m.n_storage:
            tv[m.storage_pos_topo_vect] = np.where(self._act_sto, self._bus_sto, -1)
        tv.flags.writeable = False

    def get_line_status(self) -> np.ndarray:
        return self.line_status

    def get_line_flow(self) -> np.ndarray:
        return self.a_or

    def get_topo_vect(self) -> np.ndarray:
        return self._topo_vect.copy()

    def _disconnect_line(self, id_):
        self._act_line[id_] = False
        self._topo_vect.flags.writeable = True
        self._topo_vect[self.line_or_pos_topo_vect[id_]] = -1
        self._topo_vect[self.line_ex_pos_topo_vect[id_]] = -1
        self._topo_vect.flags.writeable = False
        self.line_status.flags.writeable = True
        self.line_status[id_] = False
        self.line_status.flags.writeable = False

    def _reconnect_line(self, id_):
        self._act_line[id_] = True
        self.line_status.flags.writeable = True
        self.line_status[id_] = True
        self.line_status.flags.writeable = False

    # ---- runpf (pandaPowerBackend.py:1220-1255) ---------------------------------------------------------------------------
    def runpf(self, is_dc: bool = False) -> Tuple[bool, Union[Exception, None]]:
        if self._engine is None:
            raise BackendError("HipBackend: load_grid must be called before runpf")
        m = self._m
        eng = self._engine
        beg = time.perf_counter()
        self._refresh_status_and_topo()          # before solving, as in the reference (:1236-1237)
        inj = np.concatenate((self._gen_p, self._gen_vm, self._load_p, self._load_q, self._sto_p, self._sto_q,
                              self._sh_p, self._sh_q))
        shunt_bus = np.where(self._act_shunt, self._bus_shunt, -1).astype(np.int32)
        r = eng.solve_lane(self._lane, inj, self._topo_vect, shunt_bus if m.n_shunt else None, is_dc=bool(is_dc),
                           max_iter=self._max_iter, tol_mva=self._tol_mva)      # push state + solve + read back: one call, one sync
        self.comp_time += time.perf_counter() - beg
        st = int(r.status[0, 0])
        if st != 0:
            from .engine import STATUS_TEXT
            msg = STATUS_TEXT.get(st, f"status {st}")
            self.div_exception = msg
            self._reset_all_nan()
            return False, BackendError(f'powerflow diverged with error :"{msg}", you can check '
                                       f'`env.backend.div_exception` for more information')
        self._fetch_results(r, bool(is_dc))
        self.div_exception = None
        return True, None

    def _fetch_results(self, r, is_dc: bool) -> None:
        """``_fetch_data_pf_converged`` (pandaPowerBackend.py:1122-1218)."""
        m = self._m
        n_sub = m.n_sub
        f32 = lambda a: np.asarray(a[0], dtype=dt_float)
        self.p_or[:], self.q_or[:], self.v_or[:], self.a_or[:] = f32(r.p_or), f32(r.q_or), f32(r.v_or), f32(r.a_or)
        self.p_ex[:], self.q_ex[:], self.v_ex[:], self.a_ex[:] = f32(r.p_ex), f32(r.q_ex), f32(r.v_ex), f32(r.a_ex)
        self.theta_or[:], self.theta_ex[:] = f32(r.theta_or), f32(r.theta_ex)
        # an out-of-service line keeps its (stale) bus in the reference's tables: its theta reads the angle of that
        # bus when the bus is still energised (pandaPowerBackend.py:1163-1187)
        bus_va = r.bus_va[0]
        off = ~self._act_line
        if off.any():
            g_or = m.line_or_sub[off] + (self._bus_lor[off] - 1) * n_sub
            g_ex = m.line_ex_sub[off] + (self._bus_lex[off] - 1) * n_sub
            self.theta_or[off] = np.nan_to_num(bus_va[g_or], nan=0.0).astype(dt_float)
            self.theta_ex[off] = np.nan_to_num(bus_va[g_ex], nan=0.0).astype(dt_float)
        self.prod_p[:], self.prod_q[:], self.prod_v[:], self.gen_theta[:] = f32(r.gen_p), f32(r.gen_q), f32(r.gen_v), f32(r.gen_theta)
        self.load_p[:], self.load_q[:], self.load_v[:], self.load_theta[:] = f32(r.load_p), f32(r.load_q), f32(r.load_v), f32(r.load_theta)
        if m.n_storage:
            # p, q echo the SET-POINT table; theta is (quirk) filled with the voltage magnitude in kV (:1621-1647)
            self.storage_p[:] = self._sto_p.astype(dt_float)
            self.storage_q[:] = self._sto_q.astype(dt_float)
            self.storage_v[:] = f32(r.storage_v)
            g_st = m.storage_sub + (self._bus_sto - 1) * n_sub
            vm_st = r.bus_vm[0][g_st]
            if is_dc:
                vm_st = np.full(m.n_storage, np.nan)          # res_bus.vm_pu is NaN in DC (:1137-1139)
            self.storage_theta[:] = (vm_st * self.storage_pu_to_kv).astype(dt_float)
            # a storage unit whose voltage is not finite is switched off FOR GOOD (:1197-1201: p = q = v = 0 and
            # storage["in_service"] = False) -- in DC mode that is every unit
            dead = ~np.isfinite(vm_st) & self._act_sto
            self.storage_p[dead] = 0.0
            self.storage_q[dead] = 0.0
            self.storage_v[dead] = 0.0
            self._act_sto[dead] = False
        self._shunt_res = (f32(r.shunt_p).copy(), f32(r.shunt_q).copy(), f32(r.shunt_v).copy(),
                           np.asarray(r.shunt_bus[0], dtype=dt_int).copy())
        if is_dc:
            # pandapower leaves res_bus.vm_pu / res_line.vm_*_pu NaN in DC mode.  What PandaPowerBackend makes of it:
            # * load_v = nominal kV, or the prod_v of the first generator of the same substation that sits on the same
            #   busbar; 0 for an out-of-service load (:1137-1156)
            # * v_or = v_ex = 0 (NaN -> 0, :1161-1181)
            # * every q is 0 (:1212-1218)
            self.load_v[:] = self.load_pu_to_kv
            tv = self._topo_vect
            for l_id in range(m.n_load):
                for g_id in np.nonzero(m.gen_sub == m.load_sub[l_id])[0]:
                    if tv[m.load_pos_topo_vect[l_id]] == tv[m.gen_pos_topo_vect[g_id]]:
                        self.load_v[l_id] = self.prod_v[g_id]
                        break
            self.load_v[~self._act_load] = 0.0
            self.v_or[:] = 0.0
            self.v_ex[:] = 0.0
            self.prod_q[:] = 0.0
            self.load_q[:] = 0.0
            self.storage_q[:] = 0.0
            self.q_or[:] = 0.0
            self.q_ex[:] = 0.0

    def _reset_all_nan(self) -> None:
        """pandaPowerBackend.py:1257-1287."""
        for nm in self._RESULT_FIELDS:
            getattr(self, nm)[:] = np.nan
        self._topo_vect.flags.writeable = True
        self._topo_vect[:] = -1
        self._topo_vect.flags.writeable = False
        self.line_status.flags.writeable = True
        self.line_status[:] = False
        self.line_status.flags.writeable = False

    # ---- getters (pandaPowerBackend.py:1566-1619, 278-301): fresh copies ---------------------------------------------------------
    def generators_info(self):
        return self.prod_p.copy(), self.prod_q.copy(), self.prod_v.copy()

    def loads_info(self):
        return self.load_p.copy(), self.load_q.copy(), self.load_v.copy()

    def lines_or_info(self):
        return self.p_or.copy(), self.q_or.copy(), self.v_or.copy(), self.a_or.copy()

    def lines_ex_info(self):
        return self.p_ex.copy(), self.q_ex.copy(), self.v_ex.copy(), self.a_ex.copy()

    def storages_info(self):
        return self.storage_p.copy(), self.storage_q.copy(), self.storage_v.copy()

    def shunt_info(self):
        p, q, v, b = self._shunt_res
        return p.copy(), q.copy(), v.copy(), b.copy()

    def get_theta(self):
        return (self.cst_1 * self.theta_or, self.cst_1 * self.theta_ex, self.cst_1 * self.load_theta,
                self.cst_1 * self.gen_theta, self.cst_1 * self.storage_theta)

    def sub_from_bus_id(self, bus_id: int) -> int:
        return int(bus_id) % type(self).n_sub

    # ---- reset / copy / close (pandaPowerBackend.py:334-354, 1289-1409, 1411-1423) --------------------------------------------------
    def reset(self, path=None, grid_filename=None) -> None:
        self._restore_state(self._pristine)
        self._reset_all_nan()
        self._refresh_status_and_topo()
        self.comp_time = 0.0

    def copy(self) -> "HipBackend":
        res = type(self)(**self._my_kwargs)
        res._m = self._m
        for k in ("n_line", "n_gen", "n_load", "n_sub", "n_storage", "n_shunt", "dim_topo", "_number_true_line"):
            if hasattr(self, k):
                setattr(res, k, getattr(self, k))
        # class-level grid description is shared through the (re-typed) class; instance-level arrays are copied
        skip = {"_engine", "_m", "_my_kwargs", "_pool_key", "_lane", "_lane_finalizer"}
        for k, v in self.__dict__.items():
            if k in skip:
                continue
            if isinstance(v, np.ndarray):
                setattr(res, k, v.copy())
            elif isinstance(v, (dict, list, tuple)) and k in ("_pristine", "_shunt_res"):
                setattr(res, k, copy.deepcopy(v))
        res._topo_vect.flags.writeable = False
        res.line_status.flags.writeable = False
        res.thermal_limit_a = copy.deepcopy(self.thermal_limit_a)
        res._sh_vnkv = copy.deepcopy(self._sh_vnkv)
        res.comp_time = self.comp_time
        res.can_output_theta = self.can_output_theta
        res._is_loaded = self._is_loaded
        res.div_exception = self.div_exception
        res._missing_two_busbars_support_info = self._missing_two_busbars_support_info
        res._missing_detachment_support_info = self._missing_detachment_support_info
        res.n_busbar_per_sub = self.n_busbar_per_sub
        res.detachment_is_allowed = self.detachment_is_allowed
        if self._engine is not None:
            res._acquire_lane()        # same engine, another lane: no device allocation
        return res

    def close(self) -> None:
        self._release_lane()

    def save_file(self, full_path) -> None:
        """The reference dumps its pandapower net (``pp.to_json``, :1425-1437); here the grid description and
        the current lane state are saved as an ``.npz`` pair for debugging."""
        self._m.save_npz(str(full_path) + ".grid.npz")
        np.savez(str(full_path) + ".state.npz", **self._snapshot_state())

"""ctypes binding of ``libgridpf.so`` (C ABI: ``include/gridpf.h``).

There is NO fallback: if the shared library is missing or no HIP device is available the engine
cannot be created and every call raises `GridPFError`.  (The CPU oracle under ``oracle/`` is test
infrastructure only and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

__all__ = ["lib", "GridPFError", "GpfGridDesc", "GpfLayout", "GpfStepOpts", "library_path", "EXPORTED_SYMBOLS"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libgridpf.so"
ABI_VERSION = 323          # include/gridpf.h GPF_ABI_VERSION

EXPORTED_SYMBOLS = [
    "gpf_last_error", "gpf_version", "gpf_set_deterministic", "gpf_device_count", "gpf_create", "gpf_destroy", "gpf_get_layout", "gpf_n_lanes",
    "gpf_set_injections", "gpf_set_topology", "gpf_get_injections", "gpf_get_topology", "gpf_disconnect_line",
    "gpf_reset_lanes", "gpf_copy_lanes", "gpf_fanout_n1", "gpf_runpf", "gpf_solve_lane", "gpf_get_results", "gpf_upload_chronics",
    "gpf_upload_maintenance", "gpf_upload_hazards", "gpf_set_lane_chronics", "gpf_set_thermal_limits", "gpf_step", "gpf_step_n", "gpf_set_lane_redispatch", "gpf_set_gen_limits", "gpf_redispatch", "gpf_set_trajectory",
    "gpf_get_trajectory", "gpf_get_trajectory_obs", "gpf_upload_forecasts", "gpf_simulate_batch", "gpf_set_overflow_count",
    "gpf_set_storage_params", "gpf_set_env_dynamics", "gpf_set_lane_actions", "gpf_lane_actions_on_device", "gpf_get_env_state", "gpf_get_env_illegal", "gpf_set_env_illegal", "gpf_set_env_state", "gpf_set_gen_renewable", "gpf_set_lane_curtailment", "gpf_get_episode", "gpf_lane_capacity", "gpf_get_step_outputs", "gpf_sync",
    "gpf_get_results_pinned", "gpf_set_profiling", "gpf_get_kernel_time", "gpf_get_plan", "gpf_device_pointers", "gpf_device_pointers_n",
    "gpf_get_counters", "gpf_upload_outage_durations", "gpf_get_cooldown", "gpf_set_cooldown", "gpf_get_trajectory_cooldown", "gpf_ptdf_build", "gpf_ptdf_build_batch", "gpf_ptdf_batch_info", "gpf_ptdf_batch_get", "gpf_ptdf_get", "gpf_ptdf_flows", "gpf_get_ptdf_flows", "gpf_ptdf_flows_rows", "gpf_get_ptdf_flows_rows", "gpf_lodf_screen",
    "gpf_jit_enable", "gpf_jit_disable", "gpf_jit_info", "gpf_jit_source",
]


class GridPFError(RuntimeError):
    pass


def library_path() -> str:
    return os.environ.get("GRIDPF_LIB", os.path.join(_HERE, _LIB_NAME))


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)
_bp = C.POINTER(C.c_uint8)


class GpfGridDesc(C.Structure):
    _fields_ = [
        ("n_sub", C.c_int32), ("n_busbar", C.c_int32),
        ("n_line", C.c_int32), ("n_gen", C.c_int32), ("n_load", C.c_int32), ("n_storage", C.c_int32),
        ("n_shunt", C.c_int32), ("dim_topo", C.c_int32),
        ("sn_mva", C.c_double),
        ("sub_vn_kv", _dp),
        ("line_or_sub", _ip), ("line_ex_sub", _ip), ("line_or_pos_topo_vect", _ip), ("line_ex_pos_topo_vect", _ip),
        ("br_y", _dp), ("br_bdc", _dp),
        ("gen_sub", _ip), ("gen_pos_topo_vect", _ip), ("gen_min_q", _dp), ("gen_max_q", _dp), ("gen_slack", _bp),
        ("load_sub", _ip), ("load_pos_topo_vect", _ip),
        ("storage_sub", _ip), ("storage_pos_topo_vect", _ip),
        ("shunt_sub", _ip), ("shunt_fact", _dp),
        ("init_inj", _dp), ("init_topo", _ip), ("init_shunt_bus", _ip),
    ]


class GpfLayout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_inj", "inj_gen_p", "inj_gen_vm", "inj_load_p", "inj_load_q", "inj_storage_p", "inj_storage_q",
        "inj_shunt_p", "inj_shunt_q",
        "n_out", "out_p_or", "out_q_or", "out_v_or", "out_a_or", "out_theta_or",
        "out_p_ex", "out_q_ex", "out_v_ex", "out_a_ex", "out_theta_ex",
        "out_gen_p", "out_gen_q", "out_gen_v", "out_gen_theta",
        "out_load_p", "out_load_q", "out_load_v", "out_load_theta",
        "out_storage_p", "out_storage_q", "out_storage_v", "out_storage_theta",
        "out_shunt_p", "out_shunt_q", "out_shunt_v",
        "n_chron", "chron_load_p", "chron_load_q", "chron_prod_p", "chron_prod_v", "nb_total")]


class GpfStepOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("tol_mva", C.c_double), ("rebalance", C.c_double), ("cascade", C.c_int32),
                ("hard_overflow", C.c_float), ("soft_overflow", C.c_float), ("nb_ts_allowed", C.c_int32), ("max_rounds", C.c_int32),
                ("is_dc", C.c_int32), ("auto_reset", C.c_int32), ("warm_start", C.c_int32), ("track_cooldown", C.c_int32), ("nb_ts_reco", C.c_int32)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load ``libgridpf.so`` (built in-tree by ``__graft_entry__.build()``); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise GridPFError(
            f"{path} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(there is no CPU fallback).")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes cannot share a process: whichever
    # is loaded first serves both (same SONAME), and torch finds no GPU when it is the system one (measured on the MI355X box:
    # "No HIP GPUs are available" as soon as libgridpf.so was loaded before ``import torch``).  So when torch is installed it is
    # imported first and libgridpf.so binds to the runtime torch ships -- zero-copy views (PowerFlowEngine.device_views),
    # torch.distributed / RCCL and the engine then live on the same runtime.  GRIDPF_NO_TORCH_PRELOAD=1 skips this.
    if os.environ.get("GRIDPF_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    try:
        L = C.CDLL(path)
    except OSError as exc:
        raise GridPFError(f"cannot load {path}: {exc}") from exc
    h = C.c_void_p
    i32 = C.c_int32
    L.gpf_last_error.restype = C.c_char_p
    L.gpf_last_error.argtypes = []
    L.gpf_version.restype = C.c_int
    L.gpf_device_count.argtypes = [_ip]
    L.gpf_set_deterministic.argtypes = [h, i32]
    L.gpf_create.argtypes = [C.POINTER(GpfGridDesc), i32, i32, C.POINTER(h)]
    L.gpf_destroy.argtypes = [h]
    L.gpf_get_layout.argtypes = [h, C.POINTER(GpfLayout)]
    L.gpf_n_lanes.argtypes = [h]
    L.gpf_set_injections.argtypes = [h, i32, i32, _dp]
    L.gpf_set_topology.argtypes = [h, i32, i32, _ip, _ip]
    L.gpf_get_injections.argtypes = [h, i32, i32, _dp]
    L.gpf_get_topology.argtypes = [h, i32, i32, _ip, _ip]
    L.gpf_disconnect_line.argtypes = [h, i32, i32]
    L.gpf_reset_lanes.argtypes = [h, i32, i32]
    L.gpf_copy_lanes.argtypes = [h, i32, i32, i32]
    L.gpf_fanout_n1.argtypes = [h, i32, i32, i32, _ip]
    L.gpf_runpf.argtypes = [h, i32, i32, i32, i32, C.c_double]
    L.gpf_get_results.argtypes = [h, i32, i32, _fp, _ip, _ip, _bp, _ip, _dp, _dp]
    L.gpf_get_results_pinned.argtypes = [h, i32, i32, i32, C.POINTER(C.c_void_p)]
    L.gpf_solve_lane.argtypes = [h, i32, _dp, _ip, _ip, i32, i32, C.c_double, _fp, _ip, _ip, _bp, _ip, _dp, _dp]
    L.gpf_upload_chronics.argtypes = [h, i32, i32, _fp]
    L.gpf_upload_maintenance.argtypes = [h, i32, i32, _bp]
    L.gpf_upload_hazards.argtypes = [h, i32, i32, _bp]
    L.gpf_set_lane_chronics.argtypes = [h, _ip, _ip, _fp]
    L.gpf_set_thermal_limits.argtypes = [h, _fp]
    L.gpf_step.argtypes = [h, i32, i32, C.c_double, C.c_double, i32, C.c_float, C.c_float, i32, i32, i32]
    L.gpf_step_n.argtypes = [h, i32, i32, C.POINTER(GpfStepOpts)]
    L.gpf_set_lane_redispatch.argtypes = [h, _fp]
    L.gpf_set_gen_limits.argtypes = [h, _dp, _dp, _dp, _dp, _bp, C.c_double]
    L.gpf_redispatch.argtypes = [h, i32, i32, _dp, _dp, _dp, _dp, _bp, _dp, i32, _bp, _fp]
    L.gpf_set_trajectory.argtypes = [h, i32, i32]
    L.gpf_get_trajectory_obs.argtypes = [h, i32, i32, i32, i32, _fp, _ip, _ip, _bp]
    L.gpf_get_trajectory.argtypes = [h, i32, i32, i32, i32, _fp, C.POINTER(C.c_int8)]
    L.gpf_upload_forecasts.argtypes = [h, i32, i32, i32, _fp]
    L.gpf_simulate_batch.argtypes = [h, i32, i32, i32, _ip, i32, _ip, _ip, _ip, i32, C.POINTER(GpfStepOpts)]
    L.gpf_set_overflow_count.argtypes = [h, i32, i32, _ip]
    L.gpf_set_storage_params.argtypes = [h, _dp, _dp, _dp, _dp, _dp, _fp, C.c_double, i32]
    L.gpf_set_env_dynamics.argtypes = [h, i32, C.c_double]
    L.gpf_set_lane_actions.argtypes = [h, _fp, _fp, i32]
    L.gpf_lane_actions_on_device.argtypes = [h, i32, i32, i32, i32]
    L.gpf_get_env_state.argtypes = [h, i32, i32, _fp, _fp, _fp, _bp, _fp, _fp, _fp, _fp]
    L.gpf_get_env_illegal.argtypes = [h, i32, i32, _ip]
    L.gpf_set_env_illegal.argtypes = [h, i32, i32, _ip]
    L.gpf_set_env_state.argtypes = [h, i32, i32, _fp, _fp, _fp, _bp, _fp, _fp, _fp, _fp]
    L.gpf_set_gen_renewable.argtypes = [h, _bp]
    L.gpf_set_lane_curtailment.argtypes = [h, _fp]
    L.gpf_get_episode.argtypes = [h, i32, i32, _bp, _ip]
    L.gpf_lane_capacity.argtypes = [h]
    L.gpf_get_step_outputs.argtypes = [h, i32, i32, _fp, _ip, _ip]
    L.gpf_sync.argtypes = [h]
    L.gpf_set_profiling.argtypes = [h, i32]
    L.gpf_get_kernel_time.argtypes = [h, _dp, C.POINTER(C.c_int64)]
    L.gpf_get_plan.argtypes = [h, C.POINTER(C.c_int32)]
    L.gpf_device_pointers.argtypes = [h, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.gpf_device_pointers_n.argtypes = [h, C.POINTER(C.c_void_p), i32, C.POINTER(C.c_void_p)]
    L.gpf_ptdf_build.argtypes = [h, i32]
    L.gpf_ptdf_get.argtypes = [h, _dp]
    L.gpf_get_counters.argtypes = [h, C.POINTER(C.c_int64)]
    L.gpf_get_cooldown.argtypes = [h, i32, i32, _ip]
    L.gpf_upload_outage_durations.argtypes = [h, i32, i32, C.POINTER(C.c_uint16)]
    L.gpf_set_cooldown.argtypes = [h, i32, i32, _ip]
    L.gpf_get_trajectory_cooldown.argtypes = [h, i32, i32, i32, i32, C.POINTER(C.c_int16)]
    L.gpf_ptdf_build_batch.argtypes = [h, i32, i32, i32, _ip]
    L.gpf_ptdf_batch_info.argtypes = [h, _ip, _ip, _ip, _dp]
    L.gpf_ptdf_batch_get.argtypes = [h, i32, _dp, _dp]
    L.gpf_ptdf_flows.argtypes = [h, i32, i32]
    L.gpf_get_ptdf_flows.argtypes = [h, i32, i32, _fp]
    L.gpf_ptdf_flows_rows.argtypes = [h, i32, i32, C.c_double]
    L.gpf_get_ptdf_flows_rows.argtypes = [h, i32, i32, i32, i32, _fp]
    L.gpf_lodf_screen.argtypes = [h, i32, i32, _fp, _fp]
    L.gpf_jit_enable.argtypes = [h, C.c_char_p, C.c_char_p]
    L.gpf_jit_disable.argtypes = [h]
    L.gpf_jit_info.argtypes = [h, C.POINTER(C.c_int64), _dp, C.c_char_p, C.c_size_t]
    L.gpf_jit_source.argtypes = [h, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(L, name)
        if name not in ("gpf_last_error",):
            fn.restype = C.c_int
    got = L.gpf_version()
    if got != ABI_VERSION:               # a stale GRIDPF_LIB / an old build next to new Python: struct layouts and argument lists differ
        raise GridPFError(f"{path} has ABI version {got}, this binding needs {ABI_VERSION} (include/gridpf.h GPF_ABI_VERSION): rebuild "
                          f"with `python -c 'import __graft_entry__ as g; g.build(force=True)'`")
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gpf_last_error()
        raise GridPFError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(a: Optional[np.ndarray], ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))

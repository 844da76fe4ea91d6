"""ctypes loader of the C oracle (oracle/pf_oracle.c).  TEST INFRASTRUCTURE -- see the header of the C file.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpforacle.so")
_lib = None


def available() -> bool:
    try:
        _load()
        return True
    except Exception:
        return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    _lib = C.CDLL(_SO)
    return _lib


def _desc(m, n_busbar=2):
    """Build the gpf_grid_desc (same struct as the product ABI; only the type definition is shared)."""
    from grid2op_amd._capi import GpfGridDesc, ptr
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    br_y = np.empty((m.n_line, 8), dtype=np.float64)
    for k, y in enumerate((m.br_yff, m.br_yft, m.br_ytf, m.br_ytt)):
        br_y[:, 2 * k] = y.real
        br_y[:, 2 * k + 1] = y.imag
    keep = dict(
        sub_vn_kv=f64(m.sub_vn_kv), line_or_sub=i32(m.line_or_sub), line_ex_sub=i32(m.line_ex_sub),
        line_or_pos=i32(m.line_or_pos_topo_vect), line_ex_pos=i32(m.line_ex_pos_topo_vect), br_y=br_y,
        br_bdc=f64(m.br_bdc), gen_sub=i32(m.gen_sub), gen_pos=i32(m.gen_pos_topo_vect), gen_min_q=f64(m.gen_min_q),
        gen_max_q=f64(m.gen_max_q), gen_slack=np.ascontiguousarray(m.gen_slack, dtype=np.uint8), load_sub=i32(m.load_sub),
        load_pos=i32(m.load_pos_topo_vect), sto_sub=i32(m.storage_sub), sto_pos=i32(m.storage_pos_topo_vect),
        shunt_sub=i32(m.shunt_sub), shunt_fact=f64(m.shunt_fact),
        init_inj=np.concatenate([f64(m.gen_p0), f64(m.gen_vm0), f64(m.load_p0), f64(m.load_q0), f64(m.storage_p0),
                                 f64(m.storage_q0), f64(m.shunt_p0), f64(m.shunt_q0)]),
        init_topo=i32(m.initial_topo_vect()), init_shunt_bus=i32(m.initial_shunt_bus()))
    d = GpfGridDesc()
    d.n_sub, d.n_busbar = m.n_sub, n_busbar
    d.n_line, d.n_gen, d.n_load, d.n_storage, d.n_shunt, d.dim_topo = m.n_line, m.n_gen, m.n_load, m.n_storage, m.n_shunt, m.dim_topo
    d.sn_mva = m.sn_mva
    for field, key, ct in [("sub_vn_kv", "sub_vn_kv", C.c_double), ("line_or_sub", "line_or_sub", C.c_int32),
                           ("line_ex_sub", "line_ex_sub", C.c_int32), ("line_or_pos_topo_vect", "line_or_pos", C.c_int32),
                           ("line_ex_pos_topo_vect", "line_ex_pos", C.c_int32), ("br_y", "br_y", C.c_double),
                           ("br_bdc", "br_bdc", C.c_double), ("gen_sub", "gen_sub", C.c_int32),
                           ("gen_pos_topo_vect", "gen_pos", C.c_int32), ("gen_min_q", "gen_min_q", C.c_double),
                           ("gen_max_q", "gen_max_q", C.c_double), ("gen_slack", "gen_slack", C.c_uint8),
                           ("load_sub", "load_sub", C.c_int32), ("load_pos_topo_vect", "load_pos", C.c_int32),
                           ("storage_sub", "sto_sub", C.c_int32), ("storage_pos_topo_vect", "sto_pos", C.c_int32),
                           ("shunt_sub", "shunt_sub", C.c_int32), ("shunt_fact", "shunt_fact", C.c_double),
                           ("init_inj", "init_inj", C.c_double), ("init_topo", "init_topo", C.c_int32),
                           ("init_shunt_bus", "init_shunt_bus", C.c_int32)]:
        setattr(d, field, ptr(keep[key], ct))
    return d, keep


def set_solver(sparse: bool):
    """Linear solves of THIS thread's power flows: dense Gaussian elimination (default; the version pinned to the golden vectors) or the
    sparse LU on a minimum-degree ordering with a cached symbolic analysis (what a CPU power-flow solver does -- pandapower: scipy spsolve,
    pandaPowerBackend.py:1081-1083; lightsim2grid: KLU).  Same Newton iteration; tests/test_oracle_c.py pins the two to each other."""
    _load().pfo_set_solver(1 if sparse else 0)


class COracle:
    """One grid; `solve_rows` runs independent power flows with the result-row layout of include/gridpf.h."""

    def __init__(self, m, n_busbar=2):
        self.m = m
        self.lib = _load()
        self.desc, self._keep = _desc(m, n_busbar)
        self.n_out = self.lib.pfo_n_out(C.byref(self.desc))
        self.n_inj = self.lib.pfo_n_inj(C.byref(self.desc))
        self.nb_total = m.n_sub * n_busbar

    def solve_rows(self, inj, topo, shunt_bus, is_dc=False, max_iter=10, tol_mva=1e-8):
        from grid2op_amd._capi import ptr
        m = self.m
        inj = np.ascontiguousarray(inj, dtype=np.float64).reshape(-1, self.n_inj)
        topo = np.ascontiguousarray(topo, dtype=np.int32).reshape(-1, m.dim_topo)
        n = inj.shape[0]
        sb = np.ascontiguousarray(shunt_bus, dtype=np.int32).reshape(n, m.n_shunt) if m.n_shunt else np.zeros((n, 0), np.int32)
        out = np.empty((n, self.n_out))
        tv = np.empty((n, m.dim_topo), np.int32)
        sbo = np.empty((n, max(m.n_shunt, 1)), np.int32)
        ls = np.empty((n, m.n_line), np.uint8)
        st = np.empty((n, 4), np.int32)
        bvm = np.empty((n, self.nb_total))
        bva = np.empty((n, self.nb_total))
        for k in range(n):
            self.lib.pfo_solve(C.byref(self.desc), ptr(inj[k], C.c_double), ptr(topo[k], C.c_int32), ptr(sb[k], C.c_int32),
                               int(is_dc), int(max_iter), C.c_double(tol_mva), ptr(out[k], C.c_double), ptr(tv[k], C.c_int32),
                               ptr(sbo[k], C.c_int32), ptr(ls[k], C.c_uint8), ptr(st[k], C.c_int32), ptr(bvm[k], C.c_double),
                               ptr(bva[k], C.c_double))
        return dict(out=out, topo_vect=tv, shunt_bus=sbo[:, :m.n_shunt], line_status=ls.astype(bool), status=st, bus_vm=bvm,
                    bus_va=bva)

    def step_batch(self, chron, lane_offset, lane_scale, rebalance, t, lane0, n, want_out=False, max_iter=10, tol_mva=1e-8):
        from grid2op_amd._capi import ptr
        chron = np.ascontiguousarray(chron, dtype=np.float32)
        lo = np.ascontiguousarray(lane_offset, dtype=np.int32)
        ls = None if lane_scale is None else np.ascontiguousarray(lane_scale, dtype=np.float32)
        out = np.empty((n, self.n_out)) if want_out else None
        st = np.empty((n, 4), np.int32)
        self.lib.pfo_step_batch.argtypes = None
        nconv = self.lib.pfo_step_batch(C.byref(self.desc), ptr(chron, C.c_float), int(chron.shape[0]), ptr(lo, C.c_int32),
                                        ptr(ls, C.c_float), C.c_double(rebalance), int(t), int(lane0), int(n), int(max_iter),
                                        C.c_double(tol_mva), ptr(out, C.c_double), ptr(st, C.c_int32))
        return nconv, out, st


def time_steps(m, ch, T, budget_s=12.0, chunk=512, sparse=False):
    """bench.py cpu_baseline leg: run the synthetic DoNothing workload on ONE host thread until ~budget_s of
    solver time has been spent.  Only the C call is timed (generating the synthetic jitter is not)."""
    orc = COracle(m)
    set_solver(sparse)
    tab = np.ascontiguousarray(np.concatenate([ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]], axis=-1), np.float32)
    n_total, spent, k0 = 0, 0.0, 0
    while spent < budget_s:
        lanes = np.arange(k0, k0 + chunk)
        off = (7 * lanes) % T
        sc = np.empty((chunk, 2 * m.n_load), np.float32)
        for i, k in enumerate(lanes):
            sc[i] = 1.0 + 0.05 * np.random.default_rng(int(k)).standard_normal(2 * m.n_load)
        t1 = time.perf_counter()
        orc.step_batch(tab, off, sc, 1.02, 0, 0, chunk)
        spent += time.perf_counter() - t1
        n_total += chunk
        k0 += chunk
    set_solver(False)
    return n_total, spent


def _all_cores_worker(args):
    """One process of `time_steps_all_cores`: worker `w` of `n` solves lanes w, w+n, w+2n, ... of the same global batch."""
    grid_npz, chron_npz, w, n_workers, budget_s, start_at = args[:6]
    sparse = bool(args[6]) if len(args) > 6 else False
    import sys
    root = os.path.dirname(_HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from grid2op_amd.grid_model import GridModel
    m = GridModel.load_npz(grid_npz)
    ch = dict(np.load(chron_npz))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    T = ch["load_p"].shape[0]
    orc = COracle(m)
    set_solver(sparse)
    tab = np.ascontiguousarray(np.concatenate([ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]], axis=-1), np.float32)
    chunk = 256                              # this worker's lanes: w, w + n, w + 2n, ... stepped through t = 0, 1, 2, ...
    lanes = w + n_workers * np.arange(chunk)
    off = (7 * lanes) % T
    sc = np.empty((chunk, 2 * m.n_load), np.float32)
    for i, k in enumerate(lanes):
        sc[i] = 1.0 + 0.05 * np.random.default_rng(int(k)).standard_normal(2 * m.n_load)
    while time.time() < start_at:            # common start (interpreter start-up / input generation are not part of the sample)
        time.sleep(0.005)
    t_begin = time.time()
    n_total, t = 0, 0
    while time.time() - t_begin < budget_s:
        orc.step_batch(tab, off, sc, 1.02, t, 0, chunk)
        n_total += chunk
        t += 1
    return n_total, t_begin, time.time()


def time_steps_all_cores(grid_npz, chron_npz, n_workers, budget_s=10.0, startup_s=None, sparse=False):
    """bench.py cpu_baseline leg, all host cores: `n_workers` single-thread processes (fresh interpreters -- the parent has
    initialised HIP, which must not be forked) split the lanes of the same synthetic workload between them.  Returns
    (lane-steps, wall seconds from the common start to the last worker's end)."""
    import json
    import sys
    if startup_s is None:
        startup_s = 4.0 + 0.03 * n_workers
    start_at = time.time() + startup_s
    root = os.path.dirname(_HERE)
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.pf_oracle_c", grid_npz, chron_npz, str(w), str(n_workers), str(budget_s),
                               repr(start_at), "1" if sparse else "0"], cwd=root, stdout=subprocess.PIPE, text=True,
                              env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1"))
             for w in range(n_workers)]
    rs = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("cpu baseline worker failed")
        rs.append(json.loads(out.strip().splitlines()[-1]))
    total = sum(r[0] for r in rs)
    wall = max(r[2] for r in rs) - min(r[1] for r in rs)
    return total, wall


if __name__ == "__main__":
    import json
    import sys
    a = sys.argv[1:]
    print(json.dumps(_all_cores_worker((a[0], a[1], int(a[2]), int(a[3]), float(a[4]), float(a[5]), int(a[6]) if len(a) > 6 else 0))))

"""CPU check of the product's symbolic analysis (grid2op_amd/csrc/gridpf_symbolic.hpp): a host emulation of the device
block-LU driven by the SAME level-scheduled program must reproduce a dense solve on random block-sparse systems with the
sparsity of real grids (tests/native/sym_emul.cpp is test infrastructure compiled with g++)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

_SRC = os.path.join(ROOT, "tests", "native", "sym_emul.cpp")
_SO = os.path.join(ROOT, "tests", "native", "_build", "libsymemul.so")


@pytest.fixture(scope="module")
def emul():
    deps = [_SRC, os.path.join(ROOT, "grid2op_amd", "csrc", "gridpf_symbolic.hpp")]
    if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", _SRC, "-o", _SO])
    return C.CDLL(_SO)


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1",
                                  "l2rpn_wcci_2022_dev"])
@pytest.mark.parametrize("BS,fused", [(2, 0), (2, 1), (4, 0), (3, 1)])
def test_program_reproduces_dense_solve(emul, load_model, name, BS, fused):
    m = load_model(name)
    n = m.n_sub
    N = n * BS
    rng = np.random.default_rng(n * 10 + BS)
    A = np.zeros((N, N))
    def blk():
        return rng.standard_normal((BS, BS))
    for a, b in zip(m.line_or_sub, m.line_ex_sub):
        A[a * BS:(a + 1) * BS, b * BS:(b + 1) * BS] += blk()
        A[b * BS:(b + 1) * BS, a * BS:(a + 1) * BS] += blk()
    for s in range(n):       # block diagonally dominant, as power-flow Jacobians are in practice
        A[s * BS:(s + 1) * BS, s * BS:(s + 1) * BS] += blk() + (4.0 + np.abs(A[s * BS:(s + 1) * BS]).sum() / BS) * np.eye(BS)
    rhs = rng.standard_normal(N)
    x = np.zeros(N)
    stats = np.zeros(4, dtype=np.int32)
    lor = np.ascontiguousarray(m.line_or_sub, dtype=np.int32)
    lex = np.ascontiguousarray(m.line_ex_sub, dtype=np.int32)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    rc = emul.sym_emul_solve(n, m.n_line, lor.ctypes.data_as(ip), lex.ctypes.data_as(ip), BS, A.ctypes.data_as(dp),
                             rhs.ctypes.data_as(dp), x.ctypes.data_as(dp), stats.ctypes.data_as(ip), fused)
    assert rc >= 1
    ref = np.linalg.solve(A, rhs)
    assert np.abs(x - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    nslot_y, nslot, n_levels, _ = stats
    assert n_levels < n or n <= 2                      # level scheduling really groups pivots
    assert nslot >= nslot_y >= n


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1",
                                  "l2rpn_wcci_2022_dev"])
@pytest.mark.parametrize("gw", [16, 32, 64, 128])
def test_flat_program_reproduces_dense_solve(emul, load_model, name, gw):
    """The flat program of the single-busbar kernels (build_flat: passes of exactly gw items, byte-offset fields, right-hand
    side as pseudo-slots whose pad column holds garbage) solves the same random block-sparse systems."""
    m = load_model(name)
    n, BS = m.n_sub, 2
    N = n * BS
    rng = np.random.default_rng(n * 100 + gw)
    A = np.zeros((N, N))
    for a, b in zip(m.line_or_sub, m.line_ex_sub):
        A[a * BS:(a + 1) * BS, b * BS:(b + 1) * BS] += rng.standard_normal((BS, BS))
        A[b * BS:(b + 1) * BS, a * BS:(a + 1) * BS] += rng.standard_normal((BS, BS))
    for s in range(n):
        A[s * BS:(s + 1) * BS, s * BS:(s + 1) * BS] += rng.standard_normal((BS, BS)) + (4.0 + np.abs(A[s * BS:(s + 1) * BS]).sum() / BS) * np.eye(BS)
    rhs = rng.standard_normal(N)
    x = np.zeros(N)
    stats = np.zeros(4, dtype=np.int32)
    lor = np.ascontiguousarray(m.line_or_sub, dtype=np.int32)
    lex = np.ascontiguousarray(m.line_ex_sub, dtype=np.int32)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    rc = emul.sym_emul_solve_flat(n, m.n_line, lor.ctypes.data_as(ip), lex.ctypes.data_as(ip), gw, A.ctypes.data_as(dp),
                                  rhs.ctypes.data_as(dp), x.ctypes.data_as(dp), stats.ctypes.data_as(ip))
    assert rc >= 1, rc
    ref = np.linalg.solve(A, rhs)
    assert np.abs(x - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    n_fwd, n_tail_levels, n_back, n_words = (int(v) for v in stats)
    lv = np.zeros(4 * 64 + 2, dtype=np.int32)
    n_levels = emul.sym_level_sizes(n, m.n_line, lor.ctypes.data_as(ip), lex.ctypes.data_as(ip), lv.ctypes.data_as(ip), 64, 1)
    items = [int(lv[4 * k + 2] + lv[4 * k + 3]) for k in range(n_levels)]
    # Gauss-Jordan tail (Symbolic::gj_lv0): the last levels also eliminate their pivots' columns from the tail rows above them
    # (more forward items there) and have no back substitution: the levels before the tail take one back pass each when they fit
    assert 2 <= n_tail_levels <= n_levels and (name != "l2rpn_wcci_2022_dev" or n_tail_levels <= n_levels - 4)
    lu_passes = sum(-(-i // gw) for i in items[:n_levels - n_tail_levels] if i > 0)
    tail_lu = sum(-(-i // gw) for i in items[n_levels - n_tail_levels:] if i > 0)
    assert lu_passes + tail_lu <= n_fwd <= lu_passes + tail_lu + (128 // gw + 1) * n_tail_levels + (n_levels if gw > 64 else 0)
    assert n_levels - n_tail_levels <= n_back <= 2 * (n_levels - n_tail_levels) + (64 // gw) * 4
    if gw == 128:
        assert n_back == n_levels - n_tail_levels
    assert n_words % 4 == 0


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev"])
def test_undirected_pair_table(emul, load_model, name):
    """build_upairs: every off-diagonal block of the original pattern belongs to exactly one pair (u < v) with both its slots."""
    m = load_model(name)
    lor = np.ascontiguousarray(m.line_or_sub, dtype=np.int32)
    lex = np.ascontiguousarray(m.line_ex_sub, dtype=np.int32)
    ip = C.POINTER(C.c_int32)
    n_up = emul.sym_upairs_check(m.n_sub, m.n_line, lor.ctypes.data_as(ip), lex.ctypes.data_as(ip))
    distinct = {(min(a, b), max(a, b)) for a, b in zip(m.line_or_sub.tolist(), m.line_ex_sub.tolist()) if a != b}
    assert n_up == len(distinct)


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev"])
def test_slot_layout_search_keeps_the_program_valid(emul, load_model, name):
    """optimize_slot_layout renumbers the blocks inside their ranges (original off-diagonal / LU fill / Gauss-Jordan fill) to lower
    the LDS bank-model cost of the passes: every slot-carrying table is relabelled consistently (the flat programs of all group
    widths still reproduce dense solves, branch -> slot tables still name the right blocks) and the model cost does not go up."""
    m = load_model(name)
    n, BS = m.n_sub, 2
    N = n * BS
    lor = np.ascontiguousarray(m.line_or_sub, dtype=np.int32)
    lex = np.ascontiguousarray(m.line_ex_sub, dtype=np.int32)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    emul.sym_emul_solve_flat_opt.restype = C.c_int
    for gw in (16, 32, 64, 128):
        rng = np.random.default_rng(n * 7 + gw)
        A = np.zeros((N, N))
        for a, b in zip(m.line_or_sub, m.line_ex_sub):
            A[a * BS:(a + 1) * BS, b * BS:(b + 1) * BS] += rng.standard_normal((BS, BS))
            A[b * BS:(b + 1) * BS, a * BS:(a + 1) * BS] += rng.standard_normal((BS, BS))
        for s in range(n):
            A[s * BS:(s + 1) * BS, s * BS:(s + 1) * BS] += rng.standard_normal((BS, BS)) + (4.0 + np.abs(A[s * BS:(s + 1) * BS]).sum() / BS) * np.eye(BS)
        rhs = rng.standard_normal(N)
        x = np.zeros(N)
        stats = np.zeros(4, dtype=np.int32)
        cost = np.zeros(2, dtype=np.int64)
        rc = emul.sym_emul_solve_flat_opt(n, m.n_line, lor.ctypes.data_as(ip), lex.ctypes.data_as(ip), gw, 600, A.ctypes.data_as(dp),
                                          rhs.ctypes.data_as(dp), x.ctypes.data_as(dp), stats.ctypes.data_as(ip),
                                          cost.ctypes.data_as(C.POINTER(C.c_longlong)))
        assert rc >= 1, (gw, rc)
        ref = np.linalg.solve(A, rhs)
        assert np.abs(x - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), gw
        assert cost[1] <= cost[0], cost
    assert cost[1] < cost[0] or n <= 5


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev"])
def test_level_schedules_are_valid_and_rescheduling_reaches_the_dag_height(emul, load_model, name):
    """build_symbolic's level schedules (greedy minimum degree; re-scheduled on the elimination DAG by list scheduling): pivots of a
    level pairwise non-adjacent in the filled pattern, every updating pivot in an earlier level, each substation eliminated once.
    The re-scheduled variant has exactly as many levels as the elimination DAG is high; it is only TAKEN (resched = -1) when it has
    fewer passes at the grid's usual group width."""
    m = load_model(name)
    lor = np.ascontiguousarray(m.line_or_sub, dtype=np.int32)
    lex = np.ascontiguousarray(m.line_ex_sub, dtype=np.int32)
    ip = C.POINTER(C.c_int32)
    n = m.n_sub
    gw = 16 if n <= 8 else 32 if n <= 24 else 64 if n < 64 else 128
    res = {}
    for mode in (0, 1, -1):
        out = np.zeros(3, dtype=np.int32)
        rc = emul.sym_check_schedule(n, m.n_line, lor.ctypes.data_as(ip), lex.ctypes.data_as(ip), mode, gw, out.ctypes.data_as(ip))
        assert rc == 0, (name, mode, rc)
        res[mode] = tuple(int(v) for v in out)
    assert res[1][0] == res[1][1]                              # list scheduling: levels == height of the DAG
    assert res[1][0] <= res[0][0]
    assert res[-1][2] == min(res[0][2], res[1][2])             # auto: the variant with fewer passes
    if name == "l2rpn_wcci_2022_dev":
        assert res[1][2] < res[0][2] and res[1][0] == 13

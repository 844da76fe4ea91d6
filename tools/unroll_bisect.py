"""Developer tool (round 5, VERDICT weak #8): which loop's unrolling makes the UNROLLED specialised 36-substation kernel differ from the shipped one?
For every spec on the command line ("none", "all", "lo-hi" = loops lo..hi-1 of gridpf_sparse.hpp get `#pragma nounroll`) a copy of the
kernel sources is written to a scratch directory, the specialised kernel is compiled from it WITHOUT -fno-unroll-loops (GRIDPF_JIT_FLAGS set
but empty, scratch allowed) and 3 launches of the neurips-36 cascade workload are compared, bit for bit, with the shipped kernel.

    GRIDPF_JIT_ALLOW_SCRATCH=1 python tools/unroll_bisect.py none all 0-30 30-60 60-90 90-120 [--flags "..."] [--env NAME]
"""
import os
import re
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "grid2op_amd", "csrc")


def loops(src):
    """[(offset of 'for (', in_macro)] of the loops without an unroll pragma of their own"""
    out = []
    for mt in re.finditer(r"\bfor \(", src):
        ls = src.rfind("\n", 0, mt.start()) + 1
        le = src.find("\n", mt.start())
        line = src[ls:le]
        prev = src[src.rfind("\n", 0, ls - 1) + 1:ls - 1] if ls > 0 else ""
        if "#pragma" in prev and src[ls:mt.start()].strip() == "":
            continue
        if line.lstrip().startswith("//") or "//" in src[ls:mt.start()]:
            continue
        out.append((mt.start(), line.rstrip().endswith("\\") or prev.rstrip().endswith("\\")))
    return out


def patched(src, lo, hi):
    ls = loops(src)
    res, last = [], 0
    for idx, (off, in_macro) in enumerate(ls):
        if lo <= idx < hi:
            res.append(src[last:off])
            res.append('_Pragma("nounroll") ' if in_macro else "\n#pragma nounroll\n")
            last = off
    res.append(src[last:])
    return "".join(res), len(ls)


def main():
    argv = sys.argv[1:]
    flags, env_name, n1 = "", "l2rpn_neurips_2020_track1", 0
    while "--flags" in argv:
        i = argv.index("--flags"); flags = argv[i + 1]; del argv[i:i + 2]
    while "--n1" in argv:                 # N-1 fan-out shape of bench.py: that many envs x (1 + n_line) lanes, no cascade (islanding / diverging contingencies)
        i = argv.index("--n1"); n1 = int(argv[i + 1]); del argv[i:i + 2]
    while "--env" in argv:
        i = argv.index("--env"); env_name = argv[i + 1]; del argv[i:i + 2]
    os.environ["GRIDPF_JIT_FLAGS"] = flags
    os.environ.setdefault("GRIDPF_JIT_ALLOW_SCRATCH", "1")
    from grid2op_amd.grid_model import GridModel
    gold = os.path.join(ROOT, "tests", "golden")
    load_model_impl = lambda name: GridModel.load_npz(os.path.join(gold, f"{name}.grid.npz"))   # noqa: E731
    load_npz_impl = lambda fname: dict(np.load(os.path.join(gold, fname)))                       # noqa: E731
    import ctypes as C
    from grid2op_amd._capi import check
    from test_gpu_multistep import _setup
    from test_gpu_jit import _run, _arrays  # noqa
    src0 = open(os.path.join(CSRC, "gridpf_sparse.hpp")).read()
    n_loops = len(loops(src0))
    print(f"{n_loops} loops without a pragma of their own; flags={flags!r}", flush=True)
    kw = dict(rebalance=1.02, cascade=True, auto_reset=True) if not n1 else dict(rebalance=1.02)
    m0 = load_model_impl(env_name)
    B = 65 if not n1 else n1 * (1 + m0.n_line)

    def prepare(eng, m):
        if n1:
            fan = 1 + m.n_line
            topo = np.tile(m.initial_topo_vect().astype(np.int32), (B, 1))
            for c in range(1, fan):
                topo[c::fan, m.line_or_pos_topo_vect[c - 1]] = -1
                topo[c::fan, m.line_ex_pos_topo_vect[c - 1]] = -1
            eng.set_topology(topo)
        elif "thermal_limits" in ch:
            eng.set_thermal_limits(np.asarray(ch["thermal_limits"]) * 0.9)
    m, ch, e_ref, tab, off, scale = _setup(load_model_impl, load_npz_impl, env_name, B)
    prepare(e_ref, m)
    ref = _run(e_ref, kw, 6, 3)
    print(f"workload: {env_name}, {B} lanes, {'N-1 fan-out of ' + str(n1) + ' envs' if n1 else 'cascade on, limits x 0.9'}; converged {float(e_ref.results().converged.mean()):.3f}", flush=True)
    for spec in argv:
        lo, hi = (0, 0) if spec == "none" else (0, 10 ** 6) if spec == "all" else tuple(int(x) for x in spec.split("-"))
        d = tempfile.mkdtemp(prefix="gpf_bisect_")
        os.chmod(d, 0o700)
        srcd, cache = os.path.join(d, "csrc"), os.path.join(d, "cache")
        os.makedirs(srcd); os.makedirs(cache, mode=0o700)
        for f in ("gridpf_common.hpp", "gridpf_redispatch.hpp"):
            shutil.copy(os.path.join(CSRC, f), srcd)
        txt, _ = patched(src0, lo, hi)
        open(os.path.join(srcd, "gridpf_sparse.hpp"), "w").write(txt)
        _, _, e_jit, _, _, _ = _setup(load_model_impl, load_npz_impl, env_name, B)
        prepare(e_jit, m)
        check(e_jit._lib.gpf_jit_enable(e_jit._h, srcd.encode(), cache.encode()), "gpf_jit_enable")
        got = _run(e_jit, kw, 6, 3)
        info = e_jit.specialization()
        bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not np.array_equal(a, b, equal_nan=True)]
        worst = max((float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))) for a, b in zip(ref, got) if a.dtype.kind == "f" and a.shape == b.shape), default=0.0)
        print(f"spec {spec:>8}: {'SAME' if not bad else 'DIFFERENT in arrays ' + str(bad[:6]) + ' max |d| ' + format(worst, '.3g')}  | {info['variants']} failed={info['failed']} launches={info['launches']}",
              flush=True)
        e_jit.close()
        shutil.rmtree(d, ignore_errors=True)
    e_ref.close()


main()

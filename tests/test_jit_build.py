"""CPU checks of the run-time specialisation path (grid2op_amd/csrc/gridpf_jit.hip): the field lists behind the generated header
cover the parameter-block structs, and the kernel source compiles for gfx950 with a header the library generated on the MI355X
(tests/golden/jit_header_*.h = gpf_jit_source of two bench grids; hipcc cross-compiles here) -- the exact command line the library runs."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "grid2op_amd", "csrc")
GOLD = os.path.join(ROOT, "tests", "golden")


def _struct_fields(src, name):
    body = re.search(r"struct %s \{(.*?)\n\};" % name, src, re.S).group(1)
    body = re.sub(r"//.*", "", body)
    ints = []
    for decl in re.findall(r"^\s*int ([^;]+);", body, re.M):
        ints += [f.strip().split("[")[0] for f in decl.split(",")]
    return ints


def _macro_fields(src, macro):
    m = re.search(r"#define %s\(X\)((?:.*\\\n)*.*)\n" % macro, src)
    return re.findall(r"X\((\w+)\)", m.group(1))


def test_field_lists_cover_the_parameter_block_structs():
    common = open(os.path.join(CSRC, "gridpf_common.hpp")).read()
    sparse = open(os.path.join(CSRC, "gridpf_sparse.hpp")).read()
    for src, struct, macro in [(common, "GridDev", "GPF_GRIDDEV_INTS"), (common, "OutOff", "GPF_OUTOFF_INTS"), (sparse, "StatOff", "GPF_STATOFF_INTS"),
                               (sparse, "FlatDev", "GPF_FLATDEV_INTS"), (sparse, "SymDev", "GPF_SYMDEV_INTS")]:
        assert sorted(_struct_fields(src, struct)) == sorted(_macro_fields(src, macro)), struct
    hdr = open(os.path.join(GOLD, "jit_header_l2rpn_case14_sandbox.h")).read()
    for macro, fields in [("GPF_JIT_SET_G", _macro_fields(common, "GPF_GRIDDEV_INTS") + ["sn_mva", "inv_sn_mva"]), ("GPF_JIT_SET_OO", _macro_fields(common, "GPF_OUTOFF_INTS")),
                          ("GPF_JIT_SET_SO", _macro_fields(sparse, "GPF_STATOFF_INTS"))]:
        line = [l for l in hdr.splitlines() if l.startswith(f"#define {macro}(v)")][0]
        assert re.findall(r"\(v\)\.(\w+) =", line) == fields, macro
    assert "(v).n_sub = 14;" in hdr and "(v).n_line = 20;" in hdr


@pytest.mark.parametrize("grid,variant", [("l2rpn_case14_sandbox", "1,2,2,2,1,false,false,false"), ("l2rpn_wcci_2022_dev", "1,0,1,2,2,false,true,false")])
def test_kernel_source_compiles_with_a_generated_header(grid, variant, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this host")
    src = tmp_path / "k.hip"
    src.write_text('#include <hip/hip_runtime.h>\n#include "gridpf_common.hpp"\n#include "gridpf_sparse.hpp"\nnamespace gpf {\n'
                   f"template __global__ void step_sparse_kernel<{variant}>(const DevParamsS* __restrict__, const int* __restrict__, const int* __restrict__, "
                   "int, double, StepArgs);\n}\n")
    out = tmp_path / "k.hsaco"
    # the two builds of the library's default policy (gridpf_jit.hip: gpf_jit_get): unrolled first, -fno-unroll-loops when that one
    # spills or costs resident wavefronts; a kernel that spills to scratch memory is refused, so the fallback build must never spill
    for flags in ([], ["-fno-unroll-loops"]):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", *flags, "-DGPF_JIT", "-include",
               os.path.join(GOLD, f"jit_header_{grid}.h"), f"-I{CSRC}", str(src), "-o", str(out), "-Rpass-analysis=kernel-resource-usage"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and out.stat().st_size > 10000, p.stderr[-2000:]
        scratch = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", p.stderr)
        assert scratch, p.stderr[-1500:]
        if flags:
            assert int(scratch.group(1)) == 0, p.stderr[-1500:]

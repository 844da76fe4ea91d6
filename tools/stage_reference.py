#!/usr/bin/env python
"""Stage the UNMODIFIED reference package next to the repo for ONE gpurun call (build container only).

    python tools/stage_reference.py            # /root/reference/grid2op -> ./_stage/grid2op  (+ the pandapower import stub)
    python tools/stage_reference.py --clean    # remove ./_stage

The GPU box has no /root/reference and no network, and nothing of the reference may enter the tree: ``_stage/`` is
git-ignored scratch (it travels with the gpurun snapshot because it is NOT in .gpurunignore) and is removed again by
``tools/gpurun_staged.sh`` as soon as the call returns.  What is staged: the reference package as it is, minus
``tests/`` (kept: ``__init__.py``, ``helper_path_test.py`` and ``aaa_test_backend_interface.py`` -- the reference's own 41-test
backend API kit) and ``data_test/``; of ``data/`` the environments the staged tests use.  The pandapower stand-in of
tests/_refshim (import-time stub, no arithmetic: SURVEY.md 8(c)) is already part of the repo.

tests/test_gpu_reference_framework.py runs only when ``_stage/grid2op`` exists (skipped with that reason otherwise)."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("GRID2OP_REFERENCE", "/root/reference")
STAGE = os.path.join(ROOT, "_stage")
KEEP_TESTS = ("__init__.py", "helper_path_test.py", "aaa_test_backend_interface.py")
KEEP_ENVS = ("l2rpn_case14_sandbox", "educ_case14_storage", "educ_case14_redisp", "rte_case5_example", "l2rpn_neurips_2020_track1",
             "l2rpn_wcci_2022_dev", "l2rpn_idf_2023", "rte_case14_realistic")


def main():
    if os.path.isdir(STAGE):
        shutil.rmtree(STAGE)
    if "--clean" in sys.argv:
        return
    src = os.path.join(REFERENCE, "grid2op")
    if not os.path.isdir(src):
        sys.exit(f"stage_reference: {src} not found")
    dst = os.path.join(STAGE, "grid2op")

    def ignore(d, names):
        rel = os.path.relpath(d, src)
        if rel == ".":
            return [n for n in names if n in ("data_test", "__pycache__")]
        if rel == "tests":
            return [n for n in names if n not in KEEP_TESTS]
        if rel == "data":
            return [n for n in names if n not in KEEP_ENVS]
        return [n for n in names if n == "__pycache__"]
    shutil.copytree(src, dst, ignore=ignore)
    n = sum(len(f) for _, _, f in os.walk(dst))
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(dst) for f in fs)
    print(f"staged {n} files, {size / 1e6:.1f} MB -> {dst}")


if __name__ == "__main__":
    main()

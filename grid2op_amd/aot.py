"""Ahead-of-time specialisation of the step / power-flow kernels for a NEW grid, prepared WITHOUT a GPU.

    python -m grid2op_amd.aot path/to/grid.json  [--lanes 4096 --lanes 1] [--n-busbar 2] [--env-dynamics] [--out grid2op_amd/aot]
    python -c "import __graft_entry__ as g; g.build()"        # compiles every (header, variant) of the manifest into grid2op_amd/_aot

The header the grid-specialised kernels are compiled with is pure grid arithmetic (sizes, result-row offsets, the symbolic LU program's
pass counts, static-table offsets: `gpf_jit_source`), and which kernel variant a batch takes is host-side launch planning
(`gpf_get_plan`) -- neither needs a device, so both come from a HEADER-ONLY handle (`gpf_create(..., GPF_DEVICE_NONE)`).  The manifest
(`grid2op_amd/aot/manifest.json`: header file + kernel variants per grid) is what `__graft_entry__.build()` compiles with hipcc -- which
cross-compiles gfx950 code objects on any machine -- and what `libgridpf.so` looks up at run time by a hash of (header, variant, flags, kernel
sources): with the objects in place `PowerFlowEngine.specialize()` / `GRIDPF_JIT=1` need no compiler on the GPU box.  A grid that is not in
the manifest runs on the shipped kernels (slower by the margin DESIGN.md section 3 states: ~12 % on 14 substations) or is compiled at run
time where hipcc exists.  `tests/test_jit_build.py` regenerates the committed headers this way and compares them byte for byte.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEVICE_NONE = -1                     # include/gridpf.h GPF_DEVICE_NONE


def _model(path):
    from .grid_model import GridModel, load_grid_model
    if path.endswith(".npz"):
        return GridModel.load_npz(path)
    return load_grid_model(path)


def header_and_variants(model, lanes, n_busbar: int = 2, env_dynamics: bool = False):
    """(generated header, [[kind, variant], ...]) for batches of the given lane counts -- no device needed."""
    from .engine import PowerFlowEngine
    hdr, variants = None, []
    for n in lanes:
        eng = PowerFlowEngine(model, n_lanes=int(n), device=DEVICE_NONE, n_busbar=n_busbar)
        try:
            h = eng.specialization_header()
            assert hdr is None or h == hdr, "the header of a grid does not depend on the batch size"
            hdr = h
            p = eng.plan()
        finally:
            eng.close()
        b = lambda x: "true" if x else "false"  # noqa: E731
        nb, st, ipw, wpi, yr = p["busbars_per_block"], p["staging_tier"], p["instances_per_wavefront"], p["wavefronts_per_instance"], p["ybus_in_registers"]
        cand = [["step", f"{nb},{st},{ipw},2,{wpi},false,{b(yr)},false"], ["runpf", f"{nb},{st},{ipw},2,{wpi},false,{b(yr)}"]]
        if env_dynamics:                                  # the ENV step kernels: tables in global memory unless instance groups (gridpf_capi.hip plan_launch)
            st_e = st if ipw > 1 else 0
            cand.append(["step", f"{nb},{st_e},{ipw},2,{wpi},false,{b(yr and st_e == 0)},true"])
        for c in cand:
            if c not in variants:
                variants.append(c)
    return hdr, variants


def add_to_manifest(out_dir, label, hdr, variants):
    os.makedirs(out_dir, exist_ok=True)
    hid = hashlib.sha1(hdr.encode()).hexdigest()[:12]
    mp = os.path.join(out_dir, "manifest.json")
    man = json.load(open(mp)) if os.path.exists(mp) else {}
    ent = man.setdefault(hid, {"header": f"{hid}.h", "grids": [], "variants": []})
    with open(os.path.join(out_dir, ent["header"]), "w") as f:
        f.write(hdr)
    if label not in ent["grids"]:
        ent["grids"].append(label)
    for v in variants:
        if v not in ent["variants"]:
            ent["variants"].append(v)
    with open(mp, "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    return hid


def refresh(out_dir, grid_dir, n_busbar: int = 2):
    """Regenerate every header of the manifest from ``<grid_dir>/<label>.grid.npz`` (after a change of the kernel sources' parameter structs
    or of the symbolic programs), keeping each entry's kernel variants; headers are named by their hash, so the manifest keys change."""
    mp = os.path.join(out_dir, "manifest.json")
    man = json.load(open(mp))
    new = {}
    for ent in man.values():
        label = ent["grids"][0]
        hdr, _ = header_and_variants(_model(os.path.join(grid_dir, f"{label}.grid.npz")), [1], n_busbar)
        hid = hashlib.sha1(hdr.encode()).hexdigest()[:12]
        old = os.path.join(out_dir, ent["header"])
        if os.path.exists(old):
            os.remove(old)
        with open(os.path.join(out_dir, f"{hid}.h"), "w") as f:
            f.write(hdr)
        new[hid] = {"header": f"{hid}.h", "grids": ent["grids"], "variants": ent["variants"]}
        print(f"{label}: {ent['header']} -> {hid}.h")
    with open(mp, "w") as f:
        json.dump(new, f, indent=1, sort_keys=True)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    if argv is None:
        argv = sys.argv[1:]
    if argv and argv[0] == "--refresh":                    # python -m grid2op_amd.aot --refresh <grid dir> [out dir]
        refresh(argv[2] if len(argv) > 2 else os.path.join(HERE, "aot"), argv[1])
        return 0
    ap.add_argument("grid", help="grid.json of a grid2op environment (pandapower JSON) or a GridModel .npz")
    ap.add_argument("--lanes", type=int, action="append", help="batch size(s) the engine will be created with (default: 1 and 4096)")
    ap.add_argument("--n-busbar", type=int, default=2)
    ap.add_argument("--env-dynamics", action="store_true", help="also the step kernel with the environment's injection dynamics on")
    ap.add_argument("--out", default=os.path.join(HERE, "aot"))
    ap.add_argument("--label", default=None)
    a = ap.parse_args(argv)
    m = _model(a.grid)
    hdr, variants = header_and_variants(m, a.lanes or [1, 4096], a.n_busbar, a.env_dynamics)
    label = a.label or os.path.basename(os.path.dirname(os.path.abspath(a.grid))) or os.path.basename(a.grid)
    hid = add_to_manifest(a.out, label, hdr, variants)
    print(f"{a.out}/{hid}.h + manifest.json: {label}: {' '.join(k + '<' + v + '>' for k, v in variants)}")
    print("next: python -c \"import __graft_entry__ as g; g.build()\"   (hipcc cross-compiles the code objects; no GPU needed)")
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""Developer tool: write the substation graph of a golden grid as text for tools/lu_bench (n_sub n_line, then or ex per line)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grid2op_amd.grid_model import GridModel  # noqa: E402
env, out = sys.argv[1], sys.argv[2]
m = GridModel.load_npz(os.path.join(ROOT, "tests", "golden", f"{env}.grid.npz"))
with open(out, "w") as f:
    f.write(f"{m.n_sub} {m.n_line}\n")
    for a, b in zip(m.line_or_sub, m.line_ex_sub):
        f.write(f"{int(a)} {int(b)}\n")

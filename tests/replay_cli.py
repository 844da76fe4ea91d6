"""TEST INFRASTRUCTURE: replay recorded episodes (tests/replay.py) in a fresh interpreter on top of the grid2op STAND-IN
(tests/grid2op_stub) -- the configuration of the GPU box -- with the engine of choice.

    python tests/replay_cli.py {oracle|hip} [episode ...]
"""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "grid2op_stub"), HERE, ROOT]

import numpy as np  # noqa: E402

import grid2op  # noqa: E402
assert getattr(grid2op, "IS_STUB", False), "the real grid2op package shadows the stand-in"
from grid2op_amd.backend import HipBackend, _LanePool  # noqa: E402
import replay as R  # noqa: E402


def main():
    engine = sys.argv[1]
    names = sys.argv[2:] or sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "episodes", "*.npz")))
    if engine == "oracle":
        from oracle_engine import OracleEngine

        class Bk(HipBackend):
            def _make_engine(self, model, n_busbar, n_lanes=1):
                return OracleEngine(model, n_lanes=n_lanes, n_busbar=n_busbar)
    else:
        Bk = HipBackend
    for name in names:
        tr = dict(np.load(os.path.join(HERE, "golden", "episodes", f"{name}.npz")))
        grid = os.path.join(HERE, "golden", f"{str(tr['meta_grid'])}.grid.npz")
        n_pf, n_obs, worst = R.replay(tr, Bk, grid)
        assert not _LanePool._pools, "every lane must have been released"
        print(f"{name}: {n_pf} power flows, {n_obs} observations reproduced, worst float deviation {worst:.2e}")


if __name__ == "__main__":
    main()

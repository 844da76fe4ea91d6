"""TEST INFRASTRUCTURE: `HipBackend` with the compute engine swapped for the CPU oracle (see oracle_engine.py)."""
from grid2op_amd.backend import HipBackend
from oracle_engine import OracleEngine


class OracleHipBackend(HipBackend):
    def _make_engine(self, model, n_busbar, n_lanes=1):
        return OracleEngine(model, n_lanes=n_lanes, n_busbar=n_busbar)

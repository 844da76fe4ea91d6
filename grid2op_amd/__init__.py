"""grid2op_amd -- MI355X-native batched power-flow engine behind the grid2op Backend plugin surface.

* `grid2op_amd.engine.PowerFlowEngine`  batched API over ``libgridpf.so`` (no grid2op dependency)
* `grid2op_amd.backend.HipBackend`      single-environment drop-in ``grid2op.Backend.Backend`` (needs grid2op)
* `grid2op_amd.grid_model`              pandapower-JSON grid loader (no pandapower dependency)
"""
__version__ = "0.1.0"

from .grid_model import GridModel, load_grid_model  # noqa: F401


def __getattr__(name):
    if name == "PowerFlowEngine":
        from .engine import PowerFlowEngine
        return PowerFlowEngine
    if name == "HipBackend":
        from .backend import HipBackend
        return HipBackend
    raise AttributeError(name)

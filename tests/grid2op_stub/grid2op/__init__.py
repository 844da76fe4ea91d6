"""TEST INFRASTRUCTURE (GPU box only): the few names of the grid2op package that `grid2op_amd.backend` imports.

grid2op cannot be installed on the GPU box (no network; its reference checkout stays in the build container), so the
``-m gpu`` replay tests (tests/test_episode_replay.py) import `HipBackend` on top of this stand-in: a `Backend` base
class that only provides the bookkeeping `HipBackend` relies on (keyword storage for ``copy``, the ``can_handle_*``
flags, ``make_complete_path``, ``set_no_storage``, ``_compute_pos_big_topo``, ``assert_grid_correct``'s re-typing).  It is
written from the documented behaviour of grid2op/Backend/backend.py:130-335 and grid2op/Space/GridObjects.py, holds no
power-system logic, and is never on ``sys.path`` when the real package is importable."""
__version__ = "0.0.stub"
IS_STUB = True

"""Stand-in for ``grid2op.Backend.Backend`` (see the package docstring)."""
import copy
import os

import numpy as np

from grid2op.dtypes import dt_float, dt_int

_GRID_ATTRS = ("n_line", "n_gen", "n_load", "n_sub", "n_storage", "n_shunt", "dim_topo", "sub_info", "name_line", "name_gen",
               "name_load", "name_sub", "name_storage", "name_shunt", "load_to_subid", "gen_to_subid", "line_or_to_subid",
               "line_ex_to_subid", "storage_to_subid", "shunt_to_subid", "load_to_sub_pos", "gen_to_sub_pos",
               "line_or_to_sub_pos", "line_ex_to_sub_pos", "storage_to_sub_pos", "load_pos_topo_vect", "gen_pos_topo_vect",
               "line_or_pos_topo_vect", "line_ex_pos_topo_vect", "storage_pos_topo_vect")


class Backend:
    # class-level grid description (the real package fills it when ``assert_grid_correct`` re-types the instance)
    n_busbar_per_sub = 2
    detachment_is_allowed = False
    shunts_data_available = False
    n_line = n_gen = n_load = n_sub = n_storage = -1
    n_shunt = None
    dim_topo = -1

    def __init__(self, detailed_infos_for_cascading_failures=False, can_be_copied=True, **kwargs):
        self.detailed_infos_for_cascading_failures = detailed_infos_for_cascading_failures
        self.thermal_limit_a = None
        self._sh_vnkv = None
        self.comp_time = 0.0
        self.can_output_theta = False
        self._is_loaded = False
        self._can_be_copied = can_be_copied
        self._my_kwargs = {"detailed_infos_for_cascading_failures": detailed_infos_for_cascading_failures,
                           "can_be_copied": can_be_copied}
        self._my_kwargs.update(copy.deepcopy(kwargs))
        self._missing_two_busbars_support_info = True
        self.n_busbar_per_sub = 2
        self._missing_detachment_support_info = True
        self.detachment_is_allowed = False
        self._needs_active_bus = False

    def set_env_name(self, name):
        self.env_name = name

    def can_handle_more_than_2_busbar(self):
        self._missing_two_busbars_support_info = False
        self.n_busbar_per_sub = type(self).n_busbar_per_sub

    def can_handle_detachment(self):
        self._missing_detachment_support_info = False
        self.detachment_is_allowed = type(self).detachment_is_allowed

    def make_complete_path(self, path, filename=None):
        full = path if filename is None else os.path.join(path, filename)
        if not os.path.exists(full):
            raise RuntimeError(f'There is no grid at "{full}"')
        return full

    def set_no_storage(self):
        self.n_storage = 0
        self.name_storage = np.array([], dtype=str)
        self.storage_to_subid = np.zeros(0, dtype=dt_int)
        self.storage_to_sub_pos = np.zeros(0, dtype=dt_int)
        self.storage_pos_topo_vect = np.zeros(0, dtype=dt_int)

    def _compute_pos_big_topo(self):
        start = np.concatenate(([0], np.cumsum(self.sub_info)[:-1])).astype(dt_int)
        self.dim_topo = int(np.sum(self.sub_info))
        for el in ("load", "gen", "line_or", "line_ex", "storage"):
            sub = getattr(self, f"{el}_to_subid", None)
            if sub is None:
                continue
            setattr(self, f"{el}_pos_topo_vect", (start[sub] + getattr(self, f"{el}_to_sub_pos")).astype(dt_int))

    def assert_grid_correct(self, *a, **kw):
        """The grid description moves to a per-grid subclass and the instance is re-typed to it (copies made with
        ``type(self)(**self._my_kwargs)`` then see it as class attributes, as with the real package)."""
        if getattr(type(self), "_stub_retyped", False):
            return
        attrs = {k: getattr(self, k) for k in _GRID_ATTRS if hasattr(self, k)}
        attrs["_stub_retyped"] = True
        attrs["n_busbar_per_sub"] = self.n_busbar_per_sub
        self.__class__ = type(type(self).__name__ + "_grid", (type(self),), attrs)
        self._is_loaded = True

    def set_thermal_limit(self, limits):
        self.thermal_limit_a = np.asarray(limits, dtype=dt_float).copy()

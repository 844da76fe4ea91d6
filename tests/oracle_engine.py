"""TEST INFRASTRUCTURE: an engine with the interface of `grid2op_amd.engine.PowerFlowEngine` whose arithmetic is the
CPU oracle (oracle/pf_oracle.c).  It exists so that the HOST logic of `HipBackend` (the mirror of
PandaPowerBackend's apply_action / status / getters conventions) can be certified against the reference's own backend
conformance kit in the GPU-less build container.  It is never importable from the product package."""
import numpy as np

from grid2op_amd.engine import LaneResults, _OUT_FIELDS, _INJ_FIELDS
from oracle.pf_oracle_c import COracle


class OracleEngine:
    def __init__(self, model, n_lanes=1, device=0, n_busbar=2):
        self.model = model
        self.n_lanes = n_lanes
        self.n_busbar = n_busbar
        self._orc = COracle(model, n_busbar)
        m = model
        sizes = dict(n_line=m.n_line, n_gen=m.n_gen, n_load=m.n_load, n_storage=m.n_storage, n_shunt=m.n_shunt)
        off = 0
        self.out_slices = {}
        for name, sz in _OUT_FIELDS:
            self.out_slices[name] = slice(off, off + sizes[sz])
            off += sizes[sz]
        self.n_out = off
        off = 0
        self.inj_slices = {}
        for name, sz in _INJ_FIELDS:
            self.inj_slices[name] = slice(off, off + sizes[sz])
            off += sizes[sz]
        self.n_inj = off
        self.nb_total = m.n_sub * n_busbar
        self._inj = np.zeros((n_lanes, self.n_inj))
        self._topo = np.tile(m.initial_topo_vect(), (n_lanes, 1)).astype(np.int32)
        self._sb = np.tile(m.initial_shunt_bus(), (n_lanes, 1)).astype(np.int32)
        self._res = [None] * n_lanes

    def set_injections(self, inj, lane0=0):
        inj = np.asarray(inj, dtype=np.float64).reshape(-1, self.n_inj)
        self._inj[lane0:lane0 + inj.shape[0]] = inj

    def set_topology(self, topo, shunt_bus=None, lane0=0):
        topo = np.asarray(topo, dtype=np.int32).reshape(-1, self.model.dim_topo)
        self._topo[lane0:lane0 + topo.shape[0]] = topo
        if shunt_bus is not None and self.model.n_shunt:
            self._sb[lane0:lane0 + topo.shape[0]] = np.asarray(shunt_bus, dtype=np.int32).reshape(-1, self.model.n_shunt)

    def runpf(self, lane0=0, n=None, is_dc=False, max_iter=10, tol_mva=1e-8):
        n = self.n_lanes - lane0 if n is None else n
        r = self._orc.solve_rows(self._inj[lane0:lane0 + n], self._topo[lane0:lane0 + n], self._sb[lane0:lane0 + n],
                                 is_dc=is_dc, max_iter=max_iter, tol_mva=tol_mva)
        for k in range(n):
            self._res[lane0 + k] = {key: v[k] for key, v in r.items()}

    def results(self, lane0=0, n=None, with_bus=True):
        n = self.n_lanes - lane0 if n is None else n
        rows = self._res[lane0:lane0 + n]
        st = lambda key, dt: np.stack([np.asarray(r[key]) for r in rows]).astype(dt)
        return LaneResults(out=st("out", np.float32), topo_vect=st("topo_vect", np.int32), shunt_bus=st("shunt_bus", np.int32),
                           line_status=st("line_status", bool), status=st("status", np.int32), bus_vm=st("bus_vm", np.float64),
                           bus_va=st("bus_va", np.float64), _slices=self.out_slices)

    def solve_lane(self, lane, inj, topo, shunt_bus=None, is_dc=False, max_iter=10, tol_mva=1e-8):
        self.set_injections(np.asarray(inj)[None, :], lane0=lane)
        self.set_topology(np.asarray(topo)[None, :], None if shunt_bus is None else np.asarray(shunt_bus)[None, :], lane0=lane)
        self.runpf(lane, 1, is_dc=is_dc, max_iter=max_iter, tol_mva=tol_mva)
        return self.results(lane, 1)

    def close(self):
        pass

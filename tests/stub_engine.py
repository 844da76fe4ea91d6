"""TEST INFRASTRUCTURE: an engine with the call surface of `grid2op_amd.engine.PowerFlowEngine` and NO arithmetic.

It exists so that the process / device plumbing around the engine -- ``bench.py``'s ``--gpus N`` launcher (one rank per
GPU, barrier, max-over-ranks timing, the JSON line) and `grid2op_amd.sharding.ShardedEngine`'s routing of global lane
ranges to per-device engines -- can be exercised in the GPU-less build container.  ``bench.py --stub-engine`` labels its
output "STUB ENGINE ... not a measurement"."""
import time

import numpy as np

from grid2op_amd.engine import LaneResults, _INJ_FIELDS, _OUT_FIELDS


class StubEngine:
    def __init__(self, model, n_lanes=1, device=0, n_busbar=2):
        m = self.model = model
        self.n_lanes, self.device, self.n_busbar = int(n_lanes), int(device), n_busbar
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
        self.n_chron = 2 * m.n_load + 2 * m.n_gen
        self.nb_total = m.n_sub * n_busbar
        self.init_inj = np.zeros(self.n_inj)
        self.inj = np.zeros((self.n_lanes, self.n_inj))
        self.topo = np.tile(m.initial_topo_vect(), (self.n_lanes, 1)).astype(np.int32)
        self.lane_offset = np.zeros(self.n_lanes, np.int32)
        self.n_steps = 0
        self.n_runpf = 0
        self._t0 = None

    def pack_injections(self, n=1, **fields):
        return np.zeros((n, self.n_inj))

    def pack_chronics(self, load_p, load_q, prod_p, prod_v):
        return np.ascontiguousarray(np.concatenate([load_p, load_q, prod_p, prod_v], axis=-1), dtype=np.float32)

    def upload_chronics(self, tables):
        tables = np.asarray(tables)
        self.chron_T = tables.shape[-2]

    def set_lane_chronics(self, lane_table=None, lane_offset=None, lane_scale=None):
        if lane_offset is not None:
            assert len(lane_offset) == self.n_lanes
            self.lane_offset = np.asarray(lane_offset, np.int32)
        if lane_scale is not None:
            assert lane_scale.shape[0] == self.n_lanes

    def set_thermal_limits(self, limit_a):
        pass

    def set_injections(self, inj, lane0=0):
        inj = np.asarray(inj).reshape(-1, self.n_inj)
        self.inj[lane0:lane0 + inj.shape[0]] = inj

    def get_injections(self, lane0=0, n=None):
        n = self.n_lanes - lane0 if n is None else n
        return self.inj[lane0:lane0 + n].copy()

    def set_topology(self, topo, shunt_bus=None, lane0=0):
        topo = np.asarray(topo).reshape(-1, self.model.dim_topo)
        self.topo[lane0:lane0 + topo.shape[0]] = topo

    def reset(self, lane0=0, n=None):
        pass

    def runpf(self, lane0=0, n=None, **kw):
        self.n_runpf += 1

    def step(self, t, n_steps=1, **kw):
        self.n_steps += n_steps

    def sync(self):
        pass

    def set_profiling(self, mode):
        if mode == 3:                   # end-of-window marker of the real engine: nothing to do here
            return
        self._t0 = time.perf_counter() if mode else None
        self._k0 = self.n_steps

    def kernel_time(self):
        if self._t0 is None:
            return 0.0, 0
        return (time.perf_counter() - self._t0) * 1e3, self.n_steps - self._k0

    def results(self, lane0=0, n=None, with_bus=True):
        n = self.n_lanes - lane0 if n is None else n
        m = self.model
        st = np.zeros((n, 4), np.int32)
        st[:, 1] = 4
        st[:, 2] = m.n_sub
        # the first result column carries the lane's chronics offset so that routing can be checked end to end
        out = np.zeros((n, self.n_out), np.float32)
        out[:, 0] = self.lane_offset[lane0:lane0 + n]
        return LaneResults(out=out, topo_vect=self.topo[lane0:lane0 + n].copy(), shunt_bus=np.zeros((n, m.n_shunt), np.int32),
                           line_status=np.ones((n, m.n_line), bool), status=st, bus_vm=np.ones((n, self.nb_total)),
                           bus_va=np.zeros((n, self.nb_total)), _slices=self.out_slices)

    def step_outputs(self, lane0=0, n=None):
        n = self.n_lanes - lane0 if n is None else n
        nl = self.model.n_line
        return np.zeros((n, nl), np.float32), np.zeros((n, nl), np.int32), np.full((n, nl), -1, np.int32)

    def algorithmic_bytes_per_step(self):
        return 1

    def close(self):
        pass

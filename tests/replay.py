"""TEST INFRASTRUCTURE: recorded-episode traces of the Backend plugin boundary and their replay.

A trace (tests/golden/episodes/*.npz, written by tests/golden/make_episode_fixtures.py in the build container) is the
exact sequence of calls the UNMODIFIED reference ``Environment`` / ``ObservationSpace`` / ``N1Reward`` / ``Runner`` made on
the backend during an episode -- ``load_grid``, ``copy``, ``apply_action`` (the ``.values`` / ``.changed`` vectors of the
``_BackendAction`` it was handed), ``runpf(is_dc)``, ``_disconnect_line``, ``reset``, ``close`` -- for EVERY backend instance
of the episode (the environment's, the ``obs.simulate`` copy, the N-1 reward copies), together with everything the
getters returned after each ``runpf`` and the observation the environment built from them.  It was recorded with the
façade over the CPU oracle (tests/conformance_backend.py).  `replay` feeds the same calls to any backend factory and
compares: integer vectors bit-exact, floats within ``ABS + REL * |ref|``.

On the GPU box the factory is `grid2op_amd.backend.HipBackend` (real `PowerFlowEngine`, `_LanePool` lane sharing) on top of
the grid2op stand-in of tests/grid2op_stub (grid2op itself cannot be installed there)."""
import os

import numpy as np

EV_LOAD, EV_COPY, EV_APPLY, EV_RUNPF, EV_RESET, EV_CLOSE, EV_DISCO, EV_RECO = range(8)

ACT_FIELDS = ("prod_p", "prod_v", "load_p", "load_q", "storage", "topo", "shunt_p", "shunt_q", "shunt_bus", "storage_bus")
PF_INT = ("topo_vect", "line_status", "shunt_bus")
PF_FLOAT = ("p_or", "q_or", "v_or", "a_or", "p_ex", "q_ex", "v_ex", "a_ex", "gen_p", "gen_q", "gen_v", "load_p", "load_q", "load_v",
            "storage_p", "storage_q", "storage_v", "shunt_p", "shunt_q", "shunt_v", "theta_or", "theta_ex", "load_theta",
            "gen_theta", "storage_theta")
OBS_FLOAT = ("rho", "p_or", "q_or", "v_or", "a_or", "p_ex", "q_ex", "v_ex", "a_ex", "gen_p", "gen_q", "gen_v", "load_p", "load_q",
             "load_v", "theta_or", "theta_ex")
OBS_INT = ("line_status", "topo_vect")
ABS_TOL, REL_TOL = 2e-4, 5e-6


def read_backend(bk):
    """Everything the getters of a backend return (what the environment reads after ``runpf``)."""
    d = {"topo_vect": np.asarray(bk.get_topo_vect()).copy(), "line_status": np.asarray(bk.get_line_status()).copy()}
    d["p_or"], d["q_or"], d["v_or"], d["a_or"] = bk.lines_or_info()
    d["p_ex"], d["q_ex"], d["v_ex"], d["a_ex"] = bk.lines_ex_info()
    d["gen_p"], d["gen_q"], d["gen_v"] = bk.generators_info()
    d["load_p"], d["load_q"], d["load_v"] = bk.loads_info()
    d["storage_p"], d["storage_q"], d["storage_v"] = bk.storages_info()
    d["shunt_p"], d["shunt_q"], d["shunt_v"], d["shunt_bus"] = bk.shunt_info()
    d["theta_or"], d["theta_ex"], d["load_theta"], d["gen_theta"], d["storage_theta"] = bk.get_theta()
    return {k: np.asarray(v).copy() for k, v in d.items()}


class _VS:
    def __init__(self, values, changed):
        self.values, self.changed = values, changed


class FakeBackendAction:
    """What ``HipBackend.apply_action`` reads of a ``_BackendAction`` (Action/_backendAction.py:954-1018)."""

    def __init__(self, tr, row):
        self._v = {f: _VS(tr[f"act_{f}_values"][row], tr[f"act_{f}_changed"][row]) for f in ACT_FIELDS}

    def __call__(self):
        v = self._v
        return None, (v["prod_p"], v["prod_v"], v["load_p"], v["load_q"], v["storage"]), v["topo"], (v["shunt_p"], v["shunt_q"], v["shunt_bus"])

    def get_storages_bus(self):
        return self._v["storage_bus"]


def _cmp_float(got, ref, what):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), (what, "NaN pattern", got, ref)
    ok = ~nan_r
    err = np.abs(got[ok] - ref[ok])
    tol = ABS_TOL + REL_TOL * np.abs(ref[ok])
    assert np.all(err <= tol), (what, float(err.max()), got, ref)
    return float(err.max()) if err.size else 0.0


def replay(tr, factory, grid_path, env_dir=None):
    """Feed the recorded calls to backends made by ``factory()``; returns (n power flows checked, n observations checked,
    worst float deviation)."""
    bks = {}
    n_pf = n_obs = 0
    worst = 0.0
    obs_of_pf = {int(p): i for i, p in enumerate(tr["obs_pf_row"])} if "obs_pf_row" in tr else {}
    kinds, bids, args, rows = tr["ev_kind"], tr["ev_bid"], tr["ev_arg"], tr["ev_row"]
    for k, bid, arg, row in zip(kinds, bids, args, rows):
        k, bid, arg, row = int(k), int(bid), int(arg), int(row)
        if k == EV_LOAD:
            bk = factory()
            n_busbar = int(tr["meta_n_busbar"]) if "meta_n_busbar" in tr else 2
            cls = type(bk)                                       # what grid2op.make(..., n_busbar=N) does before load_grid
            if hasattr(cls, "set_n_busbar_per_sub"):
                cls.set_n_busbar_per_sub(n_busbar)
            else:
                cls.n_busbar_per_sub = n_busbar
            bk.set_env_name("replay_" + os.path.basename(grid_path).split(".")[0] + (f"_{n_busbar}bb" if n_busbar != 2 else ""))   # one grid class per grid (environment.py:300)
            bk.load_grid(grid_path)
            if env_dir is not None and bk.n_storage > 0:         # real grid2op: storage characteristics are part of the grid class
                bk.load_storage_data(env_dir)
            bk.assert_grid_correct()
            bks[bid] = bk
        elif k == EV_COPY:
            bks[arg] = bks[bid].copy()
        elif k == EV_APPLY:
            bks[bid].apply_action(FakeBackendAction(tr, row))
        elif k == EV_RUNPF:
            ok, exc = bks[bid].runpf(is_dc=bool(arg))
            assert bool(ok) == bool(tr["pf_ok"][row]), (row, ok, exc)
            assert (exc is None) == bool(ok)
            got = read_backend(bks[bid])
            for f in PF_INT:
                assert np.array_equal(np.asarray(got[f]).astype(np.int64), tr[f"pf_{f}"][row].astype(np.int64)), (row, f, got[f], tr[f"pf_{f}"][row])
            for f in PF_FLOAT:
                worst = max(worst, _cmp_float(got[f], tr[f"pf_{f}"][row], (row, f)))
            n_pf += 1
            if row in obs_of_pf:                     # the observation the reference Environment built from this power flow
                i = obs_of_pf[row]
                for f in OBS_INT:
                    assert np.array_equal(np.asarray(got[f]).astype(np.int64), tr[f"obs_{f}"][i].astype(np.int64)), (row, "obs", f)
                for f in OBS_FLOAT:
                    if f == "rho":
                        lim = tr["obs_thermal_limit"][i].astype(np.float64)
                        g = np.asarray(got["a_or"], dtype=np.float64) / lim
                        ref = tr["obs_rho"][i].astype(np.float64)
                        assert np.all(np.abs(g - ref) <= 1e-5 + 1e-5 * np.abs(ref)), (row, "rho", g, ref)
                    else:
                        worst = max(worst, _cmp_float(got[f], tr[f"obs_{f}"][i], (row, "obs", f)))
                n_obs += 1
        elif k == EV_RESET:
            bks[bid].reset(None, None)
        elif k == EV_CLOSE:
            bks[bid].close()
        elif k == EV_DISCO:
            bks[bid]._disconnect_line(arg)
        elif k == EV_RECO:
            bks[bid]._reconnect_line(arg)
        else:
            raise ValueError(k)
    for bk in bks.values():
        bk.close()
    return n_pf, n_obs, worst

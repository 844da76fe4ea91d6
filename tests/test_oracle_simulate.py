"""The oracle's restatement of obs.simulate (oracle/env_oracle.py: _BackendAction topology rules + forecast injections +
Backend.next_grid_state) against simulations recorded inside the UNMODIFIED reference environment
(tests/golden/make_simulate_fixtures.py: l2rpn_case14_sandbox, 8 steps x 13 candidate actions x time_step 0 / 1, default
parameters = overflow disconnections on)."""
import json

import numpy as np
import pytest

from oracle.env_oracle import maintenance_ahead, simulate
from oracle.pf_oracle import LaneState


def sim_cases(fx):
    cands = json.loads(str(fx["candidates_json"]))
    for c in cands:                                   # json turned the set_bus keys into strings
        if "set_bus" in c:
            c["set_bus"] = {int(k): v for k, v in c["set_bus"].items()}
    return cands


@pytest.mark.parametrize("grid,fixture,min_done", [("l2rpn_case14_sandbox", "simulate_case14.npz", 20),
                                                  ("l2rpn_neurips_2020_track1", "simulate_maintenance_neurips36.npz", 0)])
def test_oracle_reproduces_recorded_obs_simulate(grid, fixture, min_done, load_model, load_npz):
    """second fixture: a scenario with scheduled maintenance -- the forecast one step ahead of the observations around the first
    maintenance row has the line out already (maintenance_ahead: _ObsEnv.init, Environment/_obsEnv.py:361-385)"""
    m = load_model(grid)
    fx = load_npz(fixture)
    cands = sim_cases(fx)
    row0 = int(fx["row0"]) if "row0" in fx else 0
    n_forced = 0
    tab = {p: np.concatenate([fx[p + "_load_p"], fx[p + "_load_q"], fx[p + "_prod_p"], fx[p + "_prod_v"]], axis=1) for p in ("ch", "fc")}
    n_done = n_trip = 0
    for s in range(fx["row"].shape[0]):
        base = LaneState.from_model(m)
        base.topo = fx["topo_vect"][s].astype(np.int32)
        for ts in (0, 1):
            idx = int(fx["row"][s]) - row0
            row = tab["ch" if ts == 0 else "fc"][idx]
            mo = maintenance_ahead(fx["maintenance"], idx, ts) if "maintenance" in fx else None
            if mo is not None and ts >= 1:                   # the rule itself against what the reference's observation announces
                tnm, dnm = fx["time_next_maintenance"][s], fx["duration_next_maintenance"][s]
                assert np.array_equal(mo, (tnm != -1) & (tnm <= ts) & (tnm + dnm > ts)), (s, ts)
                n_forced += int((mo & fx["line_status"][s]).sum())
            for k, act in enumerate(cands):
                res, st, _ = simulate(m, base, row, act, fx["thermal_limit"], fx["timestep_overflow"][s], last_bus=fx["last_bus"][s],
                                      hard_overflow=float(fx["hard_overflow"]), nb_ts_allowed=int(fx["nb_ts_allowed"]), maint_out=mo,
                                      cascade=bool(fx["cascade"]) if "cascade" in fx else True)
                done = bool(fx[f"sim{ts}_done"][s, k])
                assert (not res.converged) == done, (s, ts, k, res.reason)
                n_done += done
                if done:
                    continue
                assert np.array_equal(res.topo_vect, fx[f"sim{ts}_topo_vect"][s, k]), (s, ts, k)
                assert np.array_equal(res.line_status.astype(bool), fx[f"sim{ts}_line_status"][s, k]), (s, ts, k)
                n_trip += int((res.line_status.astype(bool) != (st.topo[m.line_or_pos_topo_vect] >= 1)).any())
                for f, tol in [("p_or", 2e-4), ("q_or", 3e-4), ("p_ex", 2e-4), ("v_or", 2e-4), ("gen_p", 2e-4), ("gen_q", 3e-4), ("load_p", 1e-5)]:
                    assert np.abs(getattr(res, f) - fx[f"sim{ts}_{f}"][s, k]).max() < tol, (s, ts, k, f)
                rho = res.a_or / fx["thermal_limit"]
                assert np.abs(rho - fx[f"sim{ts}_rho"][s, k]).max() < 2e-5, (s, ts, k)
    assert n_done >= min_done
    if "maintenance" in fx:
        assert n_forced >= 1, "the fixture is meant to contain a forecast that takes a still-connected line out"

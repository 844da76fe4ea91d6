"""The oracle's PTDF (test oracle of the device DC sensitivity path) is consistent with the oracle's own DC power flow,
which is pinned to the reference's DC known answers (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from oracle.pf_oracle import LaneState, dc_bus_injection, ptdf, solve
from helpers import random_states


@pytest.mark.parametrize("name", ["rte_case5_example", "l2rpn_case14_sandbox", "educ_case14_storage", "l2rpn_wcci_2022_dev",
                                  "test_case14"])
def test_ptdf_times_injection_equals_dc_power_flow(name, load_model):
    m = load_model(name)
    rng = np.random.default_rng(7)
    n_ok = 0
    for st in [LaneState.from_model(m)] + random_states(m, 12, rng):
        r = solve(m, st, is_dc=True)
        if not r.converged:
            with pytest.raises((ValueError, np.linalg.LinAlgError)):
                T = ptdf(m, st)
                if np.isfinite(T).all():     # an island without slack is singular; an isolated element may still be solvable
                    raise ValueError("dc failed for another reason")
            continue
        T = ptdf(m, st)
        flows = T @ dc_bus_injection(m, st)
        assert np.abs(flows - r.p_or).max() < 1e-8 * max(1.0, np.abs(r.p_or).max())
        assert np.abs(flows + r.p_ex).max() < 1e-8 * max(1.0, np.abs(r.p_or).max())     # DC: no losses
        assert np.all(T[~r.line_status.astype(bool)] == 0.0)
        n_ok += 1
    assert n_ok >= 5


def test_ptdf_known_answer_dc_flows(load_model, load_npz):
    """grid2op/tests/BaseBackendTest.py:262-287 (DC p_or of test_case14) through the PTDF."""
    m = load_model("test_case14")
    ka = load_npz("known_answers.npz")
    st = LaneState.from_model(m)
    assert np.abs(ptdf(m, st) @ dc_bus_injection(m, st) - ka["p_or_dc"]).max() < 1e-7

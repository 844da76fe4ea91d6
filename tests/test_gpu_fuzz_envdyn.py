"""A small randomised sweep of the environment's injection dynamics inside the GPU suite (tests/fuzz_envdyn.py as a script runs the
large ones): per grid 48 lanes, 5 launches of 1..6 steps, every lane its own redispatch / storage / curtailment action at every launch,
held or one-step storage actions -- free-running against the oracle with the exact projection.  Infeasible projections must be reported
at exactly the step the oracle reports them (status trajectory); the states must agree to float32 rounding."""
import pytest

from fuzz_envdyn import NAMES, fuzz_envdyn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", NAMES)
def test_random_actions_free_running_vs_the_exact_oracle(name):
    res = fuzz_envdyn(name, 48, 5, seed=77 + NAMES.index(name))
    w = res["worst_abs_dev"]
    assert res["alive"] >= 8, res
    assert w["target"] < 1e-4 and w["actual"] < 2e-3 and w["charge"] < 1e-4 and w["storage_p"] < 1e-4 and w["gen_p"] < 3e-3, res

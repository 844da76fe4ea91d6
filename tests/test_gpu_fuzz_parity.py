"""A small randomised parity sweep inside the GPU suite (the large ones are run with tests/fuzz_parity.py as a script and summarised in
profiles/r05_fuzz_parity.json): per grid 768 lanes with random outages / bus splits (islanding combinations kept) and +-20 % load
jitter, 384 of them re-solved by the C oracle from the inputs the lanes hold on the device -- AC, DC, and the final state of a
three-step rollout with the protections on.  Bit-exact status / iteration count / topo_vect / line status; float32 outputs within
2e-4 + 5e-6 |x|; float64 pre-cast flows below 1e-4 pu of the grid's base."""
import pytest

from fuzz_parity import GRIDS, fuzz_grid, fuzz_multistep, fuzz_specialised

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", GRIDS)
@pytest.mark.parametrize("cascade", [False, True])
def test_random_topologies_and_injections_vs_the_c_oracle(env, cascade):
    res = fuzz_grid(env, 768, 384, seed=1234 + GRIDS.index(env) + (100 if cascade else 0), cascade=cascade)
    for key in ("ac", "dc"):
        if key not in res:
            continue
        v = res[key]
        assert v["ok"] and v["status_mismatch"] == 0 and v["n_iter_mismatch"] == 0 and v["topo_vect_mismatch"] == 0 and v["line_status_mismatch"] == 0, (key, v)
        assert v["nan_in_converged"] == 0 and v["non_nan_in_failed"] == 0, (key, v)
    assert 0 < res["ac"]["n_converged"] <= res["ac"]["n"]


@pytest.mark.parametrize("env", GRIDS)
@pytest.mark.parametrize("auto_reset", [False, True])
def test_one_multi_step_launch_equals_single_step_launches_on_random_topologies(env, auto_reset):
    """ONE 6-step launch == 6 single-step launches (integers bit for bit, floats to float32 rounding), protections on and lines tripping, on random topologies (bus
    splits -> topology-class kernels, outages); with auto-reset the failed lanes restart inside the launch as they do between launches."""
    res = fuzz_multistep(env, 320, seed=4321 + GRIDS.index(env), n_steps=6, auto_reset=auto_reset)
    assert res["ok"], res
    assert res["converged_last"] > 0


@pytest.mark.parametrize("env", ["l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev"])
def test_specialised_kernels_equal_the_shipped_ones_on_random_topologies(env, tmp_path_factory):
    """gpf_jit_enable on random topologies (outages, bus splits -> topology-class launches), protections on, auto-reset: every buffer a
    4-step + a 1-step launch leave behind is bit-identical to the shipped kernels', float64 bus voltages included."""
    res = fuzz_specialised(env, 192, seed=999 + GRIDS.index(env), cache_dir=str(tmp_path_factory.mktemp("jit_fuzz")))
    assert res["ok"], res

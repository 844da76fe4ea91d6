"""The oracle's restatement of the environment's injection dynamics (oracle/env_oracle.py InjectionDynamics: storage state of
charge, redispatch accumulation, the ramp-limited projection) against the internal state recorded step by step inside UNMODIFIED
reference environments (tests/golden/make_envdyn_fixtures.py)."""
import numpy as np
import pytest

from oracle.env_oracle import InjectionDynamics


def dyn_from_fixture(fx, exact=False):
    lim = {k: fx[k] for k in ("pmin", "pmax", "ramp_up", "ramp_down", "redispatchable")}
    lim["eps_poly"], lim["tol_poly"] = float(fx["eps_poly"]), float(fx["tol_poly"])
    if "renewable" in fx:
        lim["renewable"] = fx["renewable"]
    sto = None
    if "storage_Emax" in fx:
        sto = {"Emax": fx["storage_Emax"], "Emin": fx["storage_Emin"], "loss": fx["storage_loss"],
               "charging_efficiency": fx["storage_charging_efficiency"], "discharging_efficiency": fx["storage_discharging_efficiency"]}
    return InjectionDynamics(lim, len(fx["pmin"]), sto, float(fx["delta_time_seconds"]), fx["storage_charge0"],
                             bool(fx["activate_storage_loss"]), exact=exact)


@pytest.mark.parametrize("name", ["educ_case14_storage", "l2rpn_wcci_2022_dev", "educ_case14_storage_emin", "educ_case14_storage_illegal"])
def test_injection_dynamics_reproduce_the_reference_environment(name, load_npz):
    fx = load_npz(f"envdyn_{name}.npz")
    dyn = dyn_from_fixture(fx)
    dyn.prev_p[:] = 0.0
    n = fx["row"].shape[0]
    for t in range(n):
        if t == 0:                      # the reset step left _gen_activeprod_t_redisp = set-points of row 0 (no dispatch yet)
            dyn.prev_p[:] = fx["ch_prod_p"][fx["row"][0] - 1]
        ok, gen, sto = dyn.step(fx["new_p"][t], fx["act_redisp"][t], fx["act_storage"][t], fx["act_curtail"][t])
        assert ok
        assert bool(dyn.illegal) == bool(fx["failed_redisp"][t]), t          # (_prepare_redisp: the action was cancelled)
        assert np.abs(dyn.limit - fx["limit_curtailment"][t]).max() < 1e-6 and abs(dyn.sum_curt - fx["sum_curtailment"][t]) < 1e-4, t
        assert np.abs(dyn.target - fx["target"][t]).max() < 1e-5, t
        assert np.abs(dyn.actual - fx["actual"][t]).max() < 2e-4, (t, np.abs(dyn.actual - fx["actual"][t]).max())
        assert np.array_equal(dyn.already, fx["already_modified"][t]), t
        assert np.abs(dyn.prev_p - fx["prev_p"][t]).max() < 2e-4, t
        if len(sto):
            assert np.abs(sto - fx["storage_power"][t]).max() < 1e-5, t
            assert np.abs(dyn.charge - fx["storage_charge"][t]).max() < 1e-5, t
            assert abs(dyn.amount - fx["amount_storage"][t]) < 1e-5, t
        # the non-slack set-points are what the observation reports
        ns = np.abs(gen - fx["gen_p"][t]) < 1e-2
        assert ns.sum() >= len(gen) - 1, t


@pytest.mark.parametrize("name", ["educ_case14_storage", "l2rpn_wcci_2022_dev"])
def test_exact_projection_stays_close_to_the_reference_and_is_never_worse(name, load_npz):
    """The exact minimiser of the redispatch projection (what the device computes) vs the reference's SLSQP result, step by step
    along the recorded episode: same feasibility, the recorded dispatch within SLSQP's own inexactness (a few tenths of a MW while
    ramp limits bind -- up to 0.8 MW on the 62-generator grid, whole groups of identical generators shifted together --), objective
    never above SLSQP's."""
    from oracle.redispatch_oracle import objective_mw, qp_terms
    fx = load_npz(f"envdyn_{name}.npz")
    ex, ref = dyn_from_fixture(fx, exact=True), dyn_from_fixture(fx)
    for d in (ex, ref):
        d.prev_p[:] = fx["ch_prod_p"][fx["row"][0] - 1]
    worst = 0.0
    for t in range(fx["row"].shape[0]):
        # the exact solver from the REFERENCE's state of the previous step: one projection, compared with the recorded one
        q = None
        if t > 0:
            ex.target[:], ex.actual[:], ex.prev_p[:], ex.already[:] = ref.target, ref.actual, ref.prev_p, ref.already
            ex.charge[:], ex.amount_prev = ref.charge, ref.amount_prev
        a0, t0_, p0, al0 = ref.actual.copy(), ref.target.copy(), ref.prev_p.copy(), ref.already.copy()
        if t > 0:
            ex.limit[:], ex.sum_curt_prev = ref.limit, ref.sum_curt_prev
        ok_e, gen_e, _ = ex.step(fx["new_p"][t], fx["act_redisp"][t], fx["act_storage"][t], fx["act_curtail"][t])
        ok_r, gen_r, _ = ref.step(fx["new_p"][t], fx["act_redisp"][t], fx["act_storage"][t], fx["act_curtail"][t])
        assert ok_e and ok_r
        worst = max(worst, float(np.abs(ex.actual - ref.actual).max()))
        assert np.abs(ex.actual - ref.actual).max() < 1.0, (t, np.abs(ex.actual - ref.actual).max())     # sanity bound (tests/test_redispatch.py)
        q = qp_terms(fx["new_p"][t].astype(np.float64), p0.astype(np.float64), a0.astype(np.float64), ref.target.astype(np.float64),
                     ref.already.copy(), ref.amount, ref.sum_curt, 0.0, ref.lim)
        if q is not None and (np.abs(ex.actual - a0).max() > 0 or np.abs(ref.actual - a0).max() > 0):
            xe, xr = (ex.actual - a0)[q["part"]].astype(np.float64), (ref.actual - a0)[q["part"]].astype(np.float64)
            assert objective_mw(q, xe) <= objective_mw(q, xr) + 1e-3, (t, objective_mw(q, xe), objective_mw(q, xr))
    print(name, "worst |exact - SLSQP| along the episode", worst)

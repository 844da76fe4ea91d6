"""TEST INFRASTRUCTURE -- CPU restatement of the reference's redispatching automaton.

``BaseEnv._compute_dispatch_vect`` (grid2op/Environment/baseEnv.py:2211-2470) turns the redispatch the agents ask for
(``target_dispatch``) into the redispatch that is physically applied (``actual_dispatch``): a small quadratic program per
environment, solved there with ``scipy.optimize.minimize(method="SLSQP")``,

    minimise    sum_{i in M}  w_i (x_i - (target_i - actual_i))^2          M = generators modified by an action (all if none)
    subject to  sum_{i in G}  x_i = storage - curtailment + detached           G = participating generators
                max(pmin_i - p_i, -ramp_down_i - incr_i) <= x_i <= min(pmax_i - p_i, ramp_up_i - incr_i)

with ``p_i = new_p_i + actual_i``, ``incr_i`` the change of the chronics set-point since the previous step and
``w_i ~ 1 / (ramp_up_i + ramp_down_i)``; then ``actual += x``.  This module restates it formula by formula (same scaling,
same starting point, same solver options -- scipy is the solver the reference itself calls, :2411-2425) and is pinned against
calls recorded inside unmodified reference environments (tests/golden/redispatch_cases.npz, tests/test_redispatch.py).
Only tests/ may import it: the product path is the HIP kernel behind ``gpf_redispatch``."""
import numpy as np
from scipy.optimize import LinearConstraint, minimize


def qp_terms(new_p, prev_p, actual, target, modified, storage, curtail, detached, lim, first=False):
    """The pieces of the QP (physical units, MW): participating mask, bounds, right-hand side, weights, modified mask, targets
    -- :2225-2345.  Returns None when the infeasibility pre-check (:2246-2248, 2472-2511) refuses the step."""
    pmin, pmax, ru, rd, redisp = lim["pmin"], lim["pmax"], lim["ramp_up"], lim["ramp_down"], lim["redispatchable"]
    prev = new_p.copy() if first else prev_p                                     # :2222-2223
    part = ((new_p > 0.0) | (np.abs(actual) >= 1e-7) | (target != actual)) & redisp          # :2227-2232
    incr = new_p - (prev - actual)                                               # :2235-2237
    avail_down = np.maximum(pmin[part] - prev[part], -rd[part])                  # :2241-2245
    avail_up = np.minimum(pmax[part] - prev[part], ru[part])                     # :2247-2251
    sum_move = incr[part].sum() + storage - curtail + detached                   # :2474-2476
    if sum_move > avail_up.sum() or sum_move < avail_down.sum():
        return None
    new_p_th = new_p[part] + actual[part]                                        # :2343
    lo = np.maximum(pmin[part] - new_p_th, -rd[part] - incr[part])               # :2346-2354
    hi = np.minimum(pmax[part] - new_p_th, ru[part] - incr[part])                # :2358-2366
    coeffs = 1.0 / (ru + rd + lim["eps_poly"])                                   # :2303-2305
    w = coeffs[part] / coeffs[part].sum()
    mod = modified[part].copy()
    tv = (target - actual)[part]
    if not mod.any():                                                            # :2309-2312
        mod[:] = True
    return dict(part=part, lo=lo, hi=hi, rhs=storage - curtail + detached, w=w, mod=mod, tv=tv, actual_part=actual[part],
                target_part=target[part], modified_part=modified[part])


def compute_dispatch(new_p, prev_p, actual, target, modified, storage, curtail, detached, lim, first=False):
    """-> (ok, actual_dispatch after)"""
    q = qp_terms(new_p, prev_p, actual, target, modified, storage, curtail, detached, lim, first)
    if q is None:
        return False, actual.copy()
    eps, tol = lim["eps_poly"], lim["tol_poly"]
    n = int(q["part"].sum())
    scale_x = float(max(np.max(np.abs(actual)), 1.0))                            # :2317-2318
    tv_opt = (q["tv"][q["mod"]] / scale_x).astype(float)
    scale_obj = float(np.round(max(0.5 * np.abs(tv_opt).sum() ** 2, 1.0), decimals=4))       # :2324-2326
    w, mod = q["w"], q["mod"]
    added = 0.5 * eps
    eq = LinearConstraint(np.ones((1, n)), q["rhs"] / scale_x, q["rhs"] / scale_x)
    ineq = LinearConstraint(np.eye(n), (q["lo"] - added) / scale_x, (q["hi"] + added) / scale_x)
    x0 = np.zeros(n)                                                             # :2384-2408
    if (np.abs(target) >= 1e-7).any() or modified.any():
        g0 = (np.abs(q["target_part"]) >= 1e-7) | q["modified_part"]
        x0[g0] = (q["target_part"][g0] - q["actual_part"][g0]) / scale_x
        can = np.abs(x0) <= 1e-7
        if can.any():
            denom = (1.0 / w[can]).sum()
            if denom <= 1e-2:
                denom = 1.0
            x0[can] = -x0.sum() / (w[can] * denom)
    else:
        x0 -= q["actual_part"] / scale_x

    def fun(x):
        return (w[mod] * (x[mod] - tv_opt) ** 2).sum() / scale_obj

    def jac(x):
        g = np.zeros(n)
        g[mod] = 2.0 * w[mod] * (x[mod] - tv_opt)
        return g / scale_obj
    opt = max(float(eps / scale_x), 1e-6)
    res = minimize(fun, x0, method="SLSQP", constraints=[eq, ineq], options={"eps": opt, "ftol": opt, "disp": False}, jac=jac)
    out = actual.copy()
    if res.success:
        out[q["part"]] += res.x * scale_x
        return True, out
    vals = np.concatenate(([res.x.sum()], res.x))                                # :2433-2452: tolerate small violations
    downs = np.concatenate(([q["rhs"] / scale_x], (q["lo"] - added) / scale_x))
    ups = np.concatenate(([q["rhs"] / scale_x], (q["hi"] + added) / scale_x))
    if np.all(vals - downs >= -tol) and np.all(vals - ups <= tol):
        out[q["part"]] += res.x * scale_x
        return True, out
    return False, actual.copy()


def objective_mw(q, x):
    """weighted squared distance to the agents' targets (MW^2) of a candidate x (MW, participating generators)"""
    return float((q["w"][q["mod"]] * (x[q["mod"]] - q["tv"][q["mod"]]) ** 2).sum())


def solve_exact(q, eps_poly):
    """EXACT minimiser of the program of `qp_terms` (the reference's SLSQP stops at ftol on a scaled objective and sits up to a few
    tenths of a MW away from it): the program is separable with one coupling constraint, so x_i(lambda) = clip(t_i - lambda / (2 w_i))
    on the generators of M, the generators of G \\ M sit at a bound when lambda != 0 and share the remainder in proportion to
    1 / w_i when lambda = 0 -- the point the reference starts SLSQP from.  Plain numpy bisection (200 halvings), written
    independently of the device kernel (gridpf_redispatch.hpp / env_dynamics_step) it is the reference for.  -> x [participating]"""
    lo, hi, w, mod, tv, rhs = q["lo"] - 0.5 * eps_poly, q["hi"] + 0.5 * eps_poly, q["w"], q["mod"], q["tv"], q["rhs"]
    free = ~mod
    x = np.zeros(len(lo))

    def xm(lam):
        return np.clip(tv[mod] - lam / (2.0 * w[mod]), lo[mod], hi[mod])
    s0 = xm(0.0).sum()
    f_lo, f_hi = lo[free].sum(), hi[free].sum()
    if rhs - s0 > f_hi or rhs - s0 < f_lo:
        up = rhs - s0 > f_hi
        fb = f_hi if up else f_lo
        a, b = ((2.0 * w[mod] * (tv[mod] - hi[mod])).min(), 0.0) if up else (0.0, (2.0 * w[mod] * (tv[mod] - lo[mod])).max())
        for _ in range(200):
            mid = 0.5 * (a + b)
            if xm(mid).sum() + fb > rhs:
                a = mid
            else:
                b = mid
        x[mod] = xm(0.5 * (a + b))
        x[free] = hi[free] if up else lo[free]
        inside = mod & (x > lo) & (x < hi)
        if inside.any():
            x[inside] += (rhs - x.sum()) / inside.sum()
    else:
        x[mod] = xm(0.0)
        if free.any():
            r = rhs - s0
            a, b = np.minimum(lo[free] * w[free], hi[free] * w[free]).min(), np.maximum(lo[free] * w[free], hi[free] * w[free]).max()
            for _ in range(200):
                mid = 0.5 * (a + b)
                if np.clip(mid / w[free], lo[free], hi[free]).sum() < r:
                    a = mid
                else:
                    b = mid
            x[free] = np.clip(0.5 * (a + b) / w[free], lo[free], hi[free])
    return x


def compute_dispatch_exact(new_p, prev_p, actual, target, modified, storage, curtail, detached, lim, first=False):
    """As `compute_dispatch`, with the exact minimiser instead of SLSQP's approximate one.  -> (ok, actual_dispatch after)"""
    q = qp_terms(new_p, prev_p, actual, target, modified, storage, curtail, detached, lim, first)
    if q is None:
        return False, actual.copy()
    added = 0.5 * lim["eps_poly"]
    if q["rhs"] < (q["lo"] - added).sum() or q["rhs"] > (q["hi"] + added).sum():
        return False, actual.copy()
    out = actual.copy()
    out[q["part"]] += solve_exact(q, lim["eps_poly"])
    return True, out

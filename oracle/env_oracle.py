"""CPU restatement of the pieces of the reference ENVIRONMENT that the batched engine reproduces on the device.  TEST
INFRASTRUCTURE (like everything under oracle/): only tests/ may import it.

* `apply_topo_action`  -- `_BackendAction.__iadd__` restricted to topology (grid2op/Action/_backendAction.py:836-919;
  ValueStore.change_status / set_status / set_val / change_val :140-234; `_aux_iadd_reconcile_disco_reco` :738-765).
* `next_grid_state`    -- `Backend.next_grid_state` (grid2op/Backend/backend.py:1433-1521): overflow disconnections until the grid
  is stable, on top of `pf_oracle.solve`.
* `simulate`           -- what `obs.simulate(action, time_step)` asks of the backend (grid2op/Observation/baseObservation.py:3365-3670
  -> Environment/_obsEnv.py init / simulate -> BaseEnv.step): forecast injections in float32 as PandaPowerBackend applies them
  (pandaPowerBackend.py:925-969), the candidate topology, one protected power flow.
"""
from __future__ import annotations

import numpy as np

from .pf_oracle import LaneState, solve


def apply_topo_action(m, topo, act, last_bus=None, shunt_bus=None):
    """`act`: dict with any of set_line_status [(line, +-1)], change_line_status [lines], set_bus {pos: bus}, lines_or_bus /
    lines_ex_bus / loads_bus / gens_bus / storages_bus [(id, bus)], change_bus [positions], shunts_bus [(id, bus)].
    Returns the new topology row (and mutates `shunt_bus` in place when given)."""
    row = np.array(topo, dtype=np.int32).copy()
    lor, lex = np.asarray(m.line_or_pos_topo_vect), np.asarray(m.line_ex_pos_topo_vect)

    def old(pos):
        return int(last_bus[pos]) if (last_bus is not None and last_bus[pos] >= 1) else 1

    def reco(l):
        for p_ in (lor[l], lex[l]):
            if row[p_] < 0:
                row[p_] = old(p_)

    def disco(l):
        row[lor[l]] = row[lex[l]] = -1
    # III line status: change_status (:168-198) then set_status (:200-231)
    for l in act.get("change_line_status", ()):
        if row[lor[l]] > 0 or row[lex[l]] > 0:
            disco(l)
        else:
            reco(l)
    for l, v in act.get("set_line_status", ()):
        if v < 0:
            disco(l)
        elif v > 0:
            reco(l)
    or_before, ex_before = row[lor].copy(), row[lex].copy()
    # IV change_bus (:156-159) then set_bus (:151-154)
    modif = False
    for p_ in act.get("change_bus", ()):
        if row[p_] > 0:
            row[p_] = (1 - row[p_]) + 2
        modif = True
    pos_of = {"lines_or_bus": lor, "lines_ex_bus": lex, "loads_bus": m.load_pos_topo_vect, "gens_bus": m.gen_pos_topo_vect,
              "storages_bus": m.storage_pos_topo_vect}
    for key, pos in pos_of.items():
        for el, bus in act.get(key, ()):
            if bus != 0:
                row[pos[el]] = bus
                modif = True
    for p_, bus in dict(act.get("set_bus", {})).items():
        if bus != 0:
            row[int(p_)] = bus
            modif = True
    # V reconcile (:738-765): a line with an open end is open; an open line that got a bus is reconnected
    if modif:
        for l in range(m.n_line):
            o_, x_ = row[lor[l]], row[lex[l]]
            d_now = or_before[l] == -1 or o_ == -1 or ex_before[l] == -1 or x_ == -1
            r_now = or_before[l] == -1 and (o_ >= 1 or x_ >= 1)
            if r_now:
                reco(l)
            elif d_now:
                disco(l)
    if shunt_bus is not None:
        for sh, bus in act.get("shunts_bus", ()):
            if bus != 0:
                shunt_bus[sh] = bus
    return row


def next_grid_state(m, st: LaneState, thermal_limit, timestep_overflow, hard_overflow=2.0, soft_overflow=1.0, nb_ts_allowed=2,
                    cascade=True, is_dc=False, max_rounds=64):
    """backend.py:1476-1520.  Returns (result of the last power flow, LaneState with the tripped lines open, updated protection
    counters as BaseEnv keeps them after the step: baseEnv.py:3346-3370)."""
    st = LaneState(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.__dict__.items()})
    lor, lex = np.asarray(m.line_or_pos_topo_vect), np.asarray(m.line_ex_pos_topo_vect)
    ts = np.asarray(timestep_overflow, dtype=np.int64).copy()
    counted = np.zeros(m.n_line, bool)
    lim = np.asarray(thermal_limit, dtype=np.float32)
    res = None
    for _ in range(max_rounds + 1):
        res = solve(m, st, is_dc=is_dc)
        if not res.converged or not cascade:
            break
        a = res.a_or.astype(np.float32)
        on = res.line_status.astype(bool)
        over_soft = on & (a > np.float32(soft_overflow) * lim)
        newly = over_soft & ~counted
        counted |= newly
        local = ts + counted.astype(np.int64)
        to_disc = on & ((a > np.float32(hard_overflow) * lim) | (local > nb_ts_allowed))
        if not to_disc.any():
            break
        st.topo[lor[to_disc]] = -1
        st.topo[lex[to_disc]] = -1
    if res.converged:
        a = res.a_or.astype(np.float32)
        ts = np.where(a > np.float32(soft_overflow) * lim, ts + 1, 0)
    return res, st, ts


def forecast_state(m, base: LaneState, row):
    """chronics / forecast row [load_p | load_q | prod_p | prod_v(kV)] -> injections, float32 arithmetic as the reference"""
    st = LaneState(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base.__dict__.items()})
    nl, ng = m.n_load, m.n_gen
    row = np.asarray(row, dtype=np.float32)
    st.load_p = row[:nl].astype(np.float64)
    st.load_q = row[nl:2 * nl].astype(np.float64)
    st.gen_p = row[2 * nl:2 * nl + ng].astype(np.float64)
    st.gen_vm = (row[2 * nl + ng:2 * nl + 2 * ng] / m.sub_vn_kv[m.gen_sub].astype(np.float32)).astype(np.float64)
    return st


def maintenance_ahead(maint, idx, horizon):
    """Lines `_ObsEnv.init` forces out of a forecast `horizon` >= 1 steps ahead of the observation at chronics row `idx`
    (Environment/_obsEnv.py:361-385 with BaseEnv._update_vector_with_timestep, baseEnv.py:4768-4825): the NEXT maintenance of the
    line -- the one obs.time_next_maintenance / duration_next_maintenance describe, i.e. the first flagged run of rows of
    maintenance.csv from `idx` on (Chronics/gridValue.py:264-340) -- begins at row idx + horizon (first_ts_maintenance) or covers
    it (still_in_maintenance).  `maint`: [T, n_line] 0/1.  Returns a bool mask [n_line]."""
    maint = np.asarray(maint) != 0
    T, n_line = maint.shape
    out = np.zeros(n_line, bool)
    tgt = idx + horizon
    if horizon < 1 or tgt >= T:
        return out
    for l in range(n_line):
        on = np.nonzero(maint[idx:, l])[0]
        if on.size == 0:
            continue
        s_ = idx + int(on[0])                      # tnm = s_ - idx (0: under way)
        out[l] = s_ <= tgt and bool(maint[s_:tgt + 1, l].all())
    return out


def simulate(m, base: LaneState, row, act, thermal_limit, timestep_overflow, last_bus=None, maint_out=None, **kw):
    st = forecast_state(m, base, row)
    if maint_out is not None and np.any(maint_out):             # scheduled maintenance ahead: out BEFORE the candidate action
        lor, lex = np.asarray(m.line_or_pos_topo_vect), np.asarray(m.line_ex_pos_topo_vect)
        st.topo = st.topo.copy()
        st.topo[lor[np.asarray(maint_out, bool)]] = -1
        st.topo[lex[np.asarray(maint_out, bool)]] = -1
    sb = st.shunt_bus.copy() if m.n_shunt else None
    st.topo = apply_topo_action(m, st.topo, act, last_bus, sb)
    if sb is not None:
        st.shunt_bus = sb
    return next_grid_state(m, st, thermal_limit, timestep_overflow, **kw)


class InjectionDynamics:
    """What BaseEnv.step does to the generator / storage set-points between the chronics and the backend, for an agent that hands
    over a redispatch vector and a storage-power vector per step (zeros = do nothing): `_compute_storage` (baseEnv.py:2829-2905) +
    `_withdraw_storage_losses` (:2777-2790), `_get_already_modified_gen` (:2101-2115), the `_make_redisp` gate (:2188-2209) and
    `_compute_dispatch_vect` (:2211-2470, restated in oracle/redispatch_oracle.py with the reference's own solver), the set-points
    of `set_redispatch` / `set_storage` (:3829-3831) and `_gen_activeprod_t_redisp` (:3439).  State arrays are float32 like the
    reference's (dt_float).  Restated since round 4: the cancellation of an illegal redispatch action (`_prepare_redisp` :2140-2173,
    `self.illegal` after a step).  Not restated: detachment, generator up / down times, the dispatch of switched-off generators
    (Parameters.ALLOW_DISPATCH_GEN_SWITCH_OFF = False), LIMIT_INFEASIBLE_CURTAILMENT_STORAGE_ACTION."""

    def __init__(self, lim, n_gen, sto=None, delta_time_seconds=300.0, storage_charge0=None, activate_storage_loss=True, exact=False):
        self.lim = lim
        self.exact = bool(exact)       # the exact minimiser of the projection instead of SLSQP's approximate one (what the device computes)
        f32 = np.float32
        self.target = np.zeros(n_gen, f32)
        self.actual = np.zeros(n_gen, f32)
        self.prev_p = np.zeros(n_gen, f32)                    # _gen_activeprod_t_redisp
        self.already = np.zeros(n_gen, bool)
        self.sto = sto
        self.coeff = delta_time_seconds / 3600.0
        self.loss_on = bool(activate_storage_loss)
        n_sto = 0 if sto is None else len(sto["Emax"])
        self.charge = np.zeros(n_sto, f32) if storage_charge0 is None else np.asarray(storage_charge0, f32).copy()
        self.power = np.zeros(n_sto, f32)
        self.amount = 0.0
        self.amount_prev = 0.0
        self.limit = np.ones(n_gen, f32)                      # _limit_curtailment (ratio of pmax; 1 = not curtailed)
        self.sum_curt = 0.0                                   # _sum_curtailment_mw (change against the previous step)
        self.sum_curt_prev = 0.0
        self.fresh = True                                     # no step yet since the reset: the previous set-points are the step's own
                                                              # (nb_time_step == 0, baseEnv.py:2218-2219); assigning prev_p clears it

    def _compute_storage(self, act):
        s = self.sto
        self.charge_prev = self.charge.copy()                # _storage_previous_charge (baseEnv.py:2830)
        act = np.asarray(act, np.float32)
        sel = np.isfinite(act) & (np.abs(act) >= 1e-7)
        self.power[:] = 0.0
        if sel.any():
            a = act[sel]
            eff = np.ones(int(sel.sum()))
            if self.loss_on:
                eff[a > 0.0] *= s["charging_efficiency"][sel][a > 0.0]
                eff[a < 0.0] /= s["discharging_efficiency"][sel][a < 0.0]
            self.charge[sel] += (a * self.coeff * eff).astype(np.float32)
            self.power[sel] = a
            hi = self.charge > s["Emax"]
            if hi.any():
                t_ = (1.0 / self.coeff) * (self.charge[hi] - s["Emax"][hi])
                if self.loss_on:
                    t_ = t_ / s["charging_efficiency"][hi]
                self.power[hi] -= t_.astype(np.float32)
                self.charge[hi] = s["Emax"][hi]
            lo = self.charge < s["Emin"]
            if lo.any():
                t_ = (1.0 / self.coeff) * (self.charge[lo] - s["Emin"][lo])
                if self.loss_on:
                    t_ = t_ * s["discharging_efficiency"][lo]
                self.power[lo] -= t_.astype(np.float32)
                self.charge[lo] = s["Emin"][lo]
            self.charge[:] = np.maximum(self.charge, s["Emin"])
            self.amount = float(self.power.sum())
        else:
            self.amount = 0.0
        tmp = self.amount
        self.amount -= self.amount_prev
        self.amount_prev = tmp
        if self.loss_on:
            self.charge -= (s["loss"] * self.coeff).astype(np.float32)
            self.charge[:] = np.maximum(self.charge, 0.0)

    def _curtail(self, new_p, act_curtail):
        """_aux_handle_curtailment_without_limit (baseEnv.py:2956-2982): renewable generators are capped at limit * pmax, the change
        of the curtailed total against the previous step goes to the right-hand side of the redispatch projection"""
        ren = self.lim.get("renewable")
        if ren is None:
            return new_p
        has_act = act_curtail is not None and (np.asarray(act_curtail) != -1.0).any()
        if has_act or (np.abs(self.limit - 1.0) >= 1e-7).any():
            if has_act:
                a = np.asarray(act_curtail, np.float32)
                sel = (a != -1.0) & ren
                self.limit[sel] = a[sel]
            gc = np.abs(self.limit - 1.0) >= 1e-7
            before = new_p.copy()
            new_p = new_p.copy()
            new_p[gc] = np.minimum((self.lim["pmax"][gc] * self.limit[gc]).astype(np.float32), new_p[gc])
            tmp = float(np.float32(new_p[gc].sum() - before[gc].sum()))
            self.sum_curt = tmp - self.sum_curt_prev
            self.sum_curt_prev = tmp
        else:
            self.sum_curt = -self.sum_curt_prev
            self.sum_curt_prev = 0.0
        return new_p

    def step(self, new_p, act_redisp=None, act_storage=None, act_curtail=None):
        """-> (ok, generator set-points float32 [n_gen] = (curtailed) chronics + actual dispatch, storage power float32 [n_storage])"""
        from .redispatch_oracle import compute_dispatch, compute_dispatch_exact
        if self.exact:
            compute_dispatch = compute_dispatch_exact
        new_p = np.asarray(new_p, np.float32)
        if self.sto is not None:
            self._compute_storage(np.zeros(len(self.power), np.float32) if act_storage is None else act_storage)
        new_p = self._curtail(new_p, act_curtail)
        if act_redisp is not None and (np.asarray(act_redisp) != 0).any():
            act = np.asarray(act_redisp, np.float32)
            is_red = np.abs(act) > 1e-7
            self.target[self.already] += act[self.already]
            first_mod = (~self.already) & is_red
            self.target[first_mod] = self.actual[first_mod] + act[first_mod]
            self.already[is_red] = True
        # _prepare_redisp (baseEnv.py:2117-2186): a target dispatch beyond pmax - pmin (or below pmin - pmax) can never be met -> the
        # action is ILLEGAL: it is taken back out of the target, and BaseEnv.step replaces the whole action by do-nothing
        # (:3189-3212): the storage charge goes back to the previous step's, the storage amount is withdrawn and the losses are
        # applied again -- the storage POWER already handed to the backend stays (:3829-3831), as in the reference
        self.illegal = False
        act_v = np.zeros(len(self.target), np.float32) if act_redisp is None else np.asarray(act_redisp, np.float32)
        skip = (np.abs(act_v) <= 1e-7).all() and (np.abs(self.target) <= 1e-7).all() and (np.abs(self.actual) <= 1e-7).all()
        span = (self.lim["pmax"] - self.lim["pmin"]).astype(np.float32)
        if not skip and ((self.target > span).any() or (self.target < -span).any()):
            self.target -= act_v
            self.illegal = True
            if self.sto is not None:
                self.charge[:] = self.charge_prev
                self.amount -= self.amount_prev
                if self.loss_on:
                    self.charge -= (self.sto["loss"] * self.coeff).astype(np.float32)
                    self.charge[:] = np.maximum(self.charge, 0.0)
        tol = self.lim["tol_poly"]
        ok = True
        if self.fresh and not self.prev_p.any():
            self.prev_p[:] = new_p
        self.fresh = False
        if (abs(float(self.actual.sum())) >= tol or float(np.abs(self.actual - self.target).max()) >= tol or abs(self.amount) >= tol or
                abs(self.sum_curt) >= tol):
            ok, after = compute_dispatch(new_p.astype(np.float64), self.prev_p.astype(np.float64), self.actual.astype(np.float64),
                                         self.target.astype(np.float64), self.already.copy(), self.amount, self.sum_curt, 0.0, self.lim, first=False)
            if ok:
                self.actual[:] = after.astype(np.float32)
        gen = (new_p + self.actual).astype(np.float32)
        if ok:
            self.prev_p[:] = gen
        return ok, gen, self.power.copy()

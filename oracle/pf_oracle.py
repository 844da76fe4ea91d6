"""CPU ORACLE (numpy twin) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; nothing under ``grid2op_amd/`` does.  It is a float64 restatement of the arithmetic that the
reference delegates to the third-party package **pandapower** (pinned ``pandapower>=3.1.1`` in the
reference's ``pyproject.toml:15``; NOT vendored under /root/reference and not installable here), as
called from ``grid2op/Backend/pandaPowerBackend.py``:

* ``pp.runpp(net, check_connectivity=False, init="dc", max_iteration=10, distributed_slack=False)``
  (pandaPowerBackend.py:1097-1105)  -> `solve(..., is_dc=False)`
* ``pp.rundcpp(net, check_connectivity=True, init="flat")`` (pandaPowerBackend.py:1090)
  -> `solve(..., is_dc=True)`
* the result read-back of ``_fetch_data_pf_converged`` (pandaPowerBackend.py:1122-1218) and of the
  ``_gens_info / _loads_info / _storages_info / shunt_info`` helpers (:1526-1647)

The published algorithm restated here is pandapower's ``pd2ppc -> makeYbus -> newtonpf -> pfsoln``
chain (MATPOWER/PYPOWER lineage); formulas are listed in SURVEY.md section 8 row A4'.

PARITY PINNING: this oracle is checked (tests/test_oracle_golden.py) against the pandapower results
that the reference ships inside its own fixtures: the ``res_bus/res_line/res_trafo/res_gen/res_shunt``
tables embedded in ``grid2op/data/{rte_case5_example,l2rpn_neurips_2020_track1,l2rpn_wcci_2022_dev,
rte_case118_example,l2rpn_icaps_2021,...}/grid.json`` (extracted to ``tests/golden/*.res.npz`` by
``tests/golden/make_fixtures.py``) and the known-answer vectors of
``grid2op/tests/BaseBackendTest.py:262-313, 1584-1607``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

SQRT3 = np.sqrt(3.0)


@dataclass
class LaneState:
    """Dynamic state of ONE grid instance ("lane"): what ``apply_action`` mutates
    (pandaPowerBackend.py:902-975)."""
    topo: np.ndarray            # int32[dim_topo], local bus 1..n_busbar, -1 = disconnected
    shunt_bus: np.ndarray       # int32[n_shunt]
    gen_p: np.ndarray           # MW
    gen_vm: np.ndarray          # pu
    load_p: np.ndarray
    load_q: np.ndarray
    storage_p: np.ndarray
    storage_q: np.ndarray
    shunt_p: np.ndarray
    shunt_q: np.ndarray

    @classmethod
    def from_model(cls, m) -> "LaneState":
        return cls(topo=m.initial_topo_vect(), shunt_bus=m.initial_shunt_bus(),
                   gen_p=m.gen_p0.copy(), gen_vm=m.gen_vm0.copy(),
                   load_p=m.load_p0.copy(), load_q=m.load_q0.copy(),
                   storage_p=m.storage_p0.copy(), storage_q=m.storage_q0.copy(),
                   shunt_p=m.shunt_p0.copy(), shunt_q=m.shunt_q0.copy())

    def copy(self) -> "LaneState":
        return LaneState(**{k: v.copy() for k, v in self.__dict__.items()})


@dataclass
class PFResult:
    converged: bool = False
    n_iter: int = 0
    reason: str = ""
    # per global bus (n_sub * n_busbar); NaN where the bus is not active
    bus_vm: Optional[np.ndarray] = None
    bus_va: Optional[np.ndarray] = None       # degrees
    p_or: Optional[np.ndarray] = None
    q_or: Optional[np.ndarray] = None
    v_or: Optional[np.ndarray] = None         # kV
    a_or: Optional[np.ndarray] = None         # A
    theta_or: Optional[np.ndarray] = None     # deg
    p_ex: Optional[np.ndarray] = None
    q_ex: Optional[np.ndarray] = None
    v_ex: Optional[np.ndarray] = None
    a_ex: Optional[np.ndarray] = None
    theta_ex: Optional[np.ndarray] = None
    gen_p: Optional[np.ndarray] = None
    gen_q: Optional[np.ndarray] = None
    gen_v: Optional[np.ndarray] = None
    gen_theta: Optional[np.ndarray] = None
    load_p: Optional[np.ndarray] = None
    load_q: Optional[np.ndarray] = None
    load_v: Optional[np.ndarray] = None
    load_theta: Optional[np.ndarray] = None
    storage_p: Optional[np.ndarray] = None
    storage_q: Optional[np.ndarray] = None
    storage_v: Optional[np.ndarray] = None
    storage_theta: Optional[np.ndarray] = None
    shunt_p: Optional[np.ndarray] = None
    shunt_q: Optional[np.ndarray] = None
    shunt_v: Optional[np.ndarray] = None
    shunt_bus: Optional[np.ndarray] = None
    topo_vect: Optional[np.ndarray] = None
    line_status: Optional[np.ndarray] = None


def _global_bus(sub, local, n_sub):
    """Space/GridObjects.py:4683-4745 (``local_bus_to_global``): -1 stays -1."""
    local = np.asarray(local)
    return np.where(local >= 1, np.asarray(sub) + (local - 1) * n_sub, -1)


def element_buses(m, st: LaneState):
    """Global bus of every element (-1 = out of service) + line status.  A line is in service iff
    BOTH of its ends are connected (_backendAction.py keeps the two ends consistent)."""
    n_sub = m.n_sub
    lor_l = st.topo[m.line_or_pos_topo_vect]
    lex_l = st.topo[m.line_ex_pos_topo_vect]
    status = (lor_l >= 1) & (lex_l >= 1)
    lor = np.where(status, _global_bus(m.line_or_sub, lor_l, n_sub), -1)
    lex = np.where(status, _global_bus(m.line_ex_sub, lex_l, n_sub), -1)
    gen = _global_bus(m.gen_sub, st.topo[m.gen_pos_topo_vect], n_sub)
    load = _global_bus(m.load_sub, st.topo[m.load_pos_topo_vect], n_sub)
    sto = _global_bus(m.storage_sub, st.topo[m.storage_pos_topo_vect], n_sub) if m.n_storage else np.zeros(0, np.int64)
    sh = _global_bus(m.shunt_sub, st.shunt_bus, n_sub) if m.n_shunt else np.zeros(0, np.int64)
    return lor, lex, status, gen, load, sto, sh


def build_ybus(m, lor, lex, status, sh, st, nb_tot):
    """Dense complex Ybus over ALL global buses (pandapower ``makeYbus``; SURVEY.md A4')."""
    Y = np.zeros((nb_tot, nb_tot), dtype=np.complex128)
    for l in np.nonzero(status)[0]:
        f, t = lor[l], lex[l]
        Y[f, f] += m.br_yff[l]
        Y[f, t] += m.br_yft[l]
        Y[t, f] += m.br_ytf[l]
        Y[t, t] += m.br_ytt[l]
    for s in range(m.n_shunt):
        if sh[s] >= 0:
            # Ysh = conj(p + jq) * step * (vn_bus/vn_shunt)^2 / sn_mva
            Y[sh[s], sh[s]] += (st.shunt_p[s] - 1j * st.shunt_q[s]) * m.shunt_fact[s] / m.sn_mva
    return Y


def solve(m, st: LaneState, is_dc: bool = False, max_iter: int = 10, tol_mva: float = 1e-8,
          n_busbar: int = 2) -> PFResult:
    n_sub = m.n_sub
    nb_tot = n_sub * n_busbar
    res = PFResult()
    lor, lex, status, gbus, lbus, sbus, shbus = element_buses(m, st)

    # --- active buses (``_BackendAction._get_active_bus`` _backendAction.py:1519-1530) ---------------
    active = np.zeros(nb_tot, dtype=bool)
    for arr in (lor, lex, gbus, lbus, sbus, shbus):
        active[arr[arr >= 0]] = True

    res.line_status = status.copy()
    res.topo_vect = _topo_vect(m, st, status)

    def fail(reason):
        res.converged = False
        res.reason = reason
        _fill_nan(m, res, nb_tot)
        return res

    # --- bus types -------------------------------------------------------------------------------------
    gen_on = gbus >= 0
    slack_on = gen_on & m.gen_slack
    if not slack_on.any():
        return fail("no in-service slack generator")
    ref = np.unique(gbus[slack_on])
    is_ref = np.zeros(nb_tot, bool)
    is_ref[ref] = True
    is_pv = np.zeros(nb_tot, bool)
    is_pv[gbus[gen_on]] = True
    is_pv &= ~is_ref
    act_idx = np.nonzero(active)[0]

    # --- connectivity: every active bus must be reachable from a reference bus.
    # AC: pandapower is run with check_connectivity=False, an island makes the Jacobian singular ->
    #     NaN -> LoadflowNotConverged (pandaPowerBackend.py:1106-1120, 1241-1244);
    # DC: check_connectivity=True removes the island, its buses get NaN angles -> "Isolated bus"
    #     (pandaPowerBackend.py:1241-1244).  Both end in (False, BackendError).
    label = np.full(nb_tot, -1)
    label[ref] = 0
    changed = True
    while changed:
        changed = False
        for l in np.nonzero(status)[0]:
            f, t = lor[l], lex[l]
            if (label[f] == 0) != (label[t] == 0):
                label[f] = label[t] = 0
                changed = True
    if (label[act_idx] != 0).any():
        return fail("islanded grid")

    Y = build_ybus(m, lor, lex, status, shbus, st, nb_tot)

    # --- bus power injections (pu) and voltage setpoints ------------------------------------------------
    P = np.zeros(nb_tot)
    Q = np.zeros(nb_tot)
    vset = np.ones(nb_tot)
    for g in np.nonzero(gen_on)[0]:
        if not m.gen_slack[g]:
            P[gbus[g]] += st.gen_p[g] / m.sn_mva
        vset[gbus[g]] = st.gen_vm[g]          # last in-service generator of the bus wins
    Pd = np.zeros(nb_tot)
    Qd = np.zeros(nb_tot)
    for i in np.nonzero(lbus >= 0)[0]:
        Pd[lbus[i]] += st.load_p[i]
        Qd[lbus[i]] += st.load_q[i]
    for i in np.nonzero(sbus >= 0)[0]:
        Pd[sbus[i]] += st.storage_p[i]
        Qd[sbus[i]] += st.storage_q[i]
    P -= Pd / m.sn_mva
    Q -= Qd / m.sn_mva

    pvpq = np.nonzero(active & ~is_ref)[0]
    pq = np.nonzero(active & ~is_ref & ~is_pv)[0]

    # --- DC solve (init="dc" of runpp, or the whole of rundcpp) ---------------------------------------------
    B = np.zeros((nb_tot, nb_tot))
    for l in np.nonzero(status)[0]:
        f, t = lor[l], lex[l]
        b = m.br_bdc[l]
        B[f, f] += b
        B[t, t] += b
        B[f, t] -= b
        B[t, f] -= b
    Gs = np.zeros(nb_tot)
    for s in range(m.n_shunt):
        if shbus[s] >= 0:
            Gs[shbus[s]] += st.shunt_p[s] * m.shunt_fact[s] / m.sn_mva
    va = np.zeros(nb_tot)
    if len(pvpq):
        rhs = P[pvpq] - Gs[pvpq]
        try:
            va[pvpq] = np.linalg.solve(B[np.ix_(pvpq, pvpq)], rhs)
        except np.linalg.LinAlgError:
            return fail("singular DC matrix")
    if not np.all(np.isfinite(va)):
        return fail("non finite DC solution")

    if is_dc:
        vm = np.ones(nb_tot)
        vm[is_ref | is_pv] = vset[is_ref | is_pv]
        res.n_iter = 0
        _results_dc(m, st, res, va, vm, B, active, lor, lex, status, gbus, lbus, sbus, shbus, P, Gs, is_ref, nb_tot)
        res.converged = True
        return res

    # --- Newton-Raphson (pypower ``newtonpf`` form) ------------------------------------------------------------
    vm = np.ones(nb_tot)
    vm[is_ref | is_pv] = vset[is_ref | is_pv]
    V = vm * np.exp(1j * va)
    Sbus = P + 1j * Q
    tol = tol_mva / m.sn_mva
    npvpq, npq = len(pvpq), len(pq)

    def mismatch(V):
        mis = V * np.conj(Y @ V) - Sbus
        return np.concatenate((mis[pvpq].real, mis[pq].imag))

    F = mismatch(V)
    it = 0
    converged = (np.max(np.abs(F)) < tol) if len(F) else True
    while not converged and it < max_iter:
        it += 1
        Ibus = Y @ V
        diagV = np.diag(V)
        diagVn = np.diag(V / np.abs(V))
        dS_dVm = diagV @ np.conj(Y @ diagVn) + np.conj(np.diag(Ibus)) @ diagVn
        dS_dVa = 1j * diagV @ np.conj(np.diag(Ibus) - Y @ diagV)
        J = np.block([[dS_dVa[np.ix_(pvpq, pvpq)].real, dS_dVm[np.ix_(pvpq, pq)].real],
                      [dS_dVa[np.ix_(pq, pvpq)].imag, dS_dVm[np.ix_(pq, pq)].imag]])
        try:
            dx = -np.linalg.solve(J, F)
        except np.linalg.LinAlgError:
            return fail("singular Jacobian")
        if not np.all(np.isfinite(dx)):
            return fail("non finite Newton step")
        va_ = np.angle(V)
        vm_ = np.abs(V)
        va_[pvpq] += dx[:npvpq]
        vm_[pq] += dx[npvpq:]
        V = vm_ * np.exp(1j * va_)
        F = mismatch(V)
        converged = np.max(np.abs(F)) < tol
    res.n_iter = it
    if not converged:
        return fail(f"no convergence in {max_iter} iterations")
    _results_ac(m, st, res, V, Y, active, lor, lex, status, gbus, lbus, sbus, shbus, is_ref, nb_tot)
    res.converged = True
    return res


# ------------------------------------------------------------------------------------------------------------------
def _topo_vect(m, st, status):
    """``_get_topo_vect`` pandaPowerBackend.py:1489-1524: both ends of an out-of-service line read -1."""
    tv = st.topo.astype(np.int32).copy()
    tv[tv < 1] = -1
    tv[m.line_or_pos_topo_vect[~status]] = -1
    tv[m.line_ex_pos_topo_vect[~status]] = -1
    return tv


def _fill_nan(m, res, nb_tot):
    """``_reset_all_nan`` pandaPowerBackend.py:1257-1287."""
    nan = lambda n: np.full(n, np.nan)
    for k in ("p_or", "q_or", "v_or", "a_or", "theta_or", "p_ex", "q_ex", "v_ex", "a_ex", "theta_ex"):
        setattr(res, k, nan(m.n_line))
    for k in ("gen_p", "gen_q", "gen_v", "gen_theta"):
        setattr(res, k, nan(m.n_gen))
    for k in ("load_p", "load_q", "load_v", "load_theta"):
        setattr(res, k, nan(m.n_load))
    for k in ("storage_p", "storage_q", "storage_v", "storage_theta"):
        setattr(res, k, nan(m.n_storage))
    for k in ("shunt_p", "shunt_q", "shunt_v"):
        setattr(res, k, nan(m.n_shunt))
    res.shunt_bus = np.full(m.n_shunt, -1, np.int32)
    res.bus_vm = nan(nb_tot)
    res.bus_va = nan(nb_tot)
    res.topo_vect = np.full(m.dim_topo, -1, np.int32)
    res.line_status = np.zeros(m.n_line, bool)


def _bus_lookup(arr_bus, values, fill=0.0):
    out = np.full(len(arr_bus), fill, dtype=np.float64)
    ok = arr_bus >= 0
    out[ok] = values[arr_bus[ok]]
    return out


def _results_ac(m, st, res, V, Y, active, lor, lex, status, gbus, lbus, sbus, shbus, is_ref, nb_tot):
    sn = m.sn_mva
    vm = np.abs(V)
    va = np.degrees(np.angle(V))
    res.bus_vm = np.where(active, vm, np.nan)
    res.bus_va = np.where(active, va, np.nan)
    n_sub = m.n_sub
    # ---- branches (pandapower ``_get_branch_flows``; SURVEY.md A4' "Results") -------------------------------
    nl = m.n_line
    for k in ("p_or", "q_or", "v_or", "a_or", "theta_or", "p_ex", "q_ex", "v_ex", "a_ex", "theta_ex"):
        setattr(res, k, np.zeros(nl))
    on = np.nonzero(status)[0]
    f, t = lor[on], lex[on]
    If = m.br_yff[on] * V[f] + m.br_yft[on] * V[t]
    It = m.br_ytf[on] * V[f] + m.br_ytt[on] * V[t]
    Sf = V[f] * np.conj(If) * sn
    St = V[t] * np.conj(It) * sn
    vn_f = m.sub_vn_kv[m.line_or_sub[on]]
    vn_t = m.sub_vn_kv[m.line_ex_sub[on]]
    res.p_or[on], res.q_or[on] = Sf.real, Sf.imag
    res.p_ex[on], res.q_ex[on] = St.real, St.imag
    res.a_or[on] = np.abs(Sf) / (SQRT3 * vm[f] * vn_f) * 1000.0
    res.a_ex[on] = np.abs(St) / (SQRT3 * vm[t] * vn_t) * 1000.0
    res.v_or[on] = vm[f] * vn_f     # kV of the ORIGINAL substation (pandaPowerBackend.py:1182-1183)
    res.v_ex[on] = vm[t] * vn_t
    res.theta_or[on] = va[f]
    res.theta_ex[on] = va[t]
    # ---- loads / storages (pandaPowerBackend.py:1549-1564, 1621-1647, 1203-1207) ------------------------------
    l_on = lbus >= 0
    res.load_p = np.where(l_on, st.load_p, 0.0)
    res.load_q = np.where(l_on, st.load_q, 0.0)
    res.load_v = _bus_lookup(lbus, vm) * m.sub_vn_kv[m.load_sub]
    res.load_theta = _bus_lookup(lbus, va)
    s_on = sbus >= 0
    res.storage_p = np.where(s_on, st.storage_p, 0.0)
    res.storage_q = np.where(s_on, st.storage_q, 0.0)
    res.storage_v = _bus_lookup(sbus, vm) * (m.sub_vn_kv[m.storage_sub] if m.n_storage else 1.0)
    res.storage_theta = _bus_lookup(sbus, va)
    # ---- shunts (pandaPowerBackend.py:1596-1612) --------------------------------------------------------------
    sh_on = shbus >= 0
    vsh = _bus_lookup(shbus, vm)
    res.shunt_p = np.where(sh_on, st.shunt_p * m.shunt_fact * vsh ** 2, 0.0)
    res.shunt_q = np.where(sh_on, st.shunt_q * m.shunt_fact * vsh ** 2, 0.0)
    res.shunt_v = vsh * (m.shunt_vn_kv if m.n_shunt else 1.0)
    res.shunt_bus = np.where(sh_on, st.shunt_bus, -1).astype(np.int32)
    # ---- generators (pypower ``pfsoln``: Q split in proportion to the reactive range, slack P = bus balance)
    g_on = gbus >= 0
    Sinj = V * np.conj(Y @ V) * sn                      # MW / MVAr injected into the network at each bus
    Pd = np.zeros(nb_tot)
    Qd = np.zeros(nb_tot)
    np.add.at(Pd, lbus[l_on], st.load_p[l_on])
    np.add.at(Qd, lbus[l_on], st.load_q[l_on])
    if m.n_storage:
        np.add.at(Pd, sbus[s_on], st.storage_p[s_on])
        np.add.at(Qd, sbus[s_on], st.storage_q[s_on])
    res.gen_p = np.where(g_on, st.gen_p, 0.0)
    res.gen_q = np.zeros(m.n_gen)
    qtot = Sinj.imag + Qd
    EPS = np.finfo(np.float64).eps
    for b in np.unique(gbus[g_on]):
        gens = np.nonzero(gbus == b)[0]
        if len(gens) == 1:
            res.gen_q[gens[0]] = qtot[b]
        else:
            qmin, qmax = m.gen_min_q[gens], m.gen_max_q[gens]
            qmin_t, qmax_t = qmin.sum(), qmax.sum()
            if qmin_t == qmax_t:
                res.gen_q[gens] = qtot[b] / len(gens)
            else:
                res.gen_q[gens] = qmin + (qtot[b] - qmin_t) / (qmax_t - qmin_t + EPS) * (qmax - qmin)
        if is_ref[b]:
            slacks = [g for g in gens if m.gen_slack[g]]
            others = [g for g in gens if not m.gen_slack[g]]
            p_bus = Sinj.real[b] + Pd[b]
            p_slack = p_bus - sum(st.gen_p[g] for g in others)
            for g in slacks:
                res.gen_p[g] = p_slack / len(slacks)
    res.gen_v = _bus_lookup(gbus, vm) * m.sub_vn_kv[m.gen_sub]
    res.gen_theta = _bus_lookup(gbus, va)


def _results_dc(m, st, res, va_rad, vm, B, active, lor, lex, status, gbus, lbus, sbus, shbus, P, Gs, is_ref, nb_tot):
    """DC read-back: ``q* = 0`` (pandaPowerBackend.py:1212-1218); branch voltages are NaN -> 0 in the
    reference (:1161-1167); see Appendix A item 9 of SURVEY.md."""
    sn = m.sn_mva
    va = np.degrees(va_rad)
    res.bus_vm = np.where(active, vm, np.nan)
    res.bus_va = np.where(active, va, np.nan)
    nl = m.n_line
    for k in ("p_or", "q_or", "v_or", "a_or", "theta_or", "p_ex", "q_ex", "v_ex", "a_ex", "theta_ex"):
        setattr(res, k, np.zeros(nl))
    on = np.nonzero(status)[0]
    f, t = lor[on], lex[on]
    pf = (va_rad[f] - va_rad[t]) * m.br_bdc[on] * sn
    res.p_or[on] = pf
    res.p_ex[on] = -pf
    vn_f = m.sub_vn_kv[m.line_or_sub[on]]
    vn_t = m.sub_vn_kv[m.line_ex_sub[on]]
    res.a_or[on] = np.abs(pf) / (SQRT3 * vm[f] * vn_f) * 1000.0
    res.a_ex[on] = np.abs(pf) / (SQRT3 * vm[t] * vn_t) * 1000.0
    res.v_or[on] = vm[f] * vn_f
    res.v_ex[on] = vm[t] * vn_t
    res.theta_or[on] = va[f]
    res.theta_ex[on] = va[t]
    l_on = lbus >= 0
    res.load_p = np.where(l_on, st.load_p, 0.0)
    res.load_q = np.zeros(m.n_load)
    res.load_v = _bus_lookup(lbus, vm) * m.sub_vn_kv[m.load_sub]
    res.load_theta = _bus_lookup(lbus, va)
    s_on = sbus >= 0
    res.storage_p = np.where(s_on, st.storage_p, 0.0)
    res.storage_q = np.zeros(m.n_storage)
    res.storage_v = _bus_lookup(sbus, vm) * (m.sub_vn_kv[m.storage_sub] if m.n_storage else 1.0)
    res.storage_theta = _bus_lookup(sbus, va)
    sh_on = shbus >= 0
    vsh = _bus_lookup(shbus, vm)
    res.shunt_p = np.where(sh_on, st.shunt_p * m.shunt_fact * vsh ** 2, 0.0)
    res.shunt_q = np.zeros(m.n_shunt)
    res.shunt_v = vsh * (m.shunt_vn_kv if m.n_shunt else 1.0)
    res.shunt_bus = np.where(sh_on, st.shunt_bus, -1).astype(np.int32)
    g_on = gbus >= 0
    res.gen_p = np.where(g_on, st.gen_p, 0.0)
    res.gen_q = np.zeros(m.n_gen)
    Pinj = (B @ va_rad) * sn                              # MW leaving each bus through the branches
    Pd = np.zeros(nb_tot)
    np.add.at(Pd, lbus[l_on], st.load_p[l_on])
    if m.n_storage:
        np.add.at(Pd, sbus[s_on], st.storage_p[s_on])
    for b in np.unique(gbus[g_on]):
        if is_ref[b]:
            gens = np.nonzero(gbus == b)[0]
            slacks = [g for g in gens if m.gen_slack[g]]
            others = [g for g in gens if not m.gen_slack[g]]
            p_bus = Pinj[b] + Pd[b] + Gs[b] * sn
            p_slack = p_bus - sum(st.gen_p[g] for g in others)
            for g in slacks:
                res.gen_p[g] = p_slack / len(slacks)
    res.gen_v = _bus_lookup(gbus, vm) * m.sub_vn_kv[m.gen_sub]
    res.gen_theta = _bus_lookup(gbus, va)


def ptdf(m, st: LaneState, n_busbar: int = 2):
    """DC sensitivity matrix of the topology of ``st`` (test oracle of gpf_ptdf_build): PTDF [n_line, n_sub*n_busbar],
    origin-side MW per MW injected at the bus, slack-referenced.  Same DC model as ``solve(..., is_dc=True)``
    (pp.rundcpp, pandaPowerBackend.py:1090): flows = PTDF @ dc_bus_injection."""
    nb_tot = m.n_sub * n_busbar
    lor, lex, status, gbus, lbus, sbus, shbus = element_buses(m, st)
    active = np.zeros(nb_tot, dtype=bool)
    for arr in (lor, lex, gbus, lbus, sbus, shbus):
        active[arr[arr >= 0]] = True
    slack_on = (gbus >= 0) & m.gen_slack
    if not slack_on.any():
        raise ValueError("no in-service slack generator")
    is_ref = np.zeros(nb_tot, bool)
    is_ref[np.unique(gbus[slack_on])] = True
    B = np.zeros((nb_tot, nb_tot))
    for l in np.nonzero(status)[0]:
        f, t = lor[l], lex[l]
        if f == t:
            continue
        b = m.br_bdc[l]
        B[f, f] += b
        B[t, t] += b
        B[f, t] -= b
        B[t, f] -= b
    free = np.nonzero(active & ~is_ref)[0]
    X = np.zeros((nb_tot, nb_tot))
    X[np.ix_(free, free)] = np.linalg.inv(B[np.ix_(free, free)])
    out = np.zeros((m.n_line, nb_tot))
    for l in np.nonzero(status)[0]:
        f, t = lor[l], lex[l]
        if f != t:
            out[l] = m.br_bdc[l] * (X[f] - X[t])
    return out


def dc_bus_injection(m, st: LaneState, n_busbar: int = 2):
    """Active-power injection per bus (MW) as the DC power flow sees it: generators except the slack, minus loads,
    storages and the shunt conductances at 1 pu."""
    nb_tot = m.n_sub * n_busbar
    _, _, _, gbus, lbus, sbus, shbus = element_buses(m, st)
    P = np.zeros(nb_tot)
    for g in np.nonzero(gbus >= 0)[0]:
        if not m.gen_slack[g]:
            P[gbus[g]] += st.gen_p[g]
    for i in np.nonzero(lbus >= 0)[0]:
        P[lbus[i]] -= st.load_p[i]
    for i in np.nonzero(sbus >= 0)[0]:
        P[sbus[i]] -= st.storage_p[i]
    for s in range(m.n_shunt):
        if shbus[s] >= 0:
            P[shbus[s]] -= st.shunt_p[s] * m.shunt_fact[s]
    return P


def dc_n1_worst_loading(m, st: LaneState, cap_mw=None, n_busbar: int = 2):
    """Test oracle of gpf_lodf_screen: for every single-line outage k, re-solve the DC power flow with line k forced off
    (what ``N1Reward`` / ``obs.simulate`` do, one contingency at a time) and return max_l |p_or[l]| / cap_mw[l]
    (MW if ``cap_mw`` is None); +inf when the outage makes the DC power flow fail (islanding)."""
    out = np.empty(m.n_line)
    for k in range(m.n_line):
        s2 = LaneState.from_model(m)
        s2.topo = st.topo.copy()
        s2.shunt_bus = st.shunt_bus.copy()
        s2.load_p, s2.load_q, s2.gen_p, s2.gen_vm = st.load_p, st.load_q, st.gen_p, st.gen_vm
        s2.storage_p, s2.storage_q = st.storage_p, st.storage_q
        s2.shunt_p, s2.shunt_q = st.shunt_p, st.shunt_q
        s2.topo[m.line_or_pos_topo_vect[k]] = -1
        s2.topo[m.line_ex_pos_topo_vect[k]] = -1
        r = solve(m, s2, is_dc=True, n_busbar=n_busbar)
        if not r.converged:
            out[k] = np.inf
        else:
            f = np.abs(r.p_or)
            out[k] = (f if cap_mw is None else f / np.asarray(cap_mw)).max()
    return out

"""Static grid description (`GridModel`) shared by every lane of the batched engine.

This is the host-side restatement of what ``PandaPowerBackend.load_grid`` +
``_init_private_attrs`` (grid2op/Backend/pandaPowerBackend.py:356-617, 670-874) derive from a
pandapower-JSON grid, plus the per-unit branch model that pandapower's ``pd2ppc`` would build
(formulas: SURVEY.md section 8, row A4').  Everything is plain numpy so it can be (a) handed to the
C ABI (``include/gridpf.h``: ``gpf_grid_desc``), (b) saved/loaded as an ``.npz`` fixture that travels
to the GPU box (where neither the reference nor its data folders exist).

Conventions reproduced from the reference
-----------------------------------------
* ``n_line`` = powerlines first, then transformers; origin = ``from_bus``/``hv_bus``,
  extremity = ``to_bus``/``lv_bus``                     (pandaPowerBackend.py:462-471, 481, 691-723)
* substation id of an element = *label* of the bus it is attached to in the file;
  rows of every other table keep FILE ORDER            (pandaPowerBackend.py:691-751)
* global bus id = ``sub + (local_bus-1)*n_sub``         (Space/GridObjects.py:4683-4745)
* position of the elements in a substation: lines (or, ex) -> trafos -> gens -> loads -> storages
                                                        (pandaPowerBackend.py:691-751)
* ``topo_vect`` positions = ``cumsum(sub_info)[sub-1] + to_sub_pos``
                                                        (Space/GridObjects.py:1409-1477)
* no ``slack=True`` generator but an ``ext_grid`` row -> one generator is APPENDED on the ext_grid
  bus and becomes the slack                             (pandaPowerBackend.py:394-451)
* thermal limits: ``1000*max_i_ka`` (lines), ``1000*sn_mva/(sqrt(3)*vn_hv_kv)`` (trafos)
                                                        (pandaPowerBackend.py:806-813)
* names: table ``name`` column if it has no null, else ``{from}_{to}_{i}``, trafos
  ``{a}_{b}_{i+n_powerline}`` with (a, b) the hv/lv bus ids sorted AS STRINGS, ``gen_{bus}_{i}``,
  ``load_{bus}_{i}``, ``storage_{bus}_{i}``, ``shunt_{bus}_{i}``, ``sub_{label}``
                                                        (pandaPowerBackend.py:484-553, 762)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, fields
from typing import Optional

import numpy as np

from .pp_json import read_pandapower_json, Table

__all__ = ["GridModel", "load_grid_model"]

_I32 = np.int32
_F64 = np.float64


@dataclass
class GridModel:
    # ---- sizes -------------------------------------------------------------------------------
    n_sub: int = 0
    n_line: int = 0          # powerlines + trafos
    n_powerline: int = 0     # powerlines only (trafos come after)
    n_gen: int = 0
    n_load: int = 0
    n_storage: int = 0
    n_shunt: int = 0
    dim_topo: int = 0
    sn_mva: float = 1.0
    f_hz: float = 50.0
    slack_added: bool = False     # True when the slack generator was appended (legacy ext_grid files)
    # ---- buses / substations -------------------------------------------------------------------
    sub_vn_kv: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    sub_info: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    # ---- branches (lines then trafos), per unit on (sn_mva, vn of the substation) --------------
    line_or_sub: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    line_ex_sub: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    br_yff: np.ndarray = field(default_factory=lambda: np.zeros(0, np.complex128))
    br_yft: np.ndarray = field(default_factory=lambda: np.zeros(0, np.complex128))
    br_ytf: np.ndarray = field(default_factory=lambda: np.zeros(0, np.complex128))
    br_ytt: np.ndarray = field(default_factory=lambda: np.zeros(0, np.complex128))
    br_bdc: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))   # 1/(x*ratio) for the DC model
    line_status0: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    thermal_limit_a: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    # ---- generators ------------------------------------------------------------------------------
    gen_sub: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    gen_p0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))      # MW
    gen_vm0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))     # pu
    gen_min_q: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))   # MVAr
    gen_max_q: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    gen_slack: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    gen_status0: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    # ---- loads -----------------------------------------------------------------------------------
    load_sub: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    load_p0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    load_q0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    load_status0: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    # ---- storage units (positive p = charging = consumption) --------------------------------------
    storage_sub: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    storage_p0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    storage_q0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    storage_status0: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    # ---- shunts ------------------------------------------------------------------------------------
    shunt_sub: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    shunt_p0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))    # MW at v = 1 pu (raw table value)
    shunt_q0: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))
    shunt_fact: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64))  # step*(vn_bus/vn_shunt)^2
    shunt_vn_kv: np.ndarray = field(default_factory=lambda: np.zeros(0, _F64)) # = bus vn_kv (reference _sh_vnkv)
    shunt_status0: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    # ---- topology vector layout ----------------------------------------------------------------------
    line_or_to_sub_pos: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    line_ex_to_sub_pos: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    gen_to_sub_pos: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    load_to_sub_pos: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    storage_to_sub_pos: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    line_or_pos_topo_vect: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    line_ex_pos_topo_vect: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    gen_pos_topo_vect: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    load_pos_topo_vect: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    storage_pos_topo_vect: np.ndarray = field(default_factory=lambda: np.zeros(0, _I32))
    # ---- names (object arrays of str) -------------------------------------------------------------------
    name_line: np.ndarray = field(default_factory=lambda: np.zeros(0, object))
    name_gen: np.ndarray = field(default_factory=lambda: np.zeros(0, object))
    name_load: np.ndarray = field(default_factory=lambda: np.zeros(0, object))
    name_storage: np.ndarray = field(default_factory=lambda: np.zeros(0, object))
    name_shunt: np.ndarray = field(default_factory=lambda: np.zeros(0, object))
    name_sub: np.ndarray = field(default_factory=lambda: np.zeros(0, object))

    # ------------------------------------------------------------------------------------------------
    def initial_topo_vect(self) -> np.ndarray:
        """``topo_vect`` of the state stored in the file (all on busbar 1, -1 where out of service)."""
        topo = np.ones(self.dim_topo, dtype=_I32)
        topo[self.line_or_pos_topo_vect[~self.line_status0]] = -1
        topo[self.line_ex_pos_topo_vect[~self.line_status0]] = -1
        topo[self.gen_pos_topo_vect[~self.gen_status0]] = -1
        topo[self.load_pos_topo_vect[~self.load_status0]] = -1
        if self.n_storage:
            topo[self.storage_pos_topo_vect[~self.storage_status0]] = -1
        return topo

    def initial_shunt_bus(self) -> np.ndarray:
        return np.where(self.shunt_status0, 1, -1).astype(_I32)

    # ---- (de)serialisation: fixtures that travel to the GPU box ---------------------------------
    def save_npz(self, path: str) -> None:
        d = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, np.ndarray) and v.dtype == object:
                v = np.array([str(x) for x in v], dtype=np.str_)
            d[f.name] = np.asarray(v)
        np.savez_compressed(path, **d)

    @classmethod
    def load_npz(cls, path: str) -> "GridModel":
        z = np.load(path, allow_pickle=False)
        kw = {}
        for f in fields(cls):
            v = z[f.name]
            if f.type in ("int", int):
                v = int(v)
            elif f.type in ("float", float):
                v = float(v)
            elif f.type in ("bool", bool):
                v = bool(v)
            elif f.name.startswith("name_"):
                v = np.array([str(x) for x in v], dtype=object)
            kw[f.name] = v
        return cls(**kw)


# -----------------------------------------------------------------------------------------------------
def _names_or_default(tab: Table, default):
    if tab.n and tab.has_full_names():
        return np.array([str(x) for x in tab["name"]], dtype=object)
    return np.array(default, dtype=object)


def _fmt_bus(v) -> str:
    """pandapower tables store bus ids as uint32; ``"{from_bus}".format(**row)`` of an ``iterrows`` row
    prints them as ints when the row is all-int, but ``iterrows`` upcasts mixed rows to float/object.
    The reference's bundled names all read as plain ints (e.g. ``0_1_0``, ``gen_1_0``)."""
    return str(int(v))


def _line_pu(tab: Table, sub_vn_kv: np.ndarray, sn_mva: float, f_hz: float):
    """pi-model of the powerlines, per unit.  SURVEY.md A4' (validated vs embedded ``res_line``)."""
    fb = tab.i64("from_bus")
    length = tab.f64("length_km")
    parallel = tab.f64("parallel", 1.0)
    base_z = sub_vn_kv[fb] ** 2 / sn_mva
    r = tab.f64("r_ohm_per_km") * length / parallel / base_z
    x = tab.f64("x_ohm_per_km") * length / parallel / base_z
    b = 2.0 * math.pi * f_hz * tab.f64("c_nf_per_km", 0.0) * 1e-9 * length * parallel * base_z
    g = tab.f64("g_us_per_km", 0.0) * 1e-6 * length * parallel * base_z
    ys = 1.0 / (r + 1j * x)
    ysh = (g + 1j * b) / 2.0
    yff = ys + ysh
    ytt = ys + ysh
    yft = -ys
    ytf = -ys
    bdc = 1.0 / x
    return yff, yft, ytf, ytt, bdc


def _trafo_pu(tab: Table, sub_vn_kv: np.ndarray, sn_mva: float):
    """2-winding transformer, pandapower "t" model converted to pi by wye-delta.  SURVEY.md A4'
    (validated vs embedded ``res_trafo`` incl. the 4 trafos with ``i0_percent != 0`` of the
    118-bus grids)."""
    n = tab.n
    hv = tab.i64("hv_bus")
    lv = tab.i64("lv_bus")
    vn_bus_hv = sub_vn_kv[hv]
    vn_bus_lv = sub_vn_kv[lv]
    vn_hv = tab.f64("vn_hv_kv").copy()
    vn_lv = tab.f64("vn_lv_kv").copy()
    sn_t = tab.f64("sn_mva")
    parallel = tab.f64("parallel", 1.0)
    tap_pos = tab.f64("tap_pos", math.nan)
    tap_neutral = tab.f64("tap_neutral", math.nan)
    tap_step = tab.f64("tap_step_percent", math.nan)
    tap_side = tab.get("tap_side", np.array([None] * n, dtype=object))
    for i in range(n):
        side = tap_side[i]
        if not isinstance(side, str):
            continue
        if math.isnan(tap_pos[i]) or math.isnan(tap_neutral[i]) or math.isnan(tap_step[i]):
            continue
        fact = 1.0 + (tap_pos[i] - tap_neutral[i]) * tap_step[i] / 100.0
        if side == "hv":
            vn_hv[i] *= fact
        elif side == "lv":
            vn_lv[i] *= fact
    ratio = (vn_hv / vn_bus_hv) / (vn_lv / vn_bus_lv)
    vk = tab.f64("vk_percent")
    vkr = tab.f64("vkr_percent")
    lv_fact = (vn_lv / vn_bus_lv) ** 2
    z_sc = vk / 100.0 / sn_t * sn_mva * lv_fact
    r_sc = vkr / 100.0 / sn_t * sn_mva * lv_fact
    x_sc = np.sign(z_sc) * np.sqrt(np.maximum(z_sc ** 2 - r_sc ** 2, 0.0))
    z = (r_sc + 1j * x_sc) / parallel
    # magnetising branch (referred to the lv side)
    base_z_lv = vn_bus_lv ** 2 / sn_mva
    pfe = tab.f64("pfe_kw", 0.0)
    i0 = tab.f64("i0_percent", 0.0)
    vnl2 = tab.f64("vn_lv_kv") ** 2
    g_m = pfe / (1000.0 * vnl2) * base_z_lv
    b_m2 = (i0 / 100.0 * sn_t / vnl2) ** 2 - (pfe / (1000.0 * vnl2)) ** 2
    b_m = np.sqrt(np.maximum(b_m2, 0.0)) * base_z_lv * np.sign(i0)
    y_m = (g_m - 1j * b_m) / lv_fact * parallel
    ys = np.empty(n, dtype=np.complex128)
    ysh = np.zeros(n, dtype=np.complex128)
    x_series = np.empty(n, dtype=_F64)
    for i in range(n):
        if y_m[i] != 0:
            za = z[i] / 2.0
            zb = z[i] / 2.0
            zc = 1.0 / y_m[i]
            zsum = za * zb + za * zc + zb * zc
            z_series = zsum / zc
            ys[i] = 1.0 / z_series
            ysh[i] = zb / zsum
            x_series[i] = z_series.imag
        else:
            ys[i] = 1.0 / z[i]
            x_series[i] = z[i].imag
    yff = (ys + ysh) / ratio ** 2
    ytt = ys + ysh
    yft = -ys / ratio
    ytf = -ys / ratio
    bdc = 1.0 / (x_series * ratio)
    return yff, yft, ytf, ytt, bdc


def load_grid_model(path: str) -> GridModel:
    """Build the `GridModel` of a pandapower-JSON file (the ``load_grid`` half of the boundary)."""
    net = read_pandapower_json(path)
    T = net["tables"]
    bus, line, trafo, gen, load, sto, shunt, ext = (T[k] for k in
                                                    ("bus", "line", "trafo", "gen", "load", "storage", "shunt", "ext_grid"))
    # Elements pandapower includes in the power flow but this engine does not model: refuse the file instead of returning
    # converged-but-different results (the reference only warns, pandaPowerBackend.py:253-276, because pandapower itself
    # still solves them).
    unsupported = [t for t in ("sgen", "trafo3w", "impedance", "ward", "xward", "dcline", "motor", "asymmetric_load",
                               "asymmetric_sgen") if t in T and T[t].n]
    if trafo.n and (np.any(trafo.f64("shift_degree", 0.0) != 0.0) or np.any(trafo.boolean("tap_phase_shifter", False))):
        unsupported.append("trafo.shift_degree / tap_phase_shifter")
    for tn, tb in (("load", load), ("gen", gen), ("storage", sto)):
        if tb.n and np.any(tb.f64("scaling", 1.0) != 1.0):
            unsupported.append(f"{tn}.scaling != 1")
    if unsupported:
        raise ValueError(f"{path}: grid elements not modelled by the HIP engine: {', '.join(unsupported)} "
                         f"(pandapower would include them in the power flow; results would silently differ)")
    m = GridModel()
    m.sn_mva = net["sn_mva"]
    m.f_hz = net["f_hz"]
    # --- buses: substation id == bus LABEL (pandaPowerBackend.py:552-553, 691-751) -----------------
    labels = np.asarray(bus.index).astype(np.int64)
    m.n_sub = bus.n
    if sorted(labels.tolist()) != list(range(m.n_sub)):
        raise ValueError("bus labels must be a permutation of 0..n_bus-1")
    m.sub_vn_kv = np.zeros(m.n_sub, _F64)
    m.sub_vn_kv[labels] = bus.f64("vn_kv")
    m.name_sub = np.array([f"sub_{int(l)}" for l in labels], dtype=object)

    # --- branches -----------------------------------------------------------------------------------
    m.n_powerline = line.n
    m.n_line = line.n + trafo.n
    l_or = line.i64("from_bus") if line.n else np.zeros(0, np.int64)
    l_ex = line.i64("to_bus") if line.n else np.zeros(0, np.int64)
    t_or = trafo.i64("hv_bus") if trafo.n else np.zeros(0, np.int64)
    t_ex = trafo.i64("lv_bus") if trafo.n else np.zeros(0, np.int64)
    m.line_or_sub = np.concatenate((l_or, t_or)).astype(_I32)
    m.line_ex_sub = np.concatenate((l_ex, t_ex)).astype(_I32)
    parts = []
    if line.n:
        parts.append(_line_pu(line, m.sub_vn_kv, m.sn_mva, m.f_hz))
    if trafo.n:
        parts.append(_trafo_pu(trafo, m.sub_vn_kv, m.sn_mva))
    if parts:
        m.br_yff, m.br_yft, m.br_ytf, m.br_ytt, m.br_bdc = (np.concatenate([p[k] for p in parts]) for k in range(5))
    m.line_status0 = np.concatenate((line.boolean("in_service"), trafo.boolean("in_service")))
    lim_l = 1000.0 * line.f64("max_i_ka") if line.n else np.zeros(0)
    lim_t = (1000.0 * trafo.f64("sn_mva") / (math.sqrt(3.0) * trafo.f64("vn_hv_kv"))) if trafo.n else np.zeros(0)
    m.thermal_limit_a = np.concatenate((lim_l, lim_t))
    names_l = _names_or_default(line, [f"{_fmt_bus(a)}_{_fmt_bus(b)}_{i}" for i, (a, b) in enumerate(zip(l_or, l_ex))])
    tr_default = []
    for i, (a, b) in enumerate(zip(t_or, t_ex)):
        sa, sb = sorted((_fmt_bus(a), _fmt_bus(b)))
        tr_default.append(f"{sa}_{sb}_{i + line.n}")
    names_t = _names_or_default(trafo, tr_default)
    m.name_line = np.concatenate((names_l, names_t)).astype(object)

    # --- generators (+ slack handling) --------------------------------------------------------------------
    g_bus = gen.i64("bus") if gen.n else np.zeros(0, np.int64)
    g_p = gen.f64("p_mw", 0.0) * gen.f64("scaling", 1.0) if gen.n else np.zeros(0)
    g_vm = gen.f64("vm_pu", 1.0) if gen.n else np.zeros(0)
    g_minq = gen.f64("min_q_mvar", -1e9) if gen.n else np.zeros(0)
    g_maxq = gen.f64("max_q_mvar", 1e9) if gen.n else np.zeros(0)
    g_slack = gen.boolean("slack", False) if gen.n else np.zeros(0, bool)
    g_stat = gen.boolean("in_service") if gen.n else np.zeros(0, bool)
    g_names_full = gen.n and gen.has_full_names()
    if not g_slack.any():
        # legacy file: the slack is an ext_grid -> append a generator on its bus
        # (pandaPowerBackend.py:394-451; only the first ext_grid is kept, :451)
        ext_is = ext.boolean("in_service") if ext.n else np.zeros(0, bool)
        if not ext.n or not ext_is.any():
            raise ValueError("grid has neither a slack generator nor an in-service ext_grid")
        k = int(np.nonzero(ext_is)[0][0])
        g_bus = np.append(g_bus, ext.i64("bus")[k])
        g_p = np.append(g_p, 0.0)
        g_vm = np.append(g_vm, ext.f64("vm_pu", 1.0)[k])
        g_minq = np.append(g_minq, ext.f64("min_q_mvar", -1e9)[k])
        g_maxq = np.append(g_maxq, ext.f64("max_q_mvar", 1e9)[k])
        g_slack = np.append(g_slack, True)
        g_stat = np.append(g_stat, True)
        m.slack_added = True
        g_names_full = False  # the appended row has a null name -> every name is regenerated
    m.n_gen = len(g_bus)
    m.gen_sub = g_bus.astype(_I32)
    m.gen_p0, m.gen_vm0, m.gen_min_q, m.gen_max_q = g_p, g_vm, g_minq, g_maxq
    m.gen_slack, m.gen_status0 = g_slack.astype(bool), g_stat.astype(bool)
    if g_names_full:
        m.name_gen = np.array([str(x) for x in gen["name"]], dtype=object)
    else:
        m.name_gen = np.array([f"gen_{_fmt_bus(b)}_{i}" for i, b in enumerate(g_bus)], dtype=object)

    # --- loads ---------------------------------------------------------------------------------------------
    m.n_load = load.n
    ld_bus = load.i64("bus") if load.n else np.zeros(0, np.int64)
    m.load_sub = ld_bus.astype(_I32)
    sc = load.f64("scaling", 1.0) if load.n else np.zeros(0)
    m.load_p0 = load.f64("p_mw", 0.0) * sc if load.n else np.zeros(0)
    m.load_q0 = load.f64("q_mvar", 0.0) * sc if load.n else np.zeros(0)
    m.load_status0 = load.boolean("in_service") if load.n else np.zeros(0, bool)
    m.name_load = _names_or_default(load, [f"load_{_fmt_bus(b)}_{i}" for i, b in enumerate(ld_bus)])

    # --- storage ---------------------------------------------------------------------------------------------
    m.n_storage = sto.n
    st_bus = sto.i64("bus") if sto.n else np.zeros(0, np.int64)
    m.storage_sub = st_bus.astype(_I32)
    m.storage_p0 = sto.f64("p_mw", 0.0) if sto.n else np.zeros(0)
    m.storage_q0 = sto.f64("q_mvar", 0.0) if sto.n else np.zeros(0)
    m.storage_status0 = sto.boolean("in_service") if sto.n else np.zeros(0, bool)
    m.name_storage = _names_or_default(sto, [f"storage_{_fmt_bus(b)}_{i}" for i, b in enumerate(st_bus)])

    # --- shunts ------------------------------------------------------------------------------------------------
    m.n_shunt = shunt.n
    sh_bus = shunt.i64("bus") if shunt.n else np.zeros(0, np.int64)
    m.shunt_sub = sh_bus.astype(_I32)
    m.shunt_p0 = shunt.f64("p_mw", 0.0) if shunt.n else np.zeros(0)
    m.shunt_q0 = shunt.f64("q_mvar", 0.0) if shunt.n else np.zeros(0)
    if shunt.n:
        vn_sh = shunt.f64("vn_kv", math.nan)
        vn_bus = m.sub_vn_kv[sh_bus]
        vn_sh = np.where(np.isnan(vn_sh), vn_bus, vn_sh)
        m.shunt_fact = shunt.f64("step", 1.0) * (vn_bus / vn_sh) ** 2
        m.shunt_vn_kv = vn_bus.astype(_F64)
    m.shunt_status0 = shunt.boolean("in_service") if shunt.n else np.zeros(0, bool)
    # the reference never reads shunt names from the file (pandaPowerBackend.py:758-764)
    m.name_shunt = np.array([f"shunt_{_fmt_bus(b)}_{i}" for i, b in enumerate(sh_bus)], dtype=object)

    _compute_topo_layout(m)
    return m


def _compute_topo_layout(m: GridModel) -> None:
    """sub_info / *_to_sub_pos / *_pos_topo_vect exactly as pandaPowerBackend.py:688-753 and
    Space/GridObjects.py:1409-1477 compute them."""
    used = np.zeros(m.n_sub, dtype=np.int64)
    m.line_or_to_sub_pos = np.zeros(m.n_line, _I32)
    m.line_ex_to_sub_pos = np.zeros(m.n_line, _I32)
    for i in range(m.n_line):   # lines first then trafos: same loop order as the reference
        so, se = int(m.line_or_sub[i]), int(m.line_ex_sub[i])
        m.line_or_to_sub_pos[i] = used[so]
        used[so] += 1
        m.line_ex_to_sub_pos[i] = used[se]
        used[se] += 1
    def _seq(subs):
        pos = np.zeros(len(subs), _I32)
        for i, s in enumerate(subs):
            pos[i] = used[int(s)]
            used[int(s)] += 1
        return pos
    m.gen_to_sub_pos = _seq(m.gen_sub)
    m.load_to_sub_pos = _seq(m.load_sub)
    m.storage_to_sub_pos = _seq(m.storage_sub)
    m.sub_info = used.astype(_I32)
    m.dim_topo = int(used.sum())
    start = np.concatenate(([0], np.cumsum(used)[:-1])).astype(np.int64)
    m.line_or_pos_topo_vect = (start[m.line_or_sub] + m.line_or_to_sub_pos).astype(_I32)
    m.line_ex_pos_topo_vect = (start[m.line_ex_sub] + m.line_ex_to_sub_pos).astype(_I32)
    m.gen_pos_topo_vect = (start[m.gen_sub] + m.gen_to_sub_pos).astype(_I32)
    m.load_pos_topo_vect = (start[m.load_sub] + m.load_to_sub_pos).astype(_I32)
    m.storage_pos_topo_vect = (start[m.storage_sub] + m.storage_to_sub_pos).astype(_I32)

#!/usr/bin/env python
"""Generate the committed fixtures under tests/golden/ from the read-only reference checkout.

Run (in the build container, where /root/reference exists):

    python tests/golden/make_fixtures.py [/root/reference]

Nothing here is needed at test/bench time: the GPU box has no /root/reference, so everything the
``-m gpu`` tests, ``smoke()`` and ``bench.py`` need is extracted ONCE into small ``.npz`` files:

* ``<env>.grid.npz``      `grid2op_amd.grid_model.GridModel` of the env's pandapower-JSON grid file
* ``<env>.res.npz``       the pandapower results embedded in that grid file (``res_bus``, ``res_line``,
                          ``res_trafo``, ``res_gen``, ``res_shunt``, ``res_ext_grid``) = golden vectors
                          produced by the reference's own solver (pandapower) on the stored state
* ``<env>.chronics.npz``  one chronics scenario (float32 [T, n] in the GridModel's element order)
                          + the positional ``thermal_limits`` of the env's ``config.py``
* ``known_answers.npz``   the hard-coded vectors of grid2op/tests/BaseBackendTest.py:262-313,1584-1607
                          (AC / DC ``p_or`` and ``a_or`` on data_test/test_PandaPower/test_case14.json)

No reference SOURCE is copied: only numeric data tables.
"""
from __future__ import annotations

import bz2
import io
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

from grid2op_amd.grid_model import load_grid_model  # noqa: E402
from grid2op_amd.pp_json import read_pandapower_json  # noqa: E402

GRIDS = {
    # fixture name -> (grid file relative to <ref>/grid2op, chronics dir (or None), config.py (or None))
    "rte_case5_example": ("data/rte_case5_example/grid.json", "data/rte_case5_example/chronics/00",
                          "data/rte_case5_example/config.py"),
    "l2rpn_case14_sandbox": ("data/l2rpn_case14_sandbox/grid.json", "data/l2rpn_case14_sandbox/chronics/0000",
                             "data/l2rpn_case14_sandbox/config.py"),
    "educ_case14_storage": ("data/educ_case14_storage/grid.json", "data/educ_case14_storage/chronics/0",
                            "data/educ_case14_storage/config.py"),
    "l2rpn_neurips_2020_track1": ("data/l2rpn_neurips_2020_track1/grid.json",
                                  "data/l2rpn_neurips_2020_track1/chronics/Scenario_august_dummy",
                                  "data/l2rpn_neurips_2020_track1/config.py"),
    "l2rpn_wcci_2022_dev": ("data/l2rpn_wcci_2022_dev/grid.json", "data/l2rpn_wcci_2022_dev/chronics/2050-02-14_0",
                            "data/l2rpn_wcci_2022_dev/config.py"),
    "l2rpn_idf_2023": ("data/l2rpn_idf_2023/grid.json", "data/l2rpn_idf_2023/chronics/2035-01-15_0", "data/l2rpn_idf_2023/config.py"),
    "l2rpn_2019": ("data/l2rpn_2019/grid.json", None, None),
    "rte_case14_test": ("data/rte_case14_test/grid.json", "data/rte_case14_test/chronics/0", "data/rte_case14_test/config.py"),
    "test_case14": ("data_test/test_PandaPower/test_case14.json", None, None),
    # the other grids the reference ships (embedded pandapower results pin the oracle on each of them)
    "rte_case118_example": ("data/rte_case118_example/grid.json", None, None),
    "l2rpn_wcci_2020": ("data/l2rpn_wcci_2020/grid.json", None, None),
    "l2rpn_icaps_2021": ("data/l2rpn_icaps_2021/grid.json", None, None),
    "l2rpn_neurips_2020_track2_x1": ("data/l2rpn_neurips_2020_track2/x1/grid.json", None, None),
    "rte_case14_realistic": ("data/rte_case14_realistic/grid.json", None, None),
    "educ_case14_redisp": ("data/educ_case14_redisp/grid.json", None, None),
    "l2rpn_case14_sandbox_diff_grid": ("data/l2rpn_case14_sandbox_diff_grid/grid.json", None, None),
}
MAX_ROWS = 600


def _res_tables(path, m):
    T = read_pandapower_json(path)["tables"]
    out = {}

    def put(tab, cols, prefix):
        if tab in T and T[tab].n:
            for c in cols:
                if c in T[tab]:
                    out[f"{prefix}{c}"] = T[tab].f64(c)

    if "res_bus" in T and T["res_bus"].n:
        rb = T["res_bus"]
        lab = np.asarray(rb.index).astype(int)
        ok = lab < m.n_sub
        for c in ("vm_pu", "va_degree"):
            v = np.full(m.n_sub, np.nan)
            v[lab[ok]] = rb.f64(c)[ok]
            out["bus_" + c] = v
    put("res_line", ["p_from_mw", "q_from_mvar", "p_to_mw", "q_to_mvar", "i_from_ka", "i_to_ka",
                     "vm_from_pu", "va_from_degree", "vm_to_pu", "va_to_degree"], "line_")
    put("res_trafo", ["p_hv_mw", "q_hv_mvar", "p_lv_mw", "q_lv_mvar", "i_hv_ka", "i_lv_ka",
                      "vm_hv_pu", "va_hv_degree", "vm_lv_pu", "va_lv_degree"], "trafo_")
    put("res_gen", ["p_mw", "q_mvar", "vm_pu", "va_degree"], "gen_")
    put("res_load", ["p_mw", "q_mvar"], "load_")
    put("res_shunt", ["p_mw", "q_mvar", "vm_pu"], "shunt_")
    put("res_ext_grid", ["p_mw", "q_mvar"], "ext_grid_")
    put("res_storage", ["p_mw", "q_mvar"], "storage_")
    return out


def _read_csv_bz2(path):
    with bz2.open(path, "rt") as f:
        header = f.readline().strip().split(";")
        data = np.loadtxt(io.StringIO(f.read()), delimiter=";", ndmin=2)
    return header, data


def _chronics(cdir, m):
    out = {}
    specs = [("load_p", m.name_load), ("load_q", m.name_load), ("prod_p", m.name_gen), ("prod_v", m.name_gen)]
    for key, names in specs:
        p = os.path.join(cdir, key + ".csv.bz2")
        if not os.path.exists(p):
            continue
        header, data = _read_csv_bz2(p)
        idx = [header.index(str(n)) for n in names]      # chronics columns are matched BY NAME
        out[key] = data[:MAX_ROWS, idx].astype(np.float32)
    return out


def _thermal_limits(cfg_path):
    txt = open(cfg_path).read()
    mt = re.search(r'"thermal_limits"\s*:\s*\[(.*?)\]', txt, re.S)
    if not mt:
        return None
    return np.array([float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", mt.group(1))])


def _known_answers(ref):
    """Numbers hard-coded in the reference's own backend test-suite (tests/BaseBackendTest.py)."""
    lines = open(os.path.join(ref, "grid2op/tests/BaseBackendTest.py")).read().split("\n")

    def grab(lo, hi):
        txt = "\n".join(lines[lo - 1:hi])
        txt = txt[txt.index("["):txt.rindex("]") + 1]
        return np.array([float(x) for x in re.findall(r"[-+]?\d+\.\d*(?:[eE][-+]?\d+)?", txt)])

    out = {"p_or_dc": grab(262, 287), "p_or_ac": grab(289, 319), "a_or_init": grab(1584, 1607)}
    out.update(_observation_json_ref(ref))
    return out


def _observation_json_ref(ref):
    """The complete observation after reset of rte_case14_test recorded with PandaPowerBackend that
    grid2op/tests/test_Observation.py compares for EXACT float32 equality (``self.json_ref``, :307-...): only its
    numeric vectors are kept (prefix ``obs14_``)."""
    import ast
    txt = open(os.path.join(ref, "grid2op/tests/test_Observation.py")).read()
    a = txt.index("self.json_ref = {") + len("self.json_ref = ")
    depth, b = 0, a
    while True:
        ch = txt[b]
        depth += ch == "{"
        depth -= ch == "}"
        b += 1
        if depth == 0:
            break
    d = ast.literal_eval(txt[a:b])
    keep = ["gen_p", "gen_q", "gen_v", "load_p", "load_q", "load_v", "p_or", "q_or", "v_or", "a_or", "p_ex", "q_ex", "v_ex",
            "a_ex", "rho", "line_status", "topo_vect", "theta_or", "theta_ex", "load_theta", "gen_theta"]
    return {"obs14_" + k: np.asarray(d[k]) for k in keep if k in d}


def _runner_trajectories(ref, versions=None):
    """Trajectories recorded with PandaPowerBackend that grid2op/tests/test_Runner.py:426,540-585 loads
    (data_test/runner_data/res_agent_<ver>/{00,01}: rte_case5_example, RandomAgent).  Every recorded observation holds
    both the INPUTS of its power flow (load_p/q, gen_p, gen_v set-points, topo_vect after the random topology action)
    and the pandapower RESULTS (flows, currents, gen_q), so each row is a self-contained golden vector.  Decoded with the
    reference's own ObservationSpace (needs the reference importable: tests/_refshim); numbers only are kept."""
    try:
        sys.path.insert(0, os.path.join(HERE, "..", "_refshim"))
        sys.path.insert(0, ref)
        import json
        import warnings
        warnings.filterwarnings("ignore")
        from grid2op.Observation import ObservationSpace
    except Exception as exc:       # pragma: no cover
        print("runner trajectories skipped:", exc)
        return {}
    out = {}
    keys = ["load_p", "load_q", "gen_p", "gen_v", "gen_q", "topo_vect", "line_status", "p_or", "q_or", "v_or", "a_or", "p_ex", "q_ex",
            "v_ex", "a_ex"]
    root = os.path.join(ref, "grid2op/data_test/runner_data")
    if versions is None:          # every version the reference ships (test_Runner.py:426 loops over all of them): 46 folders, 92 episodes
        versions = sorted(d[len("res_agent_"):] for d in os.listdir(root) if d.startswith("res_agent_"))
    for ver in versions:
        base = os.path.join(root, f"res_agent_{ver}")
        if not os.path.isdir(base):
            continue
        try:
            sp = ObservationSpace.from_dict(os.path.join(base, "dict_observation_space.json"))
        except Exception as exc:   # 1.9.0: its observation-space file does not load with the reference's own ObservationSpace either
            print(f"runner trajectories: version {ver} skipped ({type(exc).__name__})")
            continue
        try:
            got = {}
            for ep in ("00", "01"):
                if not os.path.isdir(os.path.join(base, ep)):
                    continue
                meta = json.load(open(os.path.join(base, ep, "episode_meta.json")))
                n = int(meta["nb_timestep_played"])
                data = np.load(os.path.join(base, ep, "observations.npz"))["data"]
                rows = [sp.from_vect(data[t]) for t in range(n)]       # the row after the last played step is the game-over obs
                tag = f"v{ver.replace('.', '_')}_{ep}_"
                for k in keys:
                    got[tag + k] = np.stack([np.asarray(getattr(o, k)) for o in rows])
            out.update(got)
        except Exception as exc:   # 1.9.0: the recorded vectors do not match its own observation-space file (185 vs 182 elements)
            print(f"runner trajectories: version {ver} skipped ({type(exc).__name__})")
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    base = os.path.join(ref, "grid2op")
    for name, (gfile, cdir, cfg) in GRIDS.items():
        gpath = os.path.join(base, gfile)
        m = load_grid_model(gpath)
        m.save_npz(os.path.join(HERE, f"{name}.grid.npz"))
        res = _res_tables(gpath, m)
        if res:
            np.savez_compressed(os.path.join(HERE, f"{name}.res.npz"), **res)
        chron = {}
        if cdir is not None:
            chron.update(_chronics(os.path.join(base, cdir), m))
        if cfg is not None:
            th = _thermal_limits(os.path.join(base, cfg))
            if th is not None and len(th) == m.n_line:
                chron["thermal_limits"] = th.astype(np.float32)
        if chron:
            np.savez_compressed(os.path.join(HERE, f"{name}.chronics.npz"), **chron)
        print(name, "n_sub", m.n_sub, "n_line", m.n_line, "res:", sorted(res)[:3], "chron:", {k: v.shape for k, v in chron.items()})
    rt = _runner_trajectories(ref)
    if rt:
        np.savez_compressed(os.path.join(HERE, "runner_case5.npz"), **rt)
        print("runner trajectories", {k: v.shape for k, v in rt.items() if k.endswith("p_or")})
    ka = _known_answers(ref)
    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), **ka)
    print("known answers", {k: v.shape for k, v in ka.items()})


if __name__ == "__main__":
    main()

"""Batched chronics loader: decode a grid2op chronics folder ONCE into the float32 table the
engine keeps resident in HBM (``PowerFlowEngine.upload_chronics``).

On-disk format (e.g. grid2op/data/l2rpn_case14_sandbox/chronics/0000/): ``load_p.csv.bz2``, ``load_q.csv.bz2``,
``prod_p.csv.bz2``, ``prod_v.csv.bz2`` -- ``;``-separated, one header line with the ELEMENT NAMES, one row per time
step (reader in the reference: grid2op/Chronics/gridStateFromFile.py).  Columns are matched to the backend's elements
BY NAME (optionally through the ``names_chronics_to_backend`` mapping of the environment,
grid2op/Environment/environment.py:431-437).  Environments without ``prod_v.csv`` take the voltage set-points from the
``V`` column of ``prods_charac.csv`` (kV) -- what ``ControlVoltageFromFile`` ends up applying every step.
``maintenance.csv`` / ``hazards.csv`` (0/1 per line and row, header = LINE names) are read when present.

`load_chronics_multifolder` does what ``grid2op.Chronics.Multifolder`` (grid2op/Chronics/multiFolder.py) does at
``initialize``: every sub-folder of the chronics directory is one scenario, taken in SORTED order; the result is one stacked
table per quantity, ``[n_scenarios, T, n]``, i.e. the ``n_tables`` the engine indexes per lane (``lane_table``).
"""
from __future__ import annotations

import bz2
import csv
import io
import os
from typing import Dict, Optional

import numpy as np

from .grid_model import GridModel

__all__ = ["load_chronics_folder", "load_chronics_multifolder", "chronics_table"]


def _read_csv(path: str):
    opener = bz2.open if path.endswith(".bz2") else open
    with opener(path, "rt") as f:
        header = f.readline().strip().split(";")
        data = np.loadtxt(io.StringIO(f.read()), delimiter=";", ndmin=2)
    return header, data


def _find(folder: str, stem: str) -> Optional[str]:
    for ext in (".csv.bz2", ".csv"):
        p = os.path.join(folder, stem + ext)
        if os.path.exists(p):
            return p
    return None


def _columns(header, names, mapping: Optional[Dict[str, str]]):
    """Index of each backend element in the file header (chronics name -> backend name through `mapping`)."""
    if mapping:
        header = [mapping.get(h, h) for h in header]
    pos = {h: i for i, h in enumerate(header)}
    missing = [str(n) for n in names if str(n) not in pos]
    if missing:
        raise KeyError(f"chronics columns missing for elements {missing[:5]}...")
    return [pos[str(n)] for n in names]


def load_chronics_folder(folder: str, model: GridModel, names_chronics_to_backend: Optional[dict] = None,
                         prods_charac: Optional[str] = None, max_rows: Optional[int] = None,
                         forecasts: bool = False) -> Dict[str, np.ndarray]:
    """``{"load_p","load_q","prod_p","prod_v"}`` float32 ``[T, n]`` arrays in the GridModel's element order (plus
    ``maintenance`` / ``hazards`` uint8 ``[T, n_line]`` when the folder has them, and -- ``forecasts=True`` -- the
    ``*_forecasted`` tables ``obs.simulate`` injects, grid2op/Chronics/gridStateFromFileWithForecasts.py)."""
    m = model
    mp = names_chronics_to_backend or {}
    out = {}
    todo = [("load_p", m.name_load, "loads"), ("load_q", m.name_load, "loads"), ("prod_p", m.name_gen, "prods"),
            ("prod_v", m.name_gen, "prods")]
    if forecasts:
        todo += [(k + "_forecasted", n, s_) for k, n, s_ in todo]
    for key, names, sub in todo:
        path = _find(folder, key)
        if path is None:
            continue
        header, data = _read_csv(path)
        idx = _columns(header, names, mp.get(sub))
        out[key] = np.ascontiguousarray(data[:max_rows, idx], dtype=np.float32)
    for key in ("maintenance", "hazards"):
        path = _find(folder, key)
        if path is None:
            continue
        header, data = _read_csv(path)
        idx = _columns(header, m.name_line, mp.get("lines"))
        out[key] = np.ascontiguousarray(data[:max_rows, idx] != 0, dtype=np.uint8)
    if "prod_v" not in out:
        T = out["prod_p"].shape[0]
        v = (m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32)       # grid-file set-points (kV)
        if prods_charac and os.path.exists(prods_charac):
            with open(prods_charac, newline="") as f:
                rows = {r["name"]: r for r in csv.DictReader(f)}
            for i, n in enumerate(m.name_gen):
                r = rows.get(str(n))
                if r is not None and r.get("V") not in (None, ""):
                    v[i] = np.float32(float(r["V"]))
        out["prod_v"] = np.tile(v, (T, 1))
    return out


def load_chronics_multifolder(chronics_dir: str, model: GridModel, names_chronics_to_backend: Optional[dict] = None,
                              prods_charac: Optional[str] = None, max_rows: Optional[int] = None, truncate: bool = False,
                              forecasts: bool = False):
    """Every scenario of a chronics directory (sub-folders in sorted order, as ``Multifolder`` lists them).  Returns
    ``(scenario names, {"load_p", "load_q", "prod_p", "prod_v"[, "maintenance", "hazards"]: [n_scenarios, T, n]})``.  Scenarios
    of different lengths are an error unless ``truncate`` (then all are cut to the shortest)."""
    names = sorted(d for d in os.listdir(chronics_dir) if os.path.isdir(os.path.join(chronics_dir, d)))
    if not names:
        raise FileNotFoundError(f"no scenario folder under {chronics_dir}")
    per = [load_chronics_folder(os.path.join(chronics_dir, n), model, names_chronics_to_backend, prods_charac, max_rows, forecasts)
           for n in names]
    lens = [p["load_p"].shape[0] for p in per]
    T = min(lens)
    if len(set(lens)) > 1 and not truncate:
        raise ValueError(f"scenarios of different lengths {dict(zip(names, lens))}: pass truncate=True to cut them to {T} rows")
    out = {}
    keys = ["load_p", "load_q", "prod_p", "prod_v", "maintenance", "hazards"]
    keys += [k + "_forecasted" for k in keys[:4]]
    for key in keys:
        if any(key in p for p in per):
            n_col = next(p[key].shape[1] for p in per if key in p)
            dt = next(p[key].dtype for p in per if key in p)
            out[key] = np.stack([p[key][:T] if key in p else np.zeros((T, n_col), dt) for p in per])
    return names, out


def chronics_table(ch: Dict[str, np.ndarray]) -> np.ndarray:
    """``[..., T, 2*n_load + 2*n_gen]`` float32 table(s) in the engine's chronics row layout (gpf_layout.chron_*)."""
    return np.ascontiguousarray(np.concatenate([ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]], axis=-1),
                                dtype=np.float32)

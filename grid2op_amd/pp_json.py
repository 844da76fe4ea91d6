"""Reader for pandapower-JSON ``grid.json`` files -- without pandapower.

The reference loads its grids with ``pandapower.from_json``
(grid2op/Backend/pandaPowerBackend.py:377).  pandapower is a third-party dependency that is not part
of the reference tree; this module re-reads the same on-disk format directly so the backend has no
pandapower dependency.  Three on-disk variants exist in the reference's data folders
(SURVEY.md section 7 item 2):

(a) ``{"_module","_class","_object": {table: {"_class": "DataFrame", "_object": "<split json>",
    "dtype": {...}}}}``            (all of grid2op/data/*)
(b) same, but the outer ``_object`` is itself a JSON *string*  (pandapower 2.0.x files, e.g.
    grid2op/data_test/test_PandaPower/test_case14.json)
(c) tables may be missing altogether (``ext_grid``, ``storage``, ``shunt``, ``trafo`` ...).

Rows are kept in FILE ORDER (several legacy files store rows in lexicographic label order
``0,1,10,11,...,2,3``); the reference's positional conventions (element names such as ``8_9_2``,
positional ``thermal_limits`` vectors) rely on that order.
"""
from __future__ import annotations

import json
import math
from typing import Dict, List, Optional

import numpy as np

__all__ = ["Table", "read_pandapower_json"]


class Table:
    """A column store: ``cols[name] -> np.ndarray`` (object arrays for strings), file row order."""

    def __init__(self, columns: List[str], index: List, data: List[List], dtypes: Optional[dict] = None):
        self.index = np.asarray(index)
        self.n = len(data)
        self.columns = list(columns)
        self.cols: Dict[str, np.ndarray] = {}
        dtypes = dtypes or {}
        for j, c in enumerate(columns):
            raw = [row[j] for row in data]
            self.cols[c] = self._convert(raw, dtypes.get(c))

    @staticmethod
    def _convert(raw, dt):
        if dt is None:
            dt = "object"
        if dt.startswith("float"):
            return np.array([math.nan if v is None else float(v) for v in raw], dtype=np.float64)
        if dt.startswith(("uint", "int")):
            # ints with nulls happen in legacy files (tap_pos stored as float): keep as float
            if any(v is None for v in raw):
                return np.array([math.nan if v is None else float(v) for v in raw], dtype=np.float64)
            return np.array([int(v) for v in raw], dtype=np.int64)
        if dt == "bool":
            return np.array([bool(v) for v in raw], dtype=bool)
        return np.array(raw, dtype=object)

    def __contains__(self, c):
        return c in self.cols

    def __getitem__(self, c) -> np.ndarray:
        return self.cols[c]

    def get(self, c, default=None):
        return self.cols.get(c, default)

    def f64(self, c, default=math.nan) -> np.ndarray:
        """Column as float64 (missing column / nulls -> ``default``)."""
        if c not in self.cols:
            return np.full(self.n, default, dtype=np.float64)
        col = self.cols[c]
        if col.dtype == object:
            out = np.array([default if (v is None or v is False) else float(v) for v in col], dtype=np.float64)
        else:
            out = col.astype(np.float64)
        if not (isinstance(default, float) and math.isnan(default)):
            out = np.where(np.isnan(out), default, out)
        return out

    def i64(self, c) -> np.ndarray:
        return np.asarray(self.f64(c)).astype(np.int64)

    def boolean(self, c, default=True) -> np.ndarray:
        if c not in self.cols:
            return np.full(self.n, default, dtype=bool)
        col = self.cols[c]
        return np.array([default if v is None else bool(v) for v in col], dtype=bool)

    def has_full_names(self) -> bool:
        """Mirror of ``"name" in df.columns and not df["name"].isnull().any()``
        (pandaPowerBackend.py:484-487)."""
        if "name" not in self.cols:
            return False
        for v in self.cols["name"]:
            if v is None:
                return False
            if isinstance(v, float) and math.isnan(v):
                return False
        return True


def _empty_table() -> Table:
    return Table([], [], [])


def read_pandapower_json(path: str) -> dict:
    """Return ``{"tables": {name: Table}, "sn_mva": float, "f_hz": float, "version": str, ...}``."""
    with open(path, "r", encoding="utf-8") as f:
        top = json.load(f)
    obj = top.get("_object", top)
    if isinstance(obj, str):  # variant (b)
        obj = json.loads(obj)
    tables: Dict[str, Table] = {}
    scalars = {}
    for key, val in obj.items():
        if isinstance(val, dict) and val.get("_class") == "DataFrame":
            inner = val["_object"]
            if isinstance(inner, str):
                inner = json.loads(inner)
            tables[key] = Table(inner.get("columns", []), inner.get("index", []), inner.get("data", []),
                                val.get("dtype"))
        elif isinstance(val, (int, float, str, bool)) or val is None:
            scalars[key] = val
    out = {"tables": tables}
    out["sn_mva"] = float(scalars.get("sn_mva", 1.0) or 1.0)
    out["f_hz"] = float(scalars.get("f_hz", 50.0) or 50.0)
    out["version"] = str(scalars.get("version", ""))
    out["converged"] = bool(scalars.get("converged", False))
    for t in ("bus", "line", "trafo", "gen", "load", "storage", "shunt", "ext_grid", "sgen"):
        out["tables"].setdefault(t, _empty_table())
    return out

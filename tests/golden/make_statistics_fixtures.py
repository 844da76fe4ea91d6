#!/usr/bin/env python
"""Pack the reference's OWN recorded DoNothing episodes into a fixture (build container only; /root/reference is not on the GPU box).

    python tests/golden/make_statistics_fixtures.py        # -> tests/golden/stats_case5.npz

`grid2op/data/rte_case5_example/_statistics/` ships with the reference: `EpisodeStatistics.compute` (grid2op/utils/
underlying_statistics.py:680-813) ran the DoNothingAgent through the 20 scenarios of rte_case5_example with the environment's
default parameters (overflow disconnections ON: NB_TIMESTEP_OVERFLOW_ALLOWED 2, HARD_OVERFLOW_THRESHOLD 2) and **PandaPowerBackend**
(config.py of that environment), and stored every observation attribute of every step (`obs_<attr>.npz`, 7 930 rows; row 0 of an
episode is the reset observation, metadata.json gives the episode lengths).  19 episodes end in a game over after overloaded lines
tripped, one runs through its 2 016 steps.  That is the headline workload -- batched DoNothing env.step with the protections on --
recorded with pandapower itself.  Nothing is computed here: the recorded rows are sub-sampled (the first 16 and the last 48 rows of
every episode, every 8th in between; all rows' line status / protection counters are kept) and stored next to the scenarios'
chronics rows as grid2op_amd.chronics reads them.  Replayed by tests/test_oracle_golden.py (CPU oracle) and
tests/test_gpu_multistep.py (multi-step launches, one lane per scenario)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REFERENCE = os.environ.get("GRID2OP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from grid2op_amd.chronics import load_chronics_folder  # noqa: E402
from grid2op_amd.grid_model import GridModel  # noqa: E402

ENV = "rte_case5_example"
FLOATS = ["p_or", "q_or", "v_or", "a_or", "p_ex", "q_ex", "v_ex", "a_ex", "rho", "prod_p", "prod_q", "prod_v", "load_p", "load_q", "load_v"]


def main():
    base = os.path.join(REFERENCE, "grid2op", "data", ENV)
    stats = os.path.join(base, "_statistics")
    load = lambda k: np.load(os.path.join(stats, f"obs_{k}.npz"))["data"]
    with open(os.path.join(stats, "metadata.json"), encoding="utf-8") as f:
        meta = json.load(f)
    assert "DoNothingAgent" in meta["agent_type"] and meta["parameters"]["NO_OVERFLOW_DISCONNECTION"] is False
    ids = np.load(os.path.join(stats, "scenario_ids.npz"))["data"].ravel().astype(np.int64)
    n_ep = int(ids.max()) + 1
    n_rows = np.bincount(ids)
    start = np.concatenate(([0], np.cumsum(n_rows)))
    names = [meta[str(e)]["scenario_name"] for e in range(n_ep)]
    assert [meta[str(e)]["nb_step"] for e in range(n_ep)] == n_rows.tolist()
    m = GridModel.load_npz(os.path.join(HERE, f"{ENV}.grid.npz"))
    keep = []
    for e in range(n_ep):
        n = int(n_rows[e])
        k = set(range(min(16, n))) | set(range(max(0, n - 48), n)) | set(range(0, n, 8))
        keep += [int(start[e]) + i for i in sorted(k)]
    keep = np.asarray(keep, np.int64)
    out = {"row_idx": keep, "episode_start": start, "scenario": np.array(names)}
    for k in FLOATS:
        out[k] = load(k)[keep].astype(np.float32)
    out["topo_vect"] = load("topo_vect")[keep].astype(np.int8)
    out["line_status_all"] = load("line_status").astype(bool)                  # every row
    out["timestep_overflow_all"] = load("timestep_overflow").astype(np.int8)   # every row
    rho, a = load("rho"), load("a_or")
    with np.errstate(invalid="ignore", divide="ignore"):
        lim = np.nanmedian(np.where(rho > 1e-3, a / rho, np.nan), axis=0)
    out["thermal_limit"] = np.round(lim).astype(np.float32)                    # (a_or / rho of the recording: 600, 220, 160 ... A)
    assert np.abs(lim - out["thermal_limit"]).max() < 1e-2
    tabs, offs = [], [0]
    for e in range(n_ep):
        ch = load_chronics_folder(os.path.join(base, "chronics", names[e]), m, max_rows=int(n_rows[e]) + 1)
        assert not ch["maintenance"].any() and not ch["hazards"].any()
        t = np.concatenate([ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]], axis=1).astype(np.float32)
        tabs.append(t)
        offs.append(offs[-1] + t.shape[0])
    out["chronics_rows"] = np.concatenate(tabs)                                # [sum(rows), 2 n_load + 2 n_gen], ragged by episode
    out["chronics_start"] = np.asarray(offs, np.int64)
    out["parameters_json"] = np.array(json.dumps(meta["parameters"]))
    path = os.path.join(HERE, "stats_case5.npz")
    np.savez_compressed(path, **out)
    print("episodes", n_ep, "rows", int(n_rows.sum()), "kept", keep.size, "game overs", int((n_rows < 2017).sum()), "bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()

"""The driver keeps only the last 8 KB of bench.py's stdout (round 4's single 31 KB line could not be parsed): the stdout line must be
a compact record (< 4 KB) that still carries the contract's keys, `roofline`, `cpu_baseline` and one number + roofline fraction per
BASELINE config.  Built here from a recorded full record (tests/golden/bench_full_r04.json = the round-4 run of the driver's command)."""
import copy
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.fixture(scope="module")
def full():
    with open(os.path.join(ROOT, "tests", "golden", "bench_full_r04.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_compact_line_is_small_and_round_trips(full):
    rec = bench.compact_record(full, "bench_full.json")
    line = json.dumps(rec)
    assert len(line) < 4096, len(line)
    assert "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]          # the headline is passed through unrounded
    assert back["config"]["workload"].startswith("l2rpn_case14_sandbox") and back["config"]["lanes_per_gpu"] == 4096
    rf = back["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    assert rf["traffic"] and rf["kernel"].startswith("step_sparse_kernel<")
    cb = back["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["all_cores"]["cores"] == 16
    assert "unavailable" in cb["pandapower"] and "unavailable" in cb["lightsim2grid"]
    # one line per BASELINE config, each with its roofline fraction
    cfgs = back["configs"]
    for k in ("n1_fanout_36sub", "wcci_118sub", "wcci_env_dynamics", "idf_ac_118sub", "ptdf_rows"):
        assert cfgs[k]["value"] > 0, k
    for k in ("n1_fanout_36sub", "wcci_118sub", "idf_ac_118sub", "ptdf_rows"):
        assert 0 < cfgs[k]["frac"] < 1 and cfgs[k]["bound"] in ("hbm", "mfma"), k
        assert cfgs[k]["oracle_ok"] is True
    assert cfgs["wcci_118sub"]["min_med_max"][0] <= cfgs["wcci_118sub"]["min_med_max"][1] <= cfgs["wcci_118sub"]["min_med_max"][2]
    assert back["parity"]["all_ok"] is True and back["parity"]["oracle_checks"] >= 10
    assert back["full_record"] == "bench_full.json"


def test_compact_line_shrinks_inflated_records(full):
    big = copy.deepcopy(full)
    for k in ("shipped_kernels", "one_launch_per_step", "cascade_on"):
        big[k]["value_median"] = 1.23456789e8
    big["config"]["workload"] = big["config"]["workload"] + " x" * 200
    big["cpu_baseline"]["sample"] = "s" * 5000
    big["specialization"]["variants"] = ["v" * 100] * 100
    line = json.dumps(bench.compact_record(big, None))
    assert len(line) < 8192
    assert json.loads(line)["value"] == full["value"]


def test_emit_prints_one_line_and_writes_the_full_record(full, tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("GRIDPF_BENCH_FULL", str(tmp_path / "full.json"))
    line = bench.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and out.strip() == line and len(line) < 4096
    with open(tmp_path / "full.json") as f:
        assert json.load(f)["value"] == full["value"]


def test_compact_line_without_secondaries_or_baseline():
    res = {"metric": "env steps/sec (batched DoNothing)", "value": 1.0, "unit": "env steps/sec", "n_gpus": 2, "steps": 7, "warmup": 2, "ms_per_step": 1.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w", "lanes_per_gpu": 4096, "total_lanes": 8192}, "windows": {"n": 3, "value_min": 1.0, "value_median": 1.0, "value_max": 1.0},
           "roofline": None, "cpu_baseline": None}
    rec = bench.compact_record(res, None)
    assert rec["cpu_baseline"] is None and rec["roofline"] is None and rec["configs"] == {} and rec["windows"]["n"] == 3

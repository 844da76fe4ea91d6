"""The REAL multi-rank path on the hardware at hand: ``bench.py --gpus 2 --share-device --dist-backend gloo`` starts two ranks
(torch.distributed.run, one process each) that both drive HIP device 0 with real engines -- process-group init, per-rank
contiguous lane blocks of the GLOBAL batch, barrier + max-over-ranks timing, rank-0 JSON aggregation -- and every rank dumps
the result rows of its lanes.  They must equal, bit for bit, the matching halves of a 1-rank run over the same global batch
(the synthetic inputs are a pure function of the global lane id, grid2op_amd/sharding.py; single-wavefront instances are
bitwise reproducible and independent of their position in the batch).  No scaling is measured here: both ranks share one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, timeout=900, full=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if full:
        env["GRIDPF_BENCH_FULL"] = full            # the full record (the stdout line is the compact one)
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("env_name,batch", [("l2rpn_case14_sandbox", 256), ("l2rpn_neurips_2020_track1", 96),
                                            ("l2rpn_wcci_2022_dev", 64)])       # 118 substations: the 2-wavefront kernel (BASELINE configs[3])
def test_two_ranks_with_real_engines_equal_the_halves_of_one_rank(env_name, batch, tmp_path):
    common = ["--env", env_name, "--steps", "12", "--warmup", "4", "--windows", "2", "--no-secondary", "--no-cpu-baseline"]
    two = str(tmp_path / "two")
    full2 = str(tmp_path / "full2.json")
    p2 = _run(["--gpus", "2", "--share-device", "--dist-backend", "gloo", "--batch", str(batch), "--dump-results", two] + common, full=full2)
    assert p2.returncode == 0, p2.stderr[-3000:]
    lines = [l for l in p2.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, p2.stdout                # ONE compact line (rank 0)
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["total_lanes"] == 2 * batch
    assert res["data"] == "synthetic" and res["frac_converged"] == 1.0 and res["parity"]["all_ok"] is True
    assert res["value"] == pytest.approx(2 * batch * 12 / (res["ms_per_step"] * 12 * 1e-3), rel=1e-9)
    # both clocks are reported: the wall clock between the barrier + synchronize brackets (MAX over ranks; `value`) and the per-rank
    # HIP-event window around the same steps (never slower than the wall clock that contains it)
    assert res["value_hip_event_window"] >= res["value"] * 0.98
    with open(full2) as f:
        full = json.load(f)
    assert full["value"] == res["value"] and full["config"]["share_device"] is True and full["config"]["dist_backend"] == "gloo"
    assert full["oracle_check"]["ok"], full["oracle_check"]
    assert len(full["windows"]["elapsed_ms"]) == len(full["windows"]["hip_event_windows"]["elapsed_ms"]) == 2
    one = str(tmp_path / "one")
    p1 = _run(["--gpus", "1", "--batch", str(2 * batch), "--dump-results", one] + common, full=str(tmp_path / "full1.json"))
    assert p1.returncode == 0, p1.stderr[-3000:]
    whole = np.load(one + ".rank0.npz")
    r0, r1 = np.load(two + ".rank0.npz"), np.load(two + ".rank1.npz")
    assert int(r0["lane0"]) == 0 and int(r1["lane0"]) == batch and int(whole["lane0"]) == 0
    assert int(r0["t_last"]) == int(r1["t_last"]) == int(whole["t_last"])
    for key in ("out", "status", "topo_vect"):
        assert np.array_equal(np.concatenate([r0[key], r1[key]]), whole[key], equal_nan=(key == "out")), key

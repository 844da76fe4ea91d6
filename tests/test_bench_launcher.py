"""``bench.py --gpus N`` on CPU: the launcher must really start N ranks (one process per GPU in production; here gloo +
the arithmetic-free stub engine), shard the global batch, and print ONE JSON line that says so.  Also: a request for
more GPUs than are visible must fail loudly instead of silently measuring fewer, and `ShardedEngine` (the single-process
form) must route global lane ranges to the right per-device engine."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_2_launches_two_ranks_and_reports_them():
    p = _run(["--gpus", "2", "--steps", "7", "--warmup", "2", "--windows", "3", "--stub-engine", "--dist-backend", "gloo",
              "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                           # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 7 and res["warmup"] == 2
    assert res["config"]["lanes_per_gpu"] == 4096 and res["config"]["total_lanes"] == 8192
    assert res["scaling"] == "weak" and res["windows"]["n"] == 3
    # value = the units ALL ranks processed / the (max over ranks) time of the median window
    assert res["value"] == pytest.approx(8192 * 7 / (res["ms_per_step"] * 7 * 1e-3), rel=1e-9)
    assert "STUB" in res["data"]


def test_gpus_mismatch_with_world_size_fails_loudly():
    p = _run(["--gpus", "2", "--stub-engine", "--dist-backend", "gloo"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0
    assert "WORLD_SIZE" in p.stderr


def test_more_gpus_than_visible_is_refused():
    """No silent 1-GPU measurement: in this container no HIP device is visible at all."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert p.returncode != 0
    assert "refusing" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_sharded_engine_routes_global_lanes(load_model):
    from grid2op_amd.sharding import ShardedEngine, lane_range
    from stub_engine import StubEngine
    m = load_model("rte_case5_example")
    made = []

    def factory(model, n, dev, nbb):
        e = StubEngine(model, n_lanes=n, device=dev, n_busbar=nbb)
        made.append(e)
        return e
    se = ShardedEngine(m, 37, devices=[0, 1, 2], engine_factory=factory)
    assert [e.n_lanes for e in made] == [lane_range(37, 3, r)[1] for r in range(3)] and [e.device for e in made] == [0, 1, 2]
    se.set_lane_chronics(lane_offset=np.arange(37))
    r = se.results()
    assert np.array_equal(r.out[:, 0], np.arange(37))                         # global lane order
    r = se.results(10, 20)
    assert np.array_equal(r.out[:, 0], np.arange(10, 30))                     # a range spanning all three shards
    inj = np.arange(37 * se.n_inj, dtype=float).reshape(37, -1)
    se.set_injections(inj[5:30], lane0=5)
    assert np.array_equal(se.get_injections(5, 25), inj[5:30])
    topo = np.tile(m.initial_topo_vect(), (4, 1))
    topo[:, 0] = 2
    se.set_topology(topo, lane0=11)                                           # crosses the shard boundary at lane 13
    assert (se.results(11, 4).topo_vect[:, 0] == 2).all() and se.results(10, 1).topo_vect[0, 0] == 1
    se.step(0)
    se.runpf(12, 2)
    assert [e.n_steps for e in made] == [1, 1, 1] and [e.n_runpf for e in made] == [1, 1, 0]
    eng, l0 = se.owner(36)
    assert eng is made[2] and l0 == made[2].n_lanes - 1
    with pytest.raises(ValueError):
        se.results(30, 10)
    se.close()

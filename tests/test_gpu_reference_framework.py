"""The UNMODIFIED reference framework (grid2op.make -> Environment.step, obs.simulate, N1Reward, Runner, and the reference's own
AAATestBackendAPI kit) on `HipBackend` + libgridpf.so on the MI355X.

grid2op cannot be installed on the GPU box and nothing of the reference may live in the tree, so these tests run only inside a
``tools/gpurun_staged.sh`` call, which copies the reference package into the git-ignored scratch directory ``_stage/`` for that
one call (tools/stage_reference.py) and removes it afterwards; in the driver's own run the directory is absent and the tests
are skipped WITH THAT REASON.  Each test runs tests/reference_on_hip.py in a subprocess: other GPU tests of the same pytest
process put the 120-line grid2op stand-in (tests/grid2op_stub) on sys.path, which must never mix with the real package.
Logs of the last staged run: gpurun_out/reference_on_hip_*.log."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
STAGE = os.environ.get("GRID2OP_STAGE", os.path.join(ROOT, "_stage"))
staged = pytest.mark.skipif(not os.path.isdir(os.path.join(STAGE, "grid2op")),
                            reason="the unmodified reference package is not staged (tools/gpurun_staged.sh stages it for one gpurun call; "
                                   "it cannot be part of the tree and the GPU box has no /root/reference)")


def _run(cmd, timeout=900, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "REFERENCE_ON_HIP_DRYRUN")}
    env["GRID2OP_REFERENCE"] = STAGE
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "reference_on_hip.py"), cmd, *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"reference_on_hip_{cmd}.log"), "w") as f:
        f.write(p.stdout[-20000:] + "\n--- stderr ---\n" + p.stderr[-8000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


@staged
def test_unmodified_reference_framework_on_the_hip_engine():
    """ONE test (one skip line in the driver's run, where the stage cannot exist) for the five legs:"""
    # 1. the 12 recorded episodes re-run inside the real Environment / obs.simulate / N1Reward / Runner on HipBackend
    out = _run("episodes")
    assert "EPISODES OK: 12 episodes" in out
    assert len([l for l in out.splitlines() if "reproduced on the real framework + HIP engine" in l]) == 12
    # 2. the reference's own backend API kit (grid2op/tests/aaa_test_backend_interface.py, 41 tests)
    out = _run("aaa")
    assert "AAA OK" in out and "ran 41, passed 41" in out, out[-2000:]
    # 3. 288 env.step with default parameters + 78 obs.simulate calls, HIP engine vs oracle engine
    assert "LONG OK" in _run("long")
    # 4. l2rpn_wcci_2022 with storage + redispatch + curtailment actions and a bus split
    assert "WCCI OK" in _run("wcci")
    # 5. the reference's DoNothing profiler loop with both engines (numbers: gpurun_out/reference_step_loop_timing.json)
    out = _run("timing", 900, "1000")
    line = [l for l in out.splitlines() if l.startswith("TIMING ")][-1]
    d = json.loads(line[len("TIMING "):])
    assert len(d["rows"]) == 6 and all(r["steps"] == 1000 for r in d["rows"])
    with open(os.path.join(ROOT, "gpurun_out", "reference_step_loop_timing.json"), "w") as f:
        json.dump(d, f, indent=1)

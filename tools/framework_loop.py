#!/usr/bin/env python
"""Through-framework timing (build container, CPU): the reference ``Environment.step`` loop of its own DoNothing profiler
(_profiling/profiler_do_nothing.py:42-65: DoNothingAgent, NO_OVERFLOW_DISCONNECTION=True, AlwaysLegal rules) with the drop-in
façade `HipBackend` over the CPU oracle engine (no GPU here, grid2op is not installable on the GPU box).  It shows what the
single-environment plugin path costs OUTSIDE the power flow -- the Python of BaseEnv.step -- i.e. why the batched API exists.

    python tools/framework_loop.py [env] [n_steps]   -> one JSON line (committed as profiles/r02_framework_loop_cpu.json)"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("GRID2OP_REFERENCE", "/root/reference")
for p in (ROOT, os.path.join(ROOT, "tests"), REFERENCE, os.path.join(ROOT, "tests", "_refshim")):
    sys.path.insert(0, p)
os.environ.setdefault("_GRID2OP_FORCE_TEST", "1")
warnings.filterwarnings("ignore")

import grid2op  # noqa: E402
from grid2op.Agent import DoNothingAgent  # noqa: E402
from grid2op.Parameters import Parameters  # noqa: E402
from grid2op.Rules import AlwaysLegal  # noqa: E402
from conformance_backend import OracleHipBackend  # noqa: E402

env_name = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_case14_sandbox"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
param = Parameters()
param.init_from_dict({"NO_OVERFLOW_DISCONNECTION": True})
env = grid2op.make(env_name, test=True, backend=OracleHipBackend(), param=param, gamerules_class=AlwaysLegal)
agent = DoNothingAgent(action_space=env.action_space)
obs = env.reset()
done, reward, n = False, env.reward_range[0], 0
t0 = time.perf_counter()
while n < n_steps:
    act = agent.act(obs, reward, done)
    obs, reward, done, info = env.step(act)
    n += 1
    if done:
        obs = env.reset()
el = time.perf_counter() - t0
res = {"what": "reference Environment.step loop, DoNothingAgent, NO_OVERFLOW_DISCONNECTION, AlwaysLegal (profiler_do_nothing.py:42-65)",
                  "env": env_name, "backend": "HipBackend facade over the CPU oracle engine (build container, no GPU)", "steps": n,
                  "env_steps_per_sec": n / el, "ms_per_step": el / n * 1e3,
                  "time_powerflow_ms_per_step": env._time_powerflow / n * 1e3, "time_apply_act_ms_per_step": env._time_apply_act / n * 1e3,
                  "time_extract_obs_ms_per_step": env._time_extract_obs / n * 1e3, "time_step_ms_per_step": env._time_step / n * 1e3,
                  "backend_comp_time_ms_per_step": env.backend.comp_time / n * 1e3, "host_cpus": os.cpu_count()}
print(json.dumps({k: (float(v) if hasattr(v, "dtype") else v) for k, v in res.items()}))

"""Developer tool: cost of bus splits in a batch (kernel S switches every lane of a launch to NB = n_busbar blocks as soon as
one lane has a substation with two live busbars).  usage: python tools/split_bench.py [env] [lanes] [fraction_split]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grid2op_amd.grid_model import GridModel
from grid2op_amd.engine import PowerFlowEngine
env = sys.argv[1] if len(sys.argv) > 1 else "l2rpn_case14_sandbox"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
m = GridModel.load_npz(f"tests/golden/{env}.grid.npz")
ch = dict(np.load(f"tests/golden/{env}.chronics.npz"))
if "prod_v" not in ch:
    ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
eng = PowerFlowEngine(m, n_lanes=B)
eng.upload_chronics(eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]))
eng.set_lane_chronics(lane_offset=(7 * np.arange(B)) % ch["load_p"].shape[0])
NS = int(os.environ.get("NSTEPS", "1"))              # environment steps per launch
def run(label):
    for t in range(10): eng.step(t * NS, rebalance=1.02, n_steps=NS)
    eng.sync(); t0 = time.perf_counter()
    N = 100
    for t in range(N): eng.step(t * NS, rebalance=1.02, n_steps=NS)
    eng.sync(); dt = (time.perf_counter() - t0) / (N * NS)
    r = eng.results()
    print(f"{label}: {dt*1e6:.1f} us/step = {B/dt/1e6:.2f} M steps/s, converged {r.converged.mean():.3f}")
run("no split")
# split the substation with the most elements: lines alternate between busbars, injections stay on busbar 1
topo = np.tile(m.initial_topo_vect(), (B, 1))
sub = int(np.argmax(m.sub_info))
start = int(np.concatenate(([0], np.cumsum(m.sub_info)))[sub])
pos = np.arange(start, start + m.sub_info[sub])
line_pos = [p for p in pos if p in set(m.line_or_pos_topo_vect.tolist()) | set(m.line_ex_pos_topo_vect.tolist())]
k = np.random.default_rng(0).random(B) < frac
for p in line_pos[::2][:max(1, len(line_pos) // 2 - 1)]:
    topo[k, p] = 2
# keep at least one line on each busbar and a generator/load on busbar 2 is not required (a pure line bus is PQ with 0 injection)
eng.set_topology(topo)
run(f"{k.mean()*100:.0f} % of the lanes with substation {sub} split")

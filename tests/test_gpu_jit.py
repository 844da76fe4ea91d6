"""GPU tests of the grid-specialised step kernels (gpf_jit_enable, grid2op_amd/csrc/gridpf_jit.hip): the kernel source compiled at
run time with one grid's sizes and offsets as literals must reproduce the shipped (ahead-of-time) kernels BIT FOR BIT -- every
result array, every step of the observation trajectory -- on every kernel variant the batched engine launches (instance groups,
one / two wavefronts per instance, Ybus in registers, topology classes, environment dynamics, DC mode, cascades with auto-reset);
then the parity suites that compare the engine with the oracle and with recordings of the reference run once more with
GRIDPF_JIT=1, i.e. with every engine they create on specialised kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import ROOT  # noqa: E402
from test_gpu_multistep import _setup, _snapshot  # noqa: E402


@pytest.fixture(scope="module")
def jit_cache(tmp_path_factory):
    return str(tmp_path_factory.mktemp("jit_cache"))


def _arrays(eng, n_steps):
    s = _snapshot(eng)
    out = [s[k] for k in ("out", "status", "topo", "ls", "rho", "ovc", "dr", "bus_vm", "inj", "sb")]
    for r in eng.trajectory_obs(n_steps):
        out += [r.out, r.topo_vect, r.status, r.line_status]
    ep = eng.episode()
    return out + [np.asarray(x) for x in ep]


def _run(eng, kw, n_steps, n_launch, before_launch=None):
    eng.set_trajectory(n_steps, eng.TRAJ_OBS)
    got = []
    for k in range(n_launch):
        if before_launch:
            before_launch(eng, k)
        eng.step(k * n_steps, n_steps=n_steps, **kw)
        got += _arrays(eng, n_steps)
    return got


def _assert_same(ref, got, what):
    assert len(ref) == len(got)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), (what, i, a.shape)


@pytest.mark.parametrize("name,B,kw,tight", [
    ("rte_case5_example", 37, dict(rebalance=1.02), 1.0),                                  # 4 instances per wavefront, ragged tail
    ("l2rpn_case14_sandbox", 130, dict(rebalance=1.02, cascade=True, auto_reset=True), 0.8),   # 2 per wavefront, lines trip, lanes restart
    ("l2rpn_case14_sandbox", 64, dict(rebalance=1.02, is_dc=True), 1.0),
    ("l2rpn_neurips_2020_track1", 65, dict(rebalance=1.02, cascade=True, auto_reset=True), 0.9),   # 1 instance per wavefront
    ("l2rpn_wcci_2022_dev", 24, dict(rebalance=1.02), 1.0),                                # 2 wavefronts per instance, Ybus in registers
    ("l2rpn_idf_2023", 16, dict(rebalance=1.02, cascade=True, auto_reset=True), 0.9),
    ("educ_case14_storage", 16, dict(rebalance=1.0), 1.0),
])
def test_specialised_kernels_reproduce_the_shipped_kernels_bit_for_bit(name, B, kw, tight, load_model, load_npz, jit_cache):
    m, ch, e_ref, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e_jit, _, _, _ = _setup(load_model, load_npz, name, B)
    if tight != 1.0 and "thermal_limits" in ch:
        for e in (e_ref, e_jit):
            e.set_thermal_limits(np.asarray(ch["thermal_limits"]) * tight)
    info = e_jit.specialize(True, cache_dir=jit_cache, verify=False)
    assert info["enabled"]
    ref = _run(e_ref, kw, 6, 3)
    got = _run(e_jit, kw, 6, 3)
    info = e_jit.specialization()
    assert info["failed"] == 0 and info["launches"] == 3 and info["compiled"] + info["cached"] + info["aot"] >= 1, info
    assert e_ref.specialization()["launches"] == 0
    _assert_same(ref, got, (name, info["variants"]))
    # back to the shipped kernels: same engine, same state, next launch identical again
    e_jit.specialize(False)
    e_ref.step(18, n_steps=6, **kw)
    e_jit.step(18, n_steps=6, **kw)
    _assert_same(_arrays(e_ref, 6), _arrays(e_jit, 6), (name, "after disable"))
    assert e_jit.specialization()["launches"] == 3


def test_specialised_topology_class_and_mixed_batches(load_model, load_npz, jit_cache):
    """lanes with split substations run the topology-class kernels (their own symbolic program per class, the grid's static tables)"""
    name, B = "l2rpn_case14_sandbox", 48
    m, ch, e_ref, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e_jit, _, _, _ = _setup(load_model, load_npz, name, B)
    topo = np.tile(m.initial_topo_vect(), (B, 1)).astype(np.int32)
    rng = np.random.default_rng(3)
    for lane in range(0, B, 3):                      # every third lane: one substation split over two busbars
        sub = [1, 3, 4, 5][lane % 4]
        lines = np.concatenate([m.line_or_pos_topo_vect[m.line_or_sub == sub], m.line_ex_pos_topo_vect[m.line_ex_sub == sub]])
        pick = rng.permutation(lines)[: max(2, len(lines) // 2)]
        topo[lane, pick] = 2
    for e in (e_ref, e_jit):
        e.set_topology(topo)
    e_jit.specialize(True, cache_dir=jit_cache, verify=False)
    kw = dict(rebalance=1.02)
    ref = _run(e_ref, kw, 5, 3)
    got = _run(e_jit, kw, 5, 3)
    info = e_jit.specialization()
    assert info["failed"] == 0 and info["launches"] >= 3, info
    assert e_jit.plan()["topology_classes"] >= 1 and "true" in info["variants"], (info, e_jit.plan())   # the topology-class variant was specialised
    _assert_same(ref, got, info["variants"])
    conv = ref[1][:, 0] == 0
    assert conv.sum() >= B * 3 // 4


def test_specialised_environment_dynamics_kernel(load_model, load_npz, jit_cache):
    """the ENV instantiation (state of charge, ramp-limited dispatch inside the launch) on the storage grid"""
    from test_gpu_envdyn import _engine
    m = load_model("educ_case14_storage")
    fx = load_npz("envdyn_educ_case14_storage.npz")
    B = 9
    e_ref, e_jit = _engine(m, fx, B), _engine(m, fx, B)
    e_jit.specialize(True, cache_dir=jit_cache, verify=False)
    rng = np.random.default_rng(5)
    disp = np.nonzero(fx["redispatchable"])[0]
    red = np.zeros((B, m.n_gen), np.float32)
    for k in range(B):
        g2 = rng.choice(disp, 2, replace=False)
        amp = np.float32(fx["ramp_up"][g2[0]] * rng.uniform(0.1, 0.5))
        red[k, g2[0]], red[k, g2[1]] = amp, -amp
    sto = rng.uniform(-4, 4, (B, m.n_storage)).astype(np.float32)
    outs = []
    for e in (e_ref, e_jit):
        e.set_trajectory(6, e.TRAJ_OBS)
        got = []
        for k in range(3):
            e.set_lane_actions(red if k != 1 else np.zeros_like(red), sto, hold_storage=True)
            e.step(1 + k * 6, n_steps=6)
            st = e.env_state()
            got += _arrays(e, 6) + [np.asarray(st[key]) for key in sorted(st)]
        outs.append(got)
    info = e_jit.specialization()
    assert info["failed"] == 0 and info["launches"] == 3 and info["variants"].count("true") >= 1, info
    _assert_same(outs[0], outs[1], info["variants"])


@pytest.mark.parametrize("name,B", [("l2rpn_case14_sandbox", 33), ("l2rpn_neurips_2020_track1", 61), ("l2rpn_wcci_2022_dev", 5)])
def test_specialised_runpf_kernels_and_n1_scan(name, B, load_model, load_npz, jit_cache):
    """gpf_runpf / gpf_solve_lane (one power flow per lane: what HipBackend.runpf and an N-1 scan use) on specialised kernels"""
    m, ch, e_ref, tab, off, scale = _setup(load_model, load_npz, name, B)
    _, _, e_jit, _, _, _ = _setup(load_model, load_npz, name, B)
    e_jit.specialize(True, cache_dir=jit_cache, verify=False)
    rng = np.random.default_rng(2)
    inj = e_ref.pack_injections(B)
    inj[:, e_ref.inj_slices["load_p"]] *= 1 + 0.05 * rng.standard_normal((B, m.n_load))
    outs = []
    for e in (e_ref, e_jit):
        e.set_injections(inj)
        e.fanout_n1(0, 1, np.arange(min(B - 1, m.n_line)))          # lanes 1.. = lane 0 with one line out each
        e.runpf()
        r = e.results()
        got = [r.out, r.status, r.topo_vect, r.bus_vm, r.bus_va]
        e.runpf(is_dc=True)
        got += [e.results().out]
        one = e.solve_lane(0, inj[1], np.asarray(m.initial_topo_vect(), np.int32), shunt_bus=np.asarray(m.initial_shunt_bus(), np.int32) if m.n_shunt else None)
        got += [one.out, one.status]
        outs.append(got)
    info = e_jit.specialization()
    assert info["failed"] == 0 and info["launches"] >= 3 and "runpf<" in info["variants"], info
    _assert_same(outs[0], outs[1], (name, info["variants"]))


def test_specialize_self_test_header_and_refusal(load_model, load_npz, jit_cache, monkeypatch):
    from grid2op_amd.engine import GridPFError
    m, ch, eng, tab, off, scale = _setup(load_model, load_npz, "l2rpn_case14_sandbox", 32)
    info = eng.specialize(True, cache_dir=jit_cache)            # verify=True: the 64-lane twin ran both kernel sets and agreed
    assert info["enabled"]
    hdr = eng.specialization_header()
    for macro in ("GPF_JIT_SET_G", "GPF_JIT_SET_OO", "GPF_JIT_SET_SYM", "GPF_JIT_SET_SO"):
        assert f"#define {macro}(v)" in hdr
    assert f"(v).n_sub = {m.n_sub};" in hdr and f"(v).n_line = {m.n_line};" in hdr and f"(v).dim_topo = {m.dim_topo};" in hdr
    eng.step(0, n_steps=4, rebalance=1.02)
    assert eng.specialization()["launches"] == 1
    # no compiler and no ahead-of-time objects for the grid (rte_case5_example is not in grid2op_amd/aot/manifest.json): refused loudly,
    # the engine keeps its shipped kernels and keeps working
    _, _, e2, _, _, _ = _setup(load_model, load_npz, "rte_case5_example", 32)
    _, _, e3, _, _, _ = _setup(load_model, load_npz, "rte_case5_example", 32)
    monkeypatch.setenv("GRIDPF_HIPCC", "/nonexistent/hipcc")
    with pytest.raises(GridPFError, match="no compiler at run time"):
        e2.specialize(True, cache_dir=jit_cache, verify=False)
    monkeypatch.delenv("GRIDPF_HIPCC")
    for e in (e2, e3):
        e.step(0, n_steps=4, rebalance=1.02)
    assert not e2.specialization()["enabled"] and e2.specialization()["launches"] == 0
    assert np.array_equal(e2.results().out, e3.results().out, equal_nan=True) and e2.results().converged.all()


def test_parity_suites_on_specialised_kernels(jit_cache):
    """GRIDPF_JIT=1: every engine the parity tests create switches to specialised kernels at gpf_create -- the comparisons with the
    oracle, with the recorded reference episodes (obs.simulate, injection dynamics, DoNothing rollouts with pandapower's numbers) and
    of multi-step against single-step launches are then comparisons of the SPECIALISED kernels."""
    env = dict(os.environ, GRIDPF_JIT="1", GRIDPF_JIT_CACHE=jit_cache)
    sel = ["tests/test_gpu_multistep.py", "tests/test_gpu_envdyn.py", "tests/test_gpu_simulate.py", "tests/test_gpu_conditioning.py",
           "tests/test_gpu_parity.py::test_more_than_three_busbars_through_topology_classes",
           "tests/test_gpu_bench_parity.py::test_headline_launch_4096_lanes_every_step_vs_oracle"]
    n_before = len([f for f in os.listdir(jit_cache) if f.endswith(".hsaco")])
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", *sel], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=1500)
    tail = p.stdout[-3000:] + p.stderr[-2000:]
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and "failed" not in p.stdout.splitlines()[-1], tail
    n_after = len([f for f in os.listdir(jit_cache) if f.endswith(".hsaco")])
    assert n_after >= n_before + 5, (n_before, n_after)      # the suites' own grids / kernel variants were compiled: the switch was on


def test_bench_grids_run_on_ahead_of_time_objects_without_a_compiler(load_model, load_npz, tmp_path):
    """The grids of grid2op_amd/aot/manifest.json (BASELINE configs) get their specialised kernels from grid2op_amd/_aot -- built by
    __graft_entry__.build() on the CPU box, found by a hash of (generated header, variant, flags, kernel sources) -- even when NO compiler
    is available at run time (GRIDPF_HIPCC pointing nowhere, an empty cache): nothing is compiled, results equal the shipped kernels'.
    (A header that changed since the manifest was recorded simply misses the ahead-of-time objects: then this test reports it.)"""
    import glob
    import json
    import subprocess
    import sys
    if not glob.glob(os.path.join(ROOT, "grid2op_amd", "_aot", "hdr_*.ok")):
        pytest.skip("no ahead-of-time code objects in this tree (__graft_entry__.build() makes them from grid2op_amd/aot/)")
    code = r'''
import json, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import ROOT
from grid2op_amd.grid_model import GridModel
from grid2op_amd.engine import PowerFlowEngine
import os
gold = os.path.join(ROOT, "tests", "golden")
out = {}
for name, B in (("l2rpn_case14_sandbox", 4096), ("l2rpn_wcci_2022_dev", 1024)):
    m = GridModel.load_npz(os.path.join(gold, name + ".grid.npz"))
    ch = dict(np.load(os.path.join(gold, name + ".chronics.npz")))
    if "prod_v" not in ch:
        ch["prod_v"] = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    res = []
    for jit in (False, True):
        eng = PowerFlowEngine(m, n_lanes=B, device=0)
        eng.upload_chronics(eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch["prod_v"]))
        eng.set_lane_chronics(lane_offset=(7 * np.arange(B)).astype(np.int32))
        if jit:
            eng.specialize(True, verify=False)
        eng.set_trajectory(16, eng.TRAJ_OBS)
        eng.step(2, n_steps=16, rebalance=1.02)
        r = eng.results()
        res.append((r.out.copy(), r.status.copy()))
        info = eng.specialization()
        eng.close()
    out[name] = dict(info, same=bool(np.array_equal(res[0][0], res[1][0], equal_nan=True) and np.array_equal(res[0][1], res[1][1])))
print("RESULT " + json.dumps(out))
''' % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, GRIDPF_HIPCC=str(tmp_path / "no_such_hipcc"), GRIDPF_JIT_CACHE=str(tmp_path / "cache"))
    env.pop("GRIDPF_JIT", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for name, info in res.items():
        assert info["enabled"] and info["aot"] >= 1 and info["compiled"] == 0 and info["failed"] == 0 and info["launches"] == 1, (name, info)
        assert info["same"], name

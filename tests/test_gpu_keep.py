"""GPU tests of the reference-topology state shared by ONE-STEP launches (gpf::KeepArgs, gridpf_common.hpp): what a later step of a
multi-step launch finds in LDS / registers (element -> bus maps, bus types, Ybus blocks, DC factors) is loaded from one blob in HBM / L2 by
every lane whose topology row, shunt buses and shunt set-points equal the blob's key -- the engine's pristine lane -- instead of being
rebuilt by every launch: the loop of an agent that acts at every step (reference: Environment/baseEnv.py:3562-3931, one backend call per
step).  The contract: every output BIT-identical -- integers, float32 rows, float64 bus voltages -- to the same
launches with GRIDPF_KEEP=0, whatever happens to the lanes in between: host actions on some lanes (they leave the key and
come back to it), lines tripped by the protections, maintenance, multi-step launches, another kernel writing the lane's outputs, other
shunt set-points."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GRIDS = ["rte_case5_example", "l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev"]      # 4 / 2 / 1 instances per wavefront, 2 wavefronts per instance


def _engine(m, ch, B, keep, prod_v):
    from grid2op_amd.engine import PowerFlowEngine
    old = os.environ.get("GRIDPF_KEEP")
    os.environ["GRIDPF_KEEP"] = "1" if keep else "0"
    try:
        eng = PowerFlowEngine(m, n_lanes=B, device=0)
    finally:
        if old is None:
            del os.environ["GRIDPF_KEEP"]
        else:
            os.environ["GRIDPF_KEEP"] = old
    eng.upload_chronics(eng.pack_chronics(ch["load_p"], ch["load_q"], ch["prod_p"], ch.get("prod_v", prod_v))[:96])
    eng.set_lane_chronics(lane_offset=((11 * np.arange(B)) % 96).astype(np.int32))
    return eng


def _snapshot(eng):
    r = eng.results()
    rho, oc, dr = eng.step_outputs()
    topo, shb = eng.get_topology()
    return dict(out=r.out.copy(), topo_vect=r.topo_vect.copy(), shunt_bus=r.shunt_bus.copy(), line_status=r.line_status.copy(),
                status=r.status.copy(), bus_vm=r.bus_vm.copy(), bus_va=r.bus_va.copy(), rho=rho, oc=oc, dr=dr, topo=topo, shb=shb,
                inj=eng.get_injections(), cool=eng.cooldown())


def _same(a, b, what):
    """BIT-identical, floats included: a step that runs on the kept state leaves the bus sums to K1 (the accumulation order of a launch that
    rebuilds), reads Ybus blocks / DC factors that the same deterministic code produced, and skips nothing else -- so what a launch returns
    does not depend on whether the blob was there (grid2op: same seeds -> same episode, tests/test_gpu_conditioning.py)."""
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (what, k, np.argwhere(~((a[k] == b[k]) | ((a[k] != a[k]) & (b[k] != b[k]))))[:5])


@pytest.mark.parametrize("name", GRIDS)
def test_one_step_launches_with_kept_lane_state_equal_launches_without(name, load_model, load_npz):
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    prod_v = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    B = 37                                                  # an odd count: the last wavefront of the 14-substation kernel has a ghost instance
    engs = [_engine(m, ch, B, keep, prod_v) for keep in (False, True)]
    assert engs[1].plan()["busbars_per_block"] == 1
    t = [0]

    def both(fn, what):
        for e in engs:
            fn(e)
        a, b = (_snapshot(e) for e in engs)
        _same(a, b, what)
        return a

    def steps(k, what, **kw):
        for _ in range(k):
            both(lambda e: e.step(t[0], rebalance=1.02, **kw), f"{what}, t = {t[0]}")
            t[0] += 1

    steps(3, "DoNothing")                                   # launch 1 writes the blob (one lane wins the claim), 2 and 3 load it
    s = both(lambda e: None, "read-back")
    assert (s["status"][:, 0] == 0).all()
    # host actions on SOME lanes: a line out on lanes 1 and 5 (14 substations: lane 1 shares its wavefront with lane 0), later back in
    lines = {1: 3, 5: 7, 20: 3}
    topo0 = s["topo"].copy()

    def open_lines(e):
        for lane, l in lines.items():
            e.disconnect_line(lane, l)
    both(open_lines, "disconnect")
    steps(3, "lines out on three lanes")
    s = both(lambda e: None, "read-back")
    for lane, l in lines.items():
        assert not s["line_status"][lane, l] and s["line_status"][lane - 1].all()
    both(lambda e: e.set_topology(topo0, s["shb"]), "reconnect")
    steps(2, "back on the first topology")                # the three lanes are on the key again
    # another kernel writes the lanes' outputs under another topology, then the rows go back: the blob's key matches again and the step must
    # still publish ITS topology outputs
    def runpf_elsewhere(e):
        tp = topo0.copy()
        tp[:, m.line_or_pos_topo_vect[2]] = -1
        tp[:, m.line_ex_pos_topo_vect[2]] = -1
        e.set_topology(tp, s["shb"])
        e.runpf()
        e.set_topology(topo0, s["shb"])
    both(runpf_elsewhere, "runpf under another topology")
    steps(2, "after a runpf under another topology")
    s = both(lambda e: None, "read-back")
    assert s["line_status"].all() and (s["topo_vect"] == topo0).all()
    # a multi-step launch in between (does not read or write the blobs), then one-step launches again
    for e in engs:
        e.step(t[0], rebalance=1.02, n_steps=4)
    t[0] += 4
    both(lambda e: None, "multi-step launch")
    steps(2, "after a multi-step launch")
    # DC steps (no Ybus blocks: they neither read nor write the blob) and warm-started steps (the kept state holds no voltages) in between
    steps(2, "DC steps", is_dc=True)
    steps(2, "AC again")
    steps(2, "warm start requested", warm_start=True)
    # shunt set-points are part of Ybus, hence of the key
    if m.n_shunt:
        inj = s["inj"].copy()
        sl = engs[0].inj_slices["shunt_q"] if hasattr(engs[0], "inj_slices") else None
        if sl is not None:
            inj[::2, sl] *= 0.5
            both(lambda e: e.set_injections(inj), "shunt set-points")
            steps(2, "other shunt set-points on every second lane")
    # protections: limits low enough that lines trip on some lanes (cascade on): a step that trips leaves no valid blob
    s = both(lambda e: None, "read-back")
    a_or = np.abs(s["out"][:, engs[0].out_slices["a_or"]])
    lim = np.maximum(1.25 * np.median(a_or, axis=0), 1.0).astype(np.float32)
    both(lambda e: e.set_thermal_limits(lim), "limits")
    steps(5, "cascade on", cascade=True, nb_ts_allowed=1)
    s = both(lambda e: None, "read-back")
    assert (~s["line_status"]).any(), "no line tripped: the scenario does not exercise the invalidation"
    steps(2, "cascade off again")
    # maintenance table: the kernel itself takes lines out at the start of a step
    mt = np.zeros((96, m.n_line), np.uint8)
    mt[(t[0] + 1) % 96:, 4] = 1
    both(lambda e: (e.set_thermal_limits(np.full(m.n_line, 1e9, np.float32)), e.reset(), e.upload_maintenance(mt)), "maintenance table")
    steps(4, "maintenance from the second step on")
    s = both(lambda e: None, "read-back")
    assert (~s["line_status"][:, 4]).any()                 # (the lanes read different rows of the table: some are past the start of the outage)


def test_acting_agents_on_118_substations_with_kept_state(load_model, load_npz):
    """The bench's acting loop (bench.py acting_every_step: a new redispatch + storage action per lane and step, one one-step launch per
    env step, injection dynamics on the device) with and without the kept state: every observation row, the dynamics' state and the episode flags are
    bit-identical."""
    from test_gpu_envdyn import _engine as env_engine
    name = "l2rpn_wcci_2022_dev"
    m = load_model(name)
    fx = load_npz(f"envdyn_{name}.npz")
    B = 24
    rng = np.random.default_rng(5)
    red = np.zeros((B, m.n_gen), np.float32)
    disp = np.flatnonzero(fx["redispatchable"])
    for k in range(B):
        i, j = rng.choice(disp, 2, replace=False)
        red[k, i], red[k, j] = 1.0, -1.0
    sto = rng.uniform(-2, 2, (B, max(m.n_storage, 1))).astype(np.float32)
    outs = []
    for keep in (False, True):
        old = os.environ.get("GRIDPF_KEEP")
        os.environ["GRIDPF_KEEP"] = "1" if keep else "0"
        try:
            eng = env_engine(m, fx, B)
        finally:
            if old is None:
                del os.environ["GRIDPF_KEEP"]
            else:
                os.environ["GRIDPF_KEEP"] = old
        row0 = int(fx["row"][0])
        eng.set_env_state(0, prev_p=np.tile(fx["ch_prod_p"][row0 - 1], (B, 1)))
        eng.set_lane_chronics(lane_offset=(np.arange(B) % 5).astype(np.int32))
        seq = []
        for k in range(8):
            eng.set_lane_actions(red if k % 2 == 0 else -red, sto)
            eng.step(row0 + k, rebalance=0.0)
            r = eng.results()
            st = eng.env_state()
            seq.append((r.out.copy(), r.status.copy(), r.topo_vect.copy(), {k_: np.array(v) for k_, v in st.items()}))
        outs.append(seq)
    for k, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a[0], b[0], equal_nan=True), k
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), k
        for key in a[3]:
            assert np.array_equal(a[3][key], b[3][key], equal_nan=True), (k, key)
    assert (outs[1][-1][1][:, 0] == 0).all()


@pytest.mark.parametrize("name", GRIDS)
def test_multi_step_launch_is_bit_identical_to_one_step_launches(name, load_model, load_npz):
    """K1 (a step that rebuilds) and K9 (a step whose topology stands) add the set-points of a bus in the same order -- loads, generators,
    storages --, and everything downstream is the same deterministic arithmetic on the same numbers: the observation of step k of ONE
    launch equals, bit for bit, the observation of the k-th of k one-step launches that rebuild everything (GRIDPF_KEEP=0)."""
    m = load_model(name)
    ch = dict(load_npz(f"{name}.chronics.npz"))
    prod_v = np.tile((m.gen_vm0 * m.sub_vn_kv[m.gen_sub]).astype(np.float32), (ch["prod_p"].shape[0], 1))
    B, K = 21, 5
    one, multi = _engine(m, ch, B, False, prod_v), _engine(m, ch, B, True, prod_v)
    multi.set_trajectory(K, multi.TRAJ_OBS)
    multi.step(3, rebalance=1.02, n_steps=K)
    obs = multi.trajectory_obs(K)
    for k in range(K):
        one.step(3 + k, rebalance=1.02)
        r = one.results()
        assert (r.status[:, 0] == 0).all()
        assert np.array_equal(obs[k].out, r.out, equal_nan=True), k
        assert np.array_equal(obs[k].topo_vect, r.topo_vect) and np.array_equal(obs[k].line_status, r.line_status), k
    rm = multi.results()
    assert np.array_equal(rm.bus_vm, r.bus_vm, equal_nan=True) and np.array_equal(rm.bus_va, r.bus_va, equal_nan=True)
    assert np.array_equal(rm.status, r.status)

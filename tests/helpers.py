"""Shared helpers of the parity tests (state packing, random lane-state generator)."""
import numpy as np

from oracle.pf_oracle import LaneState

F32_FIELDS = ["p_or", "q_or", "v_or", "a_or", "theta_or", "p_ex", "q_ex", "v_ex", "a_ex", "theta_ex",
              "gen_p", "gen_q", "gen_v", "gen_theta", "load_p", "load_q", "load_v", "load_theta",
              "storage_p", "storage_q", "storage_v", "storage_theta", "shunt_p", "shunt_q", "shunt_v"]


def pack_states(m, states):
    inj = np.stack([np.concatenate([s.gen_p, s.gen_vm, s.load_p, s.load_q, s.storage_p, s.storage_q, s.shunt_p, s.shunt_q])
                    for s in states])
    topo = np.stack([s.topo for s in states])
    sb = np.stack([s.shunt_bus for s in states]) if m.n_shunt else None
    return inj, topo, sb


def random_states(m, n, rng, n_busbar=2, p_split=0.3, p_line_off=0.3):
    states = []
    for _ in range(n):
        s = LaneState.from_model(m)
        s.load_p = (s.load_p * rng.uniform(0.7, 1.2, m.n_load)).astype(np.float32).astype(np.float64)
        s.load_q = (s.load_q * rng.uniform(0.7, 1.2, m.n_load)).astype(np.float32).astype(np.float64)
        s.gen_p = (s.gen_p * rng.uniform(0.8, 1.1, m.n_gen)).astype(np.float32).astype(np.float64)
        s.gen_vm = s.gen_vm * rng.uniform(0.98, 1.02, m.n_gen)
        if m.n_storage:
            s.storage_p = rng.uniform(-3, 3, m.n_storage)
        if rng.random() < p_line_off:
            for l in rng.choice(m.n_line, size=rng.integers(1, 3), replace=False):
                s.topo[m.line_or_pos_topo_vect[l]] = -1
                s.topo[m.line_ex_pos_topo_vect[l]] = -1
        if rng.random() < p_split:
            # move a random subset of the elements of one substation to busbar 2
            sub = rng.integers(0, m.n_sub)
            start = int(np.concatenate(([0], np.cumsum(m.sub_info)))[sub])
            pos = np.arange(start, start + m.sub_info[sub])
            mv = pos[rng.random(len(pos)) < 0.5]
            s.topo[mv] = np.where(s.topo[mv] >= 1, 2, s.topo[mv])
        if m.n_shunt and rng.random() < 0.2:
            s.shunt_bus[rng.integers(0, m.n_shunt)] = rng.choice([-1, 2])
        states.append(s)
    return states



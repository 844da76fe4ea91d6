#!/usr/bin/env python
"""Record the inputs / outputs of the reference's redispatching automaton (build container only).

    python tests/golden/make_redispatch_fixtures.py     # -> tests/golden/redispatch_cases.npz

``BaseEnv._compute_dispatch_vect`` (grid2op/Environment/baseEnv.py:2211-2470: the SLSQP projection of the agents' redispatch /
storage actions onto pmin / pmax / ramp limits with a zero-sum constraint) is wrapped inside UNMODIFIED reference environments
driven with random redispatch and storage actions; every call is logged: new_p, the previous set-points, actual / target
dispatch before, the "already modified" mask, the storage amount -> actual dispatch after (or the refusal).  The fixtures pin
oracle/redispatch_oracle.py (CPU test) and are what the device kernel gpf_redispatch is compared with (GPU test)."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REFERENCE = os.environ.get("GRID2OP_REFERENCE", "/root/reference")
for p in (ROOT, os.path.join(ROOT, "tests"), REFERENCE, os.path.join(ROOT, "tests", "_refshim")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("_GRID2OP_FORCE_TEST", "1")
warnings.filterwarnings("ignore")

import grid2op  # noqa: E402
from grid2op.Opponent import BaseOpponent  # noqa: E402
from grid2op.Action import DontAct  # noqa: E402
from grid2op.Parameters import Parameters  # noqa: E402

from conformance_backend import OracleHipBackend  # noqa: E402

LOG = []


def record(env_name, n_steps, seed, big):
    p = Parameters()
    p.NO_OVERFLOW_DISCONNECTION = True
    env = grid2op.make(env_name, test=True, backend=OracleHipBackend(), param=p, opponent_class=BaseOpponent,
                       opponent_action_class=DontAct, opponent_init_budget=0.0, opponent_budget_per_ts=0.0)
    cls = type(env)
    orig = env._compute_dispatch_vect

    def wrapped(already_modified_gen, new_p):
        rec = dict(env=env_name, new_p=np.array(new_p, np.float64), prev_p=np.array(env._gen_activeprod_t_redisp, np.float64),
                   actual=np.array(env._actual_dispatch, np.float64), target=np.array(env._target_dispatch, np.float64),
                   modified=np.array(already_modified_gen, bool), storage=float(env._amount_storage),
                   curtail=float(env._sum_curtailment_mw), detached=float(env._detached_elements_mw), first=env.nb_time_step == 0)
        exc = orig(already_modified_gen, new_p)
        rec["ok"] = exc is None
        rec["actual_after"] = np.array(env._actual_dispatch, np.float64)
        LOG.append(rec)
        return exc
    env._compute_dispatch_vect = wrapped
    env.seed(seed)
    obs = env.reset()
    rng = np.random.default_rng(seed)
    disp = np.nonzero(cls.gen_redispatchable)[0]
    for t in range(n_steps):
        act = {}
        if rng.random() < 0.6:
            k = rng.choice(disp, size=min(len(disp), int(rng.integers(1, 4))), replace=False)
            amp = (cls.gen_max_ramp_up[k] * rng.uniform(-1.0, 1.0, len(k)) * (3.0 if big and rng.random() < 0.3 else 0.6))
            act["redispatch"] = [(int(g), float(a)) for g, a in zip(k, amp)]
        if cls.n_storage and rng.random() < 0.5:
            act["set_storage"] = [(int(i), float(rng.uniform(-3, 3))) for i in range(cls.n_storage)]
        obs, _, done, info = env.step(env.action_space(act))
        if done:
            obs = env.reset()
    meta = dict(pmin=cls.gen_pmin.astype(np.float64), pmax=cls.gen_pmax.astype(np.float64), ramp_up=cls.gen_max_ramp_up.astype(np.float64),
                ramp_down=cls.gen_max_ramp_down.astype(np.float64), redispatchable=cls.gen_redispatchable.astype(bool),
                eps_poly=float(env._epsilon_poly), tol_poly=float(env._tol_poly))
    env.close()
    return meta


def main():
    out = {}
    for env_name, n_steps, seed, big in (("l2rpn_case14_sandbox", 120, 0, True), ("l2rpn_wcci_2022_dev", 80, 1, True),
                                         ("educ_case14_storage", 80, 2, False)):
        n0 = len(LOG)
        meta = record(env_name, n_steps, seed, big)
        recs = LOG[n0:]
        tag = env_name + "__"
        for k, v in meta.items():
            out[tag + k] = np.asarray(v)
        for k in ("new_p", "prev_p", "actual", "target", "modified", "actual_after"):
            out[tag + k] = np.stack([r[k] for r in recs])
        for k in ("storage", "curtail", "detached", "ok", "first"):
            out[tag + k] = np.array([r[k] for r in recs])
        print(env_name, len(recs), "calls,", int((~out[tag + "ok"]).sum()), "refused, max |dispatch|", np.abs(out[tag + "actual_after"]).max())
    np.savez_compressed(os.path.join(HERE, "redispatch_cases.npz"), **out)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised parity sweep of the planner hooks (dev aid, GPU box): _generate_legal, _compute_prob and the fused rollouts
(random depths incl. non-multiples of 4, sims per root, all-actions policy, lane offsets, call counters) — HIP vs oracle.
usage: python tools/gpu_fuzz_planner.py [seconds]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import gym_pomdp_amd as gpa  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402
from tools.gpu_fuzz import CONFIGS  # noqa: E402


def main(budget):
    rs = np.random.RandomState(int(time.time()) & 0xFFFFFF)
    t_end, cases = time.time() + budget, 0
    while time.time() < t_end:
        name, env_id, kw = CONFIGS[rs.randint(len(CONFIGS))]
        n = int(rs.randint(2, 600))
        lane0 = int(rs.randint(0, 1 << 29)) * 4
        seed = int(rs.randint(1 << 62))
        t0 = int(rs.randint(1 << 40)) if rs.rand() < 0.5 else int(rs.randint(100))
        e = gpa.make(env_id, batch_size=n, seed=seed, lane_offset=lane0, **kw)
        e.call_counter = t0
        o = ol.OracleEnv(name, **kw)
        st = o.new_state(n)
        o.batch_reset(st, seed, lane0, t0, nthreads=4)
        e.reset()
        for k in range(int(rs.randint(0, 12))):                      # move off the start states
            a = rs.randint(o.n_actions, size=n).astype(np.int32)
            o.batch_step(st, a, seed, lane0, t0 + 1 + k, nthreads=4)
            e.step(torch.as_tensor(a, device="cuda"))
        assert np.array_equal(e.state.cpu().numpy().view(np.uint32), st)
        ctx = (name, kw, n, lane0, seed, t0)
        lst, ln = e.legal_actions()
        lo, lno = o.batch_legal(st)
        assert np.array_equal(ln.cpu().numpy(), lno), ctx
        assert np.array_equal(lst.cpu().numpy(), lo[:, : lst.shape[1]]), ctx
        a = rs.randint(o.n_actions, size=n).astype(np.int32)
        ob = rs.randint(o.n_obs, size=n).astype(np.int32)
        pg = e.compute_prob(torch.as_tensor(a, device="cuda"), torch.as_tensor(ob, device="cuda")).cpu().numpy()
        assert np.array_equal(pg, o.batch_compute_prob(st, a, ob)), ctx
        depth, sims, alla = int(rs.randint(0, 23)), int(rs.randint(1, 9)), bool(rs.rand() < 0.3)
        disc = float(rs.choice([1.0, .95, .5]))
        tc = e.call_counter
        lo2 = int(rs.randint(0, 1 << 28)) * 4
        g = e.rollout(depth, sims_per_root=sims, discount=disc, all_actions=alla, lane_offset=lo2)
        w = o.batch_rollout(st, sims, depth, disc, seed, lo2, tc, all_actions=alla, nthreads=4)
        for k in ("ret", "n_steps", "first_action", "last_ob"):
            assert np.array_equal(g[k].cpu().numpy(), w[k]), ctx + (k, depth, sims, alla)
        assert np.array_equal(g["terminated"].cpu().numpy(), w["terminated"].astype(bool)), ctx
        cases += 1
        del e
    print("planner fuzz ok: %d random cases in %.0f s" % (cases, budget))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)

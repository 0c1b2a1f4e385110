#!/usr/bin/env python3
"""Randomised parity sweep of the planner hooks (dev aid, GPU box): _generate_legal, _compute_prob and the fused rollouts
(random depths incl. non-multiples of 4, sims per root, all-actions policy, lane offsets, call counters) and the planning step on
top of them (pomdp_plan: action values bit for bit, argmax, the roots' real step) — HIP vs oracle.
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
        # ... and the planning step built on them (pomdp_plan): per-root action values in the stated float64 order, argmax, the
        # roots' real step — simulation counts on both sides of the 64-simulation chunks and of the 1024-simulation LDS tile
        sims_p = int(rs.choice([1, 3, 63, 64, 65, 100, 130, 1030])) if n <= 64 else int(rs.randint(1, 70))
        if (lane0 * sims_p) % 4 == 0 and (lane0 + n) * sims_p < 1 << 32:
            tc = e.call_counter
            r = o.batch_rollout(st, sims_p, depth, disc, seed, lane0 * sims_p, tc, all_actions=alla, nthreads=4)
            want = ol.plan_reduce(r["ret"], r["first_action"], n, sims_p, o.n_actions)
            if (want["best"] >= 0).all():
                e.auto_reset = True
                ob_g, rew_g, done_g, _, p = e.plan_step(depth, sims_per_root=sims_p, discount=disc, all_actions=alla)
                ob_o, rew_o, done_o, bad = o.batch_step(st, want["best"], seed, lane0, tc + depth, auto_reset=True, nthreads=4)
                assert np.array_equal(ob_g.cpu().numpy(), ob_o) and np.array_equal(rew_g.cpu().numpy(), rew_o), ctx + ("plan step",)
                assert np.array_equal(e.state.cpu().numpy().view(np.uint32), st), ctx + ("plan state",)
            else:
                p = e.plan(depth, sims_per_root=sims_p, discount=disc, all_actions=alla)
            for k in ("q", "value"):
                assert np.array_equal(p[k].cpu().numpy().view(np.uint64), want[k].view(np.uint64)), ctx + ("plan", k, depth, sims_p)
            for k in ("visits", "best"):
                assert np.array_equal(p[k].cpu().numpy(), want[k]), ctx + ("plan", k, depth, sims_p)
        cases += 1
        del e
    print("planner fuzz ok: %d random cases in %.0f s" % (cases, budget))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)

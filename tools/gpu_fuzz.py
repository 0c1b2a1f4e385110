#!/usr/bin/env python3
"""Randomised parity sweep (dev aid, GPU box): random env configs, batch sizes (both launch geometries), lane offsets,
call counters, auto-reset on/off, valid and invalid actions — HIP path vs the oracle, word for word; one case in eight
is a trajectory collection (fused launches, up to 2^20 + 2048 lanes, a random trajectory layout or the returns-only sink;
FUZZ_COLLECT sets the share) checked row by row against the oracle.
usage: python tools/gpu_fuzz.py [seconds [seed]]      (tests/test_gpu_parity.py runs a 30-second sweep on a fixed seed)"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import gym_pomdp_amd as gpa  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402

CONFIGS = [
    ("rock", "Rock-v0", {}), ("rock", "Rock-v0", dict(board_size=7, num_rocks=7)),
    ("rock", "Rock-v0", dict(board_size=11, num_rocks=11)), ("rock", "Rock-v0", dict(board_size=15, num_rocks=15)),
    ("rock", "Rock-v0", dict(board_size=4, num_rocks=3)), ("stochrock", "StochasticRock-v0", {}),
    ("tag", "Tag-v0", {}), ("tag", "Tag-v0", dict(num_opponents=2)), ("tag", "Tag-v0", dict(num_opponents=4)),
    ("tag", "Tag-v0", dict(move_prob=.3)), ("tag", "Tag-v0", dict(num_opponents=3, move_prob=.6)),
    ("battleship", "Battleship-v0", {}), ("battleship", "Battleship-v0", dict(board_size=(10, 10), max_len=5)),
    ("battleship", "Battleship-v0", dict(board_size=(8, 6), max_len=4)),
    ("tiger", "Tiger-v0", {}), ("network", "Network-v0", {}), ("network", "Network-v0", dict(n_machines=16, problem_type=1)),
    ("network", "Network-v0", dict(n_machines=31, problem_type=3)),
]


def collect_case(rs):
    """Trajectory collection (fused launches; from 2^20 lanes RockSample's four-lanes-per-thread loop with the table-driven
    lane step) against the oracle stepped with the synthetic policy's actions, row by row."""
    from oracle import philox_ref as px
    name, env_id, kw = CONFIGS[rs.randint(len(CONFIGS))]
    pick = rs.rand()
    # full workgroups at and above the gates of the quad-per-thread loops (2^18: Tiger, Network; 2^19: RockSample, Tag), ragged
    # batches around 2^18, small batches (any size: the fused drivers generate their own first actions)
    n = (1 << 20) + 1024 * int(rs.randint(0, 3)) if pick < 0.3 else \
        (1 << int(rs.randint(18, 20))) + 1024 * int(rs.randint(0, 3)) if pick < 0.5 else \
        int(rs.randint(1 << 18, (1 << 18) + 3000)) if pick < 0.65 else int(rs.randint(2, 6000))   # 1 is scalar mode
    lane0 = int(rs.randint(0, 1 << 30)) * 4 % ((1 << 32) - n - 8) // 4 * 4
    seed = int(rs.randint(1 << 62))
    t0 = int(rs.randint(1 << 40)) if rs.rand() < 0.5 else int(rs.randint(100))
    steps = int(rs.randint(1, 24)) if n >= (1 << 20) else int(rs.randint(1, 80))
    e = gpa.make(env_id, batch_size=n, seed=seed, lane_offset=lane0, **kw)
    e.call_counter = t0
    o = ol.OracleEnv(name, **kw)
    st = o.new_state(n)
    ob_o = o.batch_reset(st, seed, lane0, t0, nthreads=8)
    assert np.array_equal(e.reset().cpu().numpy(), ob_o), (name, kw, n, "reset ob")
    layout = ("columns", "blocked", "packed", "narrow", "returns")[rs.randint(5)]    # the sink too
    if layout == "returns":                               # no trajectory: the per-lane episode statistics (pomdp_collect_returns)
        stats = e.collect_returns(steps)
        acc, cnt = ol.new_return_stats(n)
        o.batch_collect_returns(st, acc, cnt, e._discount, seed, lane0, t0 + 1, steps, nthreads=8)
        ctx = ("returns", name, kw, n, lane0, seed, t0, steps)
        assert np.array_equal(stats.acc[:, :n].cpu().numpy().view(np.uint64), acc.view(np.uint64)), ctx
        assert np.array_equal(stats.cnt[:, :n].cpu().numpy(), cnt), ctx
        assert np.array_equal(e.state.cpu().numpy().view(np.uint32), st), ctx
        return
    tr = e.decode_trajectory(e.collect_synthetic(steps, layout=layout))
    done = np.zeros(n, np.uint8)
    for k in range(steps):
        t = t0 + 1 + k
        a = px.synthetic_actions(seed, lane0, n, t, o.n_actions)
        ob_o, rew_o, done, bad = o.batch_step(st, a, seed, lane0, t, auto_reset=True, done=done, nthreads=8)
        ctx = ("collect", layout, name, kw, n, lane0, seed, t0, steps, k)
        assert bad == 0, ctx
        assert np.array_equal(tr["action"][k].cpu().numpy(), a), ctx
        assert np.array_equal(tr["ob"][k].cpu().numpy(), ob_o), ctx
        assert np.array_equal(tr["reward"][k].cpu().numpy(), rew_o), ctx
        assert np.array_equal(tr["done"][k].cpu().numpy(), done.astype(bool)), ctx
    assert np.array_equal(e.state.cpu().numpy().view(np.uint32), st), ("collect", name, kw, n, "state")


def main(budget, seed=None):
    rs = np.random.RandomState(int(time.time()) & 0xFFFFFF if seed is None else seed)
    t_end, cases = time.time() + budget, 0
    while time.time() < t_end:
        if rs.rand() < float(os.environ.get("FUZZ_COLLECT", "0.12")):
            collect_case(rs)
            cases += 1
            continue
        name, env_id, kw = CONFIGS[rs.randint(len(CONFIGS))]
        big = rs.rand() < 0.3
        n = int(rs.randint(1 << 18, (1 << 18) + 3000)) if big else int(rs.randint(2, 6000))
        lane0 = int(rs.randint(0, 1 << 30)) * 4 % ((1 << 32) - n - 8)
        if big and rs.rand() < 0.4:             # full 1024-lane workgroups from 2^19 lanes: env.step()'s quad-per-thread kernel
            n = (1 << 19) + 1024 * int(rs.randint(0, 40))
            lane0 = lane0 // 4 * 4 if rs.rand() < 0.8 else lane0
        seed = int(rs.randint(1 << 62))
        t0 = int(rs.randint(1 << 40)) if rs.rand() < 0.5 else int(rs.randint(100))
        auto = bool(rs.rand() < 0.7)
        steps = 6 if big else 30
        e = gpa.make(env_id, batch_size=n, seed=seed, lane_offset=lane0, auto_reset=auto, **kw)
        e.call_counter = t0
        o = ol.OracleEnv(name, **kw)
        st = o.new_state(n)
        ob_o = o.batch_reset(st, seed, lane0, t0, nthreads=8)
        ob_g = e.reset()
        assert np.array_equal(ob_g.cpu().numpy(), ob_o), (name, kw, n, "reset ob")
        done = np.zeros(n, np.uint8)
        bad_total = 0
        for k in range(steps):
            t = t0 + 1 + k
            a = rs.randint(o.n_actions, size=n).astype(np.int32)
            if rs.rand() < 0.3:                                  # a few out-of-range actions
                idx = rs.randint(n, size=max(1, n // 500))
                a[idx] = rs.choice([-1, o.n_actions, 1 << 20], size=len(idx))
            ob_o, rew_o, done, bad = o.batch_step(st, a, seed, lane0, t, auto_reset=auto, done=done, nthreads=8)
            bad_total += bad
            ob_g, rew_g, done_g, _ = e.step(torch.as_tensor(a, device="cuda"))
            ctx = (name, kw, n, lane0, seed, t, auto)
            assert np.array_equal(ob_g.cpu().numpy(), ob_o), ctx
            assert np.array_equal(rew_g.cpu().numpy(), rew_o), ctx
            assert np.array_equal(done_g.cpu().numpy(), done.astype(bool)), ctx
            assert np.array_equal(e.state.cpu().numpy().view(np.uint32), st), ctx
        assert e.invalid_action_count() == bad_total, (name, kw, n)
        cases += 1
        del e
    print("fuzz ok: %d random cases in %.0f s" % (cases, budget))
    return cases


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else None)

#!/usr/bin/env python3
"""Randomised parity sweep (dev aid, GPU box): random env configs, batch sizes (both launch geometries), lane offsets,
call counters, auto-reset on/off, valid and invalid actions — HIP path vs the oracle, word for word; one case in eight
is a trajectory collection (fused launches, up to 2^20 + 2048 lanes, a random trajectory layout or the returns-only sink,
half of them driven by a random tape of the caller's actions — out-of-range bytes included — instead of the synthetic policy;
FUZZ_COLLECT sets the share) checked row by row against the oracle.
usage: python tools/gpu_fuzz.py [seconds [seed [cases]]]      (tests/test_gpu_parity.py runs a fixed number of cases on a fixed seed)"""
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
    ("stochrock", "StochasticRock-v0", dict(p_move=.4)), ("stochrock", "StochasticRock-v0", dict(board_size=11, num_rocks=11, p_move=.65)),
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
    # full workgroups at and above the gates of the quad-per-thread loops (2^18: Tiger, Network; 2^19: RockSample, Tag), across
    # the half-quad-per-thread gates (3 * 2^17 .. 3 * 2^18 lanes), ragged batches around 2^18, small batches (any size: the fused drivers generate their own first actions)
    n = (1 << 20) + 1024 * int(rs.randint(0, 3)) if pick < 0.3 else \
        (1 << int(rs.randint(18, 20))) + 1024 * int(rs.randint(0, 3)) if pick < 0.5 else \
        1024 * int(rs.randint(384, 769)) if pick < 0.58 else \
        int(rs.randint(1 << 18, (1 << 18) + 3000)) if pick < 0.7 else int(rs.randint(2, 6000))   # 1 is scalar mode
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
    # ... and where the actions come from: the synthetic policy, or a tape of the caller's (pomdp_collect_tape*), one byte in
    # ~2000 out of range
    tape = None
    if rs.rand() < 0.5:
        tape = rs.randint(0, o.n_actions, (steps, n)).astype(np.uint8)
        bad = rs.randint(0, 2000, (steps, n)) == 0
        tape[bad] = rs.randint(o.n_actions, 256, int(bad.sum())).astype(np.uint8)
        d_tape = torch.as_tensor(tape, device="cuda")
    if layout == "returns":                               # no trajectory: the per-lane episode statistics (pomdp_collect_returns)
        stats = e.collect_returns(steps) if tape is None else e.collect_tape(d_tape, layout="returns")
        acc, cnt = ol.new_return_stats(n)
        o.batch_collect_returns(st, acc, cnt, e._discount, seed, lane0, t0 + 1, steps, nthreads=8,
                                actions=None if tape is None else tape.astype(np.int32))
        ctx = ("returns", name, kw, n, lane0, seed, t0, steps)
        assert np.array_equal(stats.acc[:, :n].cpu().numpy().view(np.uint64), acc.view(np.uint64)), ctx
        assert np.array_equal(stats.cnt[:, :n].cpu().numpy(), cnt), ctx
        assert np.array_equal(e.state.cpu().numpy().view(np.uint32), st), ctx
        return
    tr = e.decode_trajectory(e.collect_synthetic(steps, layout=layout) if tape is None else e.collect_tape(d_tape, layout=layout), steps)
    done, n_bad = np.zeros(n, np.uint8), 0
    for k in range(steps):
        t = t0 + 1 + k
        a = px.synthetic_actions(seed, lane0, n, t, o.n_actions) if tape is None else tape[k].astype(np.int32)
        ob_o, rew_o, done, bad = o.batch_step(st, a, seed, lane0, t, auto_reset=True, done=done, nthreads=8)
        ctx = ("collect", layout, "tape" if tape is not None else "synthetic", name, kw, n, lane0, seed, t0, steps, k)
        n_bad += bad
        assert tape is not None or bad == 0, ctx
        if tape is None or layout != "columns":            # a tape-driven column collection has no action column (the caller holds it)
            assert np.array_equal(tr["action"][k].cpu().numpy(), a), ctx
        assert np.array_equal(tr["ob"][k].cpu().numpy(), ob_o), ctx
        assert np.array_equal(tr["reward"][k].cpu().numpy(), rew_o), ctx
        assert np.array_equal(tr["done"][k].cpu().numpy(), done.astype(bool)), ctx
    assert np.array_equal(e.state.cpu().numpy().view(np.uint32), st), ("collect", name, kw, n, "state")
    assert e.invalid_action_count() == n_bad, ("collect", name, kw, n, "invalid actions")


def main(budget, seed=None, n_cases=None):
    """`budget` seconds of random cases, or — n_cases given — exactly that many, however long they take (what the test suite
    runs: the same cases on every machine)"""
    rs = np.random.RandomState(int(time.time()) & 0xFFFFFF if seed is None else seed)
    t_start, cases = time.time(), 0
    while (cases < n_cases) if n_cases else (time.time() < t_start + budget):
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
            lane0 = min(lane0, (1 << 32) - n - 8)               # (n grew after lane0 was drawn)
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
    print("fuzz ok: %d random cases in %.0f s" % (cases, time.time() - t_start))
    return cases


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else None,
         int(sys.argv[3]) if len(sys.argv) > 3 else None)

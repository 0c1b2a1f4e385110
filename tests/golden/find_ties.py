#!/usr/bin/env python3
"""Find lanes whose RockSample draws are *ties* of the split word layout — the high word alone leaves the
comparison undecided (probability 2^-27 per draw), so the kernels must generate the low-word block — and record the
reference's behaviour on exactly those lanes (fixture ties_rock.npz).  Container-only (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/find_ties.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
from oracle import philox_ref as px  # noqa: E402
from oracle.ref_harness import harness as h  # noqa: E402

SEED = 0x7157157
CHUNK = 1 << 21


def blocks(c0, t, stream, block):
    ctr = np.zeros((len(c0), 4), dtype=np.uint64)
    ctr[:, 0] = c0
    ctr[:, 1] = t & 0xFFFFFFFF
    ctr[:, 2] = t >> 32
    ctr[:, 3] = (stream << 24) | block
    key = np.array([SEED & 0xFFFFFFFF, SEED >> 32], dtype=np.uint64)
    return px.philox4x32_10(ctr, key)


def main(n_reset=6, n_sensor=8):
    thr = json.load(open(os.path.join(HERE, "thresholds.json")))["rock_thr"]
    # first step from the start cell (0,3) of RockSample(7,8): L1 distance to rock i
    rocks = [(2, 0), (0, 1), (3, 1), (6, 3), (2, 4), (3, 4), (5, 5), (1, 6)]
    dist = [abs(0 - x) + abs(3 - y) for x, y in rocks]
    thr_hi = {d: thr[d] >> 26 for d in set(dist)}
    reset_ties, sensor_ties = [], []
    lane = 0
    while len(reset_ties) < n_reset or len(sensor_ties) < n_sensor:
        c0 = np.arange(lane, lane + CHUNK, dtype=np.uint64)
        if len(reset_ties) < n_reset:                                    # reset at t = 0: block 0 of quad c0, element e = lane c0 * 4 + e
            blk = blocks(c0, 0, px.STREAM_RESET, 0).astype(np.uint64)
            for j in range(8):                                           # rock j: the lane's word rotated right by 2 j + 2
                r = np.uint64(2 * j + 2)
                kh = (((blk >> r) | (blk << (np.uint64(32) - r))) & np.uint64(0xFFFFFFFF)) >> np.uint64(5)
                for qi, e in zip(*np.nonzero(kh == (1 << 26))):
                    reset_ties.append((int(c0[qi]) * 4 + int(e), j))
        if len(sensor_ties) < n_sensor:                                  # first step at t = 1: block 0 of quad c0
            kh = blocks(c0, 1, px.STREAM_STEP, 0) >> np.uint32(5)
            for d, th in thr_hi.items():
                for qi, e in zip(*np.nonzero(kh == th)):
                    sensor_ties.append((int(c0[qi]) * 4 + int(e), d))
        lane += CHUNK
        print("searched", lane, "found", len(reset_ties), len(sensor_ties), flush=True)
    reset_ties, sensor_ties = reset_ties[:n_reset], sensor_ties[:n_sensor]
    # reference behaviour on those lanes: reset (t = 0), then one CHECK of a rock at the tied distance (t = 1)
    lanes, actions = [], []
    for ln, _ in reset_ties:
        lanes.append(ln)
        actions.append(5)
    for ln, d in sensor_ties:
        lanes.append(ln)
        actions.append(5 + dist.index(d))
    tr = h.trace_mode_b("rock", {}, SEED, lanes, np.array(actions).reshape(-1, 1), t0=0)
    np.savez_compressed(os.path.join(HERE, "ties_rock.npz"), seed=np.int64(SEED), lanes=np.array(lanes, np.int64),
                        actions=np.array(actions, np.int64), n_reset=np.int64(len(reset_ties)),
                        tied_rock=np.array([r for _, r in reset_ties], np.int64),
                        tied_dist=np.array([d for _, d in sensor_ties], np.int64),
                        state0=tr["state0"], ob=tr["ob"][:, 0], reward=tr["reward"][:, 0], done=tr["done"][:, 0],
                        state=tr["state"][:, 0])
    print("reset ties", reset_ties, "\nsensor ties", sensor_ties)
    print("state0 of tied rocks:", [int(tr["state0"][i][2 + r]) for i, (_, r) in enumerate(reset_ties)])


def main_network(n_ties=11):
    """Network-v0 (10 machines): after reset every machine is up, so the first step draws doubles 0..9 against the
    p = .1 failure threshold and double 10 (action 0 = ping machine 0) against the .95 observation threshold.  The top 16
    bits of double j are a half of the quad's STEP block j >> 1 (philox_ref.network_step_words): a tie is a half that
    equals the threshold's top 16 bits, the low 37 bits from the lane's STEP_LO stream then decide.  One tie per draw
    index 0 .. 10 (both halves of blocks 0 .. 5: the inline blocks and the continuation of the kernels)."""
    thr = json.load(open(os.path.join(HERE, "thresholds.json")))
    t16_fail, t16_obs = thr["net_fail"]["thr"] >> 37, thr["net_obs"]["thr"] >> 37
    found, quad = {}, 0
    chunk = 1 << 16
    while len(found) < n_ties:
        cq = np.arange(quad, quad + chunk, dtype=np.uint64)
        for b in range(6):
            blk = blocks(cq, 1, px.STREAM_STEP, b)
            for half in (0, 1):
                j = 2 * b + half
                if j > 10 or j in found:
                    continue
                q16 = (blk >> np.uint32(16)) if half == 0 else (blk & np.uint32(0xFFFF))
                hit = np.nonzero(q16 == (t16_fail if j < 10 else t16_obs))
                if len(hit[0]):
                    found[j] = int(cq[hit[0][0]]) * 4 + int(hit[1][0])
        quad += chunk
        print("searched quads", quad, "found", sorted(found), flush=True)
    ties = sorted((ln, j) for j, ln in found.items())
    lanes = [ln for ln, _ in ties]
    tr = h.trace_mode_b("network", {}, SEED, lanes, np.zeros((len(lanes), 1), np.int64), t0=0)
    np.savez_compressed(os.path.join(HERE, "ties_network.npz"), seed=np.int64(SEED), lanes=np.array(lanes, np.int64),
                        tied_draw=np.array([j for _, j in ties], np.int64), ob=tr["ob"][:, 0], reward=tr["reward"][:, 0],
                        state=tr["state"][:, 0])
    print("network ties (lane, draw):", ties)


def main_auto(n_ties=4):
    """Auto-reset ties (fixture ties_rock_auto.npz).  The reset that follows a done step inside the step's call counter takes
    its rotated word pair from the step's own sensor blocks (stream STEP, blocks 0 / 1; philox_ref.rock_reset_words): find
    lanes whose element of block 0 at t = 1 puts some rock of the NEW episode on the 2^52 boundary, and end their first
    episode there — WEST from the start cell (0, 3) of RockSample(7,8) leaves the grid (-100, done)."""
    ties, lane = [], 0
    while len(ties) < n_ties:
        c0 = np.arange(lane, lane + CHUNK, dtype=np.uint64)
        blk = blocks(c0, 1, px.STREAM_STEP, 0).astype(np.uint64)
        for j in range(8):
            r = np.uint64(2 * j + 2)
            kh = (((blk >> r) | (blk << (np.uint64(32) - r))) & np.uint64(0xFFFFFFFF)) >> np.uint64(5)
            for qi, e in zip(*np.nonzero(kh == (1 << 26))):
                ties.append((int(c0[qi]) * 4 + int(e), j))
        lane += CHUNK
        print("searched", lane, "found", len(ties), flush=True)
    ties = ties[:n_ties]
    lanes = [ln for ln, _ in ties]
    tr = h.trace_mode_b("rock", {}, SEED, lanes, np.full((len(lanes), 1), 3, np.int64), t0=0)
    assert tr["done"][:, 0].all()
    np.savez_compressed(os.path.join(HERE, "ties_rock_auto.npz"), seed=np.int64(SEED), lanes=np.array(lanes, np.int64),
                        actions=np.full(len(lanes), 3, np.int64), tied_rock=np.array([r for _, r in ties], np.int64),
                        state0=tr["state0"], ob=tr["ob"][:, 0], reward=tr["reward"][:, 0], done=tr["done"][:, 0],
                        state=tr["state"][:, 0])
    print("auto-reset ties (lane, rock):", ties)
    print("new episode's status of the tied rocks:", [int(tr["state"][i, 0][1 + r]) for i, (_, r) in enumerate(ties)])


def main_tag(n_slow=4, n_tie=2):
    """Tag with one opponent (fixture ties_tag.npz): a flight and the auto-reset after a successful TAG read the lane's word W of
    the quad's STEP block (philox_ref.tag_step_words / tag_auto_reset_words).  Rare paths, at call counter 1 with action TAG:
      slow  — lanes whose reset() at t = 0 puts agent and opponent on one cell (the TAG succeeds) AND whose W has five of its
              six 5-bit fields above 28 (the auto-reset's draws run on into the lane's RESET stream: 4 x 10^-5 of the resets);
      tie   — lanes whose TAG fails and whose W >> 5 equals the move threshold's top 27 bits (2^-27: the flight's binomial is
              decided by the low word, block 1)."""
    from gym_pomdp_amd import tables
    thr_hi = int(tables.TAG_MOVE_THR) >> 26
    slow, tie, lane = [], [], 0
    while len(slow) < n_slow or len(tie) < n_tie:
        q = np.arange(lane >> 2, (lane + CHUNK) >> 2, dtype=np.uint64)
        W = blocks(q, 1, px.STREAM_STEP, 0).astype(np.uint64).reshape(-1)          # lane-major: quad q, element e -> lane 4 q + e
        lanes = np.arange(lane, lane + CHUNK, dtype=np.uint64)
        r0 = blocks(lanes, 0, px.STREAM_RESET, 0).astype(np.uint64) & np.uint64(31)   # reset() at t = 0: first two accepted fields
        ok = r0 <= 28
        first = np.argmax(ok, axis=1)
        ok2 = ok.copy(); ok2[np.arange(len(lanes)), first] = False
        second = np.argmax(ok2, axis=1)
        both = ok.sum(axis=1) >= 2
        same = both & (r0[np.arange(len(lanes)), first] == r0[np.arange(len(lanes)), second])
        acc = sum((((W >> np.uint64(5 * i)) & np.uint64(31)) <= 28).astype(np.int64) for i in range(6))
        for i in np.nonzero(same & (acc < 2))[0]:
            if len(slow) < n_slow:
                slow.append(int(lanes[i]))
        for i in np.nonzero(both & ~same & ((W >> np.uint64(5)) == thr_hi))[0]:
            if len(tie) < n_tie:
                tie.append(int(lanes[i]))
        lane += CHUNK
        if (lane // CHUNK) % 8 == 0:
            print("searched", lane, "slow", len(slow), "ties", len(tie), flush=True)
    lanes = slow + tie
    tr = h.trace_mode_b("tag", {}, SEED, lanes, np.full((len(lanes), 1), 4, np.int64), t0=0)
    assert tr["done"][: len(slow), 0].all() and not tr["done"][len(slow):, 0].any()
    np.savez_compressed(os.path.join(HERE, "ties_tag.npz"), seed=np.int64(SEED), lanes=np.array(lanes, np.int64),
                        n_slow=np.int64(len(slow)), state0=tr["state0"], ob=tr["ob"][:, 0], reward=tr["reward"][:, 0],
                        done=tr["done"][:, 0], state_pre=tr["state_pre"][:, 0], state=tr["state"][:, 0])
    print("tag: slow auto-resets at lanes", slow, "flight ties at lanes", tie)


if __name__ == "__main__":
    if "--tag" in sys.argv:
        main_tag()
    elif "--network" in sys.argv:
        main_network()
    elif "--auto" in sys.argv:
        main_auto()
    else:
        main()

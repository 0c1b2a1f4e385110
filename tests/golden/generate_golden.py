#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (inputs + expected outputs) and are committed.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

Fixtures:
  F1  modeA_<case>.npz   reference on np.random.seed(s), >= 4 seeds per case
  F2  thresholds.json    integer Bernoulli thresholds, found by bisection on
                         numpy's own binomial() with an injected MT state
  F3  modeB_<case>.npz   reference driven by the build's Philox word stream
                         (per-lane env objects; includes F4 = initial resets)
  F5  edge_cases.json    error behaviour of the reference at the boundary
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle.ref_harness import harness as h  # noqa: E402

# (case name, env, kwargs, mode-A steps, mode-B lanes, mode-B steps, P(action 4) override for tag)
CASES = [
    ("rock_7_8", "rock", {}, 1000, 512, 48),
    ("rock_7_7", "rock", dict(board_size=7, num_rocks=7), 600, 64, 32),
    ("rock_11_11", "rock", dict(board_size=11, num_rocks=11), 1000, 256, 48),
    ("rock_15_15", "rock", dict(board_size=15, num_rocks=15), 1000, 256, 48),
    ("rock_4_3", "rock", dict(board_size=4, num_rocks=3), 600, 64, 32),
    ("rock_2_1", "rock", dict(board_size=2, num_rocks=1), 300, 64, 32),
    ("stochrock_7_8", "stochrock", {}, 1500, 256, 64),
    ("stochrock_11_11", "stochrock", dict(board_size=11, num_rocks=11), 1000, 64, 64),
    ("tag_1", "tag", {}, 4000, 512, 64),
    ("tag_2", "tag", dict(num_opponents=2), 4000, 256, 64),
    ("tag_4", "tag", dict(num_opponents=4), 2000, 128, 64),
    ("stochrock_7_8_p04", "stochrock", dict(p_move=.4), 1500, 128, 64),   # ... and StochasticRockEnv(p_move <= .5) (rock.py:431, 443)
    ("tag_1_p03", "tag", dict(move_prob=.3), 2000, 128, 64),      # binomial(1, p <= .5): the other sense of numpy's inversion
    ("tag_2_p06", "tag", dict(num_opponents=2, move_prob=.6), 1000, 64, 64),
    ("battleship_5_5", "battleship", {}, 1500, 512, 48),
    ("battleship_10_10", "battleship", dict(board_size=(10, 10), max_len=5), 2000, 48, 400),
    ("battleship_8_6", "battleship", dict(board_size=(8, 6), max_len=4), 1000, 128, 64),
    ("tiger", "tiger", {}, 1000, 512, 48),
    ("network_10", "network", {}, 1000, 512, 48),
    ("network_16_ring", "network", dict(n_machines=16, problem_type=1), 600, 128, 48),
    ("network_31", "network", dict(n_machines=31, problem_type=3), 400, 64, 32),
]
MODE_A_SEEDS = [0, 1, 2, 123456789]
MODE_B_SEED = 0x5EED1234ABCD


def n_actions(env, kwargs):
    e = h.make_ref_env(env, **kwargs)
    return e.action_space.n


def action_tape(env, nA, rs, shape):
    a = rs.randint(nA, size=shape)
    if env == "tag":  # bias towards TAG so that episodes end inside the tape
        a = np.where(rs.uniform(size=shape) < 0.35, 4, a)
    return a


def small(d):
    out = {}
    for k, v in d.items():
        v = np.asarray(v)
        if k == "reward":
            out[k] = v.astype(np.float64)
        elif k in ("done",):
            out[k] = v.astype(np.uint8)
        elif k in ("lanes", "seed", "t0", "max_words_per_call"):
            out[k] = v.astype(np.int64)
        else:
            fits8 = v.size == 0 or (v.min() >= -128 and v.max() <= 127)
            out[k] = v.astype(np.int8 if fits8 else np.int16)
            assert np.array_equal(out[k], v)
    return out


def gen_mode_a(case, env, kwargs, T):
    nA = n_actions(env, kwargs)
    traces = {}
    for seed in MODE_A_SEEDS:
        tries = 0
        while True:  # RockSample(15,15)/(7,7) have crash cells: keep them out of the tape
            acts = action_tape(env, nA, np.random.RandomState(1000003 * (tries + 1) + seed), T)
            try:
                tr = h.trace_mode_a(env, kwargs, seed, acts)
                break
            except IndexError:
                tries += 1
                assert tries < 50
        for k, v in small(tr).items():
            traces["s%d_%s" % (seed, k)] = v
    traces["seeds"] = np.asarray(MODE_A_SEEDS, np.int64)
    np.savez_compressed(os.path.join(HERE, "modeA_%s.npz" % case), **traces)
    return sum(int(traces["s%d_done" % s].sum()) for s in MODE_A_SEEDS)


def gen_mode_b(case, env, kwargs, L, T):
    nA = n_actions(env, kwargs)
    # lanes: a low block and a block straddling 2^20 (exercises the lane counter word)
    lanes = list(range(L // 2)) + list(range((1 << 20) - L // 4, (1 << 20) + L // 4))
    t0 = 0x1FFFFFFF0 if case == "rock_7_8" else 7   # rock_7_8 also exercises the high counter word of t
    tries = 0
    while True:
        acts = action_tape(env, nA, np.random.RandomState(77 + tries), (len(lanes), T))
        try:
            tr = h.trace_mode_b(env, kwargs, MODE_B_SEED, lanes, acts, t0=t0)
            break
        except IndexError:
            tries += 1
            assert tries < 50
    np.savez_compressed(os.path.join(HERE, "modeB_%s.npz" % case), **small(tr))
    return int(tr["done"].sum()), int(tr["max_words_per_call"])


# (case, env, kwargs, roots, sims per root, depth, all_actions)
ROLLOUT_CASES = [
    ("rock_7_8", "rock", {}, 12, 16, 40, False),
    ("rock_7_8_all", "rock", {}, 8, 16, 40, True),
    ("stochrock_7_8", "stochrock", {}, 8, 16, 60, False),
    ("rock_15_15", "rock", dict(board_size=15, num_rocks=15), 8, 16, 60, False),
    ("rock_11_11", "rock", dict(board_size=11, num_rocks=11), 6, 16, 50, False),
    ("tag_1", "tag", {}, 8, 16, 30, False),
    ("tag_2", "tag", dict(num_opponents=2), 6, 8, 30, False),
    ("battleship_5_5", "battleship", {}, 8, 16, 40, False),
    ("battleship_10_10", "battleship", dict(board_size=(10, 10), max_len=5), 3, 6, 120, False),
    ("tiger", "tiger", {}, 8, 16, 20, False),
    ("network_10", "network", {}, 8, 8, 16, False),
]
ROLLOUT_SEED = 0xC0FFEE1234
ROLLOUT_DISCOUNT = {"rock": .95, "stochrock": .95, "tag": .95, "battleship": 1., "tiger": .95, "network": .95}  # the envs' _discount


def gen_rollout(case, env, kwargs, R, S, depth, all_actions):
    tries = 0
    while True:
        try:
            tr = h.rollout_reference(env, kwargs, ROLLOUT_SEED + tries, root_lane0=1000, n_roots=R, sims_per_root=S,
                                     depth=depth, discount=ROLLOUT_DISCOUNT[env], t_reset=3, t0=10,
                                     lane0=(1 << 20) - 64, all_actions=all_actions)
            break
        except IndexError:      # RockSample crash cells
            tries += 1
            assert tries < 50
    tr.update(seed=np.int64(ROLLOUT_SEED + tries), root_lane0=np.int64(1000), n_roots=np.int64(R),
              sims_per_root=np.int64(S), depth=np.int64(depth), discount=np.float64(ROLLOUT_DISCOUNT[env]),
              t_reset=np.int64(3), t0=np.int64(10), lane0=np.int64((1 << 20) - 64), all_actions=np.int64(all_actions))
    np.savez_compressed(os.path.join(HERE, "rollout_%s.npz" % case), **tr)
    return int(tr["terminated"].sum()), float(tr["n_steps"].mean())


# the planning step (include/pomdp_hip.h: pomdp_plan + the roots' real step): (case, env, kwargs, roots, sims per root, depth).
# Simulation counts straddle the reduction's 64-simulation chunks: 160 = two and a half, 100 = a ragged second chunk, 72.
PLAN_CASES = [
    ("rock_7_8", "rock", {}, 6, 160, 20),
    ("rock_15_15", "rock", dict(board_size=15, num_rocks=15), 4, 128, 30),
    ("stochrock_7_8", "stochrock", {}, 3, 72, 20),
    ("tag_1", "tag", {}, 4, 100, 20),
    ("battleship_5_5", "battleship", {}, 3, 80, 25),
    ("tiger", "tiger", {}, 6, 72, 10),
    ("network_10", "network", {}, 3, 64, 8),
]
PLAN_SEED = 0x9A11CE5EED


def gen_plan(case, env, kwargs, R, S, depth):
    tries = 0
    while True:
        try:
            tr = h.plan_reference(env, kwargs, PLAN_SEED + tries, root_lane0=1000, n_roots=R, sims_per_root=S, depth=depth,
                                  discount=ROLLOUT_DISCOUNT[env], t_reset=3, t0=10)
            break
        except IndexError:      # RockSample crash cells
            tries += 1
            assert tries < 50
    tr.update(seed=np.int64(PLAN_SEED + tries), root_lane0=np.int64(1000), n_roots=np.int64(R), sims_per_root=np.int64(S),
              depth=np.int64(depth), discount=np.float64(ROLLOUT_DISCOUNT[env]), t_reset=np.int64(3), t0=np.int64(10))
    np.savez_compressed(os.path.join(HERE, "plan_%s.npz" % case), **tr)
    return tr


PROB_CASES = [("rock_7_8", "rock", {}, 48, 40), ("stochrock_7_8", "stochrock", {}, 24, 40), ("rock_15_15", "rock", dict(board_size=15, num_rocks=15), 24, 40),
              ("tag_1", "tag", {}, 32, 40), ("tag_2", "tag", dict(num_opponents=2), 16, 40),
              ("battleship_5_5", "battleship", {}, 32, 40), ("tiger", "tiger", {}, 32, 30),
              ("network_10", "network", {}, 32, 30)]


def gen_prob(case, env, kwargs, L, T):
    nA = n_actions(env, kwargs)
    tries = 0
    while True:
        acts = action_tape(env, nA, np.random.RandomState(4242 + tries), (L, T))
        try:
            tr = h.compute_prob_trace(env, kwargs, MODE_B_SEED, range(64, 64 + L), acts, t0=5)
            break
        except IndexError:
            tries += 1
            assert tries < 50
    np.savez_compressed(os.path.join(HERE, "prob_%s.npz" % case), **tr)


# heuristic policy traces: (case, env, kwargs, lanes, steps)
HEUR_CASES = [("rock_7_8", "rock", {}, 96, 96), ("rock_11_11", "rock", dict(board_size=11, num_rocks=11), 32, 96),
              ("rock_15_15", "rock", dict(board_size=15, num_rocks=15), 32, 128),
              ("rock_4_3", "rock", dict(board_size=4, num_rocks=3), 32, 48),
              ("stochrock_7_8", "stochrock", {}, 48, 128),
              ("tag_1", "tag", {}, 64, 128), ("tag_2", "tag", dict(num_opponents=2), 32, 128)]
# ... and with the planner's history bounded: History(max_size) of rock.py:533-544 (case -> max_size)
HEUR_BOUNDED = [("rock_7_8_hist8", "rock", {}, 64, 160, 8), ("rock_11_11_hist2", "rock", dict(board_size=11, num_rocks=11), 32, 128, 2),
                ("rock_4_3_hist0", "rock", dict(board_size=4, num_rocks=3), 32, 64, 0)]
HEUR_SEED = 0xBE11EF5EED


def gen_heuristic(case, env, kwargs, L, T, max_size=None):
    lanes = list(range(L // 2)) + list(range((1 << 20) - L // 4, (1 << 20) + L // 4))
    tries = 0
    while True:
        try:
            tr = h.heuristic_trace(env, kwargs, HEUR_SEED + tries, lanes, T, t0=11, max_size=max_size)
            if max_size is not None:
                tr["max_size"] = np.int64(max_size)
            break
        except IndexError:      # RockSample crash cells
            tries += 1
            assert tries < 50
    out = {}
    for k, v in tr.items():
        v = np.asarray(v)
        if v.dtype == np.float64 or k in ("lanes", "seed", "t0", "max_size"):
            out[k] = v
        else:
            fits8 = v.size == 0 or (v.min() >= -128 and v.max() <= 127)
            out[k] = v.astype(np.int8 if fits8 else np.int16)
            assert np.array_equal(out[k], v)
    np.savez_compressed(os.path.join(HERE, "heur_%s.npz" % case), **out)
    return int(tr["done"].sum()), float(tr["pref_len"].mean()), tries


def gen_thresholds():
    def binom_at(p, k):
        h.inject_words([(k >> 26) << 5, (k & ((1 << 26) - 1)) << 6])
        r = int(np.random.binomial(1, p))
        assert h.consumed_words() == 2  # exactly one double per call
        return r

    def bisect(p):
        top = (1 << 53) - 1
        r0, r1 = binom_at(p, 0), binom_at(p, top)
        if r0 == r1:
            return None, r0
        lo, hi = 0, top
        while hi - lo > 1:
            m = (lo + hi) // 2
            if binom_at(p, m) == r0:
                lo = m
            else:
                hi = m
        return lo, r0

    out = {"doc": "np.random.binomial(1,p) on U=k/2^53: 'le' => 1 iff k <= thr; 'gt' => 1 iff k > thr. "
                  "rock_thr[d] is for eff(d)=(1+2^(-d/20))/2 (rock.py:383-387); d=0 (p=1.0) is always 1 "
                  "and carries 2^53."}
    rock = []
    for d in range(29):
        eff = (1 + pow(2, -d / 20)) * .5
        thr, r0 = bisect(eff)
        if thr is None:
            assert r0 == 1 and d == 0
            thr = 1 << 53
        else:
            assert r0 == 1
        rock.append(int(thr))
    out["rock_thr"] = rock
    # eff(d) itself, as the reference computes it (rock.py:383-387), for _compute_prob
    envs = h.load_reference()
    from gym_pomdp.envs.coord import Coord
    out["rock_eff_hex"] = [float(envs.RockEnv._efficiency(Coord(0, 0), Coord(d, 0))).hex() for d in range(29)]
    for name, p, sense in (("tag_move", .8, "le"), ("net_fail", .1, "gt"), ("net_fail_neighbour", .33, "gt"),
                           ("net_obs", .95, "le")):
        thr, r0 = bisect(p)
        assert r0 == (1 if sense == "le" else 0)
        out[name] = {"p": p, "sense": sense, "thr": int(thr)}
    # tiger.py:143-148 compares uniform() > .85 directly: U > .85  <=>  k > .85 * 2^53 (an exact integer)
    out["tiger_listen"] = {"p": .85, "sense": "flip iff k > thr", "thr": int(.85 * 2 ** 53)}
    assert float(out["tiger_listen"]["thr"]) / 2 ** 53 == .85
    with open(os.path.join(HERE, "thresholds.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_edge_cases():
    envs = h.load_reference()
    out = {}

    def err(fn):
        try:
            fn()
            return "ok"
        except BaseException as e:  # noqa: BLE001
            return type(e).__name__

    for name, kw in (("rock", {}), ("tag", {}), ("battleship", {}), ("tiger", {}), ("network", {})):
        e = h.make_ref_env(name, **kw)
        out["%s.step_before_reset" % name] = err(lambda: e.step(0))
        e.seed(0)
        e.reset()
        out["%s.action_out_of_range" % name] = err(lambda: e.step(e.action_space.n))
        out["%s.action_negative" % name] = err(lambda: e.step(-1))
        out["%s.action_float" % name] = err(lambda: e.step(1.0))
        out["%s.n_actions" % name] = int(e.action_space.n)
        out["%s.n_obs" % name] = int(e.observation_space.n)
    # step after done
    e = h.make_ref_env("rock")
    e.seed(0)
    e.reset()
    e.step(3)  # WEST from x=0: -100, done
    out["rock.step_after_done"] = err(lambda: e.step(0))
    # constructor validation (rock.py:101)
    for bs, k in ((7, 8), (7, 7), (7, 6), (3, 3), (11, 11), (15, 15), (2, 1), (4, 3)):
        out["rock.ctor_%d_%d" % (bs, k)] = err(lambda: envs.RockEnv(board_size=bs, num_rocks=k))
    # crash cell of RockSample(15,15): SAMPLE at (12,2)
    e = envs.RockEnv(board_size=15, num_rocks=15)
    e.seed(0)
    e.reset()
    from gym_pomdp.envs.coord import Coord
    e.state.agent_pos = Coord(12, 2)
    out["rock_15_15.sample_at_12_2"] = err(lambda: e.step(4))
    out["network.make_3legs_10"] = envs.NetworkEnv.make_3legs_neighbours(10)
    out["moves"] = {"NORTH": [0, 1], "EAST": [1, 0], "SOUTH": [0, -1], "WEST": [-1, 0]}  # coord.py:120-126
    with open(os.path.join(HERE, "edge_cases.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def main():
    assert h.reference_available(), "needs /root/reference"
    import warnings
    warnings.simplefilter("ignore", RuntimeWarning)  # reference's belief side-stats divide 0/0 (rock.py:191)
    gen_thresholds()
    if "--bounded-history-only" in sys.argv:
        for case, env, kwargs, L, T, ms in HEUR_BOUNDED:
            nd, ml, tries = gen_heuristic(case, env, kwargs, L, T, max_size=ms)
            print("heuristic %-16s dones=%4d  mean preferred-list length=%.2f  (seed retries %d)" % (case, nd, ml, tries), flush=True)
        return
    if "--heuristic-only" in sys.argv:
        for case, env, kwargs, L, T in HEUR_CASES:
            nd, ml, tries = gen_heuristic(case, env, kwargs, L, T)
            print("heuristic %-16s dones=%4d  mean preferred-list length=%.2f  (seed retries %d)" % (case, nd, ml, tries), flush=True)
        return
    if "--plans-only" in sys.argv:
        for case, env, kwargs, R, S, depth in PLAN_CASES:
            tr = gen_plan(case, env, kwargs, R, S, depth)
            print("plan %-16s best=%s visited actions per root=%s done=%d" % (case, tr["best"].tolist(),
                  (tr["visits"] > 0).sum(axis=1).tolist(), int(tr["done"].sum())), flush=True)
        write_manifest()
        return
    if "--rollouts-only" not in sys.argv:
        gen_edge_cases()
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for case, env, kwargs, TA, L, TB in (CASES if "--rollouts-only" not in sys.argv else []):
        if only and not any(case.startswith(o) for o in only):
            continue
        da = gen_mode_a(case, env, kwargs, TA)
        db, mw = gen_mode_b(case, env, kwargs, L, TB)
        print("%-18s modeA dones=%4d  modeB dones=%5d  max words/call=%d" % (case, da, db, mw), flush=True)
    wanted = lambda case: not only or any(case.startswith(o) for o in only)   # noqa: E731  (--only=<case prefix>, repeatable)
    for case, env, kwargs, L, T in PROB_CASES:
        if not wanted(case):
            continue
        gen_prob(case, env, kwargs, L, T)
        print("compute_prob %-14s ok" % case, flush=True)
    for case, env, kwargs, R, S, depth, alla in ROLLOUT_CASES:
        if not wanted(case):
            continue
        nt, ms = gen_rollout(case, env, kwargs, R, S, depth, alla)
        print("rollout %-18s terminated=%4d / %d  mean steps=%.1f" % (case, nt, R * S, ms), flush=True)
    for case, env, kwargs, L, T in HEUR_CASES:
        if not wanted(case):
            continue
        nd, ml, tries = gen_heuristic(case, env, kwargs, L, T)
        print("heuristic %-16s dones=%4d  mean preferred-list length=%.2f  (seed retries %d)" % (case, nd, ml, tries), flush=True)
    for case, env, kwargs, L, T, ms in HEUR_BOUNDED:
        if not wanted(case):
            continue
        nd, ml, tries = gen_heuristic(case, env, kwargs, L, T, max_size=ms)
        print("heuristic %-16s dones=%4d  mean preferred-list length=%.2f  (seed retries %d)" % (case, nd, ml, tries), flush=True)
    for case, env, kwargs, R, S, depth in PLAN_CASES:
        if not wanted(case):
            continue
        tr = gen_plan(case, env, kwargs, R, S, depth)
        print("plan %-16s best=%s" % (case, tr["best"].tolist()), flush=True)
    write_manifest()


def write_manifest():
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump({"cases": [[c[0], c[1], {k: (list(v) if isinstance(v, tuple) else v) for k, v in c[2].items()}]
                             for c in CASES],
                   "rollout_cases": [[c[0], c[1], {k: (list(v) if isinstance(v, tuple) else v) for k, v in c[2].items()}]
                                     for c in ROLLOUT_CASES],
                   "prob_cases": [[c[0], c[1], {k: (list(v) if isinstance(v, tuple) else v) for k, v in c[2].items()}]
                                  for c in PROB_CASES],
                   "plan_cases": [[c[0], c[1], {k: (list(v) if isinstance(v, tuple) else v) for k, v in c[2].items()}]
                                  for c in PLAN_CASES],
                   "heuristic_cases": [[c[0], c[1], {k: (list(v) if isinstance(v, tuple) else v) for k, v in c[2].items()}]
                                       for c in HEUR_CASES + HEUR_BOUNDED],
                   "mode_a_seeds": MODE_A_SEEDS, "mode_b_seed": MODE_B_SEED,
                   "numpy": np.__version__}, f, indent=1)


if __name__ == "__main__":
    main()

"""The CPU oracle (oracle/pomdp_oracle.c) against the golden traces generated
from the unmodified reference (tests/golden/generate_golden.py)."""
import json
import os

import numpy as np
import pytest

from conftest import (GOLDEN, OracleHeuristicOps, golden_cases, golden_manifest, heuristic_replay, load_golden,
                      saturate_tag_compact)

CASES = golden_cases()
KEYS = ["ob", "reward", "done", "state_pre", "state", "reset_ob"]


@pytest.mark.parametrize("case,env,kw", CASES, ids=[c[0] for c in CASES])
def test_mode_a_mt_exact(oracle_lib, case, env, kw):
    """Oracle on an emulated np.random.seed(s) MT19937 stream == the reference's own trace."""
    g = load_golden("A", case)
    o = oracle_lib.OracleEnv(env, **kw)
    for seed in g["seeds"]:
        pre = "s%d_" % seed
        got = o.trace_mt(int(seed), g[pre + "actions"])
        assert int(got["ob0"]) == int(g[pre + "ob0"])
        assert np.array_equal(got["state0"], g[pre + "state0"])
        for k in KEYS:
            assert np.array_equal(got[k], g[pre + k].astype(got[k].dtype)), (case, seed, k)


@pytest.mark.parametrize("case,env,kw", CASES, ids=[c[0] for c in CASES])
def test_mode_b_philox_injected(oracle_lib, case, env, kw):
    """Oracle batch drivers on Philox streams == the reference driven by the same words."""
    g = load_golden("B", case)
    o = oracle_lib.OracleEnv(env, **kw)
    seed, t0 = int(g["seed"]), int(g["t0"])
    lanes = g["lanes"]
    L, T = g["actions"].shape
    # lanes come in contiguous runs; run each run as its own batch with the right lane0
    starts = [0] + [i for i in range(1, L) if lanes[i] != lanes[i - 1] + 1] + [L]
    for s, e in zip(starts[:-1], starts[1:]):
        n, lane0 = e - s, int(lanes[s])
        st = o.new_state(n)
        ob0 = o.batch_reset(st, seed, lane0, t0)
        assert np.array_equal(ob0, g["ob0"][s:e])
        assert np.array_equal(o.batch_compact(st), saturate_tag_compact(env, g["state0"][s:e]))
        for i in range(T):
            ob, rew, done, bad = o.batch_step(st, g["actions"][s:e, i], seed, lane0, t0 + 1 + i)
            assert bad == 0
            assert np.array_equal(ob, g["ob"][s:e, i]), (case, i)
            assert np.array_equal(rew, g["reward"][s:e, i].astype(o.reward_dtype)), (case, i)
            assert np.array_equal(done, g["done"][s:e, i]), (case, i)
            assert np.array_equal(o.batch_compact(st), saturate_tag_compact(env, g["state"][s:e, i])), (case, i)


@pytest.mark.parametrize("case,env,kw", CASES, ids=[c[0] for c in CASES])
def test_mode_b_no_auto_reset(oracle_lib, case, env, kw):
    """auto_reset=False: the terminal state is kept (== golden state_pre) and the lane freezes."""
    g = load_golden("B", case)
    o = oracle_lib.OracleEnv(env, **kw)
    seed, t0 = int(g["seed"]), int(g["t0"])
    lanes = g["lanes"]
    n = next((i for i in range(1, len(lanes)) if lanes[i] != lanes[i - 1] + 1), len(lanes))
    st = o.new_state(n)
    o.batch_reset(st, seed, int(lanes[0]), t0)
    frozen = np.zeros(n, bool)
    done = np.zeros(n, np.uint8)
    term = np.zeros((n, o.compact_len), np.int64)
    for i in range(g["actions"].shape[1]):
        ob, rew, done, _ = o.batch_step(st, g["actions"][:n, i], seed, int(lanes[0]), t0 + 1 + i,
                                        auto_reset=False, done=done)
        live = ~frozen
        assert np.array_equal(ob[live], g["ob"][:n, i][live])
        assert np.array_equal(done[live], g["done"][:n, i][live])
        assert np.all(ob[frozen] == 0) and np.all(rew[frozen] == 0) and np.all(done[frozen] == 1)
        comp = o.batch_compact(st)
        exp = saturate_tag_compact(env, g["state_pre"][:n, i])
        assert np.array_equal(comp[live], exp[live])
        assert np.array_equal(comp[frozen], term[frozen])
        newly = live & (done == 1)
        term[newly] = comp[newly]
        frozen |= newly
        # golden lanes that auto-reset diverge from here on: stop comparing them
        frozen |= g["done"][:n, i].astype(bool)


def test_thresholds_match_fixture():
    """The constants baked into the oracle are the captured numpy thresholds (fixture F2)."""
    with open(os.path.join(GOLDEN, "thresholds.json")) as f:
        thr = json.load(f)
    src = open(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "pomdp_oracle.c")).read()
    for v in thr["rock_thr"]:
        assert "%dULL" % v in src
    for k in ("tag_move", "net_fail", "net_fail_neighbour", "net_obs", "tiger_listen"):
        assert "%dULL" % thr[k]["thr"] in src


def test_thresholds_recomputed_with_libm():
    """Informational pin: glibc exp/log/pow on this box reproduce the captured table."""
    import math
    with open(os.path.join(GOLDEN, "thresholds.json")) as f:
        thr = json.load(f)
    for d, want in enumerate(thr["rock_thr"]):
        eff = (1 + pow(2, -d / 20)) * .5
        q = 1.0 - (1.0 - eff)
        assert math.floor(math.exp(math.log(q)) * 2 ** 53) == want


def test_philox_kat(oracle_lib):
    from oracle import philox_ref as px
    for ctr, key, out in px.KAT:
        assert tuple(int(x) for x in px.philox4x32_10(np.array(ctr), np.array(key))) == out
        assert tuple(int(x) for x in oracle_lib.philox(ctr, key)) == out


def test_synthetic_actions_agree(oracle_lib):
    from oracle import philox_ref as px
    for lane0, n, t, nA in ((0, 1000, 3, 13), (1 << 20, 257, (1 << 33) + 5, 100), (5, 64, 0, 5)):
        a = oracle_lib.synthetic_actions(n, 42, lane0, t, nA, nthreads=2)
        b = px.synthetic_actions(42, lane0, n, t, nA)
        assert np.array_equal(a, b)
        assert a.min() >= 0 and a.max() < nA


def test_oracle_threads_agree(oracle_lib):
    o = oracle_lib.OracleEnv("rock")
    n = 5000
    s1, s2 = o.new_state(n), o.new_state(n)
    o.batch_reset(s1, 1, 0, 0, nthreads=1)
    o.batch_reset(s2, 1, 0, 0, nthreads=4)
    assert np.array_equal(s1, s2)
    for t in range(1, 20):
        a = oracle_lib.synthetic_actions(n, 9, 0, t, o.n_actions)
        r1 = o.batch_step(s1, a, 1, 0, t, nthreads=1)
        r2 = o.batch_step(s2, a, 1, 0, t, nthreads=4)
        for x, y in zip(r1[:3], r2[:3]):
            assert np.array_equal(x, y)
        assert np.array_equal(s1, s2)


def test_rejected_configs(oracle_lib):
    with open(os.path.join(GOLDEN, "edge_cases.json")) as f:
        edge = json.load(f)
    for bs, k in ((7, 8), (7, 7), (7, 6), (3, 3), (11, 11), (15, 15), (2, 1), (4, 3)):
        ok = edge["rock.ctor_%d_%d" % (bs, k)] == "ok"
        if ok:
            oracle_lib.OracleEnv("rock", board_size=bs, num_rocks=k)
        else:
            with pytest.raises(ValueError):
                oracle_lib.OracleEnv("rock", board_size=bs, num_rocks=k)


def test_invalid_action_counts(oracle_lib):
    o = oracle_lib.OracleEnv("rock")
    st = o.new_state(4)
    o.batch_reset(st, 0, 0, 0)
    before = st.copy()
    ob, rew, done, bad = o.batch_step(st, np.array([13, -1, 0, 99]), 0, 0, 1)
    assert bad == 3
    assert np.array_equal(st[:, [0, 1, 3]], before[:, [0, 1, 3]])
    assert np.all(ob[[0, 1, 3]] == 0) and np.all(rew[[0, 1, 3]] == 0) and np.all(done[[0, 1, 3]] == 0)


# ---- planner hooks: _generate_legal and rollouts --------------------------
def _rollout_cases():
    return [(c[0], c[1], {k: (tuple(v) if isinstance(v, list) else v) for k, v in c[2].items()})
            for c in golden_manifest()["rollout_cases"]]


ROLLOUTS = _rollout_cases()


@pytest.mark.parametrize("case,env,kw", ROLLOUTS, ids=[c[0] for c in ROLLOUTS])
def test_rollouts_and_legal_lists_match_reference(oracle_lib, case, env, kw):
    g = dict(np.load(os.path.join(GOLDEN, "rollout_%s.npz" % case)))
    o = oracle_lib.OracleEnv(env, **kw)
    R, S = int(g["n_roots"]), int(g["sims_per_root"])
    roots = o.new_state(R)
    o.batch_reset(roots, int(g["seed"]), int(g["root_lane0"]), int(g["t_reset"]))
    assert np.array_equal(o.batch_compact(roots), saturate_tag_compact(env, g["root_state"]))
    lists, lens = o.batch_legal(roots)
    assert np.array_equal(lens, g["root_legal_len"])
    assert np.array_equal(lists, g["root_legal"])
    r = o.batch_rollout(roots, S, int(g["depth"]), float(g["discount"]), int(g["seed"]), int(g["lane0"]),
                        int(g["t0"]), all_actions=bool(g["all_actions"]), nthreads=2)
    assert np.array_equal(r["ret"], g["ret"])                 # IEEE double, bit-exact
    for k in ("n_steps", "first_action", "last_ob", "terminated"):
        assert np.array_equal(r[k], g[k].astype(r[k].dtype)), k


def _plan_cases():
    return [(c[0], c[1], {k: (tuple(v) if isinstance(v, list) else v) for k, v in c[2].items()})
            for c in golden_manifest()["plan_cases"]]


PLANS = _plan_cases()


@pytest.mark.parametrize("case,env,kw", PLANS, ids=[c[0] for c in PLANS])
def test_planning_step_matches_reference(oracle_lib, case, env, kw):
    """BASELINE.json configs[4]'s planning step (fixture plan_*.npz: the reference's own step() under the simulations, their
    returns reduced in PYTHON floats in the order include/pomdp_hip.h states, the roots then stepped by the reference): the
    oracle's rollouts, or_plan_reduce and batch step reproduce every number, the float64 ones bit for bit."""
    g = dict(np.load(os.path.join(GOLDEN, "plan_%s.npz" % case)))
    o = oracle_lib.OracleEnv(env, **kw)
    R, S, depth = int(g["n_roots"]), int(g["sims_per_root"]), int(g["depth"])
    seed, lane0, t0 = int(g["seed"]), int(g["root_lane0"]), int(g["t0"])
    roots = o.new_state(R)
    o.batch_reset(roots, seed, lane0, int(g["t_reset"]))
    assert np.array_equal(o.batch_compact(roots), saturate_tag_compact(env, g["root_state"]))
    r = o.batch_rollout(roots, S, depth, float(g["discount"]), seed, lane0 * S, t0, nthreads=2)
    assert np.array_equal(r["ret"].view(np.uint64), g["sim_ret"].view(np.uint64))
    assert np.array_equal(r["first_action"], g["sim_first_action"])
    p = oracle_lib.plan_reduce(r["ret"], r["first_action"], R, S, o.n_actions)
    assert np.array_equal(p["visits"], g["visits"]) and np.array_equal(p["best"], g["best"])
    assert np.array_equal(p["q"].view(np.uint64), g["q"].view(np.uint64))             # IEEE double, bit for bit
    assert np.array_equal(p["value"].view(np.uint64), g["value"].view(np.uint64))
    assert g["visits"].sum(axis=1).tolist() == [S] * R
    ob, rew, done, bad = o.batch_step(roots, p["best"], seed, lane0, t0 + depth, auto_reset=True)
    assert bad == 0 and np.array_equal(ob, g["ob"]) and np.array_equal(done, g["done"])
    assert np.array_equal(rew.astype(np.float64), g["reward"].astype(rew.dtype).astype(np.float64))
    assert np.array_equal(o.batch_compact(roots), saturate_tag_compact(env, g["state"]))


def test_plan_reduce_order_and_edge_cases(oracle_lib):
    """or_plan_reduce against the harness's Python-float statement of the order on returns built to expose it (magnitudes 16
    orders apart: any other association rounds differently), with unvisited actions, simulations that took no step, a root
    where nothing did, ties, and simulation counts on both sides of the chunk boundaries."""
    from oracle.ref_harness import harness as h
    rng = np.random.RandomState(7)
    for R, S, A in ((3, 1, 3), (2, 63, 5), (2, 64, 5), (2, 65, 20), (3, 200, 7), (2, 1024, 20), (1, 1100, 100)):
        ret = (rng.randn(R * S) * 10.0 ** rng.randint(-8, 9, R * S)).astype(np.float64)
        fa = rng.randint(-1, A, R * S).astype(np.int32)
        fa[fa == 2] = 1                                                     # action 2 is never tried
        if R > 1:
            fa[S:2 * S] = -1                                                # root 1: no simulation took a step
        if R > 2:
            ret[2 * S:3 * S] = 1.0                                          # root 2: every visited action ties
        want = h.plan_reduce_python(ret, fa, R, S, A)
        got = oracle_lib.plan_reduce(ret, fa, R, S, A)
        for k, w in zip(("q", "visits", "best", "value"), want):
            assert np.array_equal(got[k].astype(w.dtype).view(np.uint64), w.view(np.uint64)), (R, S, A, k)
        if A > 2:
            assert (got["visits"][:, 2] == 0).all() and (got["q"][:, 2] == 0).all()
        if R > 1:
            assert got["best"][1] == -1 and got["value"][1] == 0.0
        if R > 2:
            assert got["best"][2] == int(np.flatnonzero(got["visits"][2] > 0)[0])
    # the order matters: chunk 0 = 1e16 + 63 ones (each lost to rounding), chunk 1 = -1e16 + 63 ones: the chunked sum is 0,
    # a plain left-to-right sum over the 128 simulations keeps chunk 1's ones (63)
    ret = np.ones(128)
    ret[0], ret[64] = 1e16, -1e16
    got = oracle_lib.plan_reduce(ret, np.zeros(128, np.int32), 1, 128, 1)
    seq = 0.0
    for v in ret:
        seq = seq + float(v)
    assert got["q"][0, 0] == 0.0 and seq == 63.0


def _prob_cases():
    return [(c[0], c[1], {k: (tuple(v) if isinstance(v, list) else v) for k, v in c[2].items()})
            for c in golden_manifest()["prob_cases"]]


PROBS = _prob_cases()


@pytest.mark.parametrize("case,env,kw", PROBS, ids=[c[0] for c in PROBS])
def test_compute_prob_matches_reference(oracle_lib, case, env, kw):
    """_compute_prob(a, state after the step, o) for every observation value o (fixture prob_*.npz)."""
    g = dict(np.load(os.path.join(GOLDEN, "prob_%s.npz" % case)))
    o = oracle_lib.OracleEnv(env, **kw)
    seed, t0, lane0 = int(g["seed"]), int(g["t0"]), int(g["lanes"][0])
    L, T, n_obs = g["prob"].shape
    st = o.new_state(L)
    o.batch_reset(st, seed, lane0, t0)
    for i in range(T):
        a = g["actions"][:, i]
        pre = st.copy()
        ob, _, done, _ = o.batch_step(pre, a, seed, lane0, t0 + 1 + i, auto_reset=False)   # terminal state kept
        assert np.array_equal(ob, g["ob"][:, i])
        assert np.array_equal(o.batch_compact(pre), saturate_tag_compact(env, g["state_pre"][:, i]))
        for q in range(n_obs):
            got = o.batch_compute_prob(pre, a, np.full(L, q))
            assert np.array_equal(got, g["prob"][:, i, q]), (case, i, q)
        o.batch_step(st, a, seed, lane0, t0 + 1 + i)                                       # continue with auto-reset


def test_moves_axis_convention(oracle_lib):
    """The reference's only known-answer test (coord.py:120-126, `TestCoord`): (2,2)+NORTH=(2,3), +EAST=(3,2),
    +SOUTH=(2,1), +WEST=(1,2).  Checked on the oracle's RockSample agent from its start cell (0,3)."""
    with open(os.path.join(GOLDEN, "edge_cases.json")) as f:
        moves = json.load(f)["moves"]
    assert moves == {"NORTH": [0, 1], "EAST": [1, 0], "SOUTH": [0, -1], "WEST": [-1, 0]}
    o = oracle_lib.OracleEnv("rock")
    for action, name in enumerate(("NORTH", "EAST", "SOUTH", "WEST")):
        st = o.new_state(1)
        o.batch_reset(st, 0, 0, 0)
        x0, y0 = o.batch_compact(st)[0, :2]
        ob, rew, done, _ = o.batch_step(st, [action], 0, 0, 1, auto_reset=False)
        x1, y1 = o.batch_compact(st)[0, :2]
        if name == "WEST":                      # x = 0: off-grid, -100 and done, position unchanged (rock.py:152-156)
            assert (x1, y1, int(rew[0]), int(done[0])) == (x0, y0, -100, 1)
        else:
            assert [x1 - x0, y1 - y0] == moves[name] and int(rew[0]) == 0 and int(done[0]) == 0


def test_split_layout_ties_oracle(oracle_lib):
    """RockSample draws whose high word alone leaves the comparison undecided (fixture ties_rock.npz, lanes found by
    tests/golden/find_ties.py): the low-word block decides, as in the reference."""
    g = dict(np.load(os.path.join(GOLDEN, "ties_rock.npz")))
    o = oracle_lib.OracleEnv("rock")
    seed = int(g["seed"])
    for i, lane in enumerate(g["lanes"]):
        st = o.new_state(1)
        o.batch_reset(st, seed, int(lane), 0)
        assert np.array_equal(o.batch_compact(st)[0], g["state0"][i])
        ob, rew, done, _ = o.batch_step(st, [int(g["actions"][i])], seed, int(lane), 1)
        assert (int(ob[0]), int(rew[0]), int(done[0])) == (int(g["ob"][i]), int(g["reward"][i]), int(g["done"][i]))
    # the tied rocks of the reset cases really sit on the 2^52 boundary of their high word
    from oracle import philox_ref as px
    for i in range(int(g["n_reset"])):
        w = px.rock_reset_words(seed, int(g["lanes"][i]), 0, 8)
        assert int(w[2 * int(g["tied_rock"][i])]) >> 5 == 1 << 26


def test_auto_reset_ties_oracle(oracle_lib):
    """The reset that follows a done step inside the step's call counter reads the step's own sensor blocks (stream STEP,
    rotated pair; oracle/philox_ref.py rock_reset_words): lanes whose new episode has a rock on the 2^52 boundary of its high
    word (fixture ties_rock_auto.npz, tests/golden/find_ties.py --auto: the reference, ending its first episode with WEST)."""
    g = dict(np.load(os.path.join(GOLDEN, "ties_rock_auto.npz")))
    o = oracle_lib.OracleEnv("rock")
    seed = int(g["seed"])
    from oracle import philox_ref as px
    for i, lane in enumerate(g["lanes"]):
        st = o.new_state(1)
        o.batch_reset(st, seed, int(lane), 0)
        assert np.array_equal(o.batch_compact(st)[0], g["state0"][i])
        ob, rew, done, _ = o.batch_step(st, [int(g["actions"][i])], seed, int(lane), 1)
        assert (int(ob[0]), int(rew[0]), int(done[0])) == (int(g["ob"][i]), int(g["reward"][i]), 1)
        assert np.array_equal(o.batch_compact(st)[0], g["state"][i])                    # the episode the auto-reset dealt
        w = px.rock_reset_words(seed, int(lane), 1, 8, auto_step_block=0)
        assert int(w[2 * int(g["tied_rock"][i])]) >> 5 == 1 << 26
        # ... and it is NOT what stream RESET would have dealt at that call counter
    assert any(not np.array_equal(px.rock_reset_words(seed, int(l), 1, 8), px.rock_reset_words(seed, int(l), 1, 8, auto_step_block=0))
               for l in g["lanes"])


def test_tag_quad_word_rare_paths_oracle(oracle_lib):
    """Tag with one opponent: a flight and the auto-reset after a successful TAG read the lane's word of the quad's STEP block
    (oracle/philox_ref.py tag_step_words / tag_auto_reset_words).  Fixture ties_tag.npz (tests/golden/find_ties.py --tag: the
    reference): lanes whose auto-reset's rejection draws run past the word's six fields into stream RESET, and lanes whose
    flight is decided by the double's low word."""
    g = dict(np.load(os.path.join(GOLDEN, "ties_tag.npz")))
    o = oracle_lib.OracleEnv("tag")
    seed, n_slow = int(g["seed"]), int(g["n_slow"])
    from oracle import philox_ref as px
    from gym_pomdp_amd import tables
    for i, lane in enumerate(g["lanes"]):
        lane = int(lane)
        st = o.new_state(1)
        o.batch_reset(st, seed, lane, 0)
        assert np.array_equal(o.batch_compact(st)[0], g["state0"][i])
        ob, rew, done, _ = o.batch_step(st, [4], seed, lane, 1)
        assert (int(ob[0]), float(rew[0]), int(done[0])) == (int(g["ob"][i]), float(g["reward"][i]), int(g["done"][i])), lane
        assert np.array_equal(o.batch_compact(st)[0], g["state"][i]), lane
        w = px.tag_step_words(seed, lane, 1)
        if i < n_slow:
            assert int(g["done"][i]) == 1 and sum(((int(w[0]) >> (5 * k)) & 31) <= 28 for k in range(6)) < 2
        else:
            assert int(g["done"][i]) == 0 and int(w[0]) >> 5 == tables.TAG_MOVE_THR >> 26


def test_network_split_layout_ties_oracle(oracle_lib):
    """Network draws whose quad-shared top 16 bits equal the threshold's (fixture ties_network.npz from tests/golden/
    find_ties.py --network: one lane per draw index 0 .. 10), decided by the lane's own STEP_LO words."""
    import json
    g = dict(np.load(os.path.join(GOLDEN, "ties_network.npz")))
    thr = json.load(open(os.path.join(GOLDEN, "thresholds.json")))
    o = oracle_lib.OracleEnv("network")
    seed = int(g["seed"])
    from oracle import philox_ref as px
    for i, lane in enumerate(g["lanes"]):
        st = o.new_state(1)
        o.batch_reset(st, seed, int(lane), 0)
        ob, rew, done, _ = o.batch_step(st, [0], seed, int(lane), 1)
        assert int(ob[0]) == int(g["ob"][i]) and float(rew[0]) == np.float32(g["reward"][i])
        assert np.array_equal(o.batch_compact(st)[0], g["state"][i])
        j = int(g["tied_draw"][i])
        want = thr["net_obs"]["thr"] if j == 10 else thr["net_fail"]["thr"]
        # really a tie of the quad-shared 16 bits (philox_ref.network_step_words), decided by the lane's STEP_LO words
        assert int(px.network_step_words(seed, int(lane), 1, 11)[2 * j]) >> 16 == want >> 37


# ---- heuristic policy support -------------------------------------------------------
HEUR = [tuple(c) for c in golden_manifest().get("heuristic_cases", [])]


@pytest.mark.parametrize("case,env,kw", HEUR, ids=[c[0] for c in HEUR])
def test_heuristic_policy_support(oracle_lib, case, env, kw):
    """Side statistics, history sums, _generate_preferred and _select_target of the oracle == the reference run with
    use_heuristic=True on injected Philox words (tests/golden/heur_*.npz)."""
    o = oracle_lib.OracleEnv(env, **kw)
    heuristic_replay(case, env, o, OracleHeuristicOps(oracle_lib, o))


@pytest.mark.parametrize("env,kw,n,max_size,auto", [("rock", {}, 1001, None, True), ("rock", dict(board_size=15, num_rocks=15), 515, None, True),
                                                    ("stochrock", {}, 402, None, True), ("tag", {}, 999, None, True),
                                                    ("rock", {}, 600, 8, True), ("rock", dict(board_size=11, num_rocks=11), 300, 0, True),
                                                    ("tag", {}, 500, 2, True), ("rock", {}, 700, None, False), ("tag", dict(num_opponents=2), 300, None, False),
                                                    ("tiger", {}, 400, None, True), ("battleship", {}, 300, None, True),
                                                    ("network", {}, 300, None, True)],
                         ids=["rock_7_8", "rock_15_15", "stochrock", "tag", "rock-hist8", "rock_11_11-hist0", "tag-hist2", "rock-frozen",
                              "tag_2-frozen", "tiger", "battleship", "network"])
def test_lane_major_heuristic_driver_equals_the_per_step_call_sequence(oracle_lib, env, kw, n, max_size, auto):
    """or_batch_heuristic_steps (k steps of every lane in one call, what the GPU tests at 2^20 lanes compare the fused
    pomdp_heuristic_steps launches with) == the per-step sequence preferred -> pick -> step -> side statistics ->
    history.append of the batch functions that the heur_* fixtures pin to the reference: every output row, the state, the
    statistics, the history (sums or records) and prev_ob, across two calls."""
    from conftest import OracleHeuristicOps
    ol = oracle_lib
    o = ol.OracleEnv(env, **kw)
    seed, lane0, T1, T2 = 0xC0FFEE, (1 << 20) - 256, 37, 20
    is_rock = env in ("rock", "stochrock")
    ref = OracleHeuristicOps(ol, o)
    ref.max_size = max_size
    prev = ref.reset(n, seed, lane0, 5)
    st = o.new_state(n)
    prev_d = o.batch_reset(st, seed, lane0, 5).astype(np.int32)
    assert np.array_equal(prev, prev_d)
    b = ol.Belief(o, n) if is_rock else None
    h = ol.HistorySums(o, n, max_size=max_size)
    frozen = np.zeros(n, np.uint8)
    t = 6
    for k in (T1, T2):
        got = o.batch_heuristic_steps(st, h, b, prev_d, k, seed, lane0, t, auto_reset=auto, done_in=frozen, nthreads=4)
        for s in range(k):
            lc, nc = ref.preferred()
            a = ref.pick(lc, nc, t)
            ref._pre = ref.st.copy()
            ob, rew, done, bad = o.batch_step(ref.st, a, seed, lane0, t, auto_reset=auto, done=frozen.copy() if not auto else None)
            live = frozen == 0
            if ref.b is not None:
                ref.b.update(ref.st, np.where(live, a, 0), ob, np.where(live, done, 1) if not auto else done, auto_reset=auto)
            if auto:
                ref.h.append(prev, a, ob, done)
            else:                                  # frozen lanes append nothing: replay only the live ones' records
                keep = {k_: getattr(ref.h, k_).copy() for k_ in ("size", "last_action", "last_ob", "total_sample", "total_move")}
                recs = [r.copy() for r in ref.h.rec]
                ref.h.append(prev, a, ob, done, auto_reset=False)
                for k_, v in keep.items():
                    cur = getattr(ref.h, k_)
                    cur[..., ~live] = v[..., ~live]
                for r, v in zip(ref.h.rec, recs):
                    r[..., ~live] = v[..., ~live]
            want_a = np.where(live, a, -1)
            assert np.array_equal(got["action"][s], want_a), (s, "action")
            assert np.array_equal(got["ob"][s], ob) and np.array_equal(got["reward"][s], rew), s
            assert np.array_equal(got["done"][s], done), s
            reset_ob = 0 if is_rock or env in ("battleship", "network") else (2 if env == "tiger" else ref.reset_ob())
            prev = np.where(live, np.where((done != 0) & auto, reset_ob, ob), prev).astype(np.int32)
            if not auto:
                frozen = done.copy()
            t += 1
        assert np.array_equal(st, ref.st) and np.array_equal(prev_d, prev)
        for k_ in ("size", "last_action", "last_ob") + (("total_sample", "total_move") if max_size is None else ()):
            assert np.array_equal(getattr(h, k_), getattr(ref.h, k_)), k_
        if max_size is not None:
            for r, v in zip(h.rec, ref.h.rec):
                rows = np.arange(r.shape[0])[:, None] < h.size[None, :]
                assert np.array_equal(np.where(rows, r, 0), np.where(rows, v, 0))
        if is_rock:
            for k_, _ in ol.Belief.FIELDS:
                x, y = getattr(b, k_), getattr(ref.b, k_)
                assert ((x == y) | ((x != x) & (y != y))).all(), k_


@pytest.mark.parametrize("case,env,kw", CASES, ids=[c[0] for c in CASES])
def test_collect_returns_is_the_callers_loop_over_the_reference_rewards(oracle_lib, case, env, kw):
    """or_batch_collect_returns (what the GPU's returns-only sink is held to) against the reference callers' two lines —
    `r += discount * rw; discount *= .95` per step, `eps.append(r)` per episode, `sum(eps)` (network.py:175-191,
    rock.py:569-570) — run in python floats over the REFERENCE's own rewards and done flags of the mode-B fixtures
    (float64, so Network's `base - .1` unrounded): running return and discount, last finished return, the sequential sum
    over finished episodes, bit for bit; counts; the state."""
    g = load_golden("B", case)
    o = oracle_lib.OracleEnv(env, **kw)
    seed, t0, lanes = int(g["seed"]), int(g["t0"]), g["lanes"]
    L, T = g["actions"].shape
    discount = 1.0 if env == "battleship" else .95            # battleship.py:73, else rock.py:115 / tag.py:91 / tiger.py:55 / network.py:35
    starts = [0] + [i for i in range(1, L) if lanes[i] != lanes[i - 1] + 1] + [L]
    for s, e in zip(starts[:-1], starts[1:]):
        n, lane0 = e - s, int(lanes[s])
        st, st2 = o.new_state(n), o.new_state(n)
        o.batch_reset(st, seed, lane0, t0)
        o.batch_reset(st2, seed, lane0, t0)
        acc, cnt = oracle_lib.new_return_stats(n, pitch=n + 3)
        half = T // 2                                          # two calls: the statistics carry over
        o.batch_collect_returns(st, acc, cnt, discount, seed, lane0, t0 + 1, half, actions=g["actions"][s:e, :half].T)
        o.batch_collect_returns(st, acc, cnt, discount, seed, lane0, t0 + 1 + half, T - half, actions=g["actions"][s:e, half:].T)
        for i in range(T):
            o.batch_step(st2, g["actions"][s:e, i], seed, lane0, t0 + 1 + i)
        assert np.array_equal(st, st2)
        for li in range(n):
            r, disc, eps, last = 0, 1., [], float("nan")
            for i in range(T):
                rw = float(g["reward"][s + li, i])
                r += disc * rw
                disc *= discount
                if g["done"][s + li, i]:
                    eps.append(r)
                    last, r, disc = r, 0, 1.
            tot = 0.                                           # sum(eps), left to right (python >= 3.12 compensates inside sum())
            for x in eps:
                tot += x
            want = np.array([float(r), disc, last, tot], np.float64)
            assert np.array_equal(acc[:, li].view(np.uint64), want.view(np.uint64)), (case, li, acc[:, li], want)
            assert cnt[0, li] == len(eps) and cnt[1, li] == T
        assert np.all(acc[0, n:] == 0) and np.all(acc[1, n:] == 1) and np.all(np.isnan(acc[2, n:])) and np.all(acc[3, n:] == 0)   # padding untouched
        assert np.all(cnt[:, n:] == 0)

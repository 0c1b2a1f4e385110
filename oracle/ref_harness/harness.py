"""Drive the UNMODIFIED reference (d3sm0/gym_pomdp, /root/reference) to produce
golden traces.  TEST INFRASTRUCTURE, container-only: /root/reference does not
exist on the GPU box, so nothing here is imported by tests marked `gpu`, by
smoke() or by bench.py — they consume the committed fixtures in tests/golden/.

Two trace modes:

mode A ("MT-exact")       env.seed(s) on the process-global legacy MT19937
                          stream, exactly as a user of the reference would.
mode B ("Philox-injected") before every reference reset()/step() call the
                          global RandomState's MT19937 state is overwritten so
                          that its next 624 outputs are the 32-bit words of the
                          build's Philox stream for (seed, lane, t, stream).
                          numpy's own C code then derives uniform / binomial /
                          randint / choice from those words, so no numpy
                          algorithm is restated on the reference side.

No reference file is modified, copied or monkeypatched.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
REFERENCE_ROOT = os.environ.get("GYM_POMDP_REFERENCE", "/root/reference")

if _REPO not in sys.path:
    sys.path.insert(0, _REPO)
from oracle import philox_ref as px  # noqa: E402


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gym_pomdp"))


def load_reference():
    """Import the reference package under the gym/pygame stand-ins."""
    sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
    stubs = os.path.join(_HERE, "stubs")
    for p in (REFERENCE_ROOT, stubs):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gym  # noqa: F401  (the stub)
    import gym_pomdp  # noqa: F401  (the reference: registers its ids)
    import gym_pomdp.envs as envs
    return envs


# ---------------------------------------------------------------------------
# MT19937 state injection
# ---------------------------------------------------------------------------
def untemper(y):
    """Invert MT19937's output tempering (vectorised, uint32)."""
    y = np.asarray(y, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    # y ^= y >> 18
    y = y ^ (y >> np.uint64(18))
    # y ^= (y << 15) & 0xefc60000
    y = y ^ ((y << np.uint64(15)) & np.uint64(0xEFC60000))
    # y ^= (y << 7) & 0x9d2c5680   (iterate to recover all bits)
    x = y
    for _ in range(5):
        x = y ^ ((x << np.uint64(7)) & np.uint64(0x9D2C5680))
    y = x & np.uint64(0xFFFFFFFF)
    # y ^= y >> 11  (iterate)
    x = y
    for _ in range(3):
        x = y ^ (x >> np.uint64(11))
    return (x & np.uint64(0xFFFFFFFF)).astype(np.uint32)


N_INJECT = 624


def inject_words(words, rs=None):
    """Make the next len(words) (<= 624) uint32 outputs of `rs` (default: the
    global np.random stream) equal `words`."""
    key = np.zeros(N_INJECT, dtype=np.uint32)
    w = np.asarray(words, dtype=np.uint32)
    key[: len(w)] = untemper(w)
    state = ("MT19937", key, 0, 0, 0.0)
    if rs is None:
        np.random.set_state(state)
    else:
        rs.set_state(state)


def consumed_words(rs=None):
    st = np.random.get_state() if rs is None else rs.get_state()
    return int(st[2])


def inject_stream(seed, lane, t, stream, rs=None, env=None, env_kwargs=None, auto_reset=False):
    """Inject the words of (seed, lane, t, stream).  RockSample / StochasticRock use the split, quad-shared layout
    of oracle/philox_ref.py (rock_reset_words / rock_step_words), Network's step() the quad-shared 16-bit layout
    (network_step_words), Tiger the quad's STEP blocks for all three of its streams (tiger_words), Tag with one opponent the quad's word for a
    step's flight and for the auto-reset after it (tag_step_words / tag_auto_reset_words); everything else the plain
    sequential stream."""
    if env in ("rock", "stochrock") and stream in (px.STREAM_STEP, px.STREAM_RESET):
        if stream == px.STREAM_RESET:
            # an auto-reset draws from the step's own sensor blocks (philox_ref.rock_reset_words)
            w = px.rock_reset_words(seed, lane, t, (env_kwargs or {}).get("num_rocks", 8),
                                    auto_step_block=(2 if env == "stochrock" else 0) if auto_reset else None)
        else:
            w = px.rock_step_words(seed, lane, t, 2 if env == "stochrock" else 1)
        # pad with a recognisable filler: consuming more words than the layout defines must be noticed
        inject_words(np.concatenate([w, np.full(8, 0xDEADBEEF, np.uint32)]), rs)
        return len(w)
    if stream in (px.STREAM_STEP_SPACE, px.STREAM_RESET_SPACE) or (env == "tiger" and stream == px.STREAM_STEP):
        # Tiger (the only env with a gym-space RNG): every draw of call counter t reads the quad's STEP blocks
        w = px.tiger_words(seed, lane, t)
        inject_words(np.concatenate([w, np.full(8, 0xDEADBEEF, np.uint32)]), rs)
        return len(w)
    if env == "tag" and (env_kwargs or {}).get("num_opponents", 1) == 1 and (stream == px.STREAM_STEP or (stream == px.STREAM_RESET and auto_reset)):
        # the one-opponent game: a flight and the auto-reset after a successful TAG read the quad's STEP word
        if stream == px.STREAM_STEP:
            w = np.concatenate([px.tag_step_words(seed, lane, t), np.full(8, 0xDEADBEEF, np.uint32)])
        else:
            w = px.tag_auto_reset_words(seed, lane, t)
        inject_words(w, rs)
        return len(w)
    if env == "network" and stream == px.STREAM_STEP:
        w = px.network_step_words(seed, lane, t, (env_kwargs or {}).get("n_machines", 10) + 1)
        inject_words(np.concatenate([w, np.full(8, 0xDEADBEEF, np.uint32)]), rs)
        return len(w)
    inject_words(px.stream_words(seed, lane, t, stream, N_INJECT), rs)
    return N_INJECT


def inject_auto_reset(seed, lane, t, dealt_at, env=None, env_kwargs=None):
    """Words of the reference reset() that follows a done step at call counter t.  BattleShip (board contract,
    include/pomdp_hip.h): the board that moves in then is the reference's reset() on stream NEXT of the call counter
    `dealt_at` at which the lane's PREVIOUS board was dealt (by the initial reset or by an earlier auto-reset);
    RockSample / StochasticRock: the rotated pair of the step's own sensor blocks (stream STEP of (lane, t): a step never
    draws both); every other env: stream RESET of (lane, t)."""
    if env == "battleship":
        return inject_stream(seed, lane, dealt_at, px.STREAM_NEXT, env=env, env_kwargs=env_kwargs)
    return inject_stream(seed, lane, t, px.STREAM_RESET, env=env, env_kwargs=env_kwargs, auto_reset=True)


# ---------------------------------------------------------------------------
# env construction and compact integer state
# ---------------------------------------------------------------------------
ENV_CLASSES = {
    "rock": "RockEnv",
    "stochrock": "StochasticRockEnv",   # importable class; its gym id is broken by a typo (gym_pomdp/__init__.py:35)
    "tag": "TagEnv",
    "battleship": "BattleShipEnv",
    "tiger": "TigerEnv",
    "network": "NetworkEnv",
}


def make_ref_env(name, **kwargs):
    envs = load_reference()
    return getattr(envs, ENV_CLASSES[name])(**kwargs)


def compact_state(name, env):
    """Integer view of the reference env's hidden state (the parity format).

    rock:       [x, y, status_0 .. status_{K-1}]            status in {-1, 0, +1}
    tag:        [agent_idx, opp_idx_0 .., num_opp]          TagGrid.get_index
    battleship: [total_remaining, occ_0..occ_{C-1}, vis_0..vis_{C-1}], cell a = y*X + x
    tiger:      [state]
    network:    [s_0 .. s_{M-1}]
    """
    if name in ("rock", "stochrock"):
        s = env.state
        return np.array([s.agent_pos.x, s.agent_pos.y] + [r.status for r in s.rocks], dtype=np.int64)
    if name == "tag":
        s = env.state
        g = env.grid
        return np.array([g.get_index(s.agent_pos)] + [g.get_index(o) for o in s.opponent_pos] + [s.num_opp],
                        dtype=np.int64)
    if name == "battleship":
        g = env.grid
        occ, vis = [], []
        for a in range(g.n_tiles):
            c = g.board[a % g.x_size, a // g.x_size]
            occ.append(int(c.occupied))
            vis.append(int(c.visited))
        return np.array([env.state.total_remaining] + occ + vis, dtype=np.int64)
    if name == "tiger":
        return np.array([int(env.state)], dtype=np.int64)
    if name == "network":
        return np.array([int(v) for v in env.state], dtype=np.int64)
    raise KeyError(name)


def _space_rng():
    import gym.spaces as spaces
    return spaces.np_random


def _as_float(r):
    return float(r)


# ---------------------------------------------------------------------------
# mode A
# ---------------------------------------------------------------------------
def trace_mode_a(name, kwargs, seed, actions, space_seed=None):
    """Unmodified reference on the global MT stream.  reset() is called right
    after every done step (the batched build's auto-reset)."""
    env = make_ref_env(name, **kwargs)
    if name == "tiger":
        _space_rng().seed(seed if space_seed is None else space_seed)
    env.seed(seed)
    ob0 = env.reset()
    s0 = compact_state(name, env)
    T = len(actions)
    ob = np.zeros(T, np.int64)
    rew = np.zeros(T, np.float64)
    done = np.zeros(T, np.uint8)
    state_pre = np.zeros((T, len(s0)), np.int64)
    state = np.zeros((T, len(s0)), np.int64)
    reset_ob = np.full(T, -1, np.int64)
    for i, a in enumerate(actions):
        o, r, d, _ = env.step(int(a))
        ob[i], rew[i], done[i] = int(o), _as_float(r), int(bool(d))
        state_pre[i] = compact_state(name, env)
        if d:
            reset_ob[i] = int(env.reset())
        state[i] = compact_state(name, env)
    return dict(ob0=np.int64(ob0), state0=s0, actions=np.asarray(actions, np.int64), ob=ob, reward=rew,
                done=done, state_pre=state_pre, state=state, reset_ob=reset_ob)


# ---------------------------------------------------------------------------
# mode B
# ---------------------------------------------------------------------------
def trace_mode_b(name, kwargs, seed, lanes, actions, t0=0):
    """One reference env object per lane; RNG state injected before each call.

    actions: int[len(lanes), T].  Call `c` of the batched env uses t = t0 + c:
    the initial reset is t0, step i is t0 + 1 + i, and the auto-reset that
    follows a done step shares that step's t (stream RESET instead of STEP;
    RockSample: the step's own sensor blocks, BattleShip: the cached board — see
    inject_auto_reset).
    """
    lanes = list(lanes)
    actions = np.asarray(actions)
    L, T = actions.shape
    assert L == len(lanes)
    out = None
    space = _space_rng() if name == "tiger" else None
    max_used = 0
    for li, lane in enumerate(lanes):
        env = make_ref_env(name, **kwargs)
        lim = inject_stream(seed, lane, t0, px.STREAM_RESET, env=name, env_kwargs=kwargs)
        if space is not None:
            inject_stream(seed, lane, t0, px.STREAM_RESET_SPACE, space)
        ob0 = env.reset()
        assert consumed_words() <= lim
        max_used = max(max_used, consumed_words())
        s0 = compact_state(name, env)
        if out is None:
            S = len(s0)
            out = dict(ob0=np.zeros(L, np.int64), state0=np.zeros((L, S), np.int64),
                       ob=np.zeros((L, T), np.int64), reward=np.zeros((L, T), np.float64),
                       done=np.zeros((L, T), np.uint8), state_pre=np.zeros((L, T, S), np.int64),
                       state=np.zeros((L, T, S), np.int64), reset_ob=np.full((L, T), -1, np.int64))
        out["ob0"][li] = int(ob0)
        out["state0"][li] = s0
        dealt_at = t0
        for i in range(T):
            t = t0 + 1 + i
            lim = inject_stream(seed, lane, t, px.STREAM_STEP, env=name, env_kwargs=kwargs)
            if space is not None:
                inject_stream(seed, lane, t, px.STREAM_STEP_SPACE, space)
            o, r, d, _ = env.step(int(actions[li, i]))
            assert consumed_words() <= lim
            max_used = max(max_used, consumed_words())
            out["ob"][li, i], out["reward"][li, i], out["done"][li, i] = int(o), _as_float(r), int(bool(d))
            out["state_pre"][li, i] = compact_state(name, env)
            if d:
                lim = inject_auto_reset(seed, lane, t, dealt_at, env=name, env_kwargs=kwargs)
                dealt_at = t
                if space is not None:
                    inject_stream(seed, lane, t, px.STREAM_RESET_SPACE, space)
                out["reset_ob"][li, i] = int(env.reset())
                assert consumed_words() <= lim
                max_used = max(max_used, consumed_words())
            out["state"][li, i] = compact_state(name, env)
    assert max_used < N_INJECT, "a reference call consumed more words than were injected"
    out.update(lanes=np.asarray(lanes, np.int64), actions=actions.astype(np.int64), seed=np.int64(seed),
               t0=np.int64(t0), max_words_per_call=np.int64(max_used))
    return out


# ---------------------------------------------------------------------------
# planner hooks: _generate_legal and random rollouts
# ---------------------------------------------------------------------------
MAX_LEGAL = 160


def legal_list(env):
    lst = [int(a) for a in env._generate_legal()]
    assert len(lst) <= MAX_LEGAL
    return lst + [-1] * (MAX_LEGAL - len(lst)), len(lst)


def rollout_reference(name, kwargs, seed, root_lane0, n_roots, sims_per_root, depth, discount, t_reset, t0,
                      lane0, all_actions=False, keep_envs=None):
    """Reference-side statement of the build's rollout contract (oracle/pomdp_oracle.h: or_batch_rollout):
    root r is the reference env reset on stream RESET of (seed, root_lane0 + r, t_reset); simulation s of
    root r is a deep copy of it, advanced by the reference's own step() on injected STEP words, the action
    being list[(w * len(list)) >> 32] with list = env._generate_legal() (or all actions) and w word k of
    stream ROLLOUT at (seed, lane, t0)."""
    import copy
    space = _space_rng() if name == "tiger" else None
    n = n_roots * sims_per_root
    out = dict(ret=np.zeros(n, np.float64), n_steps=np.zeros(n, np.int64), first_action=np.full(n, -1, np.int64),
               last_ob=np.zeros(n, np.int64), terminated=np.zeros(n, np.uint8), root_state=None,
               root_legal=np.full((n_roots, MAX_LEGAL), -1, np.int64), root_legal_len=np.zeros(n_roots, np.int64))
    for r in range(n_roots):
        env = make_ref_env(name, **kwargs)
        inject_stream(seed, root_lane0 + r, t_reset, px.STREAM_RESET, env=name, env_kwargs=kwargs)
        if space is not None:
            inject_stream(seed, root_lane0 + r, t_reset, px.STREAM_RESET_SPACE, space)
        env.reset()
        s0 = compact_state(name, env)
        if out["root_state"] is None:
            out["root_state"] = np.zeros((n_roots, len(s0)), np.int64)
        out["root_state"][r] = s0
        out["root_legal"][r], out["root_legal_len"][r] = legal_list(env)
        n_act = env.action_space.n
        if keep_envs is not None:
            keep_envs.append(env)          # the root itself is never stepped here: simulations run on deep copies
        for s in range(sims_per_root):
            i = r * sims_per_root + s
            lane = lane0 + i
            e = copy.deepcopy(env)
            pol_words = px.stream_words(seed, lane, t0, px.STREAM_ROLLOUT, depth)
            ret, disc, k, done, ob = 0.0, 1.0, 0, False, 0
            while k < depth and not done:
                lst = list(range(n_act)) if all_actions else [int(a) for a in e._generate_legal()]
                if not lst:
                    break
                w = int(pol_words[k])
                a = lst[(w * len(lst)) >> 32]
                if k == 0:
                    out["first_action"][i] = a
                inject_stream(seed, lane, t0 + k, px.STREAM_STEP, env=name, env_kwargs=kwargs)
                if space is not None:
                    inject_stream(seed, lane, t0 + k, px.STREAM_STEP_SPACE, space)
                ob, rw, done, _ = e.step(a)
                term = disc * float(rw)
                ret = ret + term
                disc = disc * discount
                k += 1
            out["ret"][i], out["n_steps"][i], out["last_ob"][i], out["terminated"][i] = ret, k, int(ob), int(bool(done))
    return out


PLAN_CHUNK = 64


def plan_reduce_python(ret, first_action, n_roots, sims_per_root, n_actions):
    """The planner's reduction in PYTHON floats, in the order include/pomdp_hip.h states for pomdp_plan: per root and
    action, chunks of 64 simulations by index; within a chunk the returns whose first action is `a` are added in index
    order starting from 0.0; the chunk sums are added in chunk order starting from 0.0; the mean is one division.
    best = the visited action with the largest mean, lowest index on ties (-1: none)."""
    q = np.zeros((n_roots, n_actions), np.float64)
    visits = np.zeros((n_roots, n_actions), np.int64)
    best = np.full(n_roots, -1, np.int64)
    value = np.zeros(n_roots, np.float64)
    for r in range(n_roots):
        rr = [float(v) for v in ret[r * sims_per_root:(r + 1) * sims_per_root]]
        fa = [int(v) for v in first_action[r * sims_per_root:(r + 1) * sims_per_root]]
        b, bq = -1, 0.0
        for a in range(n_actions):
            total, cnt = 0.0, 0
            for c0 in range(0, sims_per_root, PLAN_CHUNK):
                part = 0.0
                for j in range(c0, min(c0 + PLAN_CHUNK, sims_per_root)):
                    if fa[j] == a:
                        part = part + rr[j]
                        cnt += 1
                total = total + part
            qa = total / cnt if cnt else 0.0
            q[r, a], visits[r, a] = qa, cnt
            if cnt and (b < 0 or qa > bq):
                b, bq = a, qa
        best[r], value[r] = b, (bq if b >= 0 else 0.0)
    return q, visits, best, value


def plan_reference(name, kwargs, seed, root_lane0, n_roots, sims_per_root, depth, discount, t_reset, t0, all_actions=False):
    """Reference-side statement of the planning step (include/pomdp_hip.h: pomdp_plan, then pomdp_<env>_step with the
    chosen actions): root r is the reference env reset on stream RESET of (seed, root_lane0 + r, t_reset); its simulation s
    is lane (root_lane0 + r) * sims_per_root + s of rollout_reference at call counter t0; the simulations' returns —
    the reference's own float64 arithmetic over its own step() — are reduced in Python floats in the stated order; the
    root then takes its best action in the reference's own step() at call counter t0 + depth (auto-reset as in
    trace_mode_b)."""
    space = _space_rng() if name == "tiger" else None
    envs = []
    tr = rollout_reference(name, kwargs, seed, root_lane0, n_roots, sims_per_root, depth, discount, t_reset, t0,
                           lane0=root_lane0 * sims_per_root, all_actions=all_actions, keep_envs=envs)
    n_act = envs[0].action_space.n
    q, visits, best, value = plan_reduce_python(tr["ret"], tr["first_action"], n_roots, sims_per_root, n_act)
    t_step = t0 + depth
    S = tr["root_state"].shape[1]
    out = dict(sim_ret=tr["ret"], sim_first_action=tr["first_action"], root_state=tr["root_state"], q=q, visits=visits, best=best,
               value=value, ob=np.zeros(n_roots, np.int64), reward=np.zeros(n_roots, np.float64), done=np.zeros(n_roots, np.uint8),
               state_pre=np.zeros((n_roots, S), np.int64), state=np.zeros((n_roots, S), np.int64))
    for r, env in enumerate(envs):
        assert best[r] >= 0
        lane = root_lane0 + r
        lim = inject_stream(seed, lane, t_step, px.STREAM_STEP, env=name, env_kwargs=kwargs)
        if space is not None:
            inject_stream(seed, lane, t_step, px.STREAM_STEP_SPACE, space)
        o, rw, d, _ = env.step(int(best[r]))
        assert consumed_words() <= lim
        out["ob"][r], out["reward"][r], out["done"][r] = int(o), _as_float(rw), int(bool(d))
        out["state_pre"][r] = compact_state(name, env)
        if d:
            inject_auto_reset(seed, lane, t_step, t_reset, env=name, env_kwargs=kwargs)
            if space is not None:
                inject_stream(seed, lane, t_step, px.STREAM_RESET_SPACE, space)
            env.reset()
        out["state"][r] = compact_state(name, env)
    return out


def compute_prob_trace(name, kwargs, seed, lanes, actions, t0=0):
    """`env._compute_prob(a, info["state"], o)` of the reference for every observation value o, right after
    each step of a mode-B trace without auto-reset of the queried state (the query is made before reset())."""
    lanes = list(lanes)
    actions = np.asarray(actions)
    L, T = actions.shape
    space = _space_rng() if name == "tiger" else None
    out = None
    for li, lane in enumerate(lanes):
        env = make_ref_env(name, **kwargs)
        inject_stream(seed, lane, t0, px.STREAM_RESET, env=name, env_kwargs=kwargs)
        if space is not None:
            inject_stream(seed, lane, t0, px.STREAM_RESET_SPACE, space)
        env.reset()
        dealt_at = t0
        n_obs = env.observation_space.n
        if out is None:
            out = dict(prob=np.zeros((L, T, n_obs), np.float64), ob=np.zeros((L, T), np.int64),
                       state_pre=None, done=np.zeros((L, T), np.uint8))
        for i in range(T):
            t = t0 + 1 + i
            inject_stream(seed, lane, t, px.STREAM_STEP, env=name, env_kwargs=kwargs)
            if space is not None:
                inject_stream(seed, lane, t, px.STREAM_STEP_SPACE, space)
            a = int(actions[li, i])
            o, r, d, info = env.step(a)
            st = compact_state(name, env)
            if out["state_pre"] is None:
                out["state_pre"] = np.zeros((L, T, len(st)), np.int64)
            out["state_pre"][li, i] = st
            out["ob"][li, i], out["done"][li, i] = int(o), int(bool(d))
            for q in range(n_obs):
                out["prob"][li, i, q] = float(env._compute_prob(a, info["state"], q))
            if d:
                inject_auto_reset(seed, lane, t, dealt_at, env=name, env_kwargs=kwargs)
                dealt_at = t
                if space is not None:
                    inject_stream(seed, lane, t, px.STREAM_RESET_SPACE, space)
                env.reset()
    out.update(lanes=np.asarray(lanes, np.int64), actions=actions.astype(np.int64), seed=np.int64(seed), t0=np.int64(t0))
    return out


# ---------------------------------------------------------------------------
# heuristic policy: side statistics, _generate_preferred, _select_target
# ---------------------------------------------------------------------------
MAX_PREF = 32


class _TagRecord(object):
    """tag.py:233-239 reads `history.size`, `history[-1].action` and `history[-1].ob`; the module it imports them
    from (gym_pomdp.envs.history, tag.py:303) is not part of the reference tree, so the harness supplies the
    minimal container with exactly those attributes."""
    __slots__ = ("action", "ob")

    def __init__(self, action, ob):
        self.action, self.ob = action, ob


class _TagHistory(list):
    @property
    def size(self):
        return len(self)


def heuristic_trace(name, kwargs, seed, lanes, T, t0=0, max_size=None):
    """Reference envs driven by their own `_generate_preferred(history)` (use_heuristic=True for RockSample), one env
    per lane, Philox-injected like mode B.  The action of lane L at call t is list[(w * len(list)) >> 32] with w the
    synthetic policy's word (philox_ref.action_word) and list chosen by L & 3: 0, 1 -> _generate_preferred(history),
    2 -> _generate_legal(), 3 -> all actions — so that the side statistics also see what the heuristic itself never
    does.  The history is the reference's own History of Transition(observation, action, reward, next_observation,
    done) records (rock.py:525-550), rebuilt empty after every reset — `History(max_size)` when max_size is given
    (rock.py:533-544: the oldest record is popped once the list holds more than max_size).  Recorded per (lane, step): the preferred list
    *before* the step, _select_target, the action, (ob, reward, done), and after the call (post auto-reset) the
    compact state and every rock's count / measured / lkv / lkw / prob_valuable."""
    load_reference()
    import gym_pomdp.envs.rock as rk
    is_rock = name in ("rock", "stochrock")
    lanes = list(lanes)
    L = len(lanes)
    out = None
    for li, lane in enumerate(lanes):
        env = make_ref_env(name, use_heuristic=True, **kwargs) if is_rock else make_ref_env(name, **kwargs)
        inject_stream(seed, lane, t0, px.STREAM_RESET, env=name, env_kwargs=kwargs)
        ob_prev = int(env.reset())
        hist = rk.History(max_size) if is_rock else _TagHistory()
        n_act = env.action_space.n
        s0 = compact_state(name, env)
        K = len(env.state.rocks) if is_rock else 0
        if out is None:
            out = dict(pref=np.full((L, T, MAX_PREF), -1, np.int64), pref_len=np.zeros((L, T), np.int64),
                       target=np.full((L, T), -1, np.int64), action=np.zeros((L, T), np.int64),
                       ob=np.zeros((L, T), np.int64), reward=np.zeros((L, T), np.float64),
                       done=np.zeros((L, T), np.uint8), state0=np.zeros((L, len(s0)), np.int64),
                       state=np.zeros((L, T, len(s0)), np.int64),
                       count=np.zeros((L, T, K), np.int64), measured=np.zeros((L, T, K), np.int64),
                       lkv=np.zeros((L, T, K), np.float64), lkw=np.zeros((L, T, K), np.float64),
                       prob_valuable=np.zeros((L, T, K), np.float64))
        out["state0"][li] = s0
        for i in range(T):
            t = t0 + 1 + i
            pref = [int(a) for a in env._generate_preferred(hist)]
            assert 0 < len(pref) <= MAX_PREF
            out["pref"][li, i, : len(pref)] = pref
            out["pref_len"][li, i] = len(pref)
            if is_rock:
                out["target"][li, i] = int(rk.RockEnv._select_target(env.state, env.grid.x_size))
            pol = lane & 3
            lst = pref if pol < 2 else ([int(a) for a in env._generate_legal()] if pol == 2 else list(range(n_act)))
            a = lst[(px.action_word(seed, lane, t) * len(lst)) >> 32]
            lim = inject_stream(seed, lane, t, px.STREAM_STEP, env=name, env_kwargs=kwargs)
            o, r, d, _ = env.step(a)
            assert consumed_words() <= lim
            if is_rock:
                hist.append(rk.Transition(observation=ob_prev, action=a, reward=r, next_observation=int(o), done=d))
            else:
                hist.append(_TagRecord(a, int(o)))
            out["action"][li, i], out["ob"][li, i] = a, int(o)
            out["reward"][li, i], out["done"][li, i] = float(r), int(bool(d))
            if d:
                inject_auto_reset(seed, lane, t, t, env=name, env_kwargs=kwargs)   # (no BattleShip here: dealt_at unused)
                ob_prev = int(env.reset())
                hist = rk.History(max_size) if is_rock else _TagHistory()
            else:
                ob_prev = int(o)
            out["state"][li, i] = compact_state(name, env)
            for j in range(K):
                rr = env.state.rocks[j]
                out["count"][li, i, j], out["measured"][li, i, j] = rr.count, rr.measured
                out["lkv"][li, i, j], out["lkw"][li, i, j] = rr.lkv, rr.lkw
                out["prob_valuable"][li, i, j] = rr.prob_valuable
    out.update(lanes=np.asarray(lanes, np.int64), seed=np.int64(seed), t0=np.int64(t0))
    return out

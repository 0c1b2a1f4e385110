"""RockSample — batched mirror of gym_pomdp/envs/rock.py:96-407 (`RockEnv`)."""
import ctypes as C

import torch

from .. import _native, tables
from .base import BatchedEnv, staggered


class _BeliefTracker(object):
    """RockSample's side statistics (rock.py:78-86 count / measured / lkv / lkw / prob_valuable), which the
    reference's step() maintains on every CHECK (rock.py:177-191): float64 / int32 tensors [K, N] updated by one
    extra launch after each step (include/pomdp_hip.h: pomdp_rock_belief_update)."""
    FIELDS = (("count", torch.int32), ("measured", torch.int32), ("lkv", torch.float64), ("lkw", torch.float64),
              ("prob_valuable", torch.float64))

    def __init__(self, env):
        self.env = env
        k, n = env.num_rocks, env.batch_size
        cols = staggered([((k, n), dt) for _, dt in self.FIELDS] + [((n,), torch.int32)], env.device)   # one allocation, spread starts
        self.tensors = {f: c for (f, _), c in zip(self.FIELDS, cols)}
        self.check_ok = cols[-1]                                                # derived word (pomdp_rock_belief.check_ok)
        self.ptrs = _native.RockBelief(*([self.tensors[f].data_ptr() for f, _ in self.FIELDS] + [self.check_ok.data_ptr()]))
        self.ref = C.byref(self.ptrs)
        self.on_reset()

    def on_reset(self, where=None):
        env = self.env
        with torch.cuda.device(env.device):
            rc = env._lib.pomdp_rock_belief_reset(env._params_ref, self.ref, None if where is None else where.data_ptr(),
                                                  env.batch_size, env._stream())
        _native.check(rc, "pomdp_rock_belief_reset")

    def on_step(self, action, ob, done, flags):
        env = self.env
        with torch.cuda.device(env.device):
            rc = env._lib.pomdp_rock_belief_update(env._params_ref, env._state.data_ptr(), action.data_ptr(),
                                                   ob.data_ptr(), done.data_ptr(), self.ref, env.batch_size, flags,
                                                   env._stream())
        if rc:
            _native.check(rc, "pomdp_rock_belief_update")


def make_params(board_size=7, num_rocks=8, stochastic=False, p_move=.8):
    """rock.py:99-118: validates like the reference's ctor assert, builds the rock-id grid
    (every listed coordinate is stamped), start position and the sensor threshold table."""
    assert board_size in tables.ROCK_CONFIG and num_rocks in tables.ROCK_CONFIG[board_size][0], \
        "unsupported RockSample(%r, %r)" % (board_size, num_rocks)
    _, init_pos, rock_pos = tables.ROCK_CONFIG[board_size]
    if num_rocks > len(rock_pos):
        # (2,2) and (4,4) pass the reference's assert but raise IndexError in reset()
        raise IndexError("RockSample(%d,%d): only %d rock positions are configured"
                         % (board_size, num_rocks, len(rock_pos)))
    p = _native.RockParams()
    p.size, p.num_rocks = board_size, num_rocks
    p.start_x, p.start_y = init_pos
    for i in range(256):
        p.grid[i] = -1
    for idx, (x, y) in enumerate(rock_pos):
        p.grid[x * 16 + y] = idx
        p.rock_x[idx], p.rock_y[idx] = x, y
    for d in range(32):
        p.thr[d] = tables.ROCK_THR[min(d, len(tables.ROCK_THR) - 1)]
        p.eff[d] = tables.ROCK_EFF[min(d, len(tables.ROCK_EFF) - 1)]
    if stochastic:
        p.stochastic = 1
        if p_move == .8:
            p.act_thr = tables.TAG_MOVE_THR          # binomial(1, .8): the same captured threshold
        else:
            if not 0. < p_move < 1.:
                raise ValueError("StochasticRock: p_move must lie in (0, 1), got %r" % (p_move,))
            thr, sense = tables.bernoulli_threshold(p_move)
            p.act_thr, p.act_gt = thr, int(sense == "gt")   # numpy's binomial(1, p): [U > thr] for p <= .5 (rock.py:443)
    words = 1 if num_rocks <= 12 else 2
    return p, words, 5 + num_rocks, 3


class RockEnv(BatchedEnv):
    """Actions 0 N, 1 E, 2 S, 3 W, 4 SAMPLE, 5+i CHECK rock i (rock.py:18-23, 171-172);
    observations 0 NULL, 1 BAD, 2 GOOD (rock.py:12-15); reward int32 in {-100, -10, 0, 10}.

    Deliberate divergence: SAMPLE on a cell whose stamped rock id is >=
    num_rocks raises IndexError in the reference; here it is "no rock" (-100, done)."""
    env_name = "rock"
    reward_dtype = torch.int32

    def __init__(self, board_size=7, num_rocks=8, use_heuristic=False, track_belief=None, **batch_kwargs):
        """`use_heuristic` as in the reference (rock.py:99, 294).  `track_belief` (default: use_heuristic) keeps the
        per-rock side statistics the reference always keeps; off by default because they cost a second launch per
        step and nothing but the heuristic reads them."""
        self.board_size = board_size
        self.num_rocks = num_rocks
        self.p_move = getattr(self, "p_move", None)
        self._use_heuristic = use_heuristic
        self._discount = .95         # rock.py:115
        self._reward_range = 20      # rock.py:116
        self._penalization = -100    # rock.py:117
        self._setup(**batch_kwargs)
        if use_heuristic if track_belief is None else track_belief:
            with torch.cuda.device(self.device):
                self._tracker = _BeliefTracker(self)

    @property
    def belief(self):
        """dict of the side-statistic tensors [K, N] (count, measured int32; lkv, lkw, prob_valuable float64), or
        None when they are not tracked."""
        return None if self._tracker is None else self._tracker.tensors

    def _belief_ref(self):
        if self._tracker is None:
            raise RuntimeError("RockEnv: side statistics are not tracked (construct with use_heuristic=True or "
                               "track_belief=True)")
        return self._tracker.ref

    def set_belief(self, belief):
        """Overwrite the side statistics (what the reference's `_set_state(info["state"])` does through
        `rock.__dict__.update(r)`, rock.py:200-203); set_state() alone leaves fresh Rock statistics."""
        for f, _ in _BeliefTracker.FIELDS:
            self._tracker.tensors[f].copy_(torch.as_tensor(belief[f], device=self.device).reshape(self.num_rocks, -1))
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_rock_belief_refresh(self._params_ref, self._tracker.ref, self.batch_size, self._stream())
            _native.check(rc, "pomdp_rock_belief_refresh")

    def _generate_preferred(self, history):
        """rock.py:293-374.  Without use_heuristic: `_generate_legal()`."""
        if not self._use_heuristic:
            return self._generate_legal()
        return super()._generate_preferred(history)

    def select_target(self, state=None):
        """`_select_target` (rock.py:389-399) per lane: index of the nearest uncollected rock whose count is >= 0
        (straight-line distance, lowest index on ties), -1 if none -> int32[N]."""
        st = self._state if state is None else self._checked_state(state, self.batch_size, "select_target")
        out = torch.empty(self.batch_size, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_rock_select_target(self._params_ref, st.data_ptr(), self._belief_ref(), out.data_ptr(),
                                                    self.batch_size, self._stream())
            _native.check(rc, "pomdp_rock_select_target")
        return out

    def _select_target(self, rock_state=None, x_size=None):
        """Reference signature (a staticmethod there, taking the state object and the board size)."""
        t = self.select_target(rock_state)
        return int(t.item()) if self.batch_size == 1 else t

    def _build_params(self):
        return make_params(self.board_size, self.num_rocks)

    def _validate_state(self, st, what):
        """the agent stands on the board: the kernels' (position, action) tables hold the board's cells only"""
        w = st[0].to(torch.int64)
        if bool((((w & 15) >= self.board_size) | (((w >> 4) & 15) >= self.board_size)).any()):
            raise ValueError("%s: RockSample state with the agent off the %d x %d board" % (what, self.board_size, self.board_size))

    def decode_state(self):
        """Reference-format view: int64 [N, 2 + K] = [x, y, status_0..status_{K-1}], status in {-1,0,+1}."""
        s = self._state.to(torch.int64) & 0xFFFFFFFF
        v = s[0] if self.state_words == 1 else (s[0] | (s[1] << 32))
        cols = [v & 15, (v >> 4) & 15] + [((v >> (8 + 2 * j)) & 3) - 1 for j in range(self.num_rocks)]
        return torch.stack(cols, dim=1)

    def _decode_state(self, state, as_array=True):
        """Inverse of `_encode_state` for a batch (the reference's `_decode_state` builds a RockState from its
        dict encoding, rock.py:196-212): int [N, 1 + K] = [x_size * y + x, status..] -> packed int32
        [state_words, N] as set_state() takes it."""
        s = torch.as_tensor(state, device=self.device).to(torch.int64).reshape(-1, 1 + self.num_rocks)
        v = (s[:, 0] % self.board_size) | ((s[:, 0] // self.board_size) << 4)
        for j in range(self.num_rocks):
            v = v | ((s[:, 1 + j] + 1) << (8 + 2 * j))
        words = [v & 0xFFFFFFFF] + ([v >> 32] if self.state_words == 2 else [])
        words = [torch.where(w >= 1 << 31, w - (1 << 32), w).to(torch.int32) for w in words]
        return torch.stack(words, dim=0)

    def _encode_state(self, state=None):
        """The reference's array encoding of the state, `[x_size * y + x, status_0 .. status_{K-1}]`
        (rock.py:196-212 `_decode_state(as_array=True)`, 376-381 `__dict2np__`): int64 [N, 1 + K]."""
        d = self.decode_state() if state is None else state
        return torch.cat([(d[:, 1] * self.board_size + d[:, 0]).unsqueeze(1), d[:, 2:]], dim=1)


class StochasticRockEnv(RockEnv):
    """gym_pomdp/envs/rock.py:428-504: RockSample where the whole action is skipped with probability
    1 - p_move, the penalty is 0 and only the east exit terminates.  (The reference registers it as
    "StochasticRock-v0" with a typo in the entry point, gym_pomdp/__init__.py:32-36, so `gym.make` fails
    there; the class itself works and is what the fixtures were generated from.)"""

    def __init__(self, board_size=7, num_rocks=8, use_heuristic=False, p_move=.8, **batch_kwargs):
        self.p_move = p_move
        super().__init__(board_size, num_rocks, use_heuristic, **batch_kwargs)
        self._penalization = 0       # rock.py:432

    def _build_params(self):
        return make_params(self.board_size, self.num_rocks, stochastic=True, p_move=self.p_move)

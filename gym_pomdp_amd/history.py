"""History / Transition — the planner-side record `_generate_preferred(history)` reads
(gym_pomdp/envs/rock.py:525-550; tag.py:233-239 reads `history.size`, `history[-1].action`, `history[-1].ob`).

Batched, device-resident: instead of a list of records per lane the history keeps, per lane, exactly what the
heuristics take from it — size, the last action and observation, and for RockSample the two per-rock sums of
rock.py:303-310 / 327-334 (include/pomdp_hip.h: pomdp_history).  `History(env, max_size=k)` is the reference's bounded
history (rock.py:533-544: append() pops the oldest record once the list holds more than k, so it settles at k + 1
records): `size` stops at k + 1 and, for RockSample, the window itself is kept — one byte per transition in a ring of
k + 1 rows — so that a record's contribution leaves the two sums when the record leaves the window.
"""
import ctypes as C
from typing import NamedTuple

import torch

from . import _native


RING_MAX_BYTES = 64 << 30        # a bounded RockSample history's window, (max_size + 1) x N bytes: refuse what cannot fit beside the env


class Transition(NamedTuple):
    """rock.py:525-530, same field order.  Fields are int32[N] / uint8[N] tensors (python scalars when N == 1)."""
    observation: object
    action: object
    reward: object
    next_observation: object
    done: object


class History(object):
    def __init__(self, env, max_size=None, observation=None):
        """`observation`: what the agent currently sees (int32[N]); only env.heuristic_steps() needs it, as the
        `observation` field of the next transition.  Defaults to what env.reset() just returned."""
        if max_size is not None and not 0 <= int(max_size) < (1 << 31) - 1:
            raise ValueError("History: max_size must be None or a non-negative int (rock.py:533-544)")
        self._max_size = None if max_size is None else int(max_size)
        self._env = env
        n, dev = env.batch_size, env.device
        if observation is None and getattr(env, "_last_reset", None) is not None and env._last_reset[1] + 1 == env.call_counter:
            observation = env._last_reset[0]
        self.prev_ob = None if observation is None else \
            torch.as_tensor(observation, device=dev).to(torch.int32).reshape(n).clone()
        self._kind = _native.ENV_KIND[env.env_name]
        k = env.num_rocks if env.env_name == "rock" else 0
        from .envs.base import staggered           # one allocation, column starts spread over the HBM channels
        (self._size, self.last_action, self.last_ob, self.total_sample, self.total_move, self.move_ok) = staggered(
            [((n,), torch.int32), ((n,), torch.int32), ((n,), torch.int32), ((k, n), torch.int32), ((k, n), torch.int32),
             ((n if k else 0,), torch.int32)], dev)                                # move_ok: derived, bit j = total_move[j] >= 0, bit 16 + j = total_sample[j] > 0
        bounded = self._max_size is not None
        if bounded and k and (self._max_size + 1) * n > RING_MAX_BYTES:
            # the window is one byte per kept transition and lane (action in bits 0-4 — RockSample has at most 21 actions —
            # next observation in bits 5-6, "observation was BAD" in bit 7), allocated up front
            raise ValueError("History(max_size=%d) for %d lanes needs a %.1f GB window (one byte per transition and lane); "
                             "the limit is %d GB" % (self._max_size, n, (self._max_size + 1) * n / 1e9, RING_MAX_BYTES >> 30))
        self.ring = torch.zeros((self._max_size + 1, n) if bounded and k else (0, n), dtype=torch.uint8, device=dev)
        self.head = torch.zeros(n if bounded else 0, dtype=torch.int32, device=dev)
        self._ptrs = _native.HistoryPtrs(self._size.data_ptr(), self.last_action.data_ptr(), self.last_ob.data_ptr(),
                                         self.total_sample.data_ptr() if k else None,
                                         self.total_move.data_ptr() if k else None,
                                         self.move_ok.data_ptr() if k else None,
                                         self.ring.data_ptr() if bounded and k else None,
                                         self.head.data_ptr() if bounded else None,
                                         self._max_size if bounded else -1, 0)
        self._ref = C.byref(self._ptrs)
        self.clear()

    @property
    def size(self):
        """rock.py:546-548.  python int when batch_size == 1, else int32[N]."""
        return int(self._size.item()) if self._env.batch_size == 1 else self._size

    def _as(self, v, dtype):
        if isinstance(v, torch.Tensor) and v.dtype == dtype and v.device == self._env.device and v.is_contiguous():
            return v
        if isinstance(v, torch.Tensor) and v.dtype == torch.bool and dtype == torch.uint8:
            return v.to(self._env.device).contiguous().view(torch.uint8)
        return torch.as_tensor(v, device=self._env.device).to(dtype).reshape(self._env.batch_size).contiguous()

    def clear(self, where=None):
        """Empty history for every lane, or for the lanes where `where` is set (a new History() per episode)."""
        env = self._env
        w = None if where is None else self._as(where, torch.uint8)
        with torch.cuda.device(env.device):
            rc = env._lib.pomdp_history_clear(self._kind, env._params_ref, self._ref,
                                              None if w is None else w.data_ptr(), env.batch_size, env._stream())
            _native.check(rc, "pomdp_history_clear")

    def append(self, transition, auto_reset=None):
        """rock.py:541-544.  With auto_reset (default: the env's setting) a lane whose transition is terminal starts
        the next episode with an empty history, as a planner that builds a new History() per episode would."""
        env = self._env
        ar = env.auto_reset if auto_reset is None else bool(auto_reset)
        obs = self._as(transition.observation, torch.int32)
        act = self._as(transition.action, torch.int32)
        nxt = self._as(transition.next_observation, torch.int32)
        done = self._as(transition.done, torch.uint8)
        with torch.cuda.device(env.device):
            rc = env._lib.pomdp_history_append(self._kind, env._params_ref, self._ref, obs.data_ptr(), act.data_ptr(),
                                               nxt.data_ptr(), done.data_ptr(), env.batch_size,
                                               _native.POMDP_AUTO_RESET if ar else 0, env._stream())
            _native.check(rc, "pomdp_history_append")

    def __repr__(self):
        return "size:%s" % (self.size,)


class Returns(object):
    """Per-lane discounted return of the heuristic rollout loop, `r += rw * discount; discount *= env._discount`
    (rock.py:569-570), accumulated on the device by env.heuristic_steps(history, steps, returns=...): `ret`, `disc`
    (running, float64[N]) and `ret_done` (return of the lane's last finished episode, NaN until one finishes)."""

    def __init__(self, env, discount=None):
        n, dev = env.batch_size, env.device
        self.discount = float(env._discount if discount is None else discount)
        self.ret = torch.zeros(n, dtype=torch.float64, device=dev)
        self.disc = torch.ones(n, dtype=torch.float64, device=dev)
        self.ret_done = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
        self._ptrs = _native.Returns(self.discount, self.ret.data_ptr(), self.disc.data_ptr(), self.ret_done.data_ptr())
        self._ref = C.byref(self._ptrs)


class EpisodeStats(object):
    """What the reference's callers keep of a rollout under the random policy — `r += discount * rw; discount *= .95` per
    step, one return per episode, `sum(eps) / len(eps)` over episodes (network.py:175-191, rock.py:553-575) — per lane and
    device-resident, accumulated by env.collect_returns(steps, stats) with nothing written per step
    (include/pomdp_hip.h: pomdp_return_stats).  Views of the two buffers: `ret`, `disc` (running, float64[N]), `ret_done`
    (return of the lane's last finished episode; NaN until one finishes), `ret_sum` (sum of the returns of its finished
    episodes), `episodes`, `steps` (int32[N]).  The reward is the reference's own float64 value (Network: base - .1 /
    base - 2.5, not the float32 the reward column holds)."""

    def __init__(self, env, discount=None):
        n, dev = env.batch_size, env.device
        self.discount = float(env._discount if discount is None else discount)
        self.pitch = -(-n // 4) * 4
        self.acc = torch.zeros((4, self.pitch), dtype=torch.float64, device=dev)
        self.cnt = torch.zeros((2, self.pitch), dtype=torch.int32, device=dev)
        self.ret, self.disc, self.ret_done, self.ret_sum = (self.acc[q, :n] for q in range(4))
        self.episodes, self.steps = self.cnt[0, :n], self.cnt[1, :n]
        self._n = n
        self.reset()
        self._ptrs = _native.ReturnStats(self.discount, self.acc.data_ptr(), self.cnt.data_ptr(), self.pitch)
        self._ref = C.byref(self._ptrs)

    def reset(self):
        self.acc.zero_()
        self.acc[1].fill_(1.0)
        self.acc[2].fill_(float("nan"))
        self.cnt.zero_()

    def mean_return(self):
        """sum(eps) / len(eps) over every finished episode of every lane (network.py:189) -> python float.  Synchronises."""
        e = int(self.episodes.sum().item())
        return float(self.ret_sum.sum().item()) / e if e else float("nan")

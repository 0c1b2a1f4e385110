"""BatchedEnv — host-side mirror of the reference's gym.Env duck type
(`reset() -> ob`, `step(a) -> (ob, reward, done, info)`, `seed`, `action_space`,
`observation_space`; gym_pomdp/envs/rock.py:96-271 and siblings) for a batch of
independent env instances ("lanes") whose state lives in HBM as packed int32
columns and is advanced by the HIP kernels behind include/pomdp_hip.h.

Semantics added by batching:
  * lanes are globally numbered `lane_offset + i`; every random draw depends only on
    (seed, global lane, call counter t, stream id), so sharding a batch over GPUs or
    changing the launch geometry never changes a result;
  * the call counter t advances by one on every reset()/step() call;
  * `auto_reset=True` (default for batch_size > 1): a lane that reports done gets a
    fresh episode inside the same step() call — the terminal (ob, reward, done=1) is
    returned, the stored state is the new episode's;
  * `auto_reset=False` (default for batch_size == 1, the reference's behaviour): done
    lanes freeze; with batch_size == 1 stepping a done env raises AssertionError
    exactly like the reference (rock.py:126).
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _native, compat
from ..spaces import Discrete


def _random_seed():
    return int.from_bytes(os.urandom(8), "little")


def _public_raw_stream(dev_index):
    return torch.cuda.current_stream(dev_index).cuda_stream


def _resolve_fast_handles(c_module=None):
    """(raw hipStream_t of torch's current stream on a device, current device index) as plain callables.  torch keeps
    both as private C functions (no Stream object built, no device query through python); they are not a stable API, so
    they are looked up ONCE here and anything missing falls back to the public calls (tests/test_host_logic.py)."""
    c_module = torch._C if c_module is None else c_module
    raw = getattr(c_module, "_cuda_getCurrentRawStream", None)
    cur = getattr(c_module, "_cuda_getDevice", None)
    return (raw if callable(raw) else _public_raw_stream), (cur if callable(cur) else torch.cuda.current_device)


_raw_stream, _current_device = _resolve_fast_handles()


STAGGER_BYTES = 4096


def staggered(specs, device):
    """One allocation carved into the given columns — [(shape, dtype), ...] -> [tensor, ...], zero-filled — with the k-th
    column starting k * 4 KB past a 4 KB boundary of its own.  Why: separate allocations of this size all start on the same
    2 MB boundary, a kernel that touches element i of every column at the same time (every step kernel does) then keeps
    hitting the same HBM channel, and the store stream of a fused launch runs 6 % (RockSample) to 14 % (Tiger) slower than
    with the columns spread (profiles/r02d_store_layout.txt)."""
    offs, total = [], 0
    for k, (shape, dtype) in enumerate(specs):
        nbytes = int(torch.Size(shape).numel()) * torch.empty((), dtype=dtype).element_size()
        total = -(-total // STAGGER_BYTES) * STAGGER_BYTES + (k % 16) * STAGGER_BYTES
        offs.append((total, nbytes))
        total += nbytes
    pool = torch.zeros(total + STAGGER_BYTES, dtype=torch.uint8, device=device)
    return [pool[o:o + nb].view(dtype).view(shape) for (o, nb), (shape, dtype) in zip(offs, specs)]


class BatchedEnv(compat.EnvBase):
    """A `gym.Env` when an old-API gym is importable (compat.EnvBase), so that `gym.make()`'s wrappers and type checks
    accept it; either way it carries what `gym.make` touches: `unwrapped`, `spec`, `np_random`, `render_mode`, `metadata`."""
    metadata = {"render.modes": ["ansi"], "render_modes": ["ansi"]}      # gym <= 0.21 / gym >= 0.22 spelling
    reward_range = (-float("inf"), float("inf"))
    spec = None               # gym.make: `env.unwrapped.spec = spec` (gym/envs/registration.py, EnvSpec.make)
    render_mode = None
    env_name = None           # "rock", "tag", ... (C-ABI entry-point infix)
    reward_dtype = torch.int32

    # ---- subclasses provide -------------------------------------------------
    def _build_params(self):  # -> (ctypes Structure, words per lane, n_actions, n_obs)
        raise NotImplementedError

    # ---- construction ---------------------------------------------------------
    def _setup(self, batch_size=1, device=None, seed=None, auto_reset=None, lane_offset=0, reuse_buffers=False):
        batch_size = int(batch_size)
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        if lane_offset < 0 or lane_offset + batch_size > 1 << 32:
            raise ValueError("lane range must lie in [0, 2^32)")
        self.batch_size = batch_size
        self.num_envs = batch_size
        self.lane_offset = int(lane_offset)
        self._auto_reset = (batch_size > 1) if auto_reset is None else bool(auto_reset)
        self.reuse_buffers = bool(reuse_buffers)
        self._params, self.state_words, n_actions, n_obs = self._build_params()
        self._params_ref = C.byref(self._params)
        self.action_space = Discrete(n_actions)
        self.observation_space = Discrete(n_obs)
        self._seed = _random_seed() if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self._t = 0
        self._lib = _native.lib()                       # raises if the HIP library is missing
        self._reset_fn = getattr(self._lib, "pomdp_%s_reset" % self.env_name)
        self._step_fn = getattr(self._lib, "pomdp_%s_step" % self.env_name)
        if not torch.cuda.is_available():
            raise RuntimeError("gym_pomdp_amd: no GPU visible — the batched envs run on MI355X only "
                               "(there is no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (ROCm) device, got %s" % self.device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        n = batch_size
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._state, self._ob, self._reward, self._done = staggered(
            [((self.state_words, n), torch.int32), ((n,), torch.int32), ((n,), self.reward_dtype), ((n,), torch.uint8)], self.device)
        if n == 1:
            # scalar mode (the reference's usage): the step kernel writes ob / reward / done straight into 16 bytes of
            # pinned host memory (device-visible under unified addressing), so a step is one launch + one stream
            # synchronisation, no copy; the action is passed as a pointer into a device-resident table of all action
            # values, so there is no per-step H2D copy either
            self._scalar_buf = torch.zeros(4, dtype=torch.int32, device=self.device)
            self._ob = self._scalar_buf[0:1]
            self._reward = self._scalar_buf[1:2].view(self.reward_dtype)
            self._done = self._scalar_buf[2:3].view(torch.uint8)[0:1]
            self._action_table = torch.arange(n_actions, dtype=torch.int32, device=self.device)
            self._host_out = torch.zeros(4, dtype=torch.int32).pin_memory()
            self._host_reward = self._host_out[1:2].view(self.reward_dtype)
            self._host_ob_t = self._host_out[0:1]
            self._host_done_t = self._host_out[2:3].view(torch.uint8)[0:1]
            hp = self._host_out.data_ptr()
            self._host_ptrs = (hp, hp + 4, hp + 8)
            self._host_np = self._host_out.numpy()                       # same memory, cheap scalar reads
            self._host_reward_np = self._host_reward.numpy()
            self._action_base = self._action_table.data_ptr()
        self._has_reset = False
        self._last_reset = None
        self._scalar_done = False
        self._dev_flags_used = False    # a C-side driver wrote the device-side ob / done of a scalar env (see reset())
        self._collect_cache = {}        # collect_synthetic: bound argument structs by (buffer pointers, steps)
        self._tracker = None          # per-step side effects beyond (state, ob, reward, done): RockSample's side statistics
        self._done_bool = self._done.view(torch.bool)
        self._ptrs = (self._state.data_ptr(), self._ob.data_ptr(), self._reward.data_ptr(), self._done.data_ptr(),
                      self._err.data_ptr())
        # everything of a step() call that does not change between calls, bound once (include/pomdp_hip.h: pomdp_step_args)
        self._step_args = _native.StepArgs(
            env=_native.ENV_KIND[self.env_name], flags=_native.POMDP_AUTO_RESET if self._auto_reset else 0,
            params=C.addressof(self._params), state=self._ptrs[0], ob=self._ptrs[1], reward=self._ptrs[2],
            done=self._ptrs[3], err=self._ptrs[4], n=n, seed=self._seed, lane0=self.lane_offset, reserved=0)
        self._step_args_ref = C.byref(self._step_args)
        self._bound_step = self._lib.pomdp_step
        if n == 1:                       # scalar mode: the same call with ob / reward / done in pinned host memory
            self._scalar_args = _native.StepArgs(
                env=self._step_args.env, flags=self._step_args.flags, params=C.addressof(self._params), state=self._ptrs[0],
                ob=self._host_ptrs[0], reward=self._host_ptrs[1], done=self._host_ptrs[2], err=self._ptrs[4], n=1,
                seed=self._seed, lane0=self.lane_offset, reserved=0)
            self._scalar_args_ref = C.byref(self._scalar_args)
            self._bound_step_sync = self._lib.pomdp_step_sync
        self._action_shape = torch.Size((n,))
        self._dev_index = self.device.index
        self._info = {"state": self._state}

    @property
    def unwrapped(self):
        """gym.Env.unwrapped: the env under every wrapper — this object."""
        return self

    @property
    def np_random(self):
        """gym.Env.np_random (gym >= 0.22 keeps one per env).  The kernels never draw from it — their words are Philox
        streams of (seed, lane, t) — it exists for callers and wrappers that expect the attribute."""
        rng = self.__dict__.get("_np_random")
        if rng is None:
            rng = self.__dict__["_np_random"] = np.random.RandomState(getattr(self, "_seed", 0) & 0xFFFFFFFF)
        return rng

    @np_random.setter
    def np_random(self, value):
        self.__dict__["_np_random"] = value

    @property
    def auto_reset(self):
        return self._auto_reset

    @auto_reset.setter
    def auto_reset(self, value):
        if self._auto_reset and not value and self.batch_size > 1:
            # without auto-reset `done` is in/out (a set flag freezes the lane); what the buffer holds now are the flags the
            # last auto-resetting step RETURNED, and those lanes have fresh episodes
            self._done.zero_()
        self._auto_reset = bool(value)
        self._step_args.flags = _native.POMDP_AUTO_RESET if self._auto_reset else 0
        if self.batch_size == 1:
            self._scalar_args.flags = self._step_args.flags

    # ---- gym.Env surface --------------------------------------------------------
    def seed(self, seed=None):
        """Reference: np.random.seed(seed) on the global stream (rock.py:120-121).  Here: the
        Philox key of this env's lanes; the call counter restarts."""
        self._seed = _random_seed() if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self._step_args.seed = self._seed
        if self.batch_size == 1:
            self._scalar_args.seed = self._seed
        self._t = 0
        self.__dict__.pop("_np_random", None)
        return [self._seed]

    @property
    def call_counter(self):
        """t of the next reset()/step() call (the Philox counter word pair)."""
        return self._t

    @call_counter.setter
    def call_counter(self, t):
        self._t = int(t)

    def _stream(self):
        """hipStream_t of torch's current stream on this env's device (the raw handle: no Stream object is built)."""
        return _raw_stream(self._dev_index)

    def reset(self, *, seed=None, options=None, return_info=False):
        """All lanes start a new episode.  Returns ob: int32[N] tensor (python int if batch_size == 1).
        The reference's reset() takes no arguments; the keywords are what gym 0.22-0.25's `gym.make` wrappers pass through
        (`seed`: seed(seed) first; `return_info`: (ob, info) instead of ob; `options` is accepted and ignored)."""
        if seed is not None:
            self.seed(seed)
        if return_info:
            return self.reset(), self._info
        t = self._t
        self._t += 1
        if self.batch_size == 1 and not self._dev_flags_used and _current_device() == self._dev_index:
            # scalar mode: the observation goes straight to pinned host memory; launch and wait in one FFI call, read.
            # Only while nothing has written the DEVICE-side ob / done of this env: heuristic_steps() and
            # rollout_synthetic() take the device done flag as an in/out freeze flag, so after them the general path
            # below runs once (it clears that flag and refreshes the device-side observation)
            rc = self._lib.pomdp_reset_sync(self._step_args.env, self._params_ref, self._ptrs[0], self._host_ptrs[0], 1, self._seed,
                                            self.lane_offset, t, _raw_stream(self._dev_index))
            _native.check(rc, "pomdp_%s_reset" % self.env_name)
            self._host_out[2] = 0
            if self._tracker is not None:
                self._tracker.on_reset()
            ob = int(self._host_np[0])
            self._has_reset, self._scalar_done, self.done = True, False, False
            self._last_reset = (ob, t)
            return ob
        with torch.cuda.device(self.device):
            ob = self._ob if self.reuse_buffers else torch.empty_like(self._ob)
            rc = self._reset_fn(self._params_ref, self._state.data_ptr(), ob.data_ptr(), self.batch_size,
                                self._seed, self.lane_offset, t, self._stream())
            _native.check(rc, "pomdp_%s_reset" % self.env_name)
            self._done.zero_()
            if self.batch_size == 1:
                self._host_out[2] = 0            # the done flag the scalar step keeps in pinned host memory
            if self._tracker is not None:
                self._tracker.on_reset()
        self._has_reset = True
        self._scalar_done = False
        self._dev_flags_used = False
        self._last_reset = (ob, t)      # History() picks the current observation up from here
        self.done = False if self.batch_size == 1 else self._done.view(torch.bool)
        if self.batch_size == 1:
            return int(ob.item())
        return ob

    def step(self, action):
        """action: int32[N] tensor on this env's device (other integer tensors / arrays / lists are
        converted); python int when batch_size == 1.  Returns (ob, reward, done, info) like the
        reference, as tensors (python scalars when batch_size == 1); info["state"] aliases the live
        packed state, as the reference's info["state"] aliases live internals."""
        if not self._has_reset:
            # the reference raises AttributeError here: `self.done` does not exist before reset()
            raise AttributeError("%s: step() called before reset()" % type(self).__name__)
        if self.batch_size == 1:
            assert self.action_space.contains(action), "invalid action %r" % (action,)
            assert self._scalar_done is False or self._auto_reset, "step() on a done env (call reset())"
            return self._scalar_step(int(action))
        if not (type(action) is torch.Tensor and action.dtype is torch.int32 and action.shape == self._action_shape
                and action.device == self.device and action.is_contiguous()):
            action = self._as_action_tensor(action)           # slow path: convert / validate
        t = self._t
        self._t = t + 1
        if self.reuse_buffers and _current_device() == self._dev_index:
            # hot path: one FFI call with four arguments (the rest is bound in self._step_args), no allocation, no
            # device-context switch, the raw stream handle
            rc = self._bound_step(self._step_args_ref, action.data_ptr(), t, _raw_stream(self._dev_index))
            if rc:
                _native.check(rc, "pomdp_%s_step" % self.env_name)
            if self._tracker is not None:
                self._tracker.on_step(action, self._ob, self._done, self._step_args.flags)
            self.done = self._done_bool
            return self._ob, self._reward, self._done_bool, self._info
        flags = _native.POMDP_AUTO_RESET if self._auto_reset else 0
        with torch.cuda.device(self.device):
            if self.reuse_buffers:
                ob, reward = self._ob, self._reward
            else:
                ob, reward = torch.empty_like(self._ob), torch.empty_like(self._reward)
            done = self._done if (self.reuse_buffers or not self._auto_reset) else torch.empty_like(self._done)
            rc = self._step_fn(self._params_ref, self._state.data_ptr(), action.data_ptr(), ob.data_ptr(),
                               reward.data_ptr(), done.data_ptr(), self._err.data_ptr(), self.batch_size,
                               self._seed, self.lane_offset, t, flags, self._stream())
            _native.check(rc, "pomdp_%s_step" % self.env_name)
        if not self._auto_reset and not self.reuse_buffers:
            done = done.clone()
        if self._tracker is not None:
            self._tracker.on_step(action, ob, done, flags)
        self.done = self._done_bool if done is self._done else done.view(torch.bool)
        return ob, reward, self.done, self._info

    def _scalar_step(self, action):
        """batch_size == 1: one launch writing into pinned host memory and one stream synchronisation — both inside ONE
        FFI call (pomdp_step_sync) — python scalars out."""
        t = self._t
        self._t = t + 1
        if self._tracker is None and _current_device() == self._dev_index:
            rc = self._bound_step_sync(self._scalar_args_ref, self._action_base + 4 * action, t,
                                       _raw_stream(self._dev_index))
            if rc:
                _native.check(rc, "pomdp_%s_step" % self.env_name)
        else:
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device)
                rc = self._bound_step(self._scalar_args_ref, self._action_base + 4 * action, t, stream.cuda_stream)
                if rc:
                    _native.check(rc, "pomdp_%s_step" % self.env_name)
                if self._tracker is not None:
                    self._tracker.on_step(self._action_table[action:action + 1], self._host_ob_t, self._host_done_t,
                                          self._scalar_args.flags)
                stream.synchronize()
        h = self._host_np
        d = bool(h[2] & 0xFF)
        self._scalar_done = d
        self.done = d
        r = self._host_reward_np[0]
        return int(h[0]), (int(r) if self.reward_dtype == torch.int32 else float(r)), d, self._info

    def _as_action_tensor(self, action):
        if isinstance(action, torch.Tensor):
            if action.dtype.is_floating_point or action.dtype == torch.bool:
                raise AssertionError("actions must be integers")
            if action.dtype != torch.int32:
                # a wider value must not wrap into a valid action (2^32 + 1 -> 1): anything outside the action space
                # becomes -1, which the kernels count as invalid and ignore (the reference asserts, rock.py:125)
                action = torch.where((action < 0) | (action >= self.action_space.n), torch.full_like(action, -1), action)
            a = action.to(device=self.device, dtype=torch.int32)
        else:
            arr = np.asarray(action)
            if arr.dtype.kind not in "iu":
                raise AssertionError("actions must be integers")
            if arr.dtype != np.int32:
                arr = np.where((arr < 0) | (arr >= self.action_space.n), -1, arr)
            a = torch.as_tensor(arr.astype(np.int32), device=self.device)
        if a.shape != (self.batch_size,):
            raise AssertionError("actions must have shape (%d,), got %s" % (self.batch_size, tuple(a.shape)))
        return a.contiguous()

    def render(self, mode="ansi", close=False):
        if close:
            return
        raise NotImplementedError("rendering is out of scope for the batched envs (SURVEY.md §2: GUI)")

    def close(self):
        return

    # ---- batch-side extras --------------------------------------------------------
    @property
    def state(self):
        """Packed lane state, int32 [state_words, N] (layout: include/pomdp_hip.h)."""
        return self._state

    def set_state(self, state):
        """Overwrite the packed lane state (planner hook `_set_state`; tensor copy).  `state` is int32
        [state_words, N] (or [N] / a scalar-mode [state_words] for one-word layouts): anything else is rejected, so a
        state saved under another layout (BattleShip had 2 * MW words before ABI 10) cannot be misread."""
        self._state.copy_(self._checked_state(state, self.batch_size, "set_state", validate=True))
        self._done.zero_()
        if self.batch_size == 1:
            self._host_out[2] = 0
        self._scalar_done = False
        self._has_reset = True
        self.done = False if self.batch_size == 1 else self._done.view(torch.bool)
        if self._tracker is not None:
            self._tracker.on_reset()

    _set_state = set_state

    def _checked_state(self, state, lanes, what, validate=False):
        """`state` as an int32 [state_words, lanes or -1] device tensor, shape-checked (lanes=None: any lane count);
        validate: also the env's own sanity check of a state that is going to be stepped."""
        st = torch.as_tensor(state, dtype=torch.int32, device=self.device)
        if st.dim() == 1 and (self.state_words == 1 or (lanes == 1 and st.numel() == self.state_words)):
            st = st.reshape(self.state_words, -1)
        if st.dim() != 2 or st.shape[0] != self.state_words or (lanes is not None and st.shape[1] != lanes):
            raise ValueError("%s: expected a packed state of shape (%d, %s), got %s" % (
                what, self.state_words, "N" if lanes is None else lanes, tuple(st.shape)))
        if validate:
            self._validate_state(st, what)
        return st.contiguous()

    def _validate_state(self, st, what):
        """env-specific sanity of a caller-supplied packed state (BattleShip: the cached next board)"""
        return

    def invalid_action_count(self):
        """Lanes that were handed an out-of-range action since construction (they were left
        untouched; the reference would have raised AssertionError).  Synchronises."""
        return int(self._err.item())

    def synthetic_actions(self, out=None, seed=None):
        """Uniform random actions from the bench's synthetic policy (stream ACTION of the *current* call
        counter); int32[N] on device.  lane_offset must be a multiple of 4."""
        if self.lane_offset % 4:
            raise ValueError("synthetic_actions: lane_offset must be a multiple of 4 (got %d)" % self.lane_offset)
        if out is None:
            out = torch.empty(self.batch_size, dtype=torch.int32, device=self.device)
        if torch.cuda.current_device() == self.device.index:
            rc = self._lib.pomdp_synthetic_actions(out.data_ptr(), self.batch_size,
                                                   self._seed if seed is None else seed, self.lane_offset,
                                                   self._t, self.action_space.n, self._stream())
            if rc:
                _native.check(rc, "pomdp_synthetic_actions")
            return out
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_synthetic_actions(out.data_ptr(), self.batch_size,
                                                   self._seed if seed is None else seed, self.lane_offset,
                                                   self._t, self.action_space.n, self._stream())
            _native.check(rc, "pomdp_synthetic_actions")
        return out

    # ---- planner hooks (reference: _get_init_state / _set_state / _generate_legal / _discount) --------
    def _get_init_state(self):
        """A fresh batch of initial states (packed int32 [state_words, N]) without touching the live
        state — the reference's `_get_init_state()` (rock.py:266-271 etc.).  Advances the call counter."""
        t = self._t
        self._t += 1
        out = torch.empty_like(self._state)
        with torch.cuda.device(self.device):
            rc = self._reset_fn(self._params_ref, out.data_ptr(), None, self.batch_size, self._seed,
                                self.lane_offset, t, self._stream())
            _native.check(rc, "pomdp_%s_reset" % self.env_name)
        return out

    def legal_actions(self, state=None):
        """`_generate_legal()` of every lane: (list int32 [N, n_actions] in the reference's order, padded
        with -1; length int32 [N]).  `state` defaults to the live state."""
        st = self._state if state is None else self._checked_state(state, None, "legal_actions")
        n = st.shape[1]
        stride = self.action_space.n
        lst = torch.empty((n, stride), dtype=torch.int32, device=self.device)
        ln = torch.empty(n, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_legal_actions(_native.ENV_KIND[self.env_name], self._params_ref, st.data_ptr(),
                                               lst.data_ptr(), ln.data_ptr(), n, stride, self._stream())
            _native.check(rc, "pomdp_legal_actions")
        return lst, ln

    def _generate_legal(self):
        """batch_size == 1: the reference's python list; otherwise the (list, length) tensors."""
        lst, ln = self.legal_actions()
        if self.batch_size == 1:
            return lst[0, : int(ln.item())].tolist()
        return lst, ln

    def preferred_actions(self, history, state=None):
        """`_generate_preferred(history)` of every lane, in the shape of legal_actions(): (list int32 [N, n_actions]
        padded with -1, length int32 [N]).  `history` is a gym_pomdp_amd.history.History of this env."""
        st = self._state if state is None else self._checked_state(state, None, "preferred_actions")
        n = st.shape[1]
        if n != self.batch_size:
            raise ValueError("preferred_actions: the history and side statistics cover exactly batch_size lanes")
        stride = self.action_space.n
        lst = torch.empty((n, stride), dtype=torch.int32, device=self.device)
        ln = torch.empty(n, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_preferred_actions(_native.ENV_KIND[self.env_name], self._params_ref, st.data_ptr(),
                                                   self._belief_ref(), history._ref, lst.data_ptr(), ln.data_ptr(), n,
                                                   stride, self._stream())
            _native.check(rc, "pomdp_preferred_actions")
        return lst, ln

    def _belief_ref(self):
        return None

    def _generate_preferred(self, history):
        """Reference signature (tag.py:231-243; tiger.py:114-115, network.py:138-139: the legal list).
        batch_size == 1: a python list; otherwise the (list, length) tensors."""
        lst, ln = self.preferred_actions(history)
        if self.batch_size == 1:
            return lst[0, : int(ln.item())].tolist()
        return lst, ln

    def pick_actions(self, lists, lengths, seed=None, out=None):
        """The caller's `np.random.choice(list)` for every lane (rock.py:564): element (w * length) >> 32 of each
        lane's list, w being the synthetic policy's word at the *current* call counter.  -> int32[N]."""
        if out is None:
            out = torch.empty(self.batch_size, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_pick_actions(lists.data_ptr(), lengths.data_ptr(), lists.shape[1], out.data_ptr(),
                                              self.batch_size, self._seed if seed is None else seed, self.lane_offset,
                                              self._t, self._stream())
            _native.check(rc, "pomdp_pick_actions")
        return out

    def heuristic_steps(self, history, steps=1, returns=None):
        """`steps` consecutive steps under the env's own heuristic policy, one launch each
        (pomdp_heuristic_steps): a = choice(_generate_preferred(history)); step(a); side statistics;
        history.append(Transition(observation, a, reward, ob, done)) — the loop of rock.py:557-573 for every lane,
        with the results of that call sequence.  (The reference's own loop builds `Transition(ob, action, next_ob, rw,
        done)` POSITIONALLY into the fields (observation, action, reward, next_observation, done), rock.py:566 — its
        `next_observation` therefore holds the reward and `_generate_preferred`'s history sums never fire there.  This
        path, like the fixtures' harness, fills the fields by name, i.e. the intended order; DESIGN.md §3 lists it among
        the deliberate divergences.)  Returns the last step's (action, ob, reward, done) in reusable
        buffers; `history.prev_ob` holds the observation each lane sees afterwards.  `returns`
        (gym_pomdp_amd.Returns) accumulates the loop's discounted return per lane.  Asynchronous."""
        if not self._has_reset:
            raise AttributeError("%s: heuristic_steps before reset()" % type(self).__name__)
        if history.prev_ob is None:
            raise ValueError("History was built without the current observation: History(env, observation=ob)")
        if self.lane_offset % 4:
            raise ValueError("heuristic_steps: lane_offset must be a multiple of 4 (got %d): the policy's Philox block is shared "
                             "by global lanes 4q .. 4q+3" % self.lane_offset)
        if getattr(self, "_action_scratch", None) is None:
            self._action_scratch = torch.empty(self.batch_size, dtype=torch.int32, device=self.device)
        t0 = self._t
        self._t += int(steps)
        self._dev_flags_used = True
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_heuristic_steps(
                _native.ENV_KIND[self.env_name], self._params_ref, self._state.data_ptr(),
                self._belief_ref() if self.env_name == "rock" else None, history._ref, history.prev_ob.data_ptr(),
                self._action_scratch.data_ptr(), self._ob.data_ptr(), self._reward.data_ptr(), self._done.data_ptr(),
                None if returns is None else returns._ref, self.batch_size, self._seed, self.lane_offset, t0, int(steps),
                _native.POMDP_AUTO_RESET if self.auto_reset else 0, self._stream())
            _native.check(rc, "pomdp_heuristic_steps")
        return self._action_scratch, self._ob, self._reward, self._done.view(torch.bool)

    def compute_prob(self, action, ob, state=None):
        """`_compute_prob(action, next_state, ob)` per lane -> float64[N]: the likelihood of `ob` given that
        `action` led to `state` (default: the live state)."""
        st = self._state if state is None else self._checked_state(state, None, "compute_prob")
        n = st.shape[1]
        a = torch.as_tensor(action, device=self.device).to(torch.int32).reshape(n).contiguous()
        o = torch.as_tensor(ob, device=self.device).to(torch.int32).reshape(n).contiguous()
        out = torch.empty(n, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_compute_prob(_native.ENV_KIND[self.env_name], self._params_ref, st.data_ptr(),
                                              a.data_ptr(), o.data_ptr(), out.data_ptr(), n, self._stream())
            _native.check(rc, "pomdp_compute_prob")
        return out

    def _compute_prob(self, action, next_state, ob):
        """Reference signature.  batch_size == 1: python scalars in, float out."""
        if self.batch_size == 1 and not isinstance(action, torch.Tensor):
            return float(self.compute_prob([int(action)], [int(ob)], next_state).item())
        return self.compute_prob(action, ob, next_state)

    def rollout(self, depth, sims_per_root=1, roots=None, discount=None, all_actions=False, lane_offset=None, out=None):
        """Random rollouts from `roots` (packed states int32 [state_words, R]; default: the live state):
        R * sims_per_root independent simulations of at most `depth` steps under a uniform policy over
        `_generate_legal()` (or over all actions), discounted by `discount` (default: the env's
        `_discount`).  Neither `roots` nor the live state is modified; the call counter advances by
        `depth`.  Returns a dict of per-simulation tensors: ret float64, n_steps, first_action, last_ob
        int32, terminated bool — simulation i belongs to root i // sims_per_root.  `out`: the dict of an earlier
        call of the same shape, to reuse its buffers."""
        st = self._state if roots is None else self._checked_state(roots, None, "rollout")
        n_roots = st.shape[1]
        n = n_roots * int(sims_per_root)
        if (self.lane_offset if lane_offset is None else int(lane_offset)) % 4:
            raise ValueError("rollout: the simulations' first global lane must be a multiple of 4 (quad-shared Philox blocks)")
        t0 = self._t
        self._t += int(depth)
        if out is None:
            out = dict(ret=torch.empty(n, dtype=torch.float64, device=self.device),
                       n_steps=torch.empty(n, dtype=torch.int32, device=self.device),
                       first_action=torch.empty(n, dtype=torch.int32, device=self.device),
                       last_ob=torch.empty(n, dtype=torch.int32, device=self.device),
                       terminated=torch.empty(n, dtype=torch.uint8, device=self.device))
        else:                              # a dict returned by an earlier call of the same shape: buffers are reused
            out["terminated"] = out["terminated"].view(torch.uint8)
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_rollout(
                _native.ENV_KIND[self.env_name], self._params_ref, st.data_ptr(), n_roots, int(sims_per_root),
                int(depth), float(self._discount if discount is None else discount),
                _native.POMDP_ROLLOUT_ALL_ACTIONS if all_actions else 0, self._seed,
                self.lane_offset if lane_offset is None else int(lane_offset), t0, out["ret"].data_ptr(),
                out["n_steps"].data_ptr(), out["first_action"].data_ptr(), out["last_ob"].data_ptr(),
                out["terminated"].data_ptr(), self._stream())
            _native.check(rc, "pomdp_rollout")
        out["terminated"] = out["terminated"].view(torch.bool)
        return out

    def plan(self, depth, sims_per_root=1024, discount=None, all_actions=False, roots=None, out=None):
        """One planning pass of a POMCP-style caller over the live state (or `roots`): `sims_per_root` random rollouts of
        at most `depth` steps from every root (rollout()), reduced ON THE DEVICE to the roots' action values
        (pomdp_plan): {"q": float64 [R, n_actions] mean return by first action (0 where never tried), "visits": int32
        [R, n_actions], "best": int32 [R] the visited action with the largest q (lowest index on ties, -1 if no
        simulation took a step), "value": float64 [R] = q[best], "sim_ret" / "sim_first_action": the R * sims_per_root
        simulations' returns and first actions}.  The float64 sums follow the order stated in include/pomdp_hip.h, so the
        CPU restatement reproduces them bit for bit.  Root r is global lane lane_offset + r and its simulation s is
        global lane (lane_offset + r) * sims_per_root + s: the draws — and with them q / visits / best — do not depend on
        how the roots are sharded, and whole roots never straddle a shard.  The call counter advances by `depth`.
        `out`: the dict of an earlier call of the same shape, to reuse its buffers.  Asynchronous."""
        st = self._state if roots is None else self._checked_state(roots, None, "plan")
        n_roots, sims, depth = st.shape[1], int(sims_per_root), int(depth)
        n, n_act = n_roots * sims, self.action_space.n
        sim_lane0 = self.lane_offset * sims
        if sims < 1 or sim_lane0 + n > 1 << 32:
            raise ValueError("plan: the simulations' global lanes (lane_offset + r) * sims_per_root + s must lie in [0, 2^32)")
        if sim_lane0 % 4:
            raise ValueError("plan: lane_offset * sims_per_root must be a multiple of 4 (quad-shared Philox blocks)")
        if out is None:
            out = dict(q=torch.zeros((n_roots, n_act), dtype=torch.float64, device=self.device),
                       visits=torch.zeros((n_roots, n_act), dtype=torch.int32, device=self.device),
                       best=torch.empty(n_roots, dtype=torch.int32, device=self.device),
                       value=torch.empty(n_roots, dtype=torch.float64, device=self.device),
                       sim_ret=torch.empty(n, dtype=torch.float64, device=self.device),
                       sim_first_action=torch.empty(n, dtype=torch.int32, device=self.device))
        elif out["q"].shape != (n_roots, n_act) or out["sim_ret"].shape != (n,):
            raise ValueError("plan: `out` was built for another shape")
        po = out.get("_plan_out")
        if po is None:
            po = out["_plan_out"] = _native.PlanOut(q=out["q"].data_ptr(), visits=out["visits"].data_ptr(), best=out["best"].data_ptr(),
                                                    value=out["value"].data_ptr(), stride=n_act, reserved=0)
        t0 = self._t
        self._t += depth
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_plan(
                _native.ENV_KIND[self.env_name], self._params_ref, st.data_ptr(), n_roots, sims, depth,
                float(self._discount if discount is None else discount), _native.POMDP_ROLLOUT_ALL_ACTIONS if all_actions else 0,
                self._seed, sim_lane0, t0, out["sim_ret"].data_ptr(), out["sim_first_action"].data_ptr(), C.byref(po), self._stream())
            _native.check(rc, "pomdp_plan")
        return out

    def plan_step(self, depth, sims_per_root=1024, discount=None, all_actions=False, out=None):
        """One REAL step of every lane, planned: plan() from the live state, then step(best) — BASELINE.json configs[4]'s "1024-
        simulation rollout per real step".  Returns (ob, reward, done, info, plan dict).  A lane whose simulations took no
        step (best == -1: nothing legal to do, or depth == 0) is handed -1, which step() counts as an invalid action and
        leaves untouched.  batch_size == 1: python scalars as step() returns them."""
        p = self.plan(depth, sims_per_root, discount, all_actions, out=out)
        if self.batch_size == 1:
            a = int(p["best"].item())
            assert a >= 0, "plan_step: no simulation took a step"
            return self.step(a) + (p,)
        return self.step(p["best"]) + (p,)

    def rollout_synthetic(self, steps, action_seed=None, actions=None, fuse=False):
        """`steps` consecutive step() calls under the synthetic uniform policy, issued from C
        (pomdp_rollout_synthetic): the same launches per step a python loop over synthetic_actions() + step()
        makes, without the interpreter between them.  `fuse=True` (policy key == env key only): up to pomdp_fuse_max()
        (256) consecutive steps share one launch — every step's outputs are still computed and written, so all
        buffers end up exactly as after the per-step launches, but a lane's state stays in registers between
        its steps.  Outputs land in the reusable buffers; returns (ob, reward, done) of the last step.
        Asynchronous."""
        if not self._has_reset:
            raise AttributeError("%s: rollout before reset()" % type(self).__name__)
        self._check_driver_use("rollout_synthetic")
        if actions is None:
            if getattr(self, "_action_scratch", None) is None:
                self._action_scratch = torch.empty(self.batch_size, dtype=torch.int32, device=self.device)
            actions = self._action_scratch
        t0 = self._t
        self._t += int(steps)
        self._dev_flags_used = True
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_rollout_synthetic(
                _native.ENV_KIND[self.env_name], self._params_ref, self._state.data_ptr(), actions.data_ptr(),
                self._ob.data_ptr(), self._reward.data_ptr(), self._done.data_ptr(), self._err.data_ptr(),
                self.batch_size, self._seed, self._seed if action_seed is None else action_seed, self.lane_offset,
                t0, int(steps), (_native.POMDP_AUTO_RESET if self.auto_reset else 0) |
                (_native.POMDP_FUSE_STEPS if fuse else 0), self._stream())
            _native.check(rc, "pomdp_rollout_synthetic")
        return self._ob, self._reward, self._done.view(torch.bool)

    def trajectory_buffers(self, steps, layout="columns"):
        """The `out` dict of collect_synthetic(steps, layout=...).
        "columns" (the default ABI): action int32 [steps + 1, N], ob int32 [steps, N], reward [steps, N], done bool
        [steps, N] (and its uint8 view "done_u8"), carved from one allocation with staggered column starts.
        "blocked" / "packed" / "narrow" (include/pomdp_hip.h: POMDP_LAYOUT_*; one write stream per step): {"layout", "traj":
        the raw rows — uint8 [steps, pitch * 13] / int32 [steps, pitch] / uint8 [steps, 4, pitch] —, "pitch"};
        decode_trajectory() gives the four columns (narrow: its typed planes, in place)."""
        n = self.batch_size
        if layout == "columns":
            a, o, r, d = staggered([((steps + 1, n), torch.int32), ((steps, n), torch.int32), ((steps, n), self._reward.dtype),
                                    ((steps, n), torch.uint8)], self.device)
            return {"action": a, "ob": o, "reward": r, "done_u8": d, "done": d.view(torch.bool)}       # no "layout" key = columns
        if layout == "blocked":
            pitch = -(-n // 256) * 256
            return {"layout": "blocked", "pitch": pitch, "traj": torch.zeros((steps, pitch * 13), dtype=torch.uint8, device=self.device)}
        if layout == "packed":
            pitch = -(-n // 4) * 4
            return {"layout": "packed", "pitch": pitch, "traj": torch.zeros((steps, pitch), dtype=torch.int32, device=self.device)}
        if layout == "narrow":
            pitch = -(-n // 16) * 16
            return {"layout": "narrow", "pitch": pitch, "traj": torch.zeros((steps, 4, pitch), dtype=torch.uint8, device=self.device)}
        raise ValueError("unknown trajectory layout %r (columns, blocked, packed, narrow)" % (layout,))

    def packed_reward_table(self):
        """reward_code byte of a packed record -> the reward the columns hold (pomdp_packed_reward), as a [256] tensor of
        this env's reward dtype."""
        t = getattr(self, "_packed_reward_table", None)
        if t is None:
            kind = _native.ENV_KIND[self.env_name]
            vals = [self._lib.pomdp_packed_reward(kind, c) for c in range(256)]
            t = self._packed_reward_table = torch.tensor(vals, dtype=torch.float64).to(self._reward.dtype).to(self.device)
        return t

    def decode_trajectory(self, out, steps=None, into=None):
        """The four columns of a collected trajectory whatever its layout: {"action" (the action taken at each step), "ob",
        "reward", "done"}, each [steps, N].
        columns: the buffers themselves.  blocked: int32 / reward-dtype / bool COPIES gathered from the 256-lane blocks.
        packed: int32 / reward-dtype / bool columns written by ONE device pass over the records (pomdp_decode_packed: 4 bytes
        read, 13 written per lane-step) — into `into` (a trajectory_buffers(steps) dict; rows 0 .. steps - 1 of its
        "action") when given.  narrow: the planes themselves, no copy — action / ob uint8, done bool, reward int8 (the
        reward itself; Network: its values through packed_reward_table(), a gather)."""
        layout = out.get("layout", "columns")
        n = self.batch_size
        if layout == "columns":
            k = out["ob"].shape[0] if steps is None else steps
            return {"action": out["action"][:k], "ob": out["ob"][:k], "reward": out["reward"][:k], "done": out["done"][:k]}
        traj = out["traj"] if steps is None else out["traj"][:steps]
        k = traj.shape[0]
        if layout == "blocked":
            b = traj.view(k, out["pitch"] // 256, 13 * 256)
            col = lambda lo, dt: b[:, :, lo:lo + 1024].view(dt).reshape(k, -1)[:, :n]                     # noqa: E731
            return {"action": col(0, torch.int32), "ob": col(1024, torch.int32), "reward": col(2048, self._reward.dtype),
                    "done": b[:, :, 3072:].reshape(k, -1)[:, :n].view(torch.bool)}
        if layout == "narrow":
            rw = traj[:, 2, :n]
            rw = self.packed_reward_table()[rw.long()] if self.env_name == "network" else rw.view(torch.int8)
            return {"action": traj[:, 0, :n], "ob": traj[:, 1, :n], "reward": rw, "done": traj[:, 3, :n].view(torch.bool)}
        if into is None:
            into = self.trajectory_buffers(k)
        if into["ob"].shape[0] < k or into["ob"].shape[1] != n:
            raise ValueError("decode_trajectory: `into` does not have the shapes of trajectory_buffers(%d)" % k)
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_decode_packed(_native.ENV_KIND[self.env_name], traj.data_ptr(), n, k, int(out["pitch"]),
                                               into["action"].data_ptr(), into["ob"].data_ptr(), into["reward"].data_ptr(),
                                               into["done_u8"].data_ptr(), n, self._stream())
            _native.check(rc, "pomdp_decode_packed")
        return {"action": into["action"][:k], "ob": into["ob"][:k], "reward": into["reward"][:k], "done": into["done"][:k]}

    def collect_returns(self, steps, stats=None, discount=None):
        """`steps` consecutive step() calls under the synthetic uniform policy with NOTHING kept per step but the reduction
        the reference's callers apply to the stream (pomdp_collect_returns): per lane `r += discount * rw; discount *=
        _discount`, the return of every finished episode banked (network.py:175-191, rock.py:553-575).  Same policy, same
        draws and same final state as collect_synthetic(steps).  `stats`: a gym_pomdp_amd.EpisodeStats to continue
        (default: a fresh one with the env's `_discount`).  Returns it.  Asynchronous."""
        if not self._has_reset:
            raise AttributeError("%s: collect before reset()" % type(self).__name__)
        if not self.auto_reset:
            raise ValueError("collect_returns needs auto_reset=True")
        self._check_driver_use("collect_returns")
        if stats is None:
            from ..history import EpisodeStats
            stats = EpisodeStats(self, discount)
        elif stats._n != self.batch_size or stats.acc.device != self.device:
            raise ValueError("collect_returns: `stats` belongs to another batch")
        steps = int(steps)
        t0 = self._t
        self._t = t0 + steps
        with torch.cuda.device(self.device):
            rc = self._lib.pomdp_collect_returns(_native.ENV_KIND[self.env_name], self._params_ref, self._ptrs[0], stats._ref,
                                                 self._ptrs[4], self.batch_size, self._seed, self.lane_offset, t0, steps,
                                                 _native.POMDP_AUTO_RESET, self._stream())
            _native.check(rc, "pomdp_collect_returns")
        return stats

    def as_tape(self, actions):
        """`actions` ([steps, N], any integer dtype, any device) as the uint8 tape the fused launches read: a uint8 tensor on
        this env's device whose rows are contiguous is used in place (e.g. the action plane `traj["traj"][:, 0]` of a narrow
        trajectory); anything else is converted, an action that does not fit a byte becoming 255 (out of range for every env,
        so it stays the invalid action it was)."""
        if (isinstance(actions, torch.Tensor) and actions.dtype == torch.uint8 and actions.device == self.device and actions.dim() == 2
                and actions.shape[1] == self.batch_size and actions.stride(1) == 1
                and (actions.shape[0] <= 1 or actions.stride(0) >= actions.shape[1])):
            return actions                                    # already a tape: the common case costs one test
        t = torch.as_tensor(actions, device=self.device) if not isinstance(actions, torch.Tensor) else actions.to(self.device)
        if t.dim() != 2 or t.shape[1] != self.batch_size:
            raise ValueError("tape: expected actions of shape (steps, %d), got %s" % (self.batch_size, tuple(t.shape)))
        if t.dtype.is_floating_point or t.dtype == torch.bool:
            raise AssertionError("actions must be integers")
        if t.dtype != torch.uint8:
            t = torch.where((t < 0) | (t > 255), torch.full_like(t, 255), t).to(torch.uint8)
        if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
            t = t.contiguous()
        return t

    def collect_tape(self, actions, out=None, layout=None, stats=None):
        """`steps = len(actions)` consecutive step() calls with the CALLER's actions (pomdp_collect_tape*): row s of `actions`
        ([steps, N]; see as_tape) is what step() is handed at call counter t + s.  The fused launches of collect_synthetic /
        collect_returns — a lane's state in registers between its steps, up to 256 steps per launch — on actions the caller
        chose: a replayed trajectory, a table policy, another model's output.  Same results as a python loop over step(),
        row for row, including the final state.  layout "columns" (default): {"ob", "reward", "done"} [steps, N] (no action
        column: the caller holds it); "blocked" / "packed" / "narrow": as collect_synthetic; "returns" (or `stats` given): the
        episode statistics of collect_returns, returned as the EpisodeStats.  An action outside the env's range leaves the
        lane untouched for that step and is counted (invalid_action_count()).  auto_reset envs only.  Asynchronous."""
        if not self._has_reset:
            raise AttributeError("%s: collect before reset()" % type(self).__name__)
        if not self.auto_reset:
            raise ValueError("collect_tape needs auto_reset=True")
        self._check_driver_use("collect_tape")
        tape = self.as_tape(actions)
        steps, n = tape.shape[0], self.batch_size
        if stats is not None and layout is None:
            layout = "returns"
        if layout is None:
            layout = "columns" if out is None else out.get("layout", "columns")
        ct = _native.Tape(actions=tape.data_ptr(), stride=tape.stride(0) if steps > 1 else max(tape.stride(0), n))
        kind, t0 = _native.ENV_KIND[self.env_name], self._t
        self._t = t0 + steps
        with torch.cuda.device(self.device):
            if layout == "returns":
                if stats is None:
                    from ..history import EpisodeStats
                    stats = EpisodeStats(self)
                elif stats._n != n or stats.acc.device != self.device:
                    raise ValueError("collect_tape: `stats` belongs to another batch")
                rc = self._lib.pomdp_collect_tape_returns(kind, self._params_ref, self._ptrs[0], C.byref(ct), stats._ref, self._ptrs[4], n,
                                                          self._seed, self.lane_offset, t0, steps, _native.POMDP_AUTO_RESET, self._stream())
                _native.check(rc, "pomdp_collect_tape_returns")
                return stats
            if out is None:
                out = self.trajectory_buffers(steps, layout)
            if out.get("layout", "columns") != layout:
                raise ValueError("collect_tape: `out` was built for layout %r, not %r" % (out.get("layout", "columns"), layout))
            if layout == "columns":
                if not (out["ob"].shape[0] >= steps and out["ob"].shape[1] == n and out["reward"].shape == out["ob"].shape
                        and out["done_u8"].shape == out["ob"].shape):
                    raise ValueError("collect_tape: `out` does not have the shapes of trajectory_buffers(%d)" % steps)
                rc = self._lib.pomdp_collect_tape(kind, self._params_ref, self._ptrs[0], C.byref(ct), out["ob"].data_ptr(),
                                                  out["reward"].data_ptr(), out["done_u8"].data_ptr(), self._ptrs[4], n, self._seed,
                                                  self.lane_offset, t0, steps, n, _native.POMDP_AUTO_RESET, self._stream())
                _native.check(rc, "pomdp_collect_tape")
                return out
            traj, pitch = out["traj"], int(out["pitch"])
            row = {"blocked": (pitch * 13,), "packed": (pitch,), "narrow": (4, pitch)}[layout]
            if not (traj.shape[0] >= steps and tuple(traj.shape[1:]) == row and traj.is_contiguous() and pitch >= n):
                raise ValueError("collect_tape: `out` does not have the shape of trajectory_buffers(%d, %r)" % (steps, layout))
            rc = self._lib.pomdp_collect_tape_layout(kind, self._params_ref, self._ptrs[0], C.byref(ct), traj.data_ptr(), self._ptrs[4], n,
                                                     self._seed, self.lane_offset, t0, steps, pitch, _native.LAYOUTS[layout],
                                                     _native.POMDP_AUTO_RESET, self._stream())
            _native.check(rc, "pomdp_collect_tape_layout")
        return out

    def _check_driver_use(self, what):
        """The C-side episode loops advance the packed state only: they know nothing of RockSample's side statistics
        (use_heuristic / track_belief envs), and the synthetic policy shares one Philox block among global lanes
        4q .. 4q+3, so a shard has to start on such a boundary."""
        if self._tracker is not None:
            raise ValueError("%s: this env maintains RockSample's side statistics (use_heuristic / track_belief), which "
                             "the C-side drivers do not update — use step() or heuristic_steps()" % what)
        if self.lane_offset % 4:
            raise ValueError("%s: lane_offset must be a multiple of 4 (got %d): the synthetic policy's Philox block is "
                             "shared by global lanes 4q .. 4q+3" % (what, self.lane_offset))

    def collect_synthetic(self, steps, out=None, layout=None):
        """`steps` consecutive step() calls under the synthetic uniform policy with every step's results KEPT
        (pomdp_collect_synthetic): returns {"action": int32 [steps + 1, N] (row s = the actions of step s, last row = the
        next call's), "ob": int32 [steps, N], "reward": [steps, N], "done": bool [steps, N]} — row s equals what
        synthetic_actions() + step() return at that call.  The batched form of the reference callers' episode
        loops (rock.py:553-575); up to pomdp_fuse_steps(env, layout) steps per launch (256; 64 for the 13-byte layouts of RockSample /
        Tag / Tiger), auto_reset envs only.  `out`: a dict from an earlier call
        to write into.  `layout` (default: `out`'s, else "columns"): "blocked" / "packed" / "narrow" write the same
        information as one stream per step (pomdp_collect_layout; trajectory_buffers, decode_trajectory).  Asynchronous."""
        if not self._has_reset:
            raise AttributeError("%s: collect before reset()" % type(self).__name__)
        if not self.auto_reset:
            raise ValueError("collect_synthetic needs auto_reset=True")
        self._check_driver_use("collect_synthetic")
        steps, n = int(steps), self.batch_size
        if layout is None:
            layout = "columns" if out is None else out.get("layout", "columns")
        if out is None:
            out = self.trajectory_buffers(steps, layout)
        if out.get("layout", "columns") != layout:
            raise ValueError("collect_synthetic: `out` was built for layout %r, not %r" % (out.get("layout", "columns"), layout))
        if layout != "columns":
            return self._collect_traj(steps, out, layout)
        if not (out["action"].shape == (steps + 1, n) and out["ob"].shape == (steps, n) and out["reward"].shape == (steps, n)
                and out["done_u8"].shape == (steps, n)):
            raise ValueError("collect_synthetic: `out` does not have the shapes of trajectory_buffers(%d)" % steps)
        # the call's per-buffer arguments (include/pomdp_hip.h: pomdp_collect_args), bound once per set of buffers: the
        # cache lives on the env and is keyed by the buffers' addresses, so a dict handed to another env, or one whose
        # tensors were replaced, never meets a stale struct
        key = (out["action"].data_ptr(), out["ob"].data_ptr(), out["reward"].data_ptr(), out["done_u8"].data_ptr(), steps)
        bound = self._collect_cache.get(key)
        if bound is None:
            if len(self._collect_cache) >= 64:
                self._collect_cache.clear()
            a = _native.CollectArgs(env=_native.ENV_KIND[self.env_name], flags=_native.POMDP_AUTO_RESET,
                                    params=C.addressof(self._params), state=self._ptrs[0], action=key[0],
                                    ob=key[1], reward=key[2], done=key[3],
                                    err=self._ptrs[4], n=n, pitch=n, seed=self._seed,
                                    lane0=self.lane_offset, reserved=0)
            bound = self._collect_cache[key] = (a, C.byref(a), steps)
        bound[0].seed = self._seed
        t0 = self._t
        self._t = t0 + steps
        if _current_device() == self._dev_index:         # no device-context switch on the common path
            rc = self._lib.pomdp_collect(bound[1], t0, steps, _raw_stream(self._dev_index))
        else:
            with torch.cuda.device(self.device):
                rc = self._lib.pomdp_collect(bound[1], t0, steps, self._stream())
        if rc:
            _native.check(rc, "pomdp_collect_synthetic")
        return out

    def _collect_traj(self, steps, out, layout):
        """collect_synthetic into a blocked / packed / narrow trajectory: pomdp_collect_traj with its arguments bound per buffer."""
        n, traj, pitch = self.batch_size, out["traj"], int(out["pitch"])
        row = {"blocked": (pitch * 13,), "packed": (pitch,), "narrow": (4, pitch)}[layout]
        if not (traj.shape[0] >= steps and tuple(traj.shape[1:]) == row and traj.is_contiguous() and pitch >= n
                and traj.dtype == (torch.int32 if layout == "packed" else torch.uint8)):
            raise ValueError("collect_synthetic: `out` does not have the shape of trajectory_buffers(%d, %r)" % (steps, layout))
        key = (traj.data_ptr(), layout, pitch)
        bound = self._collect_cache.get(key)
        if bound is None:
            if len(self._collect_cache) >= 64:
                self._collect_cache.clear()
            a = _native.TrajArgs(env=_native.ENV_KIND[self.env_name], flags=_native.POMDP_AUTO_RESET, layout=_native.LAYOUTS[layout],
                                 reserved=0, params=C.addressof(self._params), state=self._ptrs[0], traj=key[0], err=self._ptrs[4],
                                 n=n, pitch=pitch, seed=self._seed, lane0=self.lane_offset, reserved2=0)
            bound = self._collect_cache[key] = (a, C.byref(a), steps)
        bound[0].seed = self._seed
        t0 = self._t
        self._t = t0 + steps
        if _current_device() == self._dev_index:
            rc = self._lib.pomdp_collect_traj(bound[1], t0, steps, _raw_stream(self._dev_index))
        else:
            with torch.cuda.device(self.device):
                rc = self._lib.pomdp_collect_traj(bound[1], t0, steps, self._stream())
        if rc:
            _native.check(rc, "pomdp_collect_layout")
        return out

    def __repr__(self):
        return "%s(batch_size=%d, device=%s)" % (type(self).__name__, self.batch_size, getattr(self, "device", "?"))

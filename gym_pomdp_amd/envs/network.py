"""Network — batched mirror of gym_pomdp/envs/network.py:24-168 (`NetworkEnv`)."""
import torch

from .. import _native, tables
from .base import BatchedEnv


def make_params(n_machines=10, problem_type=3):
    if not 1 <= n_machines <= 32:
        raise ValueError("Network: up to 32 machines fit the packed state word")
    nb = tables.network_neighbours(n_machines, problem_type)
    p = _native.NetworkParams()
    p.n_machines = n_machines
    deg = 0
    for i, lst in enumerate(nb):
        m = 0
        for j in lst:
            m |= 1 << j
        p.nb_mask[i] = m
        if len(lst) > 2:
            deg |= 1 << i
    p.deg_gt2_mask = deg
    p.fail_thr = tables.NET_FAIL_THR
    p.fail_nb_thr = tables.NET_FAIL_NEIGHBOUR_THR
    p.obs_thr = tables.NET_OBS_THR
    return p, 1, 2 * n_machines + 1, 3


class NetworkEnv(BatchedEnv):
    """Action 2m pings machine m, 2m+1 reboots it, 2M is a no-op (network.py:101-112); observations
    0 off, 1 on, 2 null; reward float32 = float32(#up (+1 for hubs) - .1 | 2.5).  Never terminates
    (network.py:114).  `depth` is accepted and ignored, as in the reference (network.py:31)."""
    env_name = "network"
    reward_dtype = torch.float32

    def __init__(self, n_machines=10, problem_type=3, depth=60, **batch_kwargs):
        self._n_machines = n_machines
        self.problem_type = problem_type
        self._depth = 60
        self._discount = .95                  # network.py:35
        self._reward_range = n_machines * 2   # network.py:36
        self.neighbours = tables.network_neighbours(n_machines, problem_type)
        self._setup(**batch_kwargs)

    def _build_params(self):
        return make_params(self._n_machines, self.problem_type)

    def _validate_state(self, st, what):
        """bit i = machine i: a bit at or above n_machines names no machine (the kernels index per-machine-set tables with
        the word)"""
        if self._n_machines < 32 and bool(((st[0].to(torch.int64) & 0xFFFFFFFF) >> self._n_machines).any()):
            raise ValueError("%s: Network state words may only have bits 0 .. %d set" % (what, self._n_machines - 1))

    def decode_state(self):
        w = self._state[0].to(torch.int64) & 0xFFFFFFFF
        return torch.stack([(w >> i) & 1 for i in range(self._n_machines)], dim=1)

    def step(self, action):
        if self.batch_size == 1:
            self.last_action = int(action)           # network.py:80, used by render
        return super().step(action)

    def sample_action(self):
        """network.py:141-142 `np.random.choice(self._generate_legal())` — every action is legal (network.py:129-130), so this is
        the synthetic policy's draw at the current call counter: element `(w * n_actions) >> 32` of the list, stream ACTION
        (include/pomdp_hip.h: pomdp_synthetic_actions).  int32[N]; a python int when batch_size == 1."""
        a = self.synthetic_actions()
        return int(a.item()) if self.batch_size == 1 else a

    def reset(self, **kwargs):
        self.last_action = self._n_machines * 2       # network.py:65
        self._server = 0                              # network.py:68
        return super().reset(**kwargs)

    def render(self, mode="ansi", close=False, lane=0):
        """network.py:116-120: prints `N: <machines up>, S: <server>\t<action>` for one lane (the reference's only
        text renderer)."""
        if close:
            return
        up = int(self.decode_state()[lane].sum().item())
        a = getattr(self, "last_action", self._n_machines * 2)
        act = "M: {} A: {}".format(*divmod(a, 2)) if a < self._n_machines * 2 else "Null"   # network.py:16-21
        print("N: {}, S: {}".format(up, self._server), act, sep="\t")

"""Tiger — batched mirror of gym_pomdp/envs/tiger.py:47-172 (`TigerEnv`)."""
import torch

from .. import _native, tables
from ..spaces import Discrete
from .base import BatchedEnv


def make_params():
    p = _native.TigerParams()
    p.listen_thr = tables.TIGER_LISTEN_THR
    return p, 1, 3, 3


class TigerEnv(BatchedEnv):
    """Actions 0 open-left, 1 open-right, 2 listen; observations 0 left, 1 right, 2 null
    (tiger.py:10-24).  Opening the tiger's door: -20, done, and the *state* is returned as the
    observation (tiger.py:81-83); opening the other door: +10, not done, tiger re-placed; listen: -1,
    correct w.p. .85.  `correct_prob` is stored but unused, as in the reference (tiger.py:86, 141).
    The hidden state comes from the gym-space RNG in the reference (tiger.py:64, 119), which
    np.random.seed does not control: here both generators read the lane's word of the quad's STEP block of the
    call counter (include/pomdp_hip.h, ABI 13: a call makes at most one draw that matters)."""
    env_name = "tiger"
    reward_dtype = torch.int32

    def __init__(self, seed=0, correct_prob=.85, **batch_kwargs):
        self.correct_prob = correct_prob
        self.state_space = Discrete(2)     # tiger.py:53
        self._discount = .95               # tiger.py:55
        self._reward_range = 10            # tiger.py:56
        batch_kwargs.setdefault("seed", seed)   # TigerEnv() seeds in its ctor (tiger.py:58)
        self._setup(**batch_kwargs)

    def _build_params(self):
        return make_params()

    def decode_state(self):
        return (self._state[0].to(torch.int64) & 1).unsqueeze(1)

    def step(self, action):
        if self.batch_size == 1:
            self.last_action = int(action)           # tiger.py:76, used by render
        return super().step(action)

    def render(self, mode="ansi", close=False, lane=0):
        """tiger.py:90-101, the text mode (the reference's own line indexes the integer action and raises; the
        message is kept, the action printed as it is)."""
        if close:
            return
        if mode != "ansi":
            raise NotImplementedError("only the text renderer exists here (SURVEY.md §2: GUI out of scope)")
        print("Current step: {}, tiger is in state: {}, action took: {}".format(
            self.call_counter, int(self.decode_state()[lane, 0].item()), getattr(self, "last_action", None)))

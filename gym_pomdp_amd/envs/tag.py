"""Tag — batched mirror of gym_pomdp/envs/tag.py:84-291 (`TagEnv`)."""
import torch

from .. import _native, tables
from .base import BatchedEnv


def make_params(num_opponents=1, move_prob=.8, obs_cells=29, board_size=(10, 5)):
    if tuple(board_size) != (10, 5):
        # TagGrid hard-codes the 29-cell T-shaped board (tag.py:43-66) whatever board_size says
        raise ValueError("Tag: only the reference's (10, 5) board is defined")
    if not 1 <= num_opponents <= 4:
        raise ValueError("Tag: 1..4 opponents fit the packed state word")
    p = _native.TagParams()
    p.num_opponents, p.obs_cells = num_opponents, obs_cells
    if move_prob == .8:
        p.move_thr, p.move_gt = tables.TAG_MOVE_THR, 0
    else:
        if not 0. < move_prob < 1.:
            raise ValueError("Tag: move_prob must lie in (0, 1)")
        thr, sense = tables.bernoulli_threshold(move_prob)
        p.move_thr, p.move_gt = thr, int(sense == "gt")     # numpy's binomial(1, p): [U > thr] for p <= .5 (tag.py:204)
    return p, 1, 5, obs_cells + 1


class TagEnv(BatchedEnv):
    """Actions 0 N, 1 E, 2 S, 3 W, 4 TAG (tag.py:28-33); observation = agent cell 0..28, or
    obs_cells (29) when a move leaves the agent on an opponent's cell (tag.py:219-226); reward
    float32 in {-1, -10, +10}.  The opponent only ever moves on a failed TAG (tag.py:119-131)."""
    env_name = "tag"
    reward_dtype = torch.float32

    def __init__(self, num_opponents=1, move_prob=.8, obs_cells=29, board_size=(10, 5), **batch_kwargs):
        self.num_opponents = num_opponents
        self.move_prob = move_prob
        self.obs_cells = obs_cells
        self.board_size = tuple(board_size)
        self._reward_range = 10 * num_opponents   # tag.py:90
        self._discount = .95                      # tag.py:91
        self._setup(**batch_kwargs)

    def _build_params(self):
        return make_params(self.num_opponents, self.move_prob, self.obs_cells, self.board_size)

    def _encode_state(self, state=None):
        """The reference's `_encode_state` (tag.py:158-165): int32 [N, 1 + num_opponents] = [agent cell, opponent cells..]."""
        d = self.decode_state() if state is None else state
        return d[:, : 1 + self.num_opponents].to(torch.int32)

    def _decode_state(self, state):
        """The reference's `_decode_state` (tag.py:167-179) for a batch: encoded int [N, 1 + num_opponents] ->
        packed int32 [1, N] as set_state() takes it; num_opp counts the opponent entries > -1, like the reference."""
        s = torch.as_tensor(state, device=self.device).to(torch.int64).reshape(-1, 1 + self.num_opponents)
        w = s[:, 0] & 31
        num = torch.zeros_like(w)
        for j in range(self.num_opponents):
            opp = s[:, 1 + j]
            num = num + (opp > -1).to(torch.int64)
            w = w | ((opp.clamp(min=0) & 31) << (5 + 5 * j))
        w = w | ((num & 0x7F) << 25)
        return torch.where(w >= 1 << 31, w - (1 << 32), w).to(torch.int32).reshape(1, -1)

    def decode_state(self):
        """int64 [N, 2 + num_opponents] = [agent cell, opponent cells.., num_opp] (tag.py:158-165 order)."""
        w = self._state[0].to(torch.int64) & 0xFFFFFFFF
        no = (w >> 25) & 0x7F
        no = torch.where(no >= 64, no - 128, no)
        cols = [w & 31] + [(w >> (5 + 5 * j)) & 31 for j in range(self.num_opponents)] + [no]
        return torch.stack(cols, dim=1)

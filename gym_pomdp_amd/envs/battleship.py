"""BattleShip — batched mirror of gym_pomdp/envs/battleship.py:64-211 (`BattleShipEnv`)."""
import torch

from .. import _native
from .base import BatchedEnv


def make_params(board_size=(5, 5), max_len=3):
    x, y = board_size
    cells = x * y
    if not (1 <= x <= 16 and 1 <= y <= 16 and cells <= 122):
        raise ValueError("BattleShip: boards up to 122 cells fit the packed masks")
    if not 2 <= max_len <= 10:
        raise ValueError("BattleShip: max_len must be in 2..10")
    if max(x, y) < max_len + 2:
        # collision() wants length + 2 cells in a line (battleship.py:199-201); the reference's reset() would loop forever
        raise ValueError("BattleShip: a %dx%d board cannot hold a ship of length %d (needs %d cells in a line)"
                         % (x, y, max_len, max_len + 2))
    p = _native.BattleShipParams()
    p.x_size, p.y_size, p.max_len = x, y, max_len
    col0 = sum(1 << (yy * x) for yy in range(y))
    for j in range(4):
        p.col0[j] = (col0 >> (32 * j)) & 0xFFFFFFFF
    for k in range(12):
        v = sum(1 << (i * x) for i in range(k)) & ((1 << 128) - 1)
        for j in range(4):
            p.vpat[k][j] = (v >> (32 * j)) & 0xFFFFFFFF
    mask_words = (cells + 6 + 31) // 32
    return p, 3 * mask_words, cells, 2     # occupied, visited (+ remaining), the next episode's occupied mask


class BattleShipEnv(BatchedEnv):
    """Action a shoots cell (a % X, a // X) (coord.py:64-66); observation 1 on a first hit else 0;
    reward -10 for a revisit, -1 for a new cell, + X*Y when the last ship cell is hit
    (battleship.py:91-122).  Ships of length max_len..2 are placed by rejection sampling at reset
    (battleship.py:167-211).  A lane also carries the board of its NEXT episode (state words 2*MW .. 3*MW-1; the board
    contract of include/pomdp_hip.h): an auto-reset moves it in and draws the one after it."""
    env_name = "battleship"
    reward_dtype = torch.int32

    def __init__(self, board_size=(5, 5), max_len=3, **batch_kwargs):
        self.board_size = tuple(board_size)
        self._max_len = max_len                  # the ctor argument: ships of length max_len .. 2
        # the reference's attributes of the same names hold other values (battleship.py:74-75: `self.max_len = max_len + 1`,
        # the exclusive end of `range(2, self.max_len)`, and `self.total_remaining = max_len - 1`, which reset() overwrites)
        self.max_len = max_len + 1
        self.total_remaining = max_len - 1
        self.num_obs = 2
        self._reward_range = (self.board_size[0] * self.board_size[1]) / 4.   # battleship.py:72
        self._discount = 1.                                                   # battleship.py:73
        self._setup(**batch_kwargs)

    def _build_params(self):
        return make_params(self.board_size, self._max_len)

    def _validate_state(self, st, what):
        """A caller-supplied state must carry a real next board: with an empty one the lane's next auto-reset deals an episode
        whose `remaining` starts at 0 and can never reach done again (every shot a miss)."""
        mw = self.state_words // 3
        ships = sum(range(2, self._max_len + 1))
        nxt = st[2 * mw:].to(torch.int64) & 0xFFFFFFFF
        pop = torch.zeros(st.shape[1], dtype=torch.int64, device=st.device)
        for b in range(32):
            pop += ((nxt >> b) & 1).sum(dim=0)
        if bool((pop != ships).any()):
            raise ValueError("%s: state words %d..%d (the board of the lane's NEXT episode) must hold %d ship cells per lane; "
                             "take them from reset() / _get_init_state()" % (what, 2 * mw, 3 * mw - 1, ships))

    def decode_state(self):
        """int64 [N, 1 + 2*cells] = [total_remaining, occupied_0.., visited_0..], cell a = y*X + x."""
        mw = self.state_words // 3
        cells = self.board_size[0] * self.board_size[1]
        s = self._state.to(torch.int64) & 0xFFFFFFFF
        a = torch.arange(cells, device=self.device)
        occ = (s[a // 32] >> (a % 32).unsqueeze(1)) & 1
        vis = (s[mw + a // 32] >> (a % 32).unsqueeze(1)) & 1
        rem = (s[2 * mw - 1] >> 26).unsqueeze(0)
        return torch.cat([rem, occ, vis], dim=0).t().contiguous()

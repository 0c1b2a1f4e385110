"""`gym.spaces.Discrete` (the only space the reference uses: rock.py:113-114, tag.py:92-94, battleship.py:69-70,
tiger.py:52-54, network.py:33-34): gym's own class when an old-API `gym` is importable — gym's wrappers and env checker
test `isinstance(env.action_space, gym.spaces.Space)` — and the duck type below otherwise."""
import numpy as np

from . import compat

np_random = np.random.RandomState()


def seed(s=None):
    np_random.seed(s)


class _Discrete(object):
    def __init__(self, n):
        assert n >= 0
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(np_random.randint(self.n))

    def contains(self, x):
        """gym's rule: python ints and 0-d integer numpy scalars/arrays only."""
        if isinstance(x, int):
            as_int = x
        elif isinstance(x, (np.generic, np.ndarray)) and x.dtype.kind in "iu" and x.shape == ():
            as_int = int(x)
        else:
            return False
        return 0 <= as_int < self.n

    def __contains__(self, x):
        return self.contains(x)

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return isinstance(other, _Discrete) and self.n == other.n


Discrete = compat.GymDiscrete if compat.GymDiscrete is not None else _Discrete

"""gym.spaces stand-in: Discrete only.

Old gym drew `Discrete.sample()` from a gym-owned RandomState
(`gym.spaces.prng.np_random`), which `np.random.seed` does not control.  The stub
keeps that property: `np_random` below is a module-level legacy RandomState that
the harness seeds / state-injects separately from the global `np.random`.
"""
import numpy as np

np_random = np.random.RandomState(0)


def seed(s=None):
    np_random.seed(s)


class Space(object):
    """base class gym's wrappers and env checker test spaces against"""
    shape = None
    dtype = None


class Discrete(Space):
    def __init__(self, n):
        self.n = n
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(np_random.randint(self.n))

    def contains(self, x):
        # gym's rule: python int, or integer-kind numpy generic / 0-d array
        if isinstance(x, int):
            as_int = x
        elif isinstance(x, (np.generic, np.ndarray)) and (
                x.dtype.kind in np.typecodes["AllInteger"] and x.shape == ()):
            as_int = int(x)
        else:
            return False
        return 0 <= as_int < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n

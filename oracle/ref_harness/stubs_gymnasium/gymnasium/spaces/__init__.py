import numpy as np


class Space(object):
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = shape
        self.dtype = dtype
        self._np_random = None if seed is None else np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.default_rng()
        return self._np_random

    def __contains__(self, x):
        return self.contains(x)


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        assert n > 0
        self.n = int(n)
        self.start = int(start)
        super(Discrete, self).__init__((), np.int64, seed)

    def sample(self):
        return int(self.start + self.np_random.integers(self.n))

    def contains(self, x):
        if isinstance(x, int):
            as_int = x
        elif isinstance(x, (np.generic, np.ndarray)) and np.issubdtype(x.dtype, np.integer) and x.shape == ():
            as_int = int(x)
        else:
            return False
        return self.start <= as_int < self.start + self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n and self.start == other.start

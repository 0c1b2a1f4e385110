import numpy as np


class Env(object):
    """gymnasium.Env: reset(seed=, options=) -> (ob, info); step(a) -> (ob, reward, terminated, truncated, info)."""
    metadata = {"render_modes": []}
    render_mode = None
    spec = None
    action_space = None
    observation_space = None
    _np_random = None

    def step(self, action):
        raise NotImplementedError

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.default_rng(seed)
        return None, {}

    def render(self):
        raise NotImplementedError

    def close(self):
        return

    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.default_rng()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def __str__(self):
        if self.spec is None:
            return "<{} instance>".format(type(self).__name__)
        return "<{}<{}>>".format(type(self).__name__, self.spec.id)


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError("accessing private attribute '{}' is prohibited".format(name))
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def metadata(self):
        return self.env.metadata

    @property
    def render_mode(self):
        return self.env.render_mode

    def step(self, action):
        return self.env.step(action)

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def render(self):
        return self.env.render()

    def close(self):
        return self.env.close()

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __str__(self):
        return "<{}{}>".format(type(self).__name__, self.env)

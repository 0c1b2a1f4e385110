class Env(object):
    """Old-gym (0.21) base class: reset() -> ob, step(a) -> (ob, reward, done, info)."""
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        return

    def seed(self, seed=None):
        return []

    @property
    def unwrapped(self):
        return self

    def __str__(self):
        if self.spec is None:
            return "<{} instance>".format(type(self).__name__)
        return "<{}<{}>>".format(type(self).__name__, self.spec.id)

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()
        return False


class Wrapper(Env):
    """gym.core.Wrapper of 0.21: forwards step / reset / seed / render / close and every other attribute."""

    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None
        self._reward_range = None
        self._metadata = None

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def action_space(self):
        return self.env.action_space if self._action_space is None else self._action_space

    @property
    def observation_space(self):
        return self.env.observation_space if self._observation_space is None else self._observation_space

    @property
    def reward_range(self):
        return self.env.reward_range if self._reward_range is None else self._reward_range

    @property
    def metadata(self):
        return self.env.metadata if self._metadata is None else self._metadata

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def render(self, mode="human", **kwargs):
        return self.env.render(mode, **kwargs)

    def close(self):
        return self.env.close()

    def seed(self, seed=None):
        return self.env.seed(seed)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __str__(self):
        return "<{}{}>".format(type(self).__name__, self.env)

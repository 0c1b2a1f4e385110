import gym.core


class OrderEnforcing(gym.core.Wrapper):
    """gym 0.21 wraps every env `gym.make` builds in this (EnvSpec.order_enforce defaults to True)."""

    def __init__(self, env):
        super(OrderEnforcing, self).__init__(env)
        self._has_reset = False

    def step(self, action):
        assert self._has_reset, "Cannot call env.step() before calling reset()"
        return self.env.step(action)

    def reset(self, **kwargs):
        self._has_reset = True
        return self.env.reset(**kwargs)

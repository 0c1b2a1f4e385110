from gymnasium import error
from gymnasium.core import Env, Wrapper
from gymnasium.spaces import Space


class PassiveEnvChecker(Wrapper):
    """gymnasium.wrappers.PassiveEnvChecker, the checks that RAISE (the real one also logs warnings about value types)."""

    def __init__(self, env):
        if not isinstance(env, Env):
            raise AssertionError("The environment must inherit from the gymnasium.Env class")
        super(PassiveEnvChecker, self).__init__(env)
        if not hasattr(env, "action_space"):
            raise AttributeError("The environment must specify an action space.")
        if not isinstance(env.action_space, Space):
            raise AssertionError("action space does not inherit from `gymnasium.spaces.Space`, actual type: %s" % type(env.action_space))
        if not hasattr(env, "observation_space"):
            raise AttributeError("The environment must specify an observation space.")
        if not isinstance(env.observation_space, Space):
            raise AssertionError("observation space does not inherit from `gymnasium.spaces.Space`, actual type: %s" % type(env.observation_space))
        self.checked_reset = False
        self.checked_step = False

    def reset(self, *, seed=None, options=None):
        result = self.env.reset(seed=seed, options=options)
        if not self.checked_reset:
            self.checked_reset = True
            if not isinstance(result, tuple):
                raise AssertionError("The result returned by `env.reset()` was not a tuple of the form `(obs, info)`, actual type: %s" % type(result))
            if len(result) != 2:
                raise AssertionError("Calling the reset method did not return a 2-tuple")
            if not isinstance(result[1], dict):
                raise AssertionError("The second element returned by `env.reset()` was not a dictionary, actual type: %s" % type(result[1]))
        return result

    def step(self, action):
        result = self.env.step(action)
        if not self.checked_step:
            self.checked_step = True
            if not isinstance(result, tuple):
                raise AssertionError("Expects step result to be a tuple, actual type: %s" % type(result))
            if len(result) != 5:
                raise error.Error("Expected `Env.step` to return a five element tuple, actual number of elements returned: %d." % len(result))
            if not isinstance(result[4], dict):
                raise AssertionError("The `info` returned by `step()` must be a python dictionary, actual type: %s" % type(result[4]))
        return result


class OrderEnforcing(Wrapper):
    def __init__(self, env):
        super(OrderEnforcing, self).__init__(env)
        self._has_reset = False

    def step(self, action):
        if not self._has_reset:
            raise error.ResetNeeded("Cannot call env.step() before calling env.reset()")
        return self.env.step(action)

    def reset(self, *, seed=None, options=None):
        self._has_reset = True
        return self.env.reset(seed=seed, options=options)

"""gymnasium.envs.registration, reduced to what `register` / `make` do to the object an entry point builds."""
import copy
import importlib

import gymnasium
from gymnasium import error

registry = {}


def load_env_creator(name):
    mod_name, attr_name = name.split(":")
    return getattr(importlib.import_module(mod_name), attr_name)


class EnvSpec(object):
    def __init__(self, id, entry_point=None, reward_threshold=None, nondeterministic=False, max_episode_steps=None,
                 order_enforce=True, disable_env_checker=False, kwargs=None, **more):
        self.id = id
        self.entry_point = entry_point
        self.reward_threshold = reward_threshold
        self.nondeterministic = nondeterministic
        self.max_episode_steps = max_episode_steps
        self.order_enforce = order_enforce
        self.disable_env_checker = disable_env_checker
        self.kwargs = {} if kwargs is None else dict(kwargs)

    def __repr__(self):
        return "EnvSpec({})".format(self.id)


def register(id, entry_point=None, **kwargs):
    # (the real one logs "Overriding environment ... already in registry" and overrides)
    registry[id] = EnvSpec(id, entry_point=entry_point, **kwargs)


def spec(id):
    try:
        return registry[id]
    except KeyError:
        raise error.NameNotFound("Environment `{}` doesn't exist.".format(id))


def make(id, max_episode_steps=None, disable_env_checker=None, **kwargs):
    env_spec = spec(id)
    env_spec_kwargs = copy.deepcopy(env_spec.kwargs)
    env_spec_kwargs.update(kwargs)
    if env_spec.entry_point is None:
        raise error.Error("{} registered but entry_point is not specified".format(env_spec.id))
    env_creator = env_spec.entry_point if callable(env_spec.entry_point) else load_env_creator(env_spec.entry_point)
    env = env_creator(**env_spec_kwargs)
    if not isinstance(env, gymnasium.Env):
        if str(env.__class__.__base__) == "<class 'gym.core.Env'>" or str(env.__class__.__base__) == "<class 'gym.core.Wrapper'>":
            raise TypeError("Gym is incompatible with Gymnasium, please update the environment class to `gymnasium.Env`.")
        raise TypeError("The environment must inherit from the gymnasium.Env class, actual class: {}.".format(type(env)))
    made = copy.deepcopy(env_spec)
    made.kwargs = env_spec_kwargs
    env.unwrapped.spec = made
    if disable_env_checker is False or (disable_env_checker is None and env_spec.disable_env_checker is False):
        env = gymnasium.wrappers.PassiveEnvChecker(env)
    if env_spec.order_enforce:
        env = gymnasium.wrappers.OrderEnforcing(env)
    return env

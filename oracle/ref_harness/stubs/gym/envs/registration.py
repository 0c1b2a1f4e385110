"""gym.envs.registration of gym 0.21, reduced to what `register` / `make` / `spec` do to an env class."""
import copy
import importlib

from gym import error


def load(name):
    mod_name, attr_name = name.split(":")
    return getattr(importlib.import_module(mod_name), attr_name)


class EnvSpec(object):
    def __init__(self, id, entry_point=None, reward_threshold=None, nondeterministic=False, max_episode_steps=None,
                 order_enforce=True, kwargs=None):
        self.id = id
        self.entry_point = entry_point
        self.reward_threshold = reward_threshold
        self.nondeterministic = nondeterministic
        self.max_episode_steps = max_episode_steps
        self.order_enforce = order_enforce
        self._kwargs = {} if kwargs is None else kwargs

    def make(self, **kwargs):
        """Instantiates the env with the registered kwargs overridden by the caller's (gym 0.21 EnvSpec.make)."""
        if self.entry_point is None:
            raise error.Error("Attempting to make deprecated env {}".format(self.id))
        _kwargs = self._kwargs.copy()
        _kwargs.update(kwargs)
        if callable(self.entry_point):
            env = self.entry_point(**_kwargs)
        else:
            cls = load(self.entry_point)
            env = cls(**_kwargs)
        spec = copy.deepcopy(self)
        spec._kwargs = _kwargs
        env.unwrapped.spec = spec
        if self.order_enforce:
            from gym.wrappers.order_enforcing import OrderEnforcing
            env = OrderEnforcing(env)
        return env

    def __repr__(self):
        return "EnvSpec({})".format(self.id)


class EnvRegistry(object):
    def __init__(self):
        self.env_specs = {}

    def make(self, path, **kwargs):
        return self.spec(path).make(**kwargs)

    def all(self):
        return self.env_specs.values()

    def spec(self, path):
        try:
            return self.env_specs[path]
        except KeyError:
            raise error.UnregisteredEnv("No registered env with id: {}".format(path))

    def register(self, id, **kwargs):
        if id in self.env_specs:
            raise error.Error("Cannot re-register id: {}".format(id))
        self.env_specs[id] = EnvSpec(id, **kwargs)


registry = EnvRegistry()


def register(id, **kwargs):
    return registry.register(id, **kwargs)


def make(id, **kwargs):
    return registry.make(id, **kwargs)


def spec(id):
    return registry.spec(id)

"""Which gym is there — the drop-in boundary of gym_pomdp/__init__.py:7-46 and readme.md:18-24 (`gym.make("Tag-v0")`).

The reference is written against OLD gym (`reset() -> ob`, `step(a) -> (ob, reward, done, info)`, `seed(s)`); its envs
subclass `gym.Env` and are reached through `gym.make(id)`.  What `gym.make` does with the object it builds depends on the
gym at hand, and every variant needs something of the env:

  gym <= 0.21   `EnvSpec.make`: `env = cls(**kwargs); env.unwrapped.spec = spec`, then `OrderEnforcing` / `TimeLimit`
                wrappers that forward `reset()` / `step()` and reach everything else through `__getattr__`;
  gym 0.22-0.25 the same plus `PassiveEnvChecker`, which wants `isinstance(env.action_space, gym.spaces.Space)` and calls
                `env.reset(seed=..., options=..., return_info=...)` with whatever the caller passed;
  gym 0.26      new-API checks unless the id is registered with `apply_api_compatibility=True` (then `EnvCompatibility`
                turns `seed()` + `reset()` / the 4-tuple into the new calls);
  gymnasium     `make` raises TypeError unless `isinstance(env, gymnasium.Env)` and speaks the NEW api only:
                `reset(seed=, options=) -> (ob, info)`, `step(a) -> (ob, reward, terminated, truncated, info)`.

So: the batched env classes subclass `gym.Env` when an old-API `gym` is importable (`EnvBase`), use gym's own `Discrete`
then, and always carry `unwrapped`, `spec`, `np_random`, `render_mode` and both spellings of `metadata`'s render modes;
under gymnasium the ids resolve to `GymnasiumEnv`, a thin 5-tuple adapter around the same batched env (`truncated` is
always False: the reference sets no time limit, gym_pomdp/__init__.py:9 — `max_episode_steps` is commented out).
"""
import importlib
import inspect


def _try_import(name):
    try:
        return importlib.import_module(name)
    except ImportError:
        return None


gym = _try_import("gym")
gymnasium = _try_import("gymnasium")

# the reference's base class (old API) when there is one to subclass
EnvBase = getattr(gym, "Env", object) if gym is not None else object


def _gym_discrete():
    if gym is None:
        return None
    sp = _try_import("gym.spaces")
    return getattr(sp, "Discrete", None)


GymDiscrete = _gym_discrete()


def gymnasium_discrete(n):
    sp = _try_import("gymnasium.spaces")
    return sp.Discrete(int(n))


def _truncated_like(done):
    """`truncated` of the new api for a `done` of any of the env's shapes: python bool (batch_size == 1) or a bool tensor"""
    if isinstance(done, bool):
        return False
    import torch
    return torch.zeros_like(done)


if gymnasium is not None:
    class GymnasiumEnv(gymnasium.Env):
        """The batched env behind gymnasium's api.  `env` is the old-api object (everything it has beyond reset / step —
        `collect_synthetic`, `rollout`, `plan`, `state`, ... — is reached through attribute forwarding)."""
        metadata = {"render_modes": ["ansi"]}

        def __init__(self, env):
            self.env = env
            self.action_space = gymnasium_discrete(env.action_space.n)
            self.observation_space = gymnasium_discrete(env.observation_space.n)
            self.reward_range = env.reward_range
            self.render_mode = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self.env.seed(seed)
            ob = self.env.reset()
            return ob, {"state": self.env.state}

        def step(self, action):
            ob, reward, done, info = self.env.step(action)
            return ob, reward, done, _truncated_like(done), info

        def render(self):
            return self.env.render()

        def close(self):
            return self.env.close()

        def __getattr__(self, name):
            if name.startswith("__") or name == "env":
                raise AttributeError(name)
            return getattr(self.env, name)

        def __repr__(self):
            return "GymnasiumEnv(%r)" % (self.env,)
else:
    GymnasiumEnv = None


def gymnasium_entry_point(module, attr):
    """A callable entry point for gymnasium's registry: builds the batched env, wraps it in the adapter."""
    def make(**kwargs):
        cls = getattr(importlib.import_module(module), attr)
        return GymnasiumEnv(cls(**kwargs))
    make.__name__ = "gymnasium_" + attr
    make.__qualname__ = make.__name__
    return make


def _registered_ids(reg):
    """ids a gym / gymnasium registry already holds, across the shapes the registry has had (a dict of specs; an
    EnvRegistry with .env_specs)"""
    r = getattr(reg, "registry", None)
    if r is None:
        return set()
    specs = getattr(r, "env_specs", r)
    try:
        return set(specs.keys())
    except AttributeError:
        return set()


def register_ids(registry):
    """Register the reference's ids (gym_pomdp/__init__.py:7-41) with every gym at hand.  Returns {package: [ids
    registered now]}.  An id somebody registered before (the reference itself, imported in the same process) is left
    alone; the only exception swallowed is that package's own re-registration error."""
    done = {}
    for pkg, new_api in (("gym", False), ("gymnasium", True)):
        reg = _try_import(pkg + ".envs.registration")
        if reg is None:
            continue
        err_mod = _try_import(pkg + ".error")
        reregister_error = getattr(err_mod, "Error", None) or ()
        have = _registered_ids(reg)
        try:
            params = inspect.signature(reg.register).parameters
        except (TypeError, ValueError):
            params = {}
        done[pkg] = []
        for env_id, entry in registry.items():
            if env_id in have:
                continue
            kwargs = {}
            if new_api:
                mod, attr = entry.split(":")
                entry_point = gymnasium_entry_point(mod, attr)
            else:
                entry_point = entry
                if "apply_api_compatibility" in params:      # gym 0.26: the ids speak the old api
                    kwargs["apply_api_compatibility"] = True
            try:
                reg.register(id=env_id, entry_point=entry_point, **kwargs)
                done[pkg].append(env_id)
            except reregister_error:                          # "Cannot re-register id": registered meanwhile
                pass
    return done

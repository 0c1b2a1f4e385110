"""Minimal stand-in for the `gymnasium` package (TEST INFRASTRUCTURE, container-only; gymnasium is not installed in this
image).  It reproduces what `gymnasium.make` does to the object an entry point builds — the `isinstance(env,
gymnasium.Env)` check (TypeError otherwise), `env.unwrapped.spec = spec`, the `PassiveEnvChecker` and `OrderEnforcing`
wrappers — and the NEW api those wrappers insist on: `reset(seed=, options=) -> (ob, info)`, `step(a) -> (ob, reward,
terminated, truncated, info)`.  Used only by the tests of the product's `gym.make` boundary."""
from gymnasium import error  # noqa: F401
from gymnasium.core import Env, Wrapper  # noqa: F401
from gymnasium import spaces  # noqa: F401
from gymnasium import wrappers  # noqa: F401
from gymnasium.envs.registration import make, register, registry, spec  # noqa: F401

__version__ = "1.0.0-stub"

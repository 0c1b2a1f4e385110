"""Minimal stand-in for the `gym` package (TEST INFRASTRUCTURE, container-only).

`gym` is not installed in this image.  The reference (d3sm0/gym_pomdp) only
needs `gym.Env`, `gym.core.Env`, `gym.spaces.Discrete` and
`gym.envs.registration.register/make`, so this stub provides exactly those with
old-gym (0.21) semantics — including what `gym.make` DOES with the object it
builds (`EnvSpec.make`: `env.unwrapped.spec = spec`, the `OrderEnforcing`
wrapper, the re-registration error), because that is what a drop-in env class
has to survive.  It is used by oracle/ref_harness to import the reference from
/root/reference and generate golden fixtures, and by the tests of the product's
`gym.make` boundary; it never travels into the product path.
"""
from gym import error  # noqa: F401
from gym.core import Env, Wrapper  # noqa: F401
from gym import spaces  # noqa: F401
from gym import wrappers  # noqa: F401
from gym.envs.registration import make, register, spec  # noqa: F401

__version__ = "0.21.0-stub"

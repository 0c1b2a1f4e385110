"""Minimal stand-in for the `gym` package (TEST INFRASTRUCTURE, container-only).

`gym` is not installed in this image.  The reference (d3sm0/gym_pomdp) only
needs `gym.Env`, `gym.core.Env`, `gym.spaces.Discrete` and
`gym.envs.registration.register/make`, so this stub provides exactly those with
old-gym (0.10-0.21) semantics.  It is used only by oracle/ref_harness to import
the reference from /root/reference and generate golden fixtures; it never
travels into the product path.
"""
from gym.core import Env  # noqa: F401
from gym import spaces  # noqa: F401
from gym.envs.registration import make, register  # noqa: F401

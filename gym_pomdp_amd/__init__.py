"""gym_pomdp_amd — MI355X-native batched simulator for gym_pomdp's discrete POMDP envs.

Drop-in boundary (reference: gym_pomdp/__init__.py:7-46): the same env ids resolve to batched
classes with the same constructor kwargs plus `batch_size`, `device`, `seed`, `auto_reset`,
`lane_offset`.  When `gym` or `gymnasium` is importable the ids are also registered there, so
`gym.make("Rock-v0", batch_size=1 << 20)` works; `gym_pomdp_amd.make` is always available.
"""
import importlib
import os

# The kernels take their parameter tables by value (~1 KB of kernel arguments per launch); with the arguments in device
# memory a 2^20-lane step launch is 0.3-1.8 us shorter (bench.py).  Only effective if set before the HIP runtime
# initialises, hence at import — as a DEFAULT: a value the process already has is kept, and GYM_POMDP_AMD_KEEP_ENV=1 leaves
# the environment alone altogether (the setting is process-wide: it changes where every HIP kernel of the process finds its
# arguments, not only this package's).
if not os.environ.get("GYM_POMDP_AMD_KEEP_ENV"):
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import spaces  # noqa: F401
from .history import EpisodeStats, History, Returns, Transition  # noqa: F401
from .envs import BattleShipEnv, NetworkEnv, RockEnv, StochasticRockEnv, TagEnv, TigerEnv  # noqa: F401

__version__ = "0.1.0"

# id -> entry point, as the reference registers them (gym_pomdp/__init__.py:7-41)
registry = {
    "Tiger-v0": "gym_pomdp_amd.envs:TigerEnv",
    "Tag-v0": "gym_pomdp_amd.envs:TagEnv",
    "Battleship-v0": "gym_pomdp_amd.envs:BattleShipEnv",
    "Rock-v0": "gym_pomdp_amd.envs:RockEnv",
    "StochasticRock-v0": "gym_pomdp_amd.envs:StochasticRockEnv",   # the reference's entry point has a typo (line 35)
    "Network-v0": "gym_pomdp_amd.envs:NetworkEnv",
}


def register(id, entry_point):
    registry[id] = entry_point


def make(id, **kwargs):
    """gym.make look-alike: resolves `id` through this package's registry."""
    if id not in registry:
        raise KeyError("unknown env id %r (known: %s)" % (id, sorted(registry)))
    mod, attr = registry[id].split(":")
    return getattr(importlib.import_module(mod), attr)(**kwargs)


def _register_with_gym():
    """gym_pomdp/__init__.py:7-41 for the batched classes: every gym / gymnasium that is importable gets the ids (compat.py)."""
    from . import compat
    return compat.register_ids(registry)


_register_with_gym()

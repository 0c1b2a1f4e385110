"""Batched env classes, re-exported under the reference's names
(gym_pomdp/envs/__init__.py:1-8; Pocman and TestEnv are out of scope — SURVEY.md §2)."""
from .battleship import BattleShipEnv
from .network import NetworkEnv
from .rock import RockEnv, StochasticRockEnv
from .tag import TagEnv
from .tiger import TigerEnv

__all__ = ["BattleShipEnv", "NetworkEnv", "RockEnv", "StochasticRockEnv", "TagEnv", "TigerEnv"]

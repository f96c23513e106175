"""Upkie environments on the MI355X-native batched simulation."""

from .entry_points import REGISTRY, make
from .single import UpkieBaseVelocity, UpkieGyropod, UpkiePendulum, UpkieServos
from .vec_env import (
    UpkieBaseVelocityVecEnv,
    UpkieGyropodVecEnv,
    UpkiePendulumVecEnv,
    UpkieServosVecEnv,
    UpkieVecEnv,
)


def register() -> None:
    """Register the ids with Gymnasium when it is installed
    (upkie/envs/__init__.py:24-44). ``Upkie-PyBullet-*`` ids are only claimed
    when the reference package has not registered them."""
    try:
        import gymnasium as gym  # type: ignore
    except ImportError:
        return
    for env_id, factory in REGISTRY.items():
        if env_id in gym.registry:
            continue
        gym.register(id=env_id, entry_point=f"upkie_amd.envs.entry_points:{factory}", disable_env_checker=True)


__all__ = [
    "REGISTRY",
    "UpkieBaseVelocity",
    "UpkieBaseVelocityVecEnv",
    "UpkieGyropod",
    "UpkieGyropodVecEnv",
    "UpkiePendulum",
    "UpkiePendulumVecEnv",
    "UpkieServos",
    "UpkieServosVecEnv",
    "UpkieVecEnv",
    "make",
    "register",
]

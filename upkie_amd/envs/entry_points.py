"""Environment factories and ids, after upkie/envs/entry_points.py and
upkie/envs/__init__.py:24-44. Ids follow ``<Robot>-<Backend>-<Action>``:

    Upkie-HIP-{Servos,Gyropod,Pendulum,BaseVelocity}        one robot, numpy API
    Upkie-HIP-{...}-Vec                                     num_envs robots, torch API

``Upkie-PyBullet-*`` resolves to the HIP envs when pybullet is not installed,
so agents written against the reference ids run unchanged. ``Cookie-*`` ids
(the right-wheeled sibling robot, entry_points.py:295-336) build the same envs
on the Cookie model.
"""

import os

from ..model.model import Model

from .single import UpkieBaseVelocity, UpkieGyropod, UpkiePendulum, UpkieServos
from .vec_env import (
    UpkieBaseVelocityVecEnv,
    UpkieGyropodVecEnv,
    UpkiePendulumVecEnv,
    UpkieServosVecEnv,
)

ACTIONS = {
    "Servos": (UpkieServos, UpkieServosVecEnv),
    "Gyropod": (UpkieGyropod, UpkieGyropodVecEnv),
    "Pendulum": (UpkiePendulum, UpkiePendulumVecEnv),
    "BaseVelocity": (UpkieBaseVelocity, UpkieBaseVelocityVecEnv),
}


def make_upkie_hip_servos(**kwargs):
    return UpkieServos(**kwargs)


def make_upkie_hip_gyropod(**kwargs):
    return UpkieGyropod(**kwargs)


def make_upkie_hip_pendulum(**kwargs):
    return UpkiePendulum(**kwargs)


def make_upkie_hip_base_velocity(**kwargs):
    return UpkieBaseVelocity(**kwargs)


def make_upkie_hip_servos_vec(**kwargs):
    return UpkieServosVecEnv(**kwargs)


def make_upkie_hip_gyropod_vec(**kwargs):
    return UpkieGyropodVecEnv(**kwargs)


def make_upkie_hip_pendulum_vec(**kwargs):
    return UpkiePendulumVecEnv(**kwargs)


def make_upkie_hip_base_velocity_vec(**kwargs):
    return UpkieBaseVelocityVecEnv(**kwargs)


def _get_cookie_model() -> Model:
    """entry_points.py:295-308: the Cookie URDF of `cookie_description` when
    that package is installed, else the synthetic right-wheeled model."""
    try:
        import cookie_description  # type: ignore

        return Model(urdf_path=cookie_description.URDF_PATH)
    except ImportError:
        here = os.path.dirname(os.path.abspath(__file__))
        return Model(urdf_path=os.path.join(here, "..", "model", "cookie_synthetic.urdf"))


def make_cookie(env_id: str, **kwargs):
    kwargs.setdefault("model", _get_cookie_model())
    return globals()[REGISTRY["Upkie" + env_id[len("Cookie"):]]](**kwargs)


def _snake(name: str) -> str:
    return "base_velocity" if name == "BaseVelocity" else name.lower()


REGISTRY = {}
for _action in ACTIONS:
    REGISTRY[f"Upkie-HIP-{_action}"] = f"make_upkie_hip_{_snake(_action)}"
    REGISTRY[f"Upkie-HIP-{_action}-Vec"] = f"make_upkie_hip_{_snake(_action)}_vec"
    REGISTRY[f"Upkie-PyBullet-{_action}"] = f"make_upkie_hip_{_snake(_action)}"


COOKIE_IDS = tuple("Cookie" + env_id[len("Upkie"):] for env_id in REGISTRY)


def make(env_id: str, **kwargs):
    """``gym.make`` equivalent that works without gymnasium installed."""
    if env_id in COOKIE_IDS:
        return make_cookie(env_id, **kwargs)
    if env_id not in REGISTRY:
        raise KeyError(f"unknown environment id {env_id!r}; known: {sorted(REGISTRY)}")
    return globals()[REGISTRY[env_id]](**kwargs)

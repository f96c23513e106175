"""Single-robot environments with the reference's Gymnasium surface: numpy
observations/actions, ``reset(seed=...) -> (obs, info)``, ``step(action) ->
(obs, reward, terminated, truncated, info)`` and the full spine observation
dictionary in ``info["spine_observation"]`` -- backed by the batched envs with
``num_envs=1``. Existing agents written for ``gym.make("Upkie-PyBullet-...")``
run unchanged on ``make("Upkie-HIP-...")``."""

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ..abi import ACTION_KEYS, JOINT_NAMES, SERVO_OBS_KEYS
from .spaces import Box, Dict as DictSpace
from .spine_observation import spine_observation_dict
from .vec_env import (
    UpkieBaseVelocityVecEnv,
    UpkieGyropodVecEnv,
    UpkiePendulumVecEnv,
    UpkieServosVecEnv,
)


class _SingleEnv:
    """Adapter from a ``num_envs=1`` vector env to the single-env API."""

    _vec_class = None
    metadata = {"render_modes": []}

    def __init__(self, **kwargs):
        kwargs.setdefault("regulate_frequency", False)
        if kwargs.pop("regulate_frequency"):
            from ..exceptions import UpkieException

            raise UpkieException(
                "regulate_frequency=True (the reference default) sleeps to real time; "
                "pass regulate_frequency=False to the simulated envs"
            )
        kwargs.pop("frequency_checks", None)
        self._vec = self._vec_class(num_envs=1, autoreset=False, **kwargs)
        self._vec.host_sampling = True  # reset(seed=s) reproduces the reference's np_random states (upkie_env.py:180-190)
        self.observation_space = self._vec.single_observation_space
        self.action_space = self._vec.single_action_space
        self.model = self._vec.model
        self.init_state = self._vec.init_state

    # attributes of upkie_env.py:104-119
    @property
    def dt(self) -> float:
        return self._vec.dt

    @property
    def frequency(self) -> float:
        return self._vec.frequency

    @property
    def unwrapped(self):
        return self

    @property
    def backend(self):
        return self._vec.sim

    def close(self) -> None:
        self._vec.close()

    # `with gym.make(...) as env:` as every reference example does (gymnasium.Env.__enter__/__exit__)
    def __enter__(self):
        return self

    def __exit__(self, *args) -> bool:
        self.close()
        return False

    def update_init_rand(self, **kwargs) -> None:
        self._vec.update_init_rand(**kwargs)

    def _info(self) -> dict:
        raw = self._vec.sim.observe(update_imu=True)
        return {"spine_observation": spine_observation_dict(raw, env=0)}

    def _obs(self, obs: torch.Tensor) -> np.ndarray:
        return obs[0].detach().cpu().numpy().astype(np.float32)

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None) -> Tuple[np.ndarray, Dict]:
        obs, _ = self._vec.reset(seed=seed, options=options)
        return self._obs(obs), self._info()

    def step(self, action) -> Tuple[np.ndarray, float, bool, bool, dict]:
        act = torch.as_tensor(np.asarray(action, dtype=np.float32))[None]
        obs, reward, terminated, truncated, _ = self._vec.step(act)
        return (
            self._obs(obs),
            float(reward[0].item()),  # always 0.0, upkie_env.py:230
            bool(terminated[0].item()),
            bool(truncated[0].item()),
            self._info(),
        )


class UpkiePendulum(_SingleEnv):
    """upkie/envs/upkie_pendulum.py: action [ground velocity] -> observation
    [pitch, ground position, pitch rate, ground velocity]."""

    _vec_class = UpkiePendulumVecEnv


class UpkieGyropod(_SingleEnv):
    """upkie/envs/upkie_gyropod.py."""

    _vec_class = UpkieGyropodVecEnv

    @property
    def leg_gain_scale(self) -> float:
        return self._vec.leg_gain_scale

    def set_leg_gain_scale(self, leg_gain_scale: float) -> None:
        self._vec.set_leg_gain_scale(leg_gain_scale)

    @property
    def fall_pitch(self) -> float:
        return self._vec.fall_pitch


class UpkieBaseVelocity(_SingleEnv):
    """upkie/envs/upkie_base_velocity.py."""

    _vec_class = UpkieBaseVelocityVecEnv

    @property
    def mpc_balancer(self):
        return self._vec.mpc_balancer


class UpkieServos(_SingleEnv):
    """upkie/envs/upkie_servos.py: dictionary actions and observations keyed
    by joint name, one-element float32 arrays as leaves."""

    _vec_class = UpkieServosVecEnv
    ACTION_KEYS = ACTION_KEYS

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        box = self._vec.single_action_space
        obox = self._vec.single_observation_space
        self.action_space = DictSpace(
            {
                name: DictSpace(
                    {key: Box(box.low[j, k : k + 1], box.high[j, k : k + 1], shape=(1,), dtype=np.float32) for k, key in enumerate(ACTION_KEYS)}
                )
                for j, name in enumerate(JOINT_NAMES)
            }
        )
        self.observation_space = DictSpace(
            {
                name: DictSpace(
                    {key: Box(obox.low[j, k : k + 1], obox.high[j, k : k + 1], shape=(1,), dtype=np.float32) for k, key in enumerate(SERVO_OBS_KEYS)}
                )
                for j, name in enumerate(JOINT_NAMES)
            }
        )
        neutral = self._vec._neutral
        self._neutral_action = {
            name: {key: float(neutral[j, k]) for k, key in enumerate(ACTION_KEYS)} for j, name in enumerate(JOINT_NAMES)
        }

    def get_neutral_action(self) -> dict:
        """upkie_servos.py:308-314."""
        return {name: dict(values) for name, values in self._neutral_action.items()}

    def _obs(self, obs: torch.Tensor) -> dict:
        arr = obs[0].detach().cpu().numpy()
        return {
            name: {key: np.array([arr[j, k]], dtype=np.float32) for k, key in enumerate(SERVO_OBS_KEYS)}
            for j, name in enumerate(JOINT_NAMES)
        }

    def step(self, action: dict):
        act = np.zeros((6, 6), dtype=np.float32)
        for j, name in enumerate(JOINT_NAMES):
            for k, key in enumerate(ACTION_KEYS):
                # missing keys fall back to the neutral action; a missing joint
                # is a KeyError as in the reference (upkie_servos.py:326-330)
                value = action[name][key] if key in action[name] else self._neutral_action[name][key]
                act[j, k] = value.item() if isinstance(value, np.ndarray) else float(value)
        obs, reward, terminated, truncated, _ = self._vec.step(torch.from_numpy(act)[None])
        return self._obs(obs), float(reward[0].item()), bool(terminated[0].item()), bool(truncated[0].item()), self._info()

"""Adapters between the batched envs and RL libraries.

The reference ships single-robot Gymnasium envs and leaves vectorisation to
the RL playgrounds outside the repository (README.md:107-109: they wrap the
envs in stable-baselines3 `SubprocVecEnv`s, one PyBullet process per robot).
Here the batch already lives on one GPU; these classes only translate the API:

* `NumpyVectorEnv`: `gymnasium.vector.VectorEnv` surface with numpy arrays
  (`reset(seed=, options=)`, `step(actions)`, `final_obs` with the env's
  autoreset mode);
* `Sb3VecEnv`: stable-baselines3 `VecEnv` surface (`step_async` /
  `step_wait`, `dones`, `terminal_observation`, `TimeLimit.truncated`).

Neither library is imported: the classes are duck-typed so that they work
without the packages, and `Sb3VecEnv` registers itself as a virtual subclass of
`stable_baselines3.common.vec_env.VecEnv` when that can be imported.
"""

from types import MappingProxyType
from typing import Any, List, Optional, Sequence

import numpy as np
import torch

_NO_INFO = MappingProxyType({})  # shared read-only info of envs that did not finish an episode


def _to_numpy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return t


class NumpyVectorEnv:
    """numpy in / numpy out view of a `Upkie*VecEnv` with the attributes
    `gymnasium.vector.VectorEnv` users read."""

    def __init__(self, env):
        self.env = env
        self.num_envs = env.num_envs
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.single_observation_space = env.single_observation_space
        self.single_action_space = env.single_action_space
        self.metadata = {"autoreset_mode": env.autoreset_mode}
        self.render_mode = None
        self.closed = False

    @property
    def unwrapped(self):
        return self.env

    def _info(self, info: dict) -> dict:
        out = {"spine_observation": info["spine_observation"]}  # stays lazy / on the device
        if "final_obs" in info:
            out["final_obs"] = _to_numpy(info["final_obs"])
            out["_final_obs"] = _to_numpy(info["_final_obs"])
        return out

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        obs, info = self.env.reset(seed=seed, options=options)
        return _to_numpy(obs), self._info(info)

    def step(self, actions):
        obs, reward, terminated, truncated, info = self.env.step(torch.as_tensor(np.asarray(actions), dtype=torch.float32))
        return _to_numpy(obs), _to_numpy(reward), _to_numpy(terminated), _to_numpy(truncated), self._info(info)

    def close(self, **kwargs):
        if not self.closed:
            self.env.close()
            self.closed = True


class Sb3VecEnv:
    """stable-baselines3 `VecEnv` surface over a `Upkie*VecEnv`.

    SB3 expects SAME_STEP semantics (the observation returned with `done` is
    already the first one of the next episode, the last one of the finished
    episode travels in ``infos[i]["terminal_observation"]``), so the wrapped env
    must be built with ``autoreset_mode="same_step"``."""

    def __init__(self, env):
        if env.autoreset_mode != "same_step":
            raise ValueError('Sb3VecEnv needs an env built with autoreset_mode="same_step"')
        self.env = env
        self.num_envs = env.num_envs
        self.observation_space = env.single_observation_space  # SB3 stores the per-env spaces
        self.action_space = env.single_action_space
        self.render_mode = None
        self.reset_infos: List[dict] = [{} for _ in range(self.num_envs)]
        self._seed: Optional[int] = None
        self._options: Optional[dict] = None
        self._actions = None

    # -- stepping ----------------------------------------------------------------
    def reset(self) -> np.ndarray:
        obs, _ = self.env.reset(seed=self._seed, options=self._options)
        self._seed, self._options = None, None  # SB3: seeds and options are used once
        self.reset_infos = [{} for _ in range(self.num_envs)]
        return _to_numpy(obs)

    def step_async(self, actions) -> None:
        self._actions = torch.as_tensor(np.asarray(actions), dtype=torch.float32)

    def step_wait(self):
        obs, reward, terminated, truncated, info = self.env.step(self._actions)
        done = terminated | truncated
        infos: List[Any] = [_NO_INFO] * self.num_envs
        finished = torch.nonzero(done).flatten().tolist()
        if finished:
            final = _to_numpy(info["final_obs"])
            time_limit = _to_numpy(truncated & ~terminated)
            for i in finished:
                infos[i] = {"terminal_observation": final[i], "TimeLimit.truncated": bool(time_limit[i])}
        return _to_numpy(obs), _to_numpy(reward), _to_numpy(done), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        self.env.close()

    # -- VecEnv plumbing -----------------------------------------------------------
    def seed(self, seed: Optional[int] = None) -> Sequence[Optional[int]]:
        """One seed for the batch: env i draws from the Philox stream (seed, i)."""
        self._seed = seed
        return [seed] * self.num_envs

    def set_options(self, options=None) -> None:
        self._options = options if isinstance(options, dict) or options is None else options[0]

    def _indices(self, indices) -> List[int]:
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        value = getattr(self.env, attr_name)
        return [value for _ in self._indices(indices)]

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        setattr(self.env, attr_name, value)  # one simulation: attributes are shared by the batch

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs) -> List[Any]:
        result = getattr(self.env, method_name)(*method_args, **method_kwargs)
        return [result for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None for _ in range(self.num_envs)]

    def render(self, mode: Optional[str] = None):
        return None

    @property
    def unwrapped(self):
        return self


try:  # pragma: no cover - stable-baselines3 is optional
    from stable_baselines3.common.vec_env import VecEnv as _Sb3Base

    _Sb3Base.register(Sb3VecEnv)
except Exception:  # noqa: BLE001 - any import problem just means "not installed"
    pass

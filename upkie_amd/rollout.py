"""On-device rollout buffer: the consumer of the per-step records a batch of
envs produces (BASELINE.json configs[3]: "RCCL gather of obs to a PPO rollout
consumer"). Everything stays in HBM as ``[T, N, ...]``
tensors; advantages are computed by one HIP kernel (`gae_kernel`,
csrc/rollout.hpp). Method names follow the rollout buffer of stable-baselines3,
the learner the reference's RL playgrounds use (README.md:107-109)."""

import ctypes as C
from typing import Dict, Iterator, Optional, Tuple

import torch

from . import lib
from .exceptions import UpkieRuntimeError


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def compute_gae(
    rewards: torch.Tensor,
    values: torch.Tensor,
    episode_starts: torch.Tensor,
    last_values: torch.Tensor,
    last_dones: torch.Tensor,
    gamma: float,
    gae_lambda: float,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """(advantages, returns) of a ``[T, N]`` rollout on the GPU."""
    if not rewards.is_cuda:
        raise UpkieRuntimeError("compute_gae runs on the HIP device only (there is no CPU fallback)")
    T, N = rewards.shape
    dev = rewards.device
    f = lambda t, shape: t.to(dev, torch.float32).reshape(shape).contiguous()
    u = lambda t, shape: t.to(dev, torch.uint8).reshape(shape).contiguous()
    rewards, values = f(rewards, (T, N)), f(values, (T, N))
    episode_starts = u(episode_starts, (T, N))
    last_values, last_dones = f(last_values, (N,)), u(last_dones, (N,))
    advantages = torch.empty((T, N), dtype=torch.float32, device=dev)
    returns = torch.empty((T, N), dtype=torch.float32, device=dev)
    library = lib.load()
    with torch.cuda.device(dev):
        status = library.upkie_rollout_gae(
            T, N, _ptr(rewards), _ptr(values), _ptr(episode_starts), _ptr(last_values), _ptr(last_dones),
            float(gamma), float(gae_lambda), _ptr(advantages), _ptr(returns),
            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
        )
    lib.check(status, None)
    return advantages, returns


class RolloutBuffer:
    """``buffer_size`` steps of ``n_envs`` envs, on the device."""

    def __init__(self, buffer_size: int, n_envs: int, obs_shape, action_shape, device="cuda:0", gamma: float = 0.99, gae_lambda: float = 0.95):
        self.buffer_size, self.n_envs = int(buffer_size), int(n_envs)
        self.gamma, self.gae_lambda = float(gamma), float(gae_lambda)
        self.device = torch.device(device)
        T, N = self.buffer_size, self.n_envs
        f32 = dict(dtype=torch.float32, device=self.device)
        self.observations = torch.zeros((T, N) + tuple(obs_shape), **f32)
        self.actions = torch.zeros((T, N) + tuple(action_shape), **f32)
        self.rewards = torch.zeros((T, N), **f32)
        self.values = torch.zeros((T, N), **f32)
        self.log_probs = torch.zeros((T, N), **f32)
        self.episode_starts = torch.zeros((T, N), dtype=torch.uint8, device=self.device)
        self.advantages: Optional[torch.Tensor] = None
        self.returns: Optional[torch.Tensor] = None
        self.pos = 0
        self.full = False

    def reset(self) -> None:
        self.pos, self.full = 0, False
        self.advantages = self.returns = None

    def add(self, obs, action, reward, episode_start, value, log_prob) -> None:
        if self.full:
            raise UpkieRuntimeError("rollout buffer is full")
        t = self.pos
        self.observations[t].copy_(obs.reshape(self.observations[t].shape))
        self.actions[t].copy_(action.reshape(self.actions[t].shape))
        self.rewards[t].copy_(reward.reshape(self.n_envs))
        self.episode_starts[t].copy_(episode_start.reshape(self.n_envs))
        self.values[t].copy_(value.reshape(self.n_envs))
        self.log_probs[t].copy_(log_prob.reshape(self.n_envs))
        self.pos += 1
        self.full = self.pos == self.buffer_size

    def compute_returns_and_advantage(self, last_values: torch.Tensor, dones: torch.Tensor) -> None:
        """`last_values` [N]: value estimates of the observations after the last
        stored step; `dones` [N]: whether that step ended an episode."""
        if not self.full:
            raise UpkieRuntimeError("rollout buffer is not full yet")
        self.advantages, self.returns = compute_gae(
            self.rewards, self.values, self.episode_starts, last_values, dones, self.gamma, self.gae_lambda
        )

    def get(self, batch_size: Optional[int] = None, generator: Optional[torch.Generator] = None) -> Iterator[Dict[str, torch.Tensor]]:
        """Shuffled minibatches over the flattened ``T * N`` samples."""
        if self.advantages is None:
            raise UpkieRuntimeError("call compute_returns_and_advantage() first")
        total = self.buffer_size * self.n_envs
        flat = {
            "observations": self.observations.reshape((total,) + self.observations.shape[2:]),
            "actions": self.actions.reshape((total,) + self.actions.shape[2:]),
            "old_values": self.values.reshape(total),
            "old_log_prob": self.log_probs.reshape(total),
            "advantages": self.advantages.reshape(total),
            "returns": self.returns.reshape(total),
        }
        order = torch.randperm(total, device=self.device, generator=generator)
        batch_size = total if batch_size is None else int(batch_size)
        for start in range(0, total, batch_size):
            idx = order[start : start + batch_size]
            yield {k: v[idx] for k, v in flat.items()}

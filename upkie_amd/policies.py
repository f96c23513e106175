"""A linear policy as ONE kernel launch, for the loop ``obs, ... = env.step(policy(obs))``.

The reference leaves the policy to the agent (README.md:60-67: ``action =
clamp(gains . obs)``; examples/pybullet/pd_balancing.py). Evaluated with torch
ops on the device -- ``(obs @ W).clamp(-c, c)`` -- it is two or three launches
of 2-5 us each (rocBLAS's gemv for a four-column matrix: 4.9 us) behind a
14.5 us step; `LinearPolicy` is one launch (`upkie_linear_policy`,
csrc/rollout.hpp) writing into a persistent action buffer."""

from typing import Optional

import torch

from . import lib
from .exceptions import UpkieRuntimeError


class LinearPolicy:
    """``act = clamp(obs @ weights + bias, -clip, clip)`` on the device.

    ``weights`` is ``[obs_dim, act_dim]`` (or ``[obs_dim]`` for one action),
    ``bias`` ``[act_dim]`` or None, ``clip`` a positive bound or None. The call
    returns the policy's own ``[num_envs, act_dim]`` buffer, rewritten by every
    call (as the envs' output buffers are)."""

    def __init__(self, weights, bias=None, clip: Optional[float] = None, device="cuda:0"):
        self.device = torch.device(device)
        w = torch.as_tensor(weights, dtype=torch.float32)
        if w.dim() == 1:
            w = w[:, None]
        if w.dim() != 2:
            raise ValueError("weights must be [obs_dim] or [obs_dim, act_dim]")
        self.weights = w.to(self.device).contiguous()
        self.obs_dim, self.act_dim = (int(d) for d in self.weights.shape)
        self.bias = None if bias is None else torch.as_tensor(bias, dtype=torch.float32).reshape(self.act_dim).to(self.device).contiguous()
        self.clip = 0.0 if clip is None else float(clip)
        if clip is not None and not self.clip > 0.0:
            raise ValueError("clip must be positive (None: no clamp)")
        self._act = None
        self._lib = None

    def __call__(self, obs: torch.Tensor) -> torch.Tensor:
        if not obs.is_cuda:
            raise UpkieRuntimeError("LinearPolicy runs on the HIP device only (there is no CPU fallback)")
        if obs.dtype is not torch.float32 or not obs.is_contiguous():
            obs = obs.to(torch.float32).contiguous()
        n = obs.shape[0]
        if obs.numel() != n * self.obs_dim:
            raise ValueError(f"observation rows must hold {self.obs_dim} words")
        if self._act is None or self._act.shape[0] != n or self._act.device != obs.device:
            self._act = torch.empty((n, self.act_dim), dtype=torch.float32, device=obs.device)
            self._lib = lib.load()
            if not hasattr(self._lib, "upkie_linear_policy"):
                raise UpkieRuntimeError("this build of libupkie_hip.so has no upkie_linear_policy")
        with torch.cuda.device(obs.device):
            status = self._lib.upkie_linear_policy(n, self.obs_dim, self.act_dim, obs.data_ptr(), self.weights.data_ptr(),
                                                   None if self.bias is None else self.bias.data_ptr(), self.clip, self._act.data_ptr(),
                                                   torch.cuda.current_stream(obs.device).cuda_stream)
        if status < 0:
            lib.check(status, None)
        return self._act

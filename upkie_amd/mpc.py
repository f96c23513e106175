"""Batched MPC balancer: device-side counterpart of the reference's
``MPCBalancer`` (upkie/controllers/mpc_balancer.py:127-312)."""

import ctypes as C
from typing import Optional

import torch

from . import abi, lib
from .exceptions import UpkieRuntimeError


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class BatchedMpc:
    """One condensed box-QP per env, solved every step by fixed-iteration
    ADMM on the matrix cores; warm start and commanded velocity stay on the
    device."""

    def __init__(self, config: abi.UpkieMpcConfig, device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise UpkieRuntimeError("no HIP device visible (there is no CPU fallback)")
        self._lib = lib.load()
        self.device = torch.device(device)
        self.config = config
        self.num_envs = int(config.num_envs)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            status = self._lib.upkie_mpc_create(C.byref(config), C.byref(self._handle))
        lib.check(status, None, what="mpc")
        N, B = int(config.nb_timesteps), self.num_envs
        assert self._lib.upkie_mpc_workspace_bytes(self._handle) == 2 * N * B * 4
        f32 = dict(dtype=torch.float32, device=self.device)
        self.workspace = torch.zeros((2 * N, B), **f32)
        self.commanded_velocity = torch.zeros(B, **f32)
        self.first_input = torch.zeros(B, **f32)

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.upkie_mpc_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, mask: Optional[torch.Tensor] = None) -> None:
        """MPCBalancer.reset (mpc_balancer.py:228-235) for masked envs."""
        if mask is not None:
            mask = mask.to(self.device, torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            status = self._lib.upkie_mpc_reset(
                self._handle, _ptr(self.workspace), _ptr(self.commanded_velocity), _ptr(mask), self._stream()
            )
        lib.check(status, self._handle, what="mpc")

    def step_env(self, x0: torch.Tensor, act: torch.Tensor, contact: torch.Tensor, done: Optional[torch.Tensor], dt: float):
        """First half of the fused UpkieBaseVelocity step: target velocity of
        env e is ``act[e, 0]``; envs flagged in ``done`` (a float row of the
        simulation state) are reset instead of solved."""
        with torch.cuda.device(self.device):
            status = self._lib.upkie_mpc_step_env(
                self._handle,
                _ptr(self.workspace),
                _ptr(x0),
                _ptr(act),
                _ptr(contact),
                _ptr(done),
                float(dt),
                _ptr(self.commanded_velocity),
                self._stream(),
            )
        lib.check(status, self._handle, what="mpc")
        return self.commanded_velocity

    def step(self, x0: torch.Tensor, target_velocity: torch.Tensor, contact: torch.Tensor, dt: float):
        """MPCBalancer.step (mpc_balancer.py:237-312): ``x0[B, 4]`` = ground
        position, pitch, ground velocity, pitch rate. Returns the commanded
        ground velocity ``[B]`` and ``plan.first_input`` ``[B]``."""
        x0 = x0.to(self.device, torch.float32).contiguous()
        target_velocity = target_velocity.to(self.device, torch.float32).contiguous()
        contact = contact.to(self.device, torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            status = self._lib.upkie_mpc_step(
                self._handle,
                _ptr(self.workspace),
                _ptr(x0),
                _ptr(target_velocity),
                _ptr(contact),
                float(dt),
                _ptr(self.commanded_velocity),
                _ptr(self.first_input),
                self._stream(),
            )
        lib.check(status, self._handle, what="mpc")
        return self.commanded_velocity, self.first_input

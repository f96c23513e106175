"""hipGraph capture of launch-bound inner loops.

A batched `env.step()` is one kernel of 25 microseconds; a policy written with
a handful of PyTorch element-wise ops around it costs several times that in
launch overhead when driven from Python. `GraphedLoop` records `body()` (any
mix of PyTorch ops and `BatchedSim` / `BatchedMpc` / `BatchedObservers` calls,
all of which launch on `torch.cuda.current_stream()`) into a hipGraph once and
replays it with a single launch per iteration (or per `unroll` iterations).

Rules of capture (PyTorch's): `body` must not synchronise, allocate fresh
output tensors that are read outside, or change shapes; it must read its inputs
from and write its results to tensors that exist before capture.
"""

from typing import Callable

import torch

from .exceptions import UpkieRuntimeError


class GraphedLoop:
    def __init__(self, body: Callable[[], None], unroll: int = 1, warmup: int = 3, device=None):
        if not torch.cuda.is_available():
            raise UpkieRuntimeError("no HIP device visible (there is no CPU fallback)")
        self.unroll = max(1, int(unroll))
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):  # warm-up off the capture, as torch.cuda.graphs asks
            for _ in range(max(1, int(warmup))):
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for _ in range(self.unroll):
                body()

    def replay(self) -> None:
        """`unroll` iterations of `body`, one graph launch."""
        self.graph.replay()


class GraphedEnvStep:
    """``env.step(policy(obs))`` of a batched env as ONE hipGraph launch.

    A Python RL loop around a 14 us step is host bound as soon as the policy
    is a few PyTorch ops (each costs 4-6 us of interpreter and dispatch time);
    recorded once, policy kernels + step kernel replay with a single launch::

        env = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=4096)
        obs, _ = env.reset(seed=0)                   # the env's persistent observation buffer
        step = GraphedEnvStep(env, lambda o: (o @ gain).clamp(-0.99, 0.99).unsqueeze(1))
        for _ in range(n):
            obs, reward, terminated, truncated, info = step()

    `policy` maps the env's observation buffer (`env.observation`: rewritten in
    place by every step) to an action tensor with PyTorch ops only (no
    synchronisation, no data-dependent shapes: the rules of graph capture).
    The outputs are the env's persistent buffers, as with `env.step`."""

    def __init__(self, env, policy: Callable, unroll: int = 1, warmup: int = 3):
        self.env = env
        obs = env.observation
        holder = {}

        def body():
            holder["out"] = env.step(policy(obs))

        self._loop = GraphedLoop(body, unroll=unroll, warmup=warmup, device=env.device)
        self.out = holder["out"]
        self.steps_per_call = self._loop.unroll

    def __call__(self):
        self._loop.replay()
        spine = self.out[4].get("spine_observation") if isinstance(self.out[4], dict) else None
        if spine is not None and hasattr(spine, "invalidate"):
            spine.invalidate()
        return self.out

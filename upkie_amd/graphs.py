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
    The outputs are the env's persistent buffers, as with `env.step`.

    Capture runs `warmup` real steps and records `unroll` more (torch needs
    the kernels warmed up off the capture; a captured launch is recorded, not
    executed). So that the loop above does start from the `reset()` state,
    the simulation state, the observation buffer and the step's output flags
    are SNAPSHOT before the warm-up and RESTORED after the capture: the first
    `step()` call steps the state `reset()` left, with the random streams of a
    first step (`restore_state=False` keeps the advanced state). Call `reset()`
    before constructing this object: the policy's first evaluation reads the
    env's observation buffer."""

    def __init__(self, env, policy: Callable, unroll: int = 1, warmup: int = 3, restore_state: bool = True):
        self.env = env
        obs = env.observation
        if obs is None or getattr(env, "_stepper_kind", None) is None:
            raise UpkieRuntimeError(
                f"{type(env).__name__} has no persistent observation buffer / single-call step to capture "
                "(UpkieBaseVelocityVecEnv composes its step: capture its loop with GraphedLoop instead)"
            )
        holder = {}

        def body():
            holder["out"] = env.step(policy(obs))

        sim = env.sim
        keep = [t for t in (getattr(sim, n, None) for n in ("state", "reward", "terminated", "truncated", "contact_manifold", "observer_state")) if t is not None]
        keep.append(obs)
        if getattr(env, "_final_obs", None) is not None:
            keep.append(env._final_obs)
        saved = [t.clone() for t in keep] if restore_state else []
        self._loop = GraphedLoop(body, unroll=unroll, warmup=warmup, device=env.device)
        if restore_state:
            final_obs = getattr(env, "_final_obs", None)  # (a SAME_STEP env arms this buffer on its first step: not among the saved ones then)
            for t, s in zip(keep, saved):
                t.copy_(s)
            if final_obs is not None and all(final_obs is not t for t in keep):
                final_obs.copy_(obs)
        self.out = holder["out"]
        self.steps_per_call = self._loop.unroll

    def __call__(self):
        self._loop.replay()
        spine = self.out[4].get("spine_observation") if isinstance(self.out[4], dict) else None
        if spine is not None and hasattr(spine, "invalidate"):
            spine.invalidate()
        return self.out

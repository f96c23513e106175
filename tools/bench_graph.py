"""Eager Python loop vs hipGraph replay of the same policy + env.step body."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upkie_amd import abi  # noqa: E402
from upkie_amd.graphs import GraphedLoop  # noqa: E402
from upkie_amd.sim import BatchedSim  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = abi.default_sim_config(B, seed=0)
cfg.rand_pitch = 0.1
cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
sim = BatchedSim(cfg)
sim.reset()
r = float(sim.model.wheel_radius)
servo = torch.zeros((B, 6, 6), device=sim.device)
servo[:, :, 0] = float("nan")
servo[:, [0, 1, 3, 4], 0] = 0.0
servo[:, :, 3:5] = 1.0
servo[:, 0, 5] = servo[:, 1, 5] = servo[:, 3, 5] = servo[:, 4, 5] = 16.0
servo[:, 2, 5] = servo[:, 5, 5] = 1.7


def body():  # a servo-level balancing policy written as plain PyTorch ops (8 small kernels) + the step
    st = sim.state
    pitch = 2.0 * st[abi.S_QUAT + 2]
    pos = 0.5 * (st[abi.S_Q + 2] - st[abi.S_Q + 5]) * r
    v = (10.0 * pitch + pos).clamp(-0.99, 0.99) / r
    servo[:, 2, 1] = v
    servo[:, 5, 1] = -v
    sim.step_servos(servo)


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for _ in range(100):
    body()
print(f"B={B} eager Python loop: {timeit(body, 1000):.1f} us per env.step()")
for unroll in (1, 8):
    loop = GraphedLoop(body, unroll=unroll)
    us = timeit(loop.replay, 1000 // unroll) / unroll
    print(f"B={B} hipGraph replay, {unroll} step(s) per launch: {us:.1f} us per env.step()")

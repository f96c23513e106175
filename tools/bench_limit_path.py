"""Cost of the rare paths of the step kernel when they are NOT rare: every
robot drives its knees into their stops (joint-limit rows active every
substep), or lies on the floor. Usage: python tools/bench_limit_path.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upkie_amd.envs as envs  # noqa: E402
from upkie_amd import abi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def timeit(fn, steps=500, warmup=100):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for name, setup in (
    ("standing, legs held (common path)", lambda act: None),
    ("knees driven into their stops (limit rows every substep)", lambda act: [(act[:, j, 0].fill_(float("nan")), act[:, j, 2].fill_(6.0)) for j in (1, 4)]),
    ("all joints limp: robots collapse and lie on the floor", lambda act: (act[:, :, 0].fill_(float("nan")), act[:, :, 4].fill_(0.0))),
):
    env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, autoreset_mode="disabled")
    env.reset(seed=0)
    act = env.get_neutral_action()
    act[:, [0, 1, 3, 4], 0] = 0.0
    setup(act)
    us = timeit(lambda: env.sim.step_servos(act))
    st = env.sim.state
    at_stop = ((st[abi.S_Q + 1].abs() > 2.5) | (st[abi.S_Q + 4].abs() > 2.5)).float().mean().item()
    print(f"B={B} {name}: {us:.1f} us/step, base height {st[abi.S_POS + 2].mean().item():.3f} m, knees at a stop {at_stop:.2f}")
    env.close()

"""One-off (round 6): does splitting the headline batch over TWO streams hide the queue's share of a launch (0.7-1.6 us of 15, the
stamped timeline of profiles/r06_fixed_cost.txt)? Two handles of 2048 envs, each stepping on its own stream (the PD agent inside
the launch, as bench.py's headline), against one handle of 4096 envs on one stream; wall time per env.step() of the 4096 envs."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from upkie_amd.sim import BatchedSim  # noqa: E402


def make(envs):
    sim = BatchedSim(bench.make_config(envs))
    o6 = sim.reset()
    sim.obs4.copy_(o6[:, [1, 0, 4, 3]])
    return sim


def run(sims, streams, steps):
    # one ctypes call per launch on pre-fetched addresses (BatchedSim.stepper's fast path, with the stream given explicitly:
    # torch's stream context manager costs 5 us a switch)
    calls = []
    for sim, stream in zip(sims, streams):
        fn = sim._lib.upkie_sim_step_pendulum_agent
        args = (sim._handle, sim.state.data_ptr(), sim.obs4.data_ptr(), sim.reward.data_ptr(), sim.terminated.data_ptr(), sim.truncated.data_ptr(), stream.cuda_stream)
        calls.append((fn, args))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for fn, args in calls:
            fn(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


if __name__ == "__main__":
    for split in (1, 2, 4):
        sims = [make(4096 // split) for _ in range(split)]
        streams = [torch.cuda.Stream() for _ in range(split)]
        run(sims, streams, 300)
        print(f"{split} stream(s) x {4096 // split} envs: " + "  ".join(f"{run(sims, streams, 2000):.2f}" for _ in range(3)) + " us per step of 4096 envs", flush=True)
        for sim in sims:
            sim.close()

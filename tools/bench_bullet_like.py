"""Cost of the Bullet-like contact model (`upkie_sim_set_contact_manifold`) by batch size, beside the default model:
Upkie-Pendulum, PD agent on the device, one launch per env.step().
Usage: python tools/bench_bullet_like.py [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from upkie_amd.sim import BatchedSim

sizes = [int(a) for a in sys.argv[1:]] or [4096, 32768, 65536, 131072, 262144]
for B in sizes:
    row = []
    for model in ("default", "bullet_like", "bullet_like, one lane"):
        if model.endswith("one lane"):
            os.environ["UPKIE_LANES_PER_ENV"] = "1"
        try:
            sim = BatchedSim(bench.make_config(B))
        finally:
            os.environ.pop("UPKIE_LANES_PER_ENV", None)
        if model.startswith("bullet_like"):
            sim.use_bullet_like_contacts()
        o6 = sim.reset()
        sim.obs4.copy_(o6[:, [1, 0, 4, 3]])
        for _ in range(60):
            sim.step_pendulum_agent()
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            sim.step_pendulum_agent()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        row.append(f"{model} ({sim.lanes_per_env} lane{'s' if sim.lanes_per_env > 1 else ''} per env): {us:8.1f} us per step = {B / us * 1e6:.3e} env-steps/s")
        sim.close()
    print(f"B = {B:7d}:  " + "   |   ".join(row))

"""Throughput probe of the fused Pendulum step at several batch sizes (and,
with UPKIE_LANES_PER_ENV=1/2/8, of a forced lane mapping).
Usage: python tools/quick_bench.py [batch sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tests.helpers import randomized_config
from upkie_amd.sim import BatchedSim

sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192, 16384, 32768, 65536, 262144, 1048576]
for B in sizes:
    cfg = randomized_config(B, seed=0, autoreset=True)
    sim = BatchedSim(cfg)
    sim.reset()
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    for _ in range(50):
        sim.step_pendulum_agent()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K):
        sim.step_pendulum_agent()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B:8d}  {dt / K * 1e6:9.1f} us/step  {B * K / dt:12.4e} env-steps/s  lanes/env={sim.lanes_per_env} ({os.environ.get('UPKIE_LANES_PER_ENV', 'auto')})", flush=True)

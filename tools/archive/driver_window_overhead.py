"""What a 20-step timed window (the driver's `bench.py --steps 20 --warmup 5`) costs beyond its kernels: wall time of K launches of
the bench step between two torch.cuda.synchronize() calls, against the time between two events on the stream, with the final wait
(a) left to synchronize() alone, (b) preceded by a busy poll of the stop event. Usage: python tools/driver_window_overhead.py [K] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import bench
from upkie_amd.sim import BatchedSim

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sim = BatchedSim(bench.make_config(4096))
sim.reset()
sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
for _ in range(300):
    sim.step_pendulum_agent()
torch.cuda.synchronize()
for poll in (False, True, False, True):
    wall, dev = [], []
    for _ in range(R):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record()
        for _ in range(K):
            sim.step_pendulum_agent()
        z.record()
        if poll:
            while not z.query():
                pass
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) / K * 1e6)
        dev.append(a.elapsed_time(z) * 1e3 / K)
    print(f"K = {K}, final wait {'event polled, then synchronize()' if poll else 'synchronize() alone':34s}: wall {np.median(wall):6.2f} us per step (min {np.min(wall):.2f}, p90 {np.percentile(wall, 90):.2f}); "
          f"between the events on the stream {np.median(dev):6.2f}", flush=True)

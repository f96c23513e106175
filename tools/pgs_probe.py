import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import make_config
from upkie_amd.model.default_model import default_model
from upkie_amd.sim import BatchedSim
for iters in (50, 0):
    m = default_model(); m.pgs_iterations = iters
    sim = BatchedSim(make_config(4096), m)
    sim.reset(); sim.obs4.copy_(sim.obs6[:, [1,0,4,3]])
    for phase in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): sim.step_pendulum_agent()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"pgs_iterations={iters} steps {phase*500}-{phase*500+500}: {dt/500*1e6:.1f} us/step, episodes {int(sim.state[40].sum())}", flush=True)

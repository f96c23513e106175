import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import make_config
from upkie_amd.model.default_model import default_model
from upkie_amd.sim import BatchedSim
for name in ("off", "on", "on_unbounded", "on_wide"):
    m = default_model(); m.enforce_joint_limits = 0 if name == "off" else 1
    if name == "on_unbounded":
        for j in range(6): m.joint_lower[j] = -float("inf"); m.joint_upper[j] = float("inf")
    if name == "on_wide":
        for j in (0, 1, 3, 4): m.joint_lower[j] = -100.0; m.joint_upper[j] = 100.0
    sim = BatchedSim(make_config(4096), m)
    sim.reset(); sim.obs4.copy_(sim.obs6[:, [1,0,4,3]])
    for _ in range(100): sim.step_pendulum_agent()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(800): sim.step_pendulum_agent()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    q = sim.state[13:19]
    print(f"{name}: {dt/800*1e6:.2f} us/step  max|q hips/knees| {float(q[[0,1,3,4]].abs().max()):.3f}", flush=True)

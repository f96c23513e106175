import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upkie_amd import abi
from upkie_amd.sim import BatchedSim
for B in (4096, 65536):
    cfg = abi.default_sim_config(B, frequency=200.0, seed=1); cfg.rand_pitch = 0.1; cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
    sim = BatchedSim(cfg); sim.reset()
    act = torch.zeros(B, device="cuda")
    for _ in range(20): sim.step_pendulum(act)
    for _ in range(10): sim.contact_points()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): sim.contact_points()
    e.record(); torch.cuda.synchronize()
    print(f"contact_points B={B}: {s.elapsed_time(e)*1e3/200:.1f} us per query")

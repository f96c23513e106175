"""Replay the state that breaks the pair mapping with the debug build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upkie_amd import abi, lib  # noqa: E402

lib.LIB_PATH = lib.LIB_PATH.replace("libupkie_hip.so", "libupkie_hip_debug.so")
from upkie_amd.model.default_model import default_model  # noqa: E402
from upkie_amd.sim import BatchedSim  # noqa: E402

state = [-0.06933, -1e-05, -0.02278, -0.99988, 3e-05, 0.01574, 1e-05, 0.4159, -0.0001, 1.16563, 0.00484, 6.08176, -0.00532, -1.26, -2.50664,
         -2.68964, 1.26007, 2.50653, 2.69033, -0.12588, 1.04576, 0.53095, 0.13535, -1.08255, -0.4982]
for lanes in (sys.argv[1:] or ["1", "2"]):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    cfg = abi.default_sim_config(1, frequency=1000.0, nb_substeps=1)
    cfg.autoreset_mode = abi.AUTORESET_DISABLED
    sim = BatchedSim(cfg, default_model())
    sim.reset()
    if os.path.exists("gpurun_out/bad_substate.pt"):
        sim.state[:25, 0] = torch.load("gpurun_out/bad_substate.pt")["before"].to(sim.state.device)  # exact bits
    else:
        sim.state[:25, 0] = torch.tensor(state)
    act = torch.zeros((1, 6, 6), device="cuda:0")
    act[:, :, 0] = float("nan")
    act[:, :, 3] = 1.0
    act[:, :, 5] = 16.0
    print("==== lanes", lanes, flush=True)
    sim.step_servos(act)
    torch.cuda.synchronize()
    print("after:", [round(float(v), 4) for v in sim.state[:25, 0]], flush=True)

import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tests.helpers import randomized_config
from upkie_amd.sim import BatchedSim
from upkie_amd import abi
for name, mod in (("rand", None), ("norand", "norand"), ("high", "high")):
    cfg = randomized_config(64, seed=3)
    if mod == "norand":
        cfg.rand_pitch = cfg.rand_x = cfg.rand_omega_y = 0.0; cfg.rand_linvel[0] = 0.0
    if mod == "high":
        cfg.init_pos[2] = 0.7
    sim = BatchedSim(cfg)
    sim.reset()
    st = sim.state_numpy()
    nan_envs = np.isnan(st).any(axis=0)
    print(name, "nan envs", nan_envs.sum(), "of 64; nan words:", np.where(np.isnan(st).any(axis=1))[0][:10])
    if nan_envs.any():
        e = np.where(nan_envs)[0][0]; g = np.where(~nan_envs)[0]
        print("  first nan env", e, st[:25, e])
        if len(g): print("  a good env", g[0], st[:25, g[0]])

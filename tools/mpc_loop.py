"""BASELINE C3 in a loop for profiling: UpkieBaseVelocity + MPC balancer on the fp32 MFMA, 16384 envs, 300 env.step().
Usage: python tools/mpc_loop.py [envs] [steps] [horizon, default 16; 50 = the reference's default] [--fused]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import upkie_amd.envs as envs
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 16
init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
env = envs.make("Upkie-HIP-BaseVelocity-Vec", num_envs=B, frequency=200.0, nb_timesteps=N, init_state=init)
env.reset(seed=0)
env.fuse_mpc = "--fused" in sys.argv  # default: the balancer as its own launch (mpc_step_kernel), what tools/pmc_mpc.sh profiles
act = torch.zeros(B, 2, device="cuda:0")
act[:, 0] = torch.empty(B, device="cuda:0").uniform_(-0.5, 0.5)
for _ in range(50):
    env.step(act)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    env.step(act)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"C3 B={B} N={N}: {dt * 1e6:.2f} us per env.step() = {B / dt:.4e} env-steps/s")

"""The C5 share with torque_balancing.py's law on the device (70 % of the substeps sweep), for profiling."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import upkie_amd.envs as envs
from upkie_amd import abi
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization
B = 4096
init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, init_state=init, autoreset_mode="next_step")
env.reset(seed=0)
policy = abi.torque_balancing_policy(gain=10.0, fall_pitch=1.0, left_sign=float(env.model.struct.left_sign))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    env.sim.step_servos(env.sim.servo_policy(policy))
torch.cuda.synchronize()
print("lanes", env.sim.lanes_per_env)

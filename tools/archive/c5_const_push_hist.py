"""tools/ab_step.py --c5's workload (Servos, inertia 0.2, wheel friction 0.1, a CONSTANT torso push of +-5 N, README balancer
through the wheel velocity loop as a separate policy launch) with the rare-path census: time per step and the histogram
of the sweeps a wavefront waits for. Usage: [UPKIE_HIP_LIBRARY=...] python tools/c5_const_push_hist.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import upkie_amd.envs as envs
from upkie_amd import abi
from upkie_amd.model.joint_properties import JointProperties
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization
B = 4096
init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, inertia_variation=0.2, init_state=init, autoreset_mode="next_step",
                joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")})
env.reset(seed=0)
torch.manual_seed(0)
push = torch.zeros(B, 3, device="cuda:0"); push[:, 0] = torch.empty(B, device="cuda:0").uniform_(-5, 5)
env.set_external_forces("torso", push)
policy = abi.velocity_balancing_policy(float(env.model.struct.wheel_radius), 1.0, float(env.model.struct.left_sign))
for _ in range(200): env.sim.step_servos(env.sim.servo_policy(policy))
a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(600): env.sim.step_servos(env.sim.servo_policy(policy))
z.record(); torch.cuda.synchronize()
print(f"{a.elapsed_time(z) * 1e3 / 600:.2f} us/step")
census = env.sim.enable_census()
for _ in range(300): env.sim.step_servos(env.sim.servo_policy(policy))
c = env.sim.census_counts()
print({k: v for k, v in c.items() if k != "wavefront_max_sweeps_histogram"}, "env-substeps", B * 5 * 300)
print("wavefront-substeps by largest sweep count:", c["wavefront_max_sweeps_histogram"][:30], "cap", c["wavefront_max_sweeps_histogram"][50])

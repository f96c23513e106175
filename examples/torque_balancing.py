"""The reference's examples/pybullet/torque_balancing.py for a batch, with the
agent on the device: legs held at zero by the servos, wheel torques
+-10 N.m/rad x pitch, no velocity feedback in the wheels. The law itself is a
servo-level policy evaluated inside the step's own launch
(`upkie_sim_step_servos_policy`): it flags fallen robots, which the NEXT_STEP autoreset
re-initialises: nothing returns to the host between two steps. As on the real
robot the pure pitch-to-torque law does not hold a position: the robots run
away, the tires slip, they fall and start again. The README balancer sent
through the wheels' velocity loop (`velocity_balancing_policy`) beside it."""
import torch

from _common import steps

import upkie_amd.envs as envs
from upkie_amd import abi
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

if __name__ == "__main__":
    B = 4096
    n = steps(1000)
    for name in ("torque_balancing", "velocity_balancing"):
        with envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, autoreset_mode="next_step",
                       init_state=RobotState(randomization=RobotStateRandomization(pitch=0.05, omega_y=0.1))) as env:
            env.reset(seed=0)
            model = env.model.struct
            if name == "torque_balancing":
                policy = abi.torque_balancing_policy(gain=10.0, fall_pitch=1.0, left_sign=float(model.left_sign))
            else:
                policy = abi.velocity_balancing_policy(float(model.wheel_radius), fall_pitch=1.0, left_sign=float(model.left_sign))
            for _ in range(n):
                env.step_servo_policy(policy)  # the policy inside the step's launch (one launch per step up to 8192 envs)
            torch.cuda.synchronize()
            falls = int(env.sim.state[abi.S_EPISODE].sum()) - B
            pitch = 2.0 * env.sim.state[abi.S_QUAT + 2]
            print(f"{name}: {B} robots x {n} steps on {env.sim.lanes_per_env} lanes per env, {falls} falls, mean |pitch| {float(pitch.abs().mean()):.3f} rad")

"""Domain randomisation across a batch, the reference's randomize_inertias.py,
joint_friction.py, sensor_noise.py and apply_external_forces.py at once: every
env gets its own link inertias (one factor per URDF link), joints have
friction and torque noise, and the torsos are pushed now and then."""
import torch

from _common import steps

import upkie_amd.envs as envs
from upkie_amd.model import JointProperties
from upkie_amd.utils.external_force import ExternalForce
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

if __name__ == "__main__":
    B = 2048
    noisy = JointProperties(friction=0.1, torque_control_noise=0.2, torque_measurement_noise=0.05)
    joints = ("left_hip", "left_knee", "left_wheel", "right_hip", "right_knee", "right_wheel")
    for variation in (0.0, 0.1, 0.3):
        with envs.make("Upkie-HIP-Pendulum-Vec", num_envs=B, frequency=200.0, inertia_variation=variation,
                       joint_properties={name: noisy for name in joints}, fall_pitch=1.0,
                       init_state=RobotState(randomization=RobotStateRandomization(pitch=0.05)), autoreset_mode="disabled") as env:
            obs, _ = env.reset(seed=1)
            gain = torch.tensor([10.0, 1.0, 0.0, 0.1], device=env.device)
            fallen = torch.zeros(B, dtype=torch.bool, device=env.device)
            for step in range(steps(1000)):
                if step % 400 == 100:  # a push on every torso: random horizontal direction, up to 15 N, held for 0.1 s
                    angle = torch.rand(B, device=env.device) * 6.2832
                    norm = torch.rand(B, device=env.device) * 15.0
                    force = torch.stack([norm * angle.cos(), norm * angle.sin(), torch.zeros_like(norm)], dim=1)
                    env.set_external_forces({"torso": (force, False)})
                if step % 400 == 120:
                    env.set_external_forces({"torso": ExternalForce([0.0, 0.0, 0.0])})
                action = (obs @ gain).clamp(-0.9, 0.9).unsqueeze(1)
                obs, _, terminated, _, _ = env.step(action)
                fallen |= terminated
            masses = "-" if env.sim.body_inertials is None else f"{float(env.sim.body_inertials[30].min()):.3f}..{float(env.sim.body_inertials[30].max()):.3f} kg"
            print(f"inertia_variation {variation:.1f}: {int(fallen.sum()):4d} of {B} robots fell; left wheel mass {masses}")

"""A batch of robots following velocity commands through the MPC balancer
(reference: examples/pybullet/mpc_balancing.py and UpkieBaseVelocity): two
launches per env.step(), the condensed QP of every env solved on the MFMA."""
import torch

from _common import steps

import upkie_amd.envs as envs
from upkie_amd import abi

if __name__ == "__main__":
    B = 1024
    with envs.make("Upkie-HIP-BaseVelocity-Vec", num_envs=B, frequency=200.0, nb_timesteps=16) as env:
        pose, _ = env.reset(seed=0)
        command = torch.zeros((B, 2), device=env.device)
        command[:, 0] = torch.linspace(-0.5, 0.5, B, device=env.device)  # linear velocity, m/s
        command[:, 1] = 0.3  # yaw velocity, rad/s
        for step in range(steps(600)):
            pose, _, terminated, _, _ = env.step(command)
        x, y, yaw = pose[:, 0], pose[:, 1], pose[:, 2]
        print(f"after {steps(600) / 200.0:.1f} s: dead-reckoned |(x, y)| up to {float((x * x + y * y).sqrt().max()):.2f} m, yaw {float(yaw.mean()):.2f} rad, "
              f"{int((env.sim.state[abi.S_EPISODE] > 1).sum())} of {B} robots fell along the way")

"""4096 robots balancing at once: one kernel launch per env.step(), the policy
a couple of torch ops on the device. Falls restart by themselves (NEXT_STEP
autoreset), episodes are cut after 10 s (time limit kept by the kernel)."""
import time

import torch

from _common import steps

import upkie_amd.envs as envs
from upkie_amd import abi
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

if __name__ == "__main__":
    B = 4096
    init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1))
    with envs.make("Upkie-HIP-Pendulum-Vec", num_envs=B, frequency=200.0, init_state=init, max_episode_steps=2000) as env:
        obs, _ = env.reset(seed=0)
        gain = torch.tensor([10.0, 1.0, 0.0, 0.1], device=env.device)
        n = steps(2000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            action = (obs @ gain).clamp(-0.9, 0.9).unsqueeze(1)
            obs, reward, terminated, truncated, info = env.step(action)  # nothing here waits for the device
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        restarts = int(env.sim.state[abi.S_EPISODE].sum()) - B  # falls + time limits, counted by the kernel
        print(f"{B} envs x {n} steps in {dt:.3f} s = {B * n / dt:.3e} env-steps/s "
              f"({restarts} episodes restarted, mean |pitch| {float(obs[:, 0].abs().mean()):.4f} rad)")
        # the same policy as ONE launch between two steps instead of torch's two (upkie_amd.policies.LinearPolicy)
        from upkie_amd.policies import LinearPolicy

        policy = LinearPolicy(gain, clip=0.9)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            obs, reward, terminated, truncated, info = env.step(policy(obs))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"with the policy as one launch: {B * n / dt:.3e} env-steps/s")

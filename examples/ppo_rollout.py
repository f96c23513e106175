"""The consumer side of a PPO rollout (BASELINE.json configs[3]): 128 steps of
4096 envs collected into an on-device buffer, advantages by the GAE kernel.
The policy and value function are stand-ins: the point is the data path."""
import torch

from _common import steps

import upkie_amd.envs as envs
from upkie_amd.rollout import RolloutBuffer
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

if __name__ == "__main__":
    B, T = 4096, steps(128)
    init = RobotState(randomization=RobotStateRandomization(pitch=0.1))
    with envs.make("Upkie-HIP-Pendulum-Vec", num_envs=B, frequency=200.0, init_state=init, autoreset_mode="same_step",
                   max_episode_steps=400) as env:
        buffer = RolloutBuffer(T, B, obs_shape=(4,), action_shape=(1,), device=env.device)
        gain = torch.tensor([10.0, 1.0, 0.0, 0.1], device=env.device)
        obs, _ = env.reset(seed=0)
        starts = torch.ones(B, dtype=torch.bool, device=env.device)
        for _ in range(T):
            action = ((obs @ gain) + 0.1 * torch.randn(B, device=env.device)).clamp(-0.9, 0.9).unsqueeze(1)
            value = -obs[:, 0].abs()  # stand-in critic
            next_obs, reward, terminated, truncated, info = env.step(action)
            reward = 1.0 - next_obs[:, 0].abs()  # stand-in reward (the reference's is constant, upkie_env.py:230)
            buffer.add(obs, action, reward, starts, value, torch.zeros(B, device=env.device))
            starts = terminated | truncated
            obs = next_obs.clone()
        buffer.compute_returns_and_advantage(last_values=-obs[:, 0].abs(), dones=starts)
        print(f"rollout of {T} x {B}: mean advantage {float(buffer.advantages.mean()):+.4f}, mean return {float(buffer.returns.mean()):+.3f}")

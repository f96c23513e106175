"""Python-level throughput of the vector env API (what an RL loop sees):
`env.step(policy(obs))` from a Python loop, policy evaluated with torch ops on
the device. Usage: python tools/bench_vec_env.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import upkie_amd.envs as envs
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for mode, limit in (("next_step", None), ("same_step", None), ("next_step", 300), ("same_step", 300)):
    env = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=B, frequency=200.0, autoreset_mode=mode, max_episode_steps=limit,
                    init_state=RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1)))
    obs, _ = env.reset(seed=0)
    gain = torch.tensor([10.0, 1.0, 0.0, 0.1], device=env.device)
    for phase in ("warmup", "timed"):
        n = 200 if phase == "warmup" else steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            act = (obs @ gain).clamp(-0.99, 0.99).unsqueeze(1)
            obs, reward, terminated, truncated, info = env.step(act)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"B={B} autoreset={mode} max_episode_steps={limit}: {dt / steps * 1e6:.1f} us per env.step() from Python, {B * steps / dt:.3e} env-steps/s")
    env.close()

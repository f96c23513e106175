"""Python-level throughput of the vector env API (what an RL loop sees):
`env.step(policy(obs))` from a Python loop, policy evaluated with torch ops on
the device; the same recorded into a hipGraph (`GraphedEnvStep`, 1 / 4 / 16
steps per graph launch).
With --policy one_launch the same policy is ONE kernel (`upkie_amd.policies.LinearPolicy`), with --policy in_launch it is
evaluated inside the step's launch (`env.step_linear_policy`, NEXT_STEP only).
Usage: python tools/bench_vec_env.py [B] [steps] [modes, e.g. next_step,same_step] [--no-graph] [--limit] [--policy torch|one_launch|in_launch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import upkie_amd.envs as envs
from upkie_amd.graphs import GraphedEnvStep
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

policy_kind = "torch"
if "--policy" in sys.argv:
    i = sys.argv.index("--policy")
    policy_kind = sys.argv[i + 1]
    del sys.argv[i:i + 2]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if len(args) > 0 else 4096
steps = int(args[1]) if len(args) > 1 else 2000
modes = args[2].split(",") if len(args) > 2 else ["next_step", "same_step"]
limits = (None, 300) if "--limit" in sys.argv else (None,)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for mode in modes:
    for limit in limits:
        env = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=B, frequency=200.0, autoreset_mode=mode, max_episode_steps=limit,
                        init_state=RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1)))
        obs, _ = env.reset(seed=0)
        gain = torch.tensor([10.0, 1.0, 0.0, 0.1], device=env.device)
        policy = lambda o: (o @ gain).clamp(-0.99, 0.99).unsqueeze(1)
        if policy_kind == "one_launch":
            from upkie_amd.policies import LinearPolicy

            policy = LinearPolicy(gain, clip=0.99)
        state = {"obs": obs}
        host_gains = [10.0, 1.0, 0.0, 0.1]

        def eager():
            if policy_kind == "in_launch":
                state["obs"], reward, terminated, truncated, info = env.step_linear_policy(host_gains, clip=0.99)
            else:
                state["obs"], reward, terminated, truncated, info = env.step(policy(state["obs"]))

        timed(eager, 200)
        us = timed(eager, steps)
        print(f"B={B} autoreset={mode} max_episode_steps={limit} policy={policy_kind}: {us:.1f} us per env.step() from Python, {B / us * 1e6:.3e} env-steps/s")
        # where the loop's time goes: the policy's kernels alone (same ops, result dropped), and the host side alone
        us_policy = timed(lambda: policy(state["obs"]), steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            eager()
        host = (time.perf_counter() - t0) / steps * 1e6  # (no synchronisation: what the interpreter needs to ISSUE a step)
        torch.cuda.synchronize()
        print(f"    policy ops alone {us_policy:.1f} us per call; host time to issue one loop iteration {host:.1f} us")
        if "--no-graph" not in sys.argv and policy_kind != "in_launch":
            for unroll in (1, 4, 16):
                graphed = GraphedEnvStep(env, policy, unroll=unroll)
                timed(graphed, 50)
                us = timed(graphed, max(1, steps // unroll)) / unroll
                print(f"    hipGraph of policy + step, {unroll} step(s) per graph launch: {us:.1f} us per env.step()")
        env.close()

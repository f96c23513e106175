"""Where the time of the C5-like Servos step goes: variants of the scenario."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upkie_amd.envs as envs  # noqa: E402
from upkie_amd import abi  # noqa: E402
from upkie_amd.model.joint_properties import JointProperties  # noqa: E402
from upkie_amd.utils.robot_state import RobotState  # noqa: E402
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization  # noqa: E402

B = 4096


def run(name, inertia=0.0, push=0.0, friction=0.0, reset_fallen=False, steps=600):
    kw = {}
    if friction:
        kw["joint_properties"] = {n: JointProperties(friction=friction) for n in ("left_wheel", "right_wheel")}
    env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, inertia_variation=inertia, autoreset_mode="disabled",
                    init_state=RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0]))), **kw)
    env.reset(seed=0)
    if push:
        f = torch.zeros(B, 3, device=env.device)
        f[:, 0] = torch.empty(B, device=env.device).uniform_(-push, push)
        env.set_external_forces("torso", f)
    act = env.get_neutral_action()
    act[:, [0, 1, 3, 4], 0] = 0.0
    act[:, :, 4] = 1.0
    r = env.model.wheel_radius

    def step():
        st = env.sim.state
        pitch = 2.0 * st[abi.S_QUAT + 2]
        pos = 0.5 * (st[abi.S_Q + 2] - st[abi.S_Q + 5]) * r
        v = (10.0 * pitch + pos).clamp(-0.99, 0.99) / r
        act[:, 2, 1] = v
        act[:, 5, 1] = -v
        env.sim.step_servos(act)
        if reset_fallen:
            env.sim.reset(mask=(pitch.abs() > 1.0))

    for _ in range(200):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    pitch = 2.0 * env.sim.state[abi.S_QUAT + 2]
    print(f"{name}: {us:.1f} us/step, fallen {float((pitch.abs() > 1.0).float().mean()):.2f}")
    env.close()


run("balanced, nothing else")
run("balanced + inertia variation 0.2", inertia=0.2)
run("balanced + wheel friction 0.1", friction=0.1)
run("constant torso push +-20 N", push=20.0)
run("constant torso push +-20 N, fallen robots reset", push=20.0, reset_fallen=True)
run("constant torso push +-5 N, fallen robots reset", push=5.0, reset_fallen=True)

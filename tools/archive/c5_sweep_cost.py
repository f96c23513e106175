"""What the Gauss-Seidel sweeps cost the C5 share (bench.py's secondary_c5_share workload, torque_balancing.py law by default): the same
workload with the sweeps capped at 1, 2, 4, 8 and at the model's 50 (`pgs_iterations`). The slope is what one more sweep of the slowest
wavefront costs a launch, the intercept what entering the sweeps costs (gathering the system, warm start, projection).
Usage: python tools/c5_sweep_cost.py [--law torque|velocity] [--caps 1,2,4,8,50] [--steps 600]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import bench
import upkie_amd.envs as envs_mod
from upkie_amd import abi
from upkie_amd.model.joint_properties import JointProperties
from upkie_amd.model.model import Model
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

args = sys.argv[1:]


def take(flag, default):
    if flag in args:
        i = args.index(flag)
        v = args[i + 1]
        del args[i:i + 2]
        return v
    return default


law, caps, steps = take("--law", "torque"), [int(c) for c in take("--caps", "1,2,4,8,50").split(",")], int(take("--steps", "600"))
envs, warmup = 4096, 200
for cap in caps:
    model = Model()
    model.struct.pgs_iterations = cap
    init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
    env = envs_mod.make("Upkie-HIP-Servos-Vec", num_envs=envs, frequency=200.0, inertia_variation=0.2, init_state=init, autoreset_mode="next_step", seed=0, model=model,
                        joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")})
    env.reset(seed=0)
    sim = env.sim
    push = torch.zeros((3, envs), dtype=torch.float32, device=env.device)
    sim.set_external_force(push)
    m = env.model.struct
    policy = (abi.torque_balancing_policy(10.0, 1.0, float(m.left_sign)) if law == "torque" else abi.velocity_balancing_policy(float(m.wheel_radius), 1.0, float(m.left_sign)))

    def step(k):
        phase = k % bench.PUSH_PERIOD
        if phase == 0:
            sim.sample_pushes(k // bench.PUSH_PERIOD, bench.PUSH_MAX_NORM, out=push)
        elif phase == bench.PUSH_HOLD:
            push.zero_()
        sim.step_servos_policy(policy)

    wall, _ = bench._timed_loop(step, steps, warmup)
    sim.enable_census()
    for k in range(warmup + steps, warmup + steps + 200):
        step(k)
    c = sim.census_counts()
    sub = envs * 5 * 200
    hist = c["wavefront_max_sweeps_histogram"]
    waves = max(sum(hist), 1)
    print(f"law {law} sweeps capped at {cap:2d}: {wall / steps * 1e6:6.2f} us per step; env-substeps sweeping {c['friction_cone'] / sub:.3f}, "
          f"sweeps per sweeping env-substep {c['sweeps_total'] / max(c['friction_cone'], 1):.2f}, wavefront-substep maximum: mean {sum(i * h for i, h in enumerate(hist)) / waves:.2f}, "
          f"wavefront-substeps that swept {waves / (envs // 8 * 5 * 200):.3f}, episodes {int(sim.state[40].sum().item())}", flush=True)
    env.close()

import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import upkie_amd.envs as envs
from upkie_amd import abi
from upkie_amd.model.joint_properties import JointProperties
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization
B=4096
init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
import os
for variant, lanes in (("full", "8"), ("full", "2"), ("plain", "8"), ("plain", "2")):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    kw = dict(num_envs=B, frequency=200.0, init_state=init, autoreset_mode="next_step")
    if variant in ("full", "no_push"): kw["inertia_variation"] = 0.2
    if variant != "plain": kw["joint_properties"] = {n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")}
    env = envs.make("Upkie-HIP-Servos-Vec", **kw)
    env.reset(seed=0)
    if variant in ("full", "no_inertia"):
        push = torch.zeros(B, 3, device="cuda:0"); push[:, 0] = torch.empty(B, device="cuda:0").uniform_(-5, 5)
        env.set_external_forces("torso", push)
    policy = abi.torque_balancing_policy(gain=10.0, fall_pitch=1.0, left_sign=float(env.model.struct.left_sign))
    census = env.sim.enable_census()
    print('mapping', env.sim.lanes_per_env)
    for w in range(4):
        census.zero_()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(300):
            env.sim.step_servos(env.sim.servo_policy(policy))
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/300
        c = env.sim.census_counts(); es = B*5*300; ws = B*8//64*5*300
        print(variant, w, f"{dt*1e6:.1f} us/step lanes {env.sim.lanes_per_env}", {k: (v/es if 'wave' not in k else v/ws) for k,v in c.items()}, "episodes", int(env.sim.state[40].sum()), flush=True)
    # kernel-only timing of the two launches
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    act = env.sim.servo_policy(policy)
    torch.cuda.synchronize(); s.record()
    for _ in range(100): env.sim.servo_policy(policy)
    e.record(); torch.cuda.synchronize(); print(" policy kernel", s.elapsed_time(e)*10, "us")
    s.record()
    for _ in range(100): env.sim.step_servos(act)
    e.record(); torch.cuda.synchronize(); print(" step_servos", s.elapsed_time(e)*10, "us")

"""A/B of builds of the library on ONE box: the bench workload's step (4096 envs, fused agent, one launch per
step) or, with --c5, one GPU's share of BASELINE's C5 (Servos, randomised inertias, pushes, servo policy on the
device: plenty of Gauss-Seidel sweeps); each build in its own process, interleaved, several rounds (box-to-box
differences are larger than the few-percent effects this is for).
With --rollout the same workload as 32 env.step() per launch (upkie_sim_step_pendulum_agent_rollout).
Usage: python tools/ab_step.py libA.so libB.so ... [--rounds N] [--c5] [--fall] [--rollout]"""
import os, subprocess, sys

CHILD_C5 = r'''
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), ".."))
import numpy as np, torch
import upkie_amd.envs as envs
from upkie_amd import abi
from upkie_amd.model.joint_properties import JointProperties
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization
B = 4096
init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, inertia_variation=0.2, init_state=init, autoreset_mode="next_step",
                joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")})
env.reset(seed=0)
torch.manual_seed(0)
push = torch.zeros(B, 3, device="cuda:0"); push[:, 0] = torch.empty(B, device="cuda:0").uniform_(-5, 5)
env.set_external_forces("torso", push)
policy = abi.velocity_balancing_policy(float(env.model.struct.wheel_radius), 1.0, float(env.model.struct.left_sign))
for _ in range(200): env.sim.step_servos(env.sim.servo_policy(policy))
out = []
for rep in range(3):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(600): env.sim.step_servos(env.sim.servo_policy(policy))
    z.record(); torch.cuda.synchronize()
    out.append(a.elapsed_time(z) * 1e3 / 600)
print(" ".join(f"{t:.2f}" for t in out))
'''

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), ".."))
import torch, bench
from upkie_amd.sim import BatchedSim
sim = BatchedSim(bench.make_config(4096)); sim.reset(); sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
for _ in range(int(os.environ.get("AB_WARMUP", "100"))): sim.step_pendulum_agent()
out = []
for rep in range(3):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(400): sim.step_pendulum_agent()
    z.record(); torch.cuda.synchronize()
    out.append(a.elapsed_time(z) * 1e3 / 400)
print(" ".join(f"{t:.2f}" for t in out))
'''

CHILD_ROLLOUT = r'''
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), ".."))
import torch, bench
from upkie_amd.sim import BatchedSim
sim = BatchedSim(bench.make_config(4096)); sim.reset()
prev = torch.zeros(4096, 8, device="cuda:0"); prev[:, :4] = sim.obs6[:, [1, 0, 4, 3]]
ring = torch.zeros(32, 4096, 8, device="cuda:0")
def launch():
    sim.rollout_pendulum_records(prev, ring)
    prev.copy_(ring[-1])
for _ in range(int(os.environ.get("AB_WARMUP", "100")) // 32 + 1): launch()
out = []
for rep in range(3):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): launch()
    z.record(); torch.cuda.synchronize()
    out.append(a.elapsed_time(z) * 1e3 / 640)
print(" ".join(f"{t:.2f}" for t in out))
'''

args = sys.argv[1:]
rounds = 3
if "--rounds" in args:
    i = args.index("--rounds")
    rounds = int(args[i + 1])
    del args[i:i + 2]
c5 = "--c5" in args
if c5:
    args.remove("--c5")
if "--fall" in args:  # the windows of the bench workload in which robots fall (steps 1200-2400)
    args.remove("--fall")
    os.environ["AB_WARMUP"] = "1200"
rollout = "--rollout" in args
if rollout:
    args.remove("--rollout")
libs = args
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, UPKIE_HIP_LIBRARY=os.path.abspath(lib))
        res = subprocess.run([sys.executable, "-c", CHILD_C5 if c5 else (CHILD_ROLLOUT if rollout else CHILD), os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print(f"round {r} {os.path.basename(lib):28s} us/step (three consecutive windows): {res.stdout.strip() or res.stderr[-300:]}", flush=True)

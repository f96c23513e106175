"""Secondary workloads of BASELINE.json (configs[2] and one GPU's share of
configs[4]); prints one JSON line per config. Not the headline bench."""
import json, sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import upkie_amd.envs as envs
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

def rand_state():
    return RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))

def timeit(fn, steps, warmup):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

out = []
# C3: UpkieGroundVelocity/BaseVelocity, 16384 envs, MPC balancer N = 16
B = 16384
env = envs.make("Upkie-HIP-BaseVelocity-Vec", num_envs=B, frequency=200.0, nb_timesteps=16, init_state=rand_state())
env.reset(seed=0)
act = torch.zeros(B, 2, device="cuda:0"); act[:, 0] = torch.empty(B, device="cuda:0").uniform_(-0.5, 0.5)
dt = timeit(lambda: env.step(act), 1000, 100)
out.append(dict(config="C3 UpkieBaseVelocity + MPC N=16 (ADMM 30 it, MFMA)", envs=B, us_per_step=dt * 1e6, env_steps_per_s=B / dt,
                episodes=int(env.sim.state[40].sum()), algorithmic_bytes_per_env_step=554))
# C5 share: UpkieServos 4096 envs, inertia randomisation 0.2, push force, joint friction 0.1
B = 4096
from upkie_amd.model.joint_properties import JointProperties
env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, inertia_variation=0.2, init_state=rand_state(),
                joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")})
obs, _ = env.reset(seed=0)
push = torch.zeros(B, 3, device="cuda:0"); push[:, 0] = torch.empty(B, device="cuda:0").uniform_(-20, 20)
env.set_external_forces("torso", push)
act = env.get_neutral_action(); act[:, [0, 1, 3, 4], 0] = 0.0; act[:, :, 4] = 1.0
def servo_step():
    pitch = env.sim.state[5] * 2.0  # ~ pitch from quaternion y for small angles
    act[:, 2, 2] = (10.0 * pitch).clamp(-1.7, 1.7); act[:, 5, 2] = -(10.0 * pitch).clamp(-1.7, 1.7)
    env.step(act)
dt = timeit(servo_step, 1000, 100)
out.append(dict(config="C5 share: UpkieServos, inertia_variation 0.2, torso push, wheel friction 0.1", envs=B, us_per_step=dt * 1e6,
                env_steps_per_s=B / dt, algorithmic_bytes_per_env_step=630))
for line in out: print(json.dumps(line))

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
env.fuse_mpc = True
act = torch.zeros(B, 2, device="cuda:0"); act[:, 0] = torch.empty(B, device="cuda:0").uniform_(-0.5, 0.5)
dt = timeit(lambda: env.step(act), 1000, 100)
out.append(dict(config="C3 UpkieBaseVelocity + MPC N=16 (ADMM 30 it, MFMA), balancer and step in ONE launch (upkie_sim_step_base_velocity_mpc)", envs=B, us_per_step=dt * 1e6, env_steps_per_s=B / dt,
                episodes=int(env.sim.state[40].sum()), lanes_per_env=env.sim.lanes_per_env, algorithmic_bytes_per_env_step=554))
env.fuse_mpc = False
dt = timeit(lambda: env.step(act), 1000, 100)
out.append(dict(config="C3, two launches (upkie_mpc_step_env + upkie_sim_step_base_velocity)", envs=B, us_per_step=dt * 1e6, env_steps_per_s=B / dt))
# C5 share: UpkieServos 4096 envs (one GPU's share of 32768 over 8), inertia randomisation 0.2, wheel
# friction 0.1, a +-5 N push on the torso per env; a servo-level balancing law written as PyTorch
# ops; fallen robots are reset (as an RL loop does). Captured in a hipGraph (8 steps per launch) so
# that the figure is GPU time, not Python launch overhead; the eager loop is timed next to it.
B = 4096
from upkie_amd import abi
from upkie_amd.graphs import GraphedLoop
from upkie_amd.model.joint_properties import JointProperties
env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, inertia_variation=0.2, init_state=rand_state(), autoreset_mode="disabled",
                joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")})
obs, _ = env.reset(seed=0)
push = torch.zeros(B, 3, device="cuda:0"); push[:, 0] = torch.empty(B, device="cuda:0").uniform_(-5, 5)
env.set_external_forces("torso", push)
act = env.get_neutral_action(); act[:, [0, 1, 3, 4], 0] = 0.0; act[:, :, 4] = 1.0
r = float(env.model.wheel_radius)
fallen = torch.zeros(B, dtype=torch.uint8, device="cuda:0")
def servo_step():
    st = env.sim.state
    pitch = 2.0 * st[abi.S_QUAT + 2]
    pos = 0.5 * (st[abi.S_Q + 2] - st[abi.S_Q + 5]) * r
    v = (10.0 * pitch + pos).clamp(-0.99, 0.99) / r
    act[:, 2, 1] = v; act[:, 5, 1] = -v
    env.sim.step_servos(act)
    torch.gt(pitch.abs(), 1.0, out=fallen.view(torch.bool))
    env.sim.reset(mask=fallen)
dt_eager = timeit(servo_step, 600, 100)
loop = GraphedLoop(servo_step, unroll=8)
dt = timeit(loop.replay, 100, 10) / 8
out.append(dict(config="C5 share: UpkieServos, inertia_variation 0.2, +-5 N torso push, wheel friction 0.1, PyTorch balancing law, fallen robots reset; hipGraph, 8 steps per launch",
                envs=B, us_per_step=dt * 1e6, env_steps_per_s=B / dt, eager_python_us_per_step=dt_eager * 1e6, algorithmic_bytes_per_env_step=630))
# C5 share with the servo-level policy on the device (upkie_sim_servo_policy, fallen robots flagged for the NEXT_STEP
# autoreset): two launches per step, nothing on the host in between; same randomisation and push. Two laws: the README's
# balancer through the wheels' velocity loop (what the PyTorch law above does), and examples/pybullet/torque_balancing.py:15-37
# (wheel torques +-10 x pitch, no velocity feedback: the robots run away, slip and fall -- most substeps go through the
# Gauss-Seidel sweeps, see the census).
for label, make_policy in (("README balancer through the wheel velocity loop", lambda m: abi.velocity_balancing_policy(float(m.wheel_radius), 1.0, float(m.left_sign))),
                           ("examples/pybullet/torque_balancing.py (pitch -> wheel torque)", lambda m: abi.torque_balancing_policy(10.0, 1.0, float(m.left_sign)))):
    env2 = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, inertia_variation=0.2, init_state=rand_state(), autoreset_mode="next_step",
                     joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")})
    env2.reset(seed=0)
    env2.set_external_forces("torso", push)
    policy = make_policy(env2.model.struct)
    def policy_step():
        env2.sim.step_servos_policy(policy)  # the policy inside the step's launch (eight lanes per env)
    def two_launches():
        env2.sim.step_servos(env2.sim.servo_policy(policy))
    dt_two = timeit(two_launches, 1000, 400)
    dt_policy = timeit(policy_step, 2000, 400)
    # the rare-path census afterwards, on its own steps: its atomics (five per wavefront-substep that sweeps) are not free
    census = env2.sim.enable_census()
    census_steps = 400
    timeit(policy_step, census_steps, 0)
    c = env2.sim.census_counts()
    env2.sim.enable_census(False)
    out.append(dict(config=f"C5 share, servo-level policy on the device ({label}): upkie_sim_step_servos_policy (policy evaluated inside the step's launch), NEXT_STEP autoreset of fallen robots, ONE launch per step, Python loop",
                    us_per_step_as_two_launches=dt_two * 1e6,
                    envs=B, us_per_step=dt_policy * 1e6, env_steps_per_s=B / dt_policy, lanes_per_env=env2.sim.lanes_per_env, episodes=int(env2.sim.state[40].sum()),
                    env_substeps_in_gauss_seidel_sweeps=c["friction_cone"] / (B * 5 * census_steps), env_substeps_with_a_joint_at_its_stop=c["joint_limit"] / (B * 5 * census_steps),
                    sweeps_per_infeasible_env_substep=c["sweeps_total"] / max(c["friction_cone"], 1), infeasible_env_substeps_at_the_sweep_cap=c["sweep_cap_hits"] / max(c["friction_cone"], 1),
                    algorithmic_bytes_per_env_step=630 + 2 * 144))
for line in out: print(json.dumps(line))

"""What the reference's OWN env classes send through the Backend boundary.

`UpkiePendulum` / `UpkieGyropod` / `UpkieServos` imported unmodified from
/root/reference/upkie (gymnasium / loop_rate_limiters / upkie_description
stubbed as in tools/make_golden_envs.py) run on `HipBackend` -- the drop-in for
`PyBulletBackend` (upkie/envs/backends/backend.py:11-50) -- here on the fp64
CPU double of the simulation handle, with scripted agent actions. Recorded:
the `RobotState` the reference samples and hands to `Backend.reset`, every
spine action dictionary it hands to `Backend.step`, and the observation /
termination it returns to the agent. tests/test_reference_interop_gpu.py
replays those very calls against `HipBackend` on libupkie_hip.so on the GPU
box, where the reference tree does not exist.

Output: tests/golden/reference_interop.json (committed). Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_interop.py
"""

import importlib.util
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True  # never write into /root/reference
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
JOINTS = ["left_hip", "left_knee", "left_wheel", "right_hip", "right_knee", "right_wheel"]
ACTION_KEYS = ["position", "velocity", "feedforward_torque", "kp_scale", "kd_scale", "maximum_torque"]


def reference_classes():
    spec = importlib.util.spec_from_file_location("make_golden_envs", os.path.join(ROOT, "tools", "make_golden_envs.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    tool.install_stubs()
    import upkie.envs.upkie_gyropod as ref_gyropod
    import upkie.envs.upkie_pendulum as ref_pendulum
    import upkie.envs.upkie_servos as ref_servos
    import upkie.utils.robot_state as ref_state
    import upkie.utils.robot_state_randomization as ref_rand

    return ref_servos.UpkieServos, ref_gyropod.UpkieGyropod, ref_pendulum.UpkiePendulum, ref_state.RobotState, ref_rand.RobotStateRandomization


class Recorder:
    """Backend that forwards to `inner` and keeps what crossed the boundary."""

    def __init__(self, inner):
        self.inner, self.resets, self.steps = inner, [], []

    def reset(self, init_state):
        self.resets.append({
            "position_base_in_world": [float(x) for x in init_state.position_base_in_world],
            "orientation_base_in_world_xyzw": [float(x) for x in init_state.orientation_base_in_world.as_quat()],
            "linear_velocity_base_to_world_in_world": [float(x) for x in init_state.linear_velocity_base_to_world_in_world],
            "angular_velocity_base_in_base": [float(x) for x in init_state.angular_velocity_base_in_base],
            "joint_configuration": [float(x) for x in init_state.joint_configuration],
        })
        return self.inner.reset(init_state)

    def step(self, action):
        servo = action["servo"]
        self.steps.append([[float(servo[j].get(k, {"feedforward_torque": 0.0, "kp_scale": 1.0, "kd_scale": 1.0}.get(k))) for k in ACTION_KEYS] for j in JOINTS])
        return self.inner.step(action)

    def get_spine_observation(self):
        return self.inner.get_spine_observation()

    def close(self):
        self.inner.close()


def run(kind, seed, steps):
    Servos, Gyropod, Pendulum, RobotState, Randomization = reference_classes()
    from tests.fake_sim import oracle_sim_factory
    from upkie_amd.envs.backends import HipBackend

    rand = dict(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0.0, 0.0]))
    init = RobotState(position_base_in_world=np.array([0.0, 0.0, 0.6]), randomization=Randomization(**rand))
    backend = Recorder(HipBackend(dt=1.0 / 200.0, sim_factory=oracle_sim_factory, device="cpu"))
    servos = Servos(backend=backend, frequency=200.0, frequency_checks=False, init_state=init, regulate_frequency=False)
    env = {"pendulum": lambda: Pendulum(servos), "gyropod": lambda: Gyropod(servos)}[kind]()
    obs, _ = env.reset(seed=seed)
    rng = np.random.default_rng(seed)
    actions, observations, terminated = [], [[float(x) for x in obs]], []
    for _ in range(steps):
        # scripted from the observation the reference returned on the CPU double: the README law plus a wobble
        if kind == "pendulum":
            pitch, pos, _, vel = obs
            action = np.array([np.clip(10.0 * pitch + pos + 0.1 * vel, -0.9, 0.9)], dtype=np.float32)
        else:
            action = np.array([np.clip(10.0 * obs[1] + obs[0], -0.9, 0.9), rng.uniform(-0.5, 0.5)], dtype=np.float32)
        obs, _, term, _, _ = env.step(action)
        actions.append([float(x) for x in action])
        observations.append([float(x) for x in obs])
        terminated.append(bool(term))
    env.close()
    return {"kind": kind, "seed": seed, "reset": backend.resets[0], "agent_actions": actions, "spine_actions": backend.steps,
            "observations": observations, "terminated": terminated}


if __name__ == "__main__":
    cases = [run("pendulum", 3, 100), run("gyropod", 11, 100)]
    out = os.path.join(ROOT, "tests", "golden", "reference_interop.json")
    with open(out, "w") as f:
        json.dump({"joints": JOINTS, "action_keys": ACTION_KEYS, "frequency": 200.0, "cases": cases}, f)
    print(out, os.path.getsize(out), "bytes")

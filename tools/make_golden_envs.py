"""Golden vectors from the reference's OWN env wrappers.

The reference package cannot be imported as is here (gymnasium,
loop_rate_limiters and upkie_description are not installed), but those three
are thin dependencies of the wrappers: this script stubs them (a bare
`gymnasium.Env` / `Wrapper` / `spaces.Box` / `spaces.Dict`, a no-op
`RateLimiter`, and `upkie_description.URDF_PATH` pointing at this repository's
synthetic URDF), imports `/root/reference/upkie` unmodified and runs
`UpkieServos`, `UpkieGyropod` and `UpkiePendulum` on a recording backend: the
backend feeds scripted spine observations and records the spine actions the
reference's code sends. The result pins SURVEY section 8 rows a1-a5 (action
maps, clamps, leg low-pass, observation maps, fall detection, yaw
integration) to the reference's actual behaviour.

Output: tests/golden/reference_envs.json (committed). Run in the build
container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_envs.py
"""

import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True  # never write into /root/reference
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
JOINTS = ["left_hip", "left_knee", "left_wheel", "right_hip", "right_knee", "right_wheel"]
ACTION_KEYS = ["position", "velocity", "feedforward_torque", "kp_scale", "kd_scale", "maximum_torque"]
OBS_KEYS = ["position", "velocity", "torque", "temperature", "voltage"]


def install_stubs():
    gym = types.ModuleType("gymnasium")

    class Env:
        metadata = {}

        def reset(self, *, seed=None, options=None):
            self.np_random = np.random.default_rng(seed)

        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)

        @property
        def unwrapped(self):
            return self.env.unwrapped

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
            if shape is not None:
                low, high = np.broadcast_to(low, shape).copy(), np.broadcast_to(high, shape).copy()
            self.low, self.high, self.shape, self.dtype = low, high, low.shape, dtype

    class Dict(dict):
        def __init__(self, spaces=None, **kwargs):
            super().__init__(spaces or {}, **kwargs)
            self.spaces = self

    gym.Env, gym.Wrapper, gym.Space = Env, Wrapper, object
    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box, spaces.Dict = Box, Dict
    gym.spaces = spaces
    envs_mod = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    registration.register = lambda **kwargs: None
    envs_mod.registration = registration
    gym.envs = envs_mod
    sys.modules.update(
        {"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs_mod, "gymnasium.envs.registration": registration}
    )
    limiters = types.ModuleType("loop_rate_limiters")

    class RateLimiter:
        def __init__(self, *args, **kwargs):
            self.slack = 0.0
            self.period = 0.0

        def sleep(self):
            pass

        def remaining(self):
            return 0.0

    limiters.RateLimiter = RateLimiter
    sys.modules["loop_rate_limiters"] = limiters
    description = types.ModuleType("upkie_description")
    description.URDF_PATH = os.path.join(ROOT, "upkie_amd", "model", "upkie_synthetic.urdf")
    sys.modules["upkie_description"] = description
    sys.path.insert(0, REFERENCE)


def spine_observation(rng, pitch=None):
    """A spine observation with arbitrary (seeded) values in every field the
    wrappers read."""
    pitch = float(rng.uniform(-0.6, 0.6)) if pitch is None else float(pitch)
    return {
        "base_orientation": {
            "pitch": pitch,
            "angular_velocity": rng.uniform(-2.0, 2.0, 3).tolist(),
            "linear_velocity": rng.uniform(-1.0, 1.0, 3).tolist(),
        },
        "floor_contact": {"contact": bool(rng.integers(0, 2))},
        "imu": {
            "orientation": [1.0, 0.0, 0.0, 0.0],
            "angular_velocity": rng.uniform(-1.0, 1.0, 3).tolist(),
            "linear_acceleration": rng.uniform(-1.0, 1.0, 3).tolist(),
        },
        "number": 0,
        "servo": {
            joint: {
                "position": float(rng.uniform(-1.0, 1.0)),
                "velocity": float(rng.uniform(-5.0, 5.0)),
                "torque": float(rng.uniform(-3.0, 3.0)),
                "temperature": 42.0,
                "voltage": 18.0,
            }
            for joint in JOINTS
        },
        "wheel_odometry": {"position": float(rng.uniform(-2.0, 2.0)), "velocity": float(rng.uniform(-1.5, 1.5))},
    }


def slim(observation: dict) -> dict:
    """The fields of a spine observation the wrappers read (keeps the fixture small)."""
    return {
        "pitch": observation["base_orientation"]["pitch"],
        "angular_velocity": observation["base_orientation"]["angular_velocity"],
        "wheel_odometry": [observation["wheel_odometry"]["position"], observation["wheel_odometry"]["velocity"]],
        "servo": [[observation["servo"][j][k] for k in ("position", "velocity", "torque")] for j in JOINTS],
    }


def flat_servo_action(action: dict):
    return [[float(action["servo"][j].get(k, float("nan"))) if k in action["servo"][j] else None for k in ACTION_KEYS] for j in JOINTS]


def main():
    install_stubs()
    from upkie.envs import UpkieGyropod, UpkiePendulum, UpkieServos
    from upkie.envs.backends import Backend
    from upkie.model import Model

    class RecordingBackend(Backend):
        def __init__(self, observations):
            self.observations = list(observations)
            self.cursor = 0
            self.actions = []

        def _next(self):
            obs = self.observations[min(self.cursor, len(self.observations) - 1)]
            self.cursor += 1
            return obs

        def reset(self, init_state=None):
            return self._next()

        def step(self, action):
            self.actions.append(action)
            return self._next()

        def get_spine_observation(self):
            return self.observations[min(max(self.cursor - 1, 0), len(self.observations) - 1)]

        def close(self):
            pass

    model = Model()
    golden = {
        "source": "reference UpkieServos / UpkieGyropod / UpkiePendulum run by tools/make_golden_envs.py on a recording backend",
        "model": {
            "wheel_radius": model.wheel_radius,
            "wheel_base": model.wheel_base,
            "left_wheeled": bool(model.left_wheeled),
            "joint_limits": {
                j.name: [j.limit.lower, j.limit.upper, j.limit.velocity, j.limit.effort] for j in model.joints
            },
        },
    }
    rng = np.random.default_rng(2024)
    kw = dict(frequency=200.0, frequency_checks=False, regulate_frequency=False, model=model)

    # ---- UpkieServos: clamps of get_spine_action (upkie_servos.py:316-344), observation map (:288-306)
    n = 24
    observations = [spine_observation(rng) for _ in range(n + 1)]
    backend = RecordingBackend(observations)
    servos = UpkieServos(backend=backend, **kw)
    servos.reset(seed=0)
    cases = []
    for i in range(n):
        action = {}
        for joint in JOINTS:
            wheel = "wheel" in joint
            full = {
                "position": float("nan") if (wheel or rng.uniform() < 0.2) else float(rng.uniform(-4.0, 4.0)),
                "velocity": float(rng.uniform(-150.0, 150.0)),
                "feedforward_torque": float(rng.uniform(-30.0, 30.0)),
                "kp_scale": float(rng.uniform(-1.0, 8.0)),
                "kd_scale": float(rng.uniform(-1.0, 8.0)),
                "maximum_torque": float(rng.uniform(-5.0, 40.0)),
            }
            if i % 3 == 2:  # partial dictionaries: missing keys come from the neutral action (:255-262)
                for key in list(full):
                    if rng.uniform() < 0.4:
                        del full[key]
            action[joint] = full
        obs, reward, terminated, truncated, info = servos.step(action)
        sent = backend.actions[-1]
        cases.append(
            {
                "action": {j: a for j, a in action.items()},
                "spine_servo": [[float(sent["servo"][j][k]) for k in ACTION_KEYS] for j in JOINTS],
                "spine_observation_servo": slim(observations[i + 1])["servo"],
                "observation": [[float(np.asarray(obs[j][k]).reshape(-1)[0]) for k in OBS_KEYS] for j in JOINTS],
                "reward": float(reward),
                "terminated": bool(terminated),
                "truncated": bool(truncated),
            }
        )
    neutral = servos.get_neutral_action()
    golden["servos"] = {
        "cases": cases,
        "neutral_action": {j: {k: float(np.asarray(neutral[j][k]).reshape(-1)[0]) for k in ACTION_KEYS} for j in JOINTS},
        "action_low": {j: {k: float(servos.action_space[j][k].low.reshape(-1)[0]) for k in ACTION_KEYS} for j in JOINTS},
        "action_high": {j: {k: float(servos.action_space[j][k].high.reshape(-1)[0]) for k in ACTION_KEYS} for j in JOINTS},
    }

    # ---- UpkieGyropod: action map, leg low-pass, observation, fall detection, yaw integration
    for name, extra in (("gyropod", {}), ("gyropod_scaled", dict(leg_gain_scale=2.5, max_ground_velocity=1.0, max_yaw_velocity=0.5, fall_pitch=0.4))):
        n = 30
        observations = [spine_observation(rng) for _ in range(n + 1)]
        # the first observation seeds the leg filters with the measured hip / knee angles (upkie_gyropod.py:236-240)
        backend = RecordingBackend(observations)
        servos = UpkieServos(backend=backend, **kw)
        env = UpkieGyropod(servos, **extra)
        obs0, _ = env.reset(seed=1)
        steps = []
        for i in range(n):
            action = np.array([rng.uniform(-4.0, 4.0), rng.uniform(-2.0, 2.0)], dtype=np.float32)
            obs, reward, terminated, truncated, info = env.step(action)
            sent = backend.actions[-1]
            steps.append(
                {
                    "action": [float(action[0]), float(action[1])],
                    "spine_servo": [[float(sent["servo"][j][k]) for k in ACTION_KEYS] for j in JOINTS],
                    "observation": [float(v) for v in obs],
                    "terminated": bool(terminated),
                    "truncated": bool(truncated),
                    "reward": float(reward),
                }
            )
        golden[name] = {
            "kwargs": extra,
            "dt": float(servos.dt),
            "spine_observations": [slim(o) for o in observations],
            "reset_observation": [float(v) for v in obs0],
            "steps": steps,
        }

    # ---- UpkiePendulum: action padding and observation permutation (upkie_pendulum.py:17,104-142)
    n = 12
    observations = [spine_observation(rng) for _ in range(n + 1)]
    backend = RecordingBackend(observations)
    servos = UpkieServos(backend=backend, **kw)
    env = UpkiePendulum(servos, fall_pitch=0.5, max_ground_velocity=2.0)
    obs0, _ = env.reset(seed=2)
    steps = []
    for i in range(n):
        action = np.array([rng.uniform(-3.0, 3.0)], dtype=np.float32)
        obs, reward, terminated, truncated, info = env.step(action)
        sent = backend.actions[-1]
        steps.append(
            {
                "action": [float(action[0])],
                "spine_servo": [[float(sent["servo"][j][k]) for k in ACTION_KEYS] for j in JOINTS],
                "observation": [float(v) for v in obs],
                "terminated": bool(terminated),
            }
        )
    golden["pendulum"] = {
        "kwargs": dict(fall_pitch=0.5, max_ground_velocity=2.0),
        "spine_observations": [slim(o) for o in observations],
        "reset_observation": [float(v) for v in obs0],
        "steps": steps,
    }

    def clean(value):
        if isinstance(value, dict):
            return {k: clean(v) for k, v in value.items()}
        if isinstance(value, (list, tuple)):
            return [clean(v) for v in value]
        if isinstance(value, float) and np.isnan(value):
            return "nan"
        if isinstance(value, float) and np.isinf(value):
            return "inf" if value > 0 else "-inf"
        return value

    out = os.path.join(ROOT, "tests", "golden", "reference_envs.json")
    with open(out, "w") as f:
        json.dump(clean(golden), f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

"""Golden vectors of the reference's OWN initial-state sampler.

`RobotState.sample_state(np_random)` (upkie/utils/robot_state.py:175-196) with
the `RobotStateRandomization.sample_*` methods
(robot_state_randomization.py:135-193) is pure numpy + scipy: both modules are
loaded by file path from /root/reference, unmodified, and driven by the
generator gymnasium's `Env.reset(seed=s)` builds
(`np.random.Generator(np.random.PCG64(np.random.SeedSequence(s)))`, which is
what `np.random.default_rng(s)` returns). Each case records the states of
three consecutive `reset()` calls of one env, i.e. three consecutive draws
from one generator (upkie_env.py:180-190). Pins SURVEY section 8 row a15.

Output: tests/golden/reference_sample_state.json (committed). Build container
only:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_state.py
"""

import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True  # never write bytecode caches into /root/reference
REF = "/root/reference/upkie"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_sample_state.json")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


upkie = types.ModuleType("upkie"); upkie.__path__ = [REF]; sys.modules["upkie"] = upkie
utils = types.ModuleType("upkie.utils"); utils.__path__ = [REF + "/utils"]; sys.modules["upkie.utils"] = utils
randomization = load("upkie.utils.robot_state_randomization", REF + "/utils/robot_state_randomization.py")
robot_state = load("upkie.utils.robot_state", REF + "/utils/robot_state.py")
from scipy.spatial.transform import Rotation as ScipyRotation

CASES = [
    # (seed, init kwargs, randomization kwargs)
    (0, {}, dict(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=[0.05, 0.0, 0.0])),  # bench / SURVEY 8d C2
    (1, {}, dict(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=[0.05, 0.0, 0.0])),
    (42, {}, dict(roll=0.3, pitch=0.2, x=0.1, z=0.05, omega_x=0.4, omega_y=0.5, linear_velocity=[0.3, 0.2, 0.1])),
    (
        7,
        dict(
            position_base_in_world=[0.1, -0.2, 0.7],
            orientation_base_in_world_zyx=[0.4, -0.2, 0.1],
            linear_velocity_base_to_world_in_world=[0.1, 0.0, -0.1],
            angular_velocity_base_in_base=[0.0, 0.3, 0.1],
            joint_configuration=[0.1, -0.2, 0.3, -0.1, 0.2, -0.3],
        ),
        dict(roll=0.05, pitch=0.15, x=0.02, z=0.01, omega_x=0.1, omega_y=0.2, linear_velocity=[0.1, 0.05, 0.02]),
    ),
    (123456789, {}, {}),  # no randomisation at all: the draws are still made
]

golden = {"source": "tools/make_golden_state.py: /root/reference/upkie/utils/robot_state.py + robot_state_randomization.py, unmodified", "cases": []}
for seed, init, rand in CASES:
    kwargs = {k: np.array(v, dtype=float) for k, v in init.items() if k != "orientation_base_in_world_zyx"}
    if "orientation_base_in_world_zyx" in init:
        kwargs["orientation_base_in_world"] = ScipyRotation.from_euler("ZYX", init["orientation_base_in_world_zyx"])
    r = dict(rand)
    if "linear_velocity" in r:
        r["linear_velocity"] = np.array(r["linear_velocity"], dtype=float)
    state = robot_state.RobotState(randomization=randomization.RobotStateRandomization(**r), **kwargs)
    rng = np.random.default_rng(seed)
    draws = []
    for _ in range(3):
        s = state.sample_state(rng)
        x, y, z, w = s.orientation_base_in_world.as_quat()
        draws.append(
            dict(
                angular_velocity_base_in_base=s.angular_velocity_base_in_base.tolist(),
                linear_velocity_base_to_world_in_world=s.linear_velocity_base_to_world_in_world.tolist(),
                orientation_wxyz=[w, x, y, z],
                position_base_in_world=s.position_base_in_world.tolist(),
                joint_configuration=np.asarray(s.joint_configuration, dtype=float).tolist(),
            )
        )
    bx, by, bz, bw = state.orientation_base_in_world.as_quat()
    golden["cases"].append(
        dict(
            seed=seed,
            init=dict(
                position_base_in_world=state.position_base_in_world.tolist(),
                orientation_wxyz=[bw, bx, by, bz],
                linear_velocity_base_to_world_in_world=state.linear_velocity_base_to_world_in_world.tolist(),
                angular_velocity_base_in_base=state.angular_velocity_base_in_base.tolist(),
                joint_configuration=np.asarray(state.joint_configuration, dtype=float).tolist(),
            ),
            randomization=dict(rand),
            draws=draws,
        )
    )

with open(OUT, "w") as f:
    json.dump(golden, f, indent=1)
print("wrote", OUT)

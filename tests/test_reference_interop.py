"""The reference's OWN classes on this repository's side of the boundary
(build container only: skipped where /root/reference is
absent, e.g. on the GPU box).

* `UpkiePendulum(UpkieServos(backend=...))` (its own UpkieGyropod inside) imported unmodified
  from /root/reference/upkie (gymnasium / loop_rate_limiters / upkie_description
  stubbed as in tools/make_golden_envs.py) running on top of
  `upkie_amd.envs.backends.HipBackend` -- the drop-in for `PyBulletBackend`
  (upkie/envs/backends/backend.py:11-50) -- against this repository's fused
  `UpkiePendulum` / `UpkieGyropod` / `UpkieServos`, same seed, same actions.
  Both run on the fp64 CPU double of the simulation handle, so the comparison
  isolates the wrapper arithmetic: the reference's Python (command maps, leg
  low-pass, clamps, observation maps, fall detection, init-state sampling)
  versus the step functions that restate it.
* the reference's `SpineInterface` (upkie/envs/backends/spine/spine_interface.py)
  driving `HipSpine` over real shared memory.
"""

import importlib.util
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "upkie")), reason="the reference tree is not here")


@pytest.fixture(scope="module")
def reference():
    """The reference package, imported unmodified behind the three stubs."""
    spec = importlib.util.spec_from_file_location("make_golden_envs", os.path.join(ROOT, "tools", "make_golden_envs.py"))
    tool = importlib.util.module_from_spec(spec)
    saved_modules, saved_path = dict(sys.modules), list(sys.path)
    sys.dont_write_bytecode = True  # never write into /root/reference
    spec.loader.exec_module(tool)
    tool.install_stubs()
    import upkie.envs.upkie_gyropod as ref_gyropod
    import upkie.envs.upkie_pendulum as ref_pendulum
    import upkie.envs.upkie_servos as ref_servos
    import upkie.utils.robot_state as ref_state
    import upkie.utils.robot_state_randomization as ref_rand

    yield {"servos": ref_servos.UpkieServos, "gyropod": ref_gyropod.UpkieGyropod, "pendulum": ref_pendulum.UpkiePendulum,
           "RobotState": ref_state.RobotState, "RobotStateRandomization": ref_rand.RobotStateRandomization}
    for name in list(sys.modules):
        if name not in saved_modules:
            del sys.modules[name]
    sys.modules.update(saved_modules)
    sys.path[:] = saved_path


def both_envs(reference, kind):
    import upkie_amd.envs as envs
    from upkie_amd.envs.backends import HipBackend
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    from .fake_sim import oracle_sim_factory

    rand = dict(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0.0, 0.0]))
    ref_init = reference["RobotState"](position_base_in_world=np.array([0.0, 0.0, 0.6]), randomization=reference["RobotStateRandomization"](**rand))
    backend = HipBackend(dt=1.0 / 200.0, sim_factory=oracle_sim_factory, device="cpu")
    servos = reference["servos"](backend=backend, frequency=200.0, frequency_checks=False, init_state=ref_init, regulate_frequency=False)
    if kind == "servos":
        ref_env = servos
    elif kind == "gyropod":
        ref_env = reference["gyropod"](servos)
    else:
        ref_env = reference["pendulum"](servos)  # (wraps its own UpkieGyropod, upkie_pendulum.py:80-90)
    init = RobotState(position_base_in_world=np.array([0.0, 0.0, 0.6]), randomization=RobotStateRandomization(**rand))
    name = {"servos": "Upkie-HIP-Servos", "gyropod": "Upkie-HIP-Gyropod", "pendulum": "Upkie-HIP-Pendulum"}[kind]
    mine = envs.make(name, frequency=200.0, init_state=init, sim_factory=oracle_sim_factory)
    return ref_env, mine


@pytest.mark.parametrize("kind", ["pendulum", "gyropod"])
def test_reference_wrappers_on_hip_backend_equal_the_fused_envs(reference, kind):
    ref_env, mine = both_envs(reference, kind)
    for seed in (3, 11):
        obs_r, info_r = ref_env.reset(seed=seed)
        obs_m, info_m = mine.reset(seed=seed)
        np.testing.assert_allclose(obs_m, obs_r, atol=2e-6)
        assert set(info_m["spine_observation"]) >= {"base_orientation", "floor_contact", "imu", "servo", "wheel_odometry"}
        assert set(info_r["spine_observation"]) == set(info_m["spine_observation"])
        rng = np.random.default_rng(seed)
        for step in range(120):
            if kind == "pendulum":
                pitch, pos, _, vel = obs_r
                action = np.array([np.clip(10.0 * pitch + pos + 0.1 * vel, -0.9, 0.9)], dtype=np.float32)
            else:
                action = np.array([np.clip(10.0 * obs_r[1] + obs_r[0], -0.9, 0.9), rng.uniform(-0.5, 0.5)], dtype=np.float32)
            obs_r, rew_r, term_r, trunc_r, info_r = ref_env.step(action)
            obs_m, rew_m, term_m, trunc_m, info_m = mine.step(action)
            # (the reference hands float32 observations to the agent: compare there)
            np.testing.assert_allclose(obs_m, obs_r, atol=5e-6, err_msg=f"step {step}")
            assert rew_m == rew_r == 0.0 and term_m == term_r and trunc_m == trunc_r
            sr, sm = info_r["spine_observation"]["servo"], info_m["spine_observation"]["servo"]
            for joint in sr:
                for key in ("position", "velocity", "torque"):
                    assert sm[joint][key] == pytest.approx(sr[joint][key], abs=2e-5), (step, joint, key)
    ref_env.close()
    mine.close()


def test_reference_servos_on_hip_backend_equal_the_fused_env(reference):
    """Dictionary actions through the reference's UpkieServos (clamps, neutral
    action, missing keys) onto HipBackend.step, against this repository's
    UpkieServos, and the observation dictionaries they return."""
    ref_env, mine = both_envs(reference, "servos")
    obs_r, _ = ref_env.reset(seed=5)
    obs_m, _ = mine.reset(seed=5)
    rng = np.random.default_rng(5)
    for step in range(60):
        action = ref_env.get_neutral_action()
        for joint in action:
            if "wheel" in joint:
                action[joint]["velocity"] = float(rng.uniform(-3, 3))
                action[joint]["kd_scale"] = float(rng.uniform(0.0, 7.0))  # beyond max_gain_scale: clamped
            else:
                action[joint]["position"] = float(rng.uniform(-0.3, 0.3))
                action[joint]["kp_scale"] = float(rng.uniform(0.0, 2.0))
                action[joint]["maximum_torque"] = float(rng.uniform(0.0, 20.0))  # beyond the effort limit: clamped
        obs_r, _, term_r, _, _ = ref_env.step(action)
        obs_m, _, term_m, _, _ = mine.step(action)
        assert term_m == term_r
        for joint in obs_r:
            for key in obs_r[joint]:
                assert float(np.asarray(obs_m[joint][key]).reshape(-1)[0]) == pytest.approx(float(np.asarray(obs_r[joint][key]).reshape(-1)[0]), abs=2e-5), (step, joint, key)
    ref_env.close()
    mine.close()


def test_reference_spine_backend_drives_hip_spine(reference):
    """The reference's SpineBackend / SpineInterface (shared memory opened with
    multiprocessing.shared_memory, msgpack dictionaries, the request word
    protocol and the 100 ms busy-wait of spine_interface.py:66-169) as the agent
    side of a `HipSpine` running in its own process and serving env #1 of a
    batch: start with the reference's own spine configuration and reset state,
    act, stop; the reference's UpkieServos on that backend balances the robot."""
    import upkie.envs.backends.spine_backend as ref_spine_backend
    from upkie.exceptions import UpkieTimeoutError

    # The reference gives the spine 100 ms per request (spine_interface.py:128-145) and busy-waits meanwhile: on a host
    # that is compiling in every core the server process may not be scheduled in time. That is the host's load, not the
    # protocol: the scenario is tried up to three times, each with its own server and shared-memory file.
    for attempt in range(3):
        try:
            _drive_hip_spine_with_the_reference_agent(reference, ref_spine_backend)
            return
        except UpkieTimeoutError:
            if attempt == 2:
                raise


def _drive_hip_spine_with_the_reference_agent(reference, ref_spine_backend):
    import subprocess
    import uuid

    name = f"/upkie_ref_{os.getpid()}_{uuid.uuid4().hex[:8]}"
    server = subprocess.Popen([sys.executable, "-m", "tests.spine_server", name], cwd=ROOT, stdout=subprocess.PIPE, text=True)
    try:
        assert server.stdout.readline().strip() == "ready"
        backend = ref_spine_backend.SpineBackend(shm_name=name)
        init = reference["RobotState"](position_base_in_world=np.array([0.1, 0.0, 0.58]), joint_configuration=np.array([0.1, -0.2, 0.0, 0.1, -0.2, 0.0]))
        servos = reference["servos"](backend=backend, frequency=200.0, frequency_checks=False, init_state=init, regulate_frequency=False)
        obs, info = servos.reset(seed=1)
        spine_obs = info["spine_observation"]
        assert set(spine_obs) >= {"servo", "imu", "base_orientation", "floor_contact", "wheel_odometry", "time"}
        assert spine_obs["time"] == 0.0
        assert float(obs["left_hip"]["position"][0]) == pytest.approx(0.1, abs=2e-3) and float(obs["right_knee"]["position"][0]) == pytest.approx(-0.2, abs=2e-3)
        for k in range(60):
            so = info["spine_observation"]
            ground_velocity = 10.0 * so["base_orientation"]["pitch"] + so["wheel_odometry"]["position"] + 0.1 * so["wheel_odometry"]["velocity"]
            action = servos.get_neutral_action()
            for joint, target in (("left_hip", 0.1), ("left_knee", -0.2), ("right_hip", 0.1), ("right_knee", -0.2)):
                action[joint]["position"] = target
            action["left_wheel"]["velocity"] = ground_velocity / 0.05
            action["right_wheel"]["velocity"] = -ground_velocity / 0.05
            obs, _, terminated, truncated, info = servos.step(action)
            assert not terminated and not truncated
        so = info["spine_observation"]
        assert so["time"] == pytest.approx(60 * 0.005)
        assert abs(so["base_orientation"]["pitch"]) < 0.2 and so["floor_contact"]["contact"] is True
        assert float(obs["left_hip"]["position"][0]) == pytest.approx(0.1, abs=0.05)
        servos.close()  # SpineBackend.close: stop request, shared memory closed on the agent side
    finally:
        server.terminate()
        tail = server.communicate(timeout=30)[0]
    over = tail.strip().splitlines()[-1].split()
    assert over[0] == "over"  # the spine left its loop in order (State.kOver) ...
    # ... env #0 was stepped alongside from its own reset state (a passive robot tipping over near x = 0), env #1 from x = 0.1
    assert abs(float(over[1])) < 0.05 and float(over[2]) == pytest.approx(0.1, abs=0.3) and abs(float(over[2]) - float(over[1])) > 0.02
    assert not os.path.exists(f"/dev/shm{name}")  # unlinked by the spine (AgentInterface.cpp:81-83)

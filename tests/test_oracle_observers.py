"""The observer oracle against the reference's own observer tests
(upkie/cpp/observers/tests/{FloorContactTest,WheelOdometryObserverTest,
BaseOrientationTest}.cpp), restated case by case. This pins the fp64
restatement in oracle/upkie_oracle_observers.c; the GPU parity tests then
compare the HIP pipeline with it."""

import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from upkie_amd import abi

JOINT = {"left_hip": 0, "left_knee": 1, "left_wheel": 2, "right_hip": 3, "right_knee": 4, "right_wheel": 5}


def servo_obs(**joints):
    """[1, 6, 5] servo block from {joint: (velocity, torque)}; joints the
    reference test leaves out read 0 (their KeyError branch adds nothing)."""
    s = np.zeros((1, 6, 5))
    for name, (velocity, torque) in joints.items():
        s[0, JOINT[name], 1] = velocity
        s[0, JOINT[name], 2] = torque
    return s


@pytest.fixture
def floor_contact():
    # FloorContactTest::SetUp, FloorContactTest.cpp:17-35
    dt = 1.0 / 250.0
    cfg = abi.default_observer_config(1, dt)
    cfg.upper_leg_torque_threshold = 10.0
    cfg.liftoff_inertia = 0.001
    cfg.min_touchdown_acceleration = 2.0
    cfg.min_touchdown_torque = 0.015
    cfg.touchdown_inertia = 0.004
    cfg.wheel_cutoff_period = 3 * dt
    return O.ObserverOracle(cfg), dt


def test_no_torque_no_contact(floor_contact):  # FloorContactTest.cpp:67-79
    obs, _ = floor_contact
    out = obs.step(servo_obs(left_wheel=(0.1, 0.0), right_wheel=(0.1, 0.0)))
    assert not out["floor_contact"][0]


def test_big_wheel_accel_torque_means_contact(floor_contact):  # FloorContactTest.cpp:81-108
    obs, dt = floor_contact
    vel = 10.0
    obs.step(servo_obs(left_wheel=(vel, 10.0), right_wheel=(vel, 10.0)))
    new_vel = vel + 5.0 * dt
    out = obs.step(servo_obs(left_wheel=(new_vel, 10.0), right_wheel=(new_vel, 10.0)))
    assert out["floor_contact"][0]
    # joystick button resets wheel observers: would be true without the reset
    out = obs.step(servo_obs(left_wheel=(new_vel, 10.0), right_wheel=(new_vel, 10.0)), cross_button=np.array([1], dtype=np.uint8))
    assert not out["floor_contact"][0]
    assert out["wheel_contact"][0, 0, 2] == 0.0 and out["wheel_contact"][0, 1, 2] == 0.0


def test_small_leg_torque_no_contact(floor_contact):  # FloorContactTest.cpp:110-129
    obs, _ = floor_contact
    out = obs.step(servo_obs(right_hip=(0.0, 1.0), right_knee=(0.0, 1.0)))
    assert not out["floor_contact"][0]
    assert out["upper_leg_torque"][0] == pytest.approx(0.4 * np.sqrt(2.0))


def test_big_leg_torque_means_contact(floor_contact):  # FloorContactTest.cpp:131-151
    obs, _ = floor_contact
    out = obs.step(
        servo_obs(left_hip=(0.0, 100.0), left_knee=(0.0, 100.0), right_hip=(0.0, 100.0), right_knee=(0.0, 100.0))
    )
    assert out["floor_contact"][0]
    # contact through the legs only: "Contact detected, but no wheel in contact?", WheelOdometry.cpp:47-50
    assert out["wheel_odometry"][0, 1] == 0.0


def test_unconfigured_wheel_contact_is_idle():  # WheelContact.cpp:21-24
    cfg = abi.default_observer_config(1, 1.0 / 250.0)
    cfg.wheel_cutoff_period = 0.0
    obs = O.ObserverOracle(cfg)
    for _ in range(5):
        out = obs.step(servo_obs(left_wheel=(10.0, 10.0), right_wheel=(-3.0, 10.0)))
    assert not out["floor_contact"][0]
    assert np.all(obs.state[:10] == 0.0)


def test_filter_error_when_cutoff_below_nyquist():  # low_pass_filter.h:22-30
    cfg = abi.default_observer_config(1, 1.0 / 200.0)  # 0.01 s leg filter <= 2 dt
    with pytest.raises(ValueError):
        O.ObserverOracle(cfg)
    cfg = abi.default_observer_config(1, 1.0 / 1000.0)
    cfg.wheel_cutoff_period = 0.002
    with pytest.raises(ValueError):
        O.ObserverOracle(cfg)
    O.ObserverOracle(abi.default_observer_config(1, 1.0 / 1000.0))


# ---- WheelOdometryObserverTest.cpp ------------------------------------------------


def odometry_oracle():
    # WheelOdometryTest::SetUp, WheelOdometryObserverTest.cpp:17-27
    cfg = abi.default_observer_config(1, 1.0 / 1000.0)
    cfg.signed_radius[0] = +0.50
    cfg.signed_radius[1] = -0.50
    return O.ObserverOracle(cfg)


def force_contacts(obs, left, right, floor):
    obs.state[abi.O_WHEEL + 4, 0] = 1.0 if left else 0.0
    obs.state[abi.O_WHEEL + 9, 0] = 1.0 if right else 0.0


def run_odometry(left_contact, right_contact, floor, vl, vr):
    """The reference test feeds floor_contact flags straight to WheelOdometry;
    here they are produced by wheel estimators latched in contact (inertia kept
    above the lift-off threshold by a torque)."""
    obs = odometry_oracle()
    force_contacts(obs, left_contact, right_contact, floor)
    for w, c in enumerate((left_contact, right_contact)):
        if c:  # abs_torque / (abs_acc + 1e-4) stays >> liftoff_inertia
            obs.state[abi.O_WHEEL + 5 * w + 2, 0] = 10.0
            obs.state[abi.O_WHEEL + 5 * w + 0, 0] = (vl, vr)[w]
    torque = 10.0
    out = obs.step(
        servo_obs(left_wheel=(vl, torque if left_contact else 0.0), right_wheel=(vr, torque if right_contact else 0.0))
    )
    return out


def test_odometry_go_forward():  # WheelOdometryObserverTest.cpp:48-61
    out = run_odometry(True, True, True, 1.0, -1.0)
    assert out["floor_contact"][0]
    assert out["wheel_odometry"][0, 1] == 0.50 * 1.0
    assert out["wheel_odometry"][0, 0] == pytest.approx(0.50 * 1e-3)


def test_odometry_turn_in_place():  # WheelOdometryObserverTest.cpp:63-74
    out = run_odometry(True, True, True, 1.0, 1.0)
    assert out["wheel_odometry"][0, 1] == 0.0


def test_odometry_zero_velocity_when_no_contact():  # WheelOdometryObserverTest.cpp:76-88
    out = run_odometry(False, False, False, 1.0, 1.0)
    assert not out["floor_contact"][0]
    assert out["wheel_odometry"][0, 0] == 0.0 and out["wheel_odometry"][0, 1] == 0.0


def test_odometry_single_wheel_average():  # compute_average_velocity, WheelOdometry.cpp:27-52
    out = run_odometry(True, False, True, 2.0, 5.0)
    assert out["wheel_odometry"][0, 1] == pytest.approx(0.50 * 2.0)


# ---- BaseOrientationTest.cpp ------------------------------------------------------


def test_zero_pitch():  # BaseOrientationTest.cpp:39-46
    phi = 0.42
    R = np.array([[np.cos(phi), -np.sin(phi), 0.0], [np.sin(phi), np.cos(phi), 0.0], [0.0, 0.0, 1.0]])
    assert O.pitch_frame_in_parent(R) == 0.0


def test_close_to_zero():  # BaseOrientationTest.cpp:49-56
    theta = 1e-3
    R = np.array([[np.cos(theta), 0.0, np.sin(theta)], [0.0, 1.0, 0.0], [-np.sin(theta), 0.0, np.cos(theta)]])
    assert O.pitch_frame_in_parent(R) == pytest.approx(theta, abs=1e-6)


def test_orientation_not_neatly_normalized():  # BaseOrientationTest.cpp:60-68
    theta = 1e-3
    R = np.array([[np.cos(theta), 0.0, np.sin(theta)], [0.0, 1.0, 0.0], [-np.sin(theta), 0.0, np.cos(theta)]])
    R[:, 0] *= 1.0 - 1e-2
    assert O.pitch_frame_in_parent(R) == pytest.approx(theta, abs=1e-6)


QUAT_IMU_IN_ARS = [0.008472769239730098, -0.9953038144146671, -0.09639792825405252, -0.002443076206500708]


def test_base_pitch_from_imu():  # BaseOrientationTest.cpp:70-86
    base_to_imu = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    ars_to_world = np.diag([1.0, -1.0, -1.0])
    R = O.base_orientation_from_imu(QUAT_IMU_IN_ARS, base_to_imu, ars_to_world)
    assert O.pitch_frame_in_parent(R) == pytest.approx(-0.016, abs=1e-3)


def test_neutral_values():  # BaseOrientationTest.cpp:95-123
    obs = O.ObserverOracle(abi.default_observer_config(1, 1e-3))
    out = obs.step(np.zeros((1, 6, 5)), imu_orientation=[[1.0, 0.0, 0.0, 0.0]], imu_angular_velocity=[[0.0, 0.0, 0.0]])
    assert out["base_pitch"][0] == 0.0
    assert np.all(out["base_angular_velocity"] == 0.0)
    expected = np.diag([1.0, -1.0, -1.0]) @ np.eye(3) @ np.diag([-1.0, 1.0, -1.0])
    assert np.allclose(out["rotation_base_to_world"].reshape(3, 3), expected)


def test_angular_velocity_is_rotated_to_base():  # BaseOrientation.h:144-148
    obs = O.ObserverOracle(abi.default_observer_config(1, 1e-3))
    out = obs.step(np.zeros((1, 6, 5)), imu_orientation=[[1.0, 0.0, 0.0, 0.0]], imu_angular_velocity=[[0.1, -0.2, 0.3]])
    assert np.allclose(out["base_angular_velocity"][0], [-0.1, -0.2, -0.3])


def test_touchdown_and_liftoff_hysteresis():
    """WheelContact.cpp:36-47 driven through a touchdown then a lift-off, against
    an independent plain-Python transcription of the same recursion."""
    dt = 1e-3
    cfg = abi.default_observer_config(1, dt)
    obs = O.ObserverOracle(cfg)
    rng = np.random.default_rng(1)
    v = a = t = inertia = 0.0
    contact = False
    alpha = dt / 0.2
    saw = set()
    for k in range(3000):
        loaded = 500 <= k < 1800
        torque = (1.0 + 0.2 * rng.standard_normal()) if loaded else 0.0
        velocity = 5.0 * np.sin(0.02 * k) * (0.3 if loaded else 3.0)
        prev = v
        v = v + alpha * (velocity - v)
        acc = (v - prev) / dt
        a = a + alpha * (abs(acc) - a)
        t = t + alpha * (abs(torque) - t)
        if contact or not (a < 2.0 or t < 0.015):
            inertia = t / (a + 1e-4)
            if inertia < 0.001:
                contact = False
            elif inertia > 0.004:
                contact = True
        out = obs.step(servo_obs(left_wheel=(velocity, torque)))
        assert bool(out["wheel_contact"][0, 0, 2]) == contact
        assert out["wheel_contact"][0, 0, 3] == pytest.approx(inertia, rel=1e-12, abs=1e-15)
        saw.add(contact)
    assert saw == {False, True}

"""The non-finite guard (include/upkie_hip.h, "Non-finite commands and states")
as the fp64 checker states it: what the device is held to in
tests/test_nonfinite_guard_gpu.py. The reference itself asserts on a NaN
velocity target (pybullet_backend.py:519): one robot, one process; a batch
replaces the word by the neutral action's (upkie_servos.py:255-262) and goes on."""

import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from upkie_amd import abi
from upkie_amd.model.default_model import default_model


def guard_counts(reset=False):
    counts = (C.c_int64 * 2).in_dll(O.lib(), "oracle_guard_counts")
    out = (int(counts[0]), int(counts[1]))
    if reset:
        counts[0] = counts[1] = 0
    return out


def make(B=8, autoreset=abi.AUTORESET_NEXT_STEP):
    cfg = abi.default_sim_config(B, frequency=200.0, seed=3)
    cfg.rand_pitch = 0.05
    cfg.autoreset_mode = autoreset
    model = default_model()
    ref = O.Oracle(model, cfg)
    ref.reset()
    return ref, model, cfg


def neutral(model, B):
    act = np.zeros((B, 6, 6))
    act[:, :, 0] = np.nan
    act[:, :, 3] = 1.0
    act[:, :, 4] = 1.0
    act[:, :, 5] = np.array(model.joint_effort[:])[None, :]
    return act


def test_nonfinite_servo_words_are_the_neutral_actions():
    ref, model, _ = make()
    twin, _, _ = make()
    guard_counts(reset=True)
    B = ref.B
    act = neutral(model, B)
    poisoned = act.copy()
    poisoned[0, 2, 1] = np.nan  # velocity of the left wheel
    poisoned[1, 5, 2] = np.nan  # feedforward torque
    poisoned[2, 0, 3] = np.nan  # kp_scale
    poisoned[3, 1, 4] = np.inf  # kd_scale: clamped to max_gain_scale by the reference's clamp, not replaced
    poisoned[4, 4, 5] = np.nan  # maximum torque
    poisoned[5, 2, 0] = np.inf  # position of a joint without position limits
    poisoned[6, 0, 0] = -np.inf  # position of a bounded joint: clamped to its stop
    expected = act.copy()
    expected[3, 1, 4] = 1e30
    expected[6, 0, 0] = -1e30
    for _ in range(5):
        obs_p, _, term_p, _ = ref.step_servos(poisoned)
        obs_e, _, term_e, _ = twin.step_servos(expected)
        assert np.isfinite(obs_p).all() and not term_p.any()
        np.testing.assert_array_equal(obs_p, obs_e)
    np.testing.assert_array_equal(ref.state, twin.state)
    assert guard_counts() == (5 * 5, 0)


def test_nan_ground_velocity_is_zero_and_an_infinite_yaw_velocity_is_the_limit():
    ref, _, cfg = make()
    twin, _, _ = make()
    guard_counts(reset=True)
    act = np.zeros((ref.B, 2))
    act[:, 0] = 0.1
    clean = act.copy()
    act[0, 0] = np.nan
    clean[0, 0] = 0.0
    act[1, 1] = np.nan
    act[2, 1] = np.inf
    clean[2, 1] = cfg.max_yaw_velocity
    act[3, 0] = -np.inf
    clean[3, 0] = -1e30
    for _ in range(3):
        obs_p, _, _, _ = ref.step_gyropod(act)
        obs_c, _, _, _ = twin.step_gyropod(clean)
        assert np.isfinite(obs_p).all()
        np.testing.assert_array_equal(obs_p, obs_c)
    assert guard_counts() == (3 * 3, 0)


@pytest.mark.parametrize("kind", ["servos", "gyropod"])
def test_a_state_that_is_not_finite_ends_the_episode_in_the_initial_state(kind):
    ref, model, cfg = make()
    guard_counts(reset=True)
    B = ref.B
    ref.state[abi.S_LINVEL + 1, 2] = np.nan
    ref.state[abi.S_QD + 4, 5] = np.inf
    before = ref.state.copy()
    if kind == "servos":
        obs, _, term, _ = ref.step_servos(neutral(model, B))
    else:
        obs, _, term, _ = ref.step_gyropod(np.zeros((B, 2)))
    assert np.isfinite(obs).all() and np.isfinite(ref.state).all()
    np.testing.assert_array_equal(term, np.isin(np.arange(B), (2, 5)).astype(np.uint8))
    for e in (2, 5):
        np.testing.assert_array_equal(ref.state[abi.S_POS:abi.S_POS + 3, e], cfg.init_pos[:])
        np.testing.assert_array_equal(ref.state[abi.S_QUAT:abi.S_QUAT + 4, e], cfg.init_quat[:])
        assert (ref.state[abi.S_QD:abi.S_QD + 6, e] == 0).all() and (ref.state[abi.S_TORQUE:abi.S_TORQUE + 6, e] == 0).all()
        assert ref.state[abi.S_DONE, e] == 1.0
    assert guard_counts() == (0, 2)
    # NEXT_STEP autoreset: the next step re-initialises them like fallen robots
    episodes = before[abi.S_EPISODE].copy()
    if kind == "servos":
        obs, _, term, _ = ref.step_servos(neutral(model, B))
    else:
        obs, _, term, _ = ref.step_gyropod(np.zeros((B, 2)))
    assert not term.any() and np.isfinite(obs).all()
    assert (ref.state[abi.S_EPISODE, [2, 5]] == episodes[[2, 5]] + 1).all() and (ref.state[abi.S_DONE] == 0).all()

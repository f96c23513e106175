"""Pin the oracle on golden vectors generated from the reference's own helper
modules (tools/make_golden.py -> tests/golden/reference_helpers.json) and on
the reference's known-answer tests, restated 1:1."""

import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _vec(n):
    return (C.c_double * n)()


def test_quaternion_to_matrix(golden):
    lib = O.lib()
    for q, R in zip(golden["quat_wxyz"], golden["matrix_from_quat"]):
        out = _vec(9)
        lib.oracle_quat_to_matrix((C.c_double * 4)(*q), out)
        np.testing.assert_allclose(np.array(out).reshape(3, 3), R, atol=1e-15)


def test_matrix_to_quaternion_scipy_convention(golden):
    lib = O.lib()
    for R, q in zip(golden["matrix_from_quat"], golden["quat_from_matrix"]):
        out = _vec(4)
        lib.oracle_matrix_to_quat((C.c_double * 9)(*np.array(R).ravel()), out)
        np.testing.assert_allclose(list(out), q, atol=1e-14)  # same sign choice too


def test_euler_zyx_composition(golden):
    lib = O.lib()
    bx, by, bz, bw = golden["base_quat_xyzw"]
    for ypr, q_xyzw in zip(golden["euler_zyx"], golden["composed_quat_xyzw"]):
        out = _vec(4)
        lib.oracle_euler_zyx_compose((C.c_double * 4)(bw, bx, by, bz), (C.c_double * 3)(*ypr), out)
        expected = [q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]]
        got = np.array(out)
        if np.dot(got, expected) < 0:  # q and -q are the same rotation
            got = -got
        np.testing.assert_allclose(got, expected, atol=1e-14)


def test_low_pass_filter(golden):
    lib = O.lib()
    lib.oracle_low_pass_filter.restype = C.c_double
    lib.oracle_low_pass_filter.argtypes = [C.c_double] * 4
    x = 0.37
    for expected in golden["low_pass_leg"]:
        x = lib.oracle_low_pass_filter(x, 1.0, 0.0, 0.005)
        assert x == pytest.approx(expected, abs=1e-16)
    x = 1.3
    for expected in golden["low_pass_mpc"]:
        x = lib.oracle_low_pass_filter(x, 0.1, 0.0, 0.005)
        assert x == pytest.approx(expected, abs=1e-16)


def test_clamp_and_warn_semantics(golden):
    lib = O.lib()
    lib.oracle_clamp.restype = C.c_double
    lib.oracle_clamp.argtypes = [C.c_double] * 3
    for v, expected in zip(golden["clamp_inputs"], golden["clamp_outputs"]):
        out = lib.oracle_clamp(v, -1.0, 1.0)
        if math.isnan(expected):
            assert math.isnan(out)  # NaN passes through, clamp.py:52-58
        else:
            assert out == expected


# --- tests/envs/backends/test_pybullet_backend_mock.py, restated ------------
def test_pd_law_known_answer():
    """:157-173: 1.0 + 1*(0 - 0.005) + 20*(0 - 0.1) = -1.005"""
    tau = O.joint_torque(
        0.1, 0.005,
        dict(position=0.0, velocity=0.0, feedforward_torque=1.0, kp_scale=1.0, kd_scale=1.0, maximum_torque=10.0),
    )
    assert tau == pytest.approx(-1.005, abs=1e-12)


def test_pd_law_friction_known_answers():
    """:205-247: friction 0.1 with velocity +-0.005 (> stiction) -> -+0.105"""
    cmd = dict(position=float("nan"), velocity=0.0, feedforward_torque=0.0, kp_scale=1.0, kd_scale=1.0, maximum_torque=10.0)
    assert O.joint_torque(0.0, 0.005, cmd, friction=0.1) == pytest.approx(-0.105, abs=1e-12)
    assert O.joint_torque(0.0, -0.005, cmd, friction=0.1) == pytest.approx(+0.105, abs=1e-12)


def test_pd_law_stiction_threshold_is_strict():
    """:271-313: |velocity| must be strictly above 1e-3 for kinetic friction"""
    cmd = dict(position=float("nan"), velocity=0.0, feedforward_torque=0.0, kp_scale=1.0, kd_scale=1.0, maximum_torque=10.0)
    at = O.joint_torque(0.0, 1e-3, cmd, friction=0.1)
    above = O.joint_torque(0.0, 1.1e-3, cmd, friction=0.1)
    assert at == pytest.approx(-1e-3, abs=1e-12)  # damping only
    assert above == pytest.approx(-1.1e-3 - 0.1, abs=1e-12)


def test_pd_law_full_step_known_answer():
    """:561-619: q=0.1, qd=0.05, q*=0.2, qd*=0.1 -> tau = 2.05"""
    tau = O.joint_torque(
        0.1, 0.05,
        dict(position=0.2, velocity=0.1, feedforward_torque=0.0, kp_scale=1.0, kd_scale=1.0, maximum_torque=10.0),
    )
    assert tau == pytest.approx(2.05, abs=1e-12)


def test_pd_law_clips_to_maximum_torque():
    cmd = dict(position=1.0, velocity=0.0, feedforward_torque=0.0, kp_scale=1.0, kd_scale=1.0, maximum_torque=1.7)
    assert O.joint_torque(0.0, 0.0, cmd) == 1.7
    cmd["position"] = -1.0
    assert O.joint_torque(0.0, 0.0, cmd) == -1.7


# --- upkie/cpp/observers/tests/BaseOrientationTest.cpp:40-87 ---------------
def test_pitch_from_quaternion_known_answers():
    lib = O.lib()
    lib.oracle_pitch_from_quat.restype = C.c_double
    for angle in (-1.2, -0.3, 0.0, 0.42, 1.0):
        q = (C.c_double * 4)(math.cos(angle / 2), 0.0, math.sin(angle / 2), 0.0)
        assert lib.oracle_pitch_from_quat(q) == pytest.approx(angle, abs=1e-12)
    # yaw does not change the pitch
    yaw = 0.7
    qz = np.array([math.cos(yaw / 2), 0, 0, math.sin(yaw / 2)])
    qy = np.array([math.cos(0.2), 0, math.sin(0.2), 0])
    w = qz[0] * qy[0] - qz[3] * qy[3] * 0
    q = np.array([qz[0] * qy[0], -qz[3] * qy[2], qz[0] * qy[2], qz[3] * qy[0]])
    assert lib.oracle_pitch_from_quat((C.c_double * 4)(*q)) == pytest.approx(0.4, abs=1e-12)


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox4x32-10."""
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_the_oracle_sizes_its_thread_team_to_the_cores_this_process_may_use(monkeypatch):
    """A GPU box reports 256 hardware threads under a 16-core quota: the
    default OpenMP team would be 256 threads taking turns on 16 cores."""
    import ctypes

    omp = ctypes.CDLL("libgomp.so.1")
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    cores = O.usable_cores()
    assert 1 <= cores <= (os.cpu_count() or 1)
    assert O.set_threads() == cores and omp.omp_get_max_threads() == cores
    assert O.set_threads(1) == 1 and omp.omp_get_max_threads() == 1
    monkeypatch.setenv("OMP_NUM_THREADS", "2")
    assert O.set_threads() == 2 and omp.omp_get_max_threads() == 2
    monkeypatch.delenv("OMP_NUM_THREADS")
    O.set_threads()

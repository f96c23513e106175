"""Host-side helpers with the reference's names (upkie/utils/*) against the
reference's own unit tests, restated."""

import numpy as np


def test_point_contact_like_the_reference():
    """tests/utils/test_point_contact.py:15-43: fields and repr."""
    from upkie_amd.utils.point_contact import PointContact, point_contacts

    c = PointContact(link_name="left_wheel_link", position_contact_in_world=np.array([1.0, 2.0, 3.0]), force_in_world=np.array([0.0, 0.0, 10.0]))
    assert c.link_name == "left_wheel_link"
    np.testing.assert_array_equal(c.position_contact_in_world, [1.0, 2.0, 3.0])
    np.testing.assert_array_equal(c.force_in_world, [0.0, 0.0, 10.0])
    r = repr(PointContact(link_name="imu", position_contact_in_world=np.array([0.0, 0.0, 0.1]), force_in_world=np.array([0.0, 0.0, -50.0])))
    assert "PointContact" in r and "link_name='imu'" in r
    assert "position_contact_in_world=[0.0, 0.0, 0.1]" in r and "force_in_world=[0.0, 0.0, -50.0]" in r
    # rows of BatchedSim.contact_points() -> the reference's list (pybullet_backend.py:660-716)
    rows = np.array([[1.0, 0.1, 0.15, 0.0, 1.0, 0.0, 26.0, 0.0], [0.0] * 8])
    both = point_contacts(rows)
    assert len(both) == 1 and both[0].link_name == "left_wheel_tire" and both[0].force_in_world[2] == 26.0
    assert point_contacts(rows, "right_wheel_tire") == [] and point_contacts(rows, "nope") == []


# --- tests/utils/test_external_force.py (19 cases), restated ------------------
import pytest  # noqa: E402


def test_external_force_holds_a_copy_of_a_3d_vector():
    """:15-34, :83-117: list or array in, float array out, world frame by
    default, the caller's list is not aliased."""
    from upkie_amd.utils.external_force import ExternalForce

    for values in ([1.0, 2.0, 3.0], [0.0, 0.0, 1.0], [-10.0, -5.0, -1.0], [0.0, 0.0, 0.0], [1000.0, 2000.0, 3000.0], [1.23456789, 2.34567891, 3.45678912]):
        f = ExternalForce(values)
        assert isinstance(f.force, np.ndarray) and np.array_equal(f.force, np.array(values)) and f.local is False
    given = np.array([4.0, 5.0, 6.0])
    f = ExternalForce(given, local=True)
    assert np.array_equal(f.force, given) and f.local is True
    original = [1.0, 2.0, 3.0]
    f = ExternalForce(original)
    original[0] = 999.0
    assert f.force[0] == 1.0


@pytest.mark.parametrize("bad, shape", [([1.0, 2.0], "(2,)"), ([1.0, 2.0, 3.0, 4.0], "(4,)"), ([], "(0,)"), (np.array([[1.0, 2.0], [3.0, 4.0]]), "(2, 2)"), (5.0, "()")])
def test_external_force_refuses_anything_but_a_3d_vector(bad, shape):
    """:36-68: ValueError naming the offending shape."""
    from upkie_amd.utils.external_force import ExternalForce

    with pytest.raises(ValueError) as raised:
        ExternalForce(bad)
    assert "Force must be a 3D vector" in str(raised.value) and shape in str(raised.value)


def test_external_force_repr_and_frames():
    """:119-171: repr names the class and both fields; frames compare as given."""
    from upkie_amd.utils.external_force import ExternalForce

    r = repr(ExternalForce([1.0, 2.0, 3.0], local=True))
    assert "ExternalForce" in r and "force=[1.0, 2.0, 3.0]" in r and "local=True" in r
    r = repr(ExternalForce([4.0, 5.0, 6.0]))
    assert "force=[4.0, 5.0, 6.0]" in r and "local=False" in r
    a, b, c, d = ExternalForce([1.0, 2.0, 3.0], local=True), ExternalForce([1.0, 2.0, 3.0], local=True), ExternalForce([1.0, 2.0, 3.0]), ExternalForce([2.0, 2.0, 3.0], local=True)
    assert a.local == b.local and np.array_equal(a.force, b.force)
    assert a.local != c.local and not np.array_equal(a.force, d.force)


# --- tests/utils/test_clamp.py, restated -------------------------------------
def test_clamp_family():
    """:21-40: both bounds, absolute bound, the warning variants, infinite bounds."""
    from upkie_amd.utils.clamp import clamp, clamp_abs, clamp_abs_and_warn, clamp_and_warn

    assert clamp(-2.1, -2.0, 2.0) == -2.0 and clamp(3.1, -2.0, 2.0) == 2.0
    assert clamp_abs(-2.1, 2.0) == -2.0 and clamp_abs(3.1, 2.0) == 2.0
    assert clamp_and_warn(-2.1, -2.0, 2.0, "x") == -2.0 and clamp_and_warn(3.1, -2.0, 2.0, "x") == 2.0
    assert clamp_abs_and_warn(-2.1, 2.0, "y") == -2.0 and clamp_abs_and_warn(3.1, 2.0, "y") == 2.0
    assert clamp(1.42, -np.inf, np.inf) == 1.42 and clamp_abs(1.42, np.inf) == 1.42


# --- tests/utils/test_rotations.py, restated -----------------------------------
@pytest.mark.parametrize(
    "rpy, expected",
    [
        ((0.0, 0.0, 0.0), np.eye(3)),
        ((np.pi, 0.0, 0.0), np.diag([1.0, -1.0, -1.0])),
        ((0.0, np.pi, 0.0), np.diag([-1.0, 1.0, -1.0])),
        ((0.0, 0.0, np.pi), np.diag([-1.0, -1.0, 1.0])),
        ((np.pi, 0.0, np.pi), np.diag([-1.0, 1.0, -1.0])),  # Upkie's IMU placement
        ((0.0, 0.0, np.pi / 2), np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])),
    ],
)
def test_rotation_matrix_from_rpy_known_answers(rpy, expected):
    """:15-71."""
    from upkie_amd.utils.rotations import rotation_matrix_from_rpy

    np.testing.assert_allclose(rotation_matrix_from_rpy(rpy), expected, atol=1e-10)


def test_rotation_matrix_from_rpy_is_a_proper_rotation():
    """:73-78, plus the quaternion helpers the observation path uses
    (rotations.py:14-71): round trip through the rotation matrix."""
    from upkie_amd.utils.rotations import quaternion_from_rotation_matrix, rotation_matrix_from_quaternion, rotation_matrix_from_rpy

    R = rotation_matrix_from_rpy((0.3, -0.7, 1.2))
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-10)
    assert abs(np.linalg.det(R) - 1.0) < 1e-10
    q = quaternion_from_rotation_matrix(R)
    np.testing.assert_allclose(rotation_matrix_from_quaternion(q), R, atol=1e-12)


# --- tests/utils/test_robot_state.py, restated --------------------------------
def test_robot_state_without_randomization_samples_the_identity():
    """:15-21: zero roll / pitch randomisation leaves the orientation alone."""
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    state = RobotState(randomization=RobotStateRandomization(roll=0.0, pitch=0.0))
    zyx = state.sample_orientation(np.random).as_euler("ZYX")
    assert np.allclose(zyx, np.zeros(3))
    # the other three samplers of the reference's class (robot_state.py:109-173): offsets around the state's own values
    rng = np.random.default_rng(5)
    moved = RobotState(
        position_base_in_world=np.array([1.0, 2.0, 0.6]),
        linear_velocity_base_to_world_in_world=np.array([0.1, 0.0, 0.0]),
        angular_velocity_base_in_base=np.array([0.0, 0.2, 0.0]),
        randomization=RobotStateRandomization(x=0.05, z=0.02, omega_x=0.1, omega_y=0.1, linear_velocity=[0.3, 0.0, 0.1]),
    )
    p, v, w = moved.sample_position(rng), moved.sample_linear_velocity(rng), moved.sample_angular_velocity(rng)
    assert abs(p[0] - 1.0) <= 0.05 and p[1] == 2.0 and 0.6 <= p[2] <= 0.62
    assert abs(v[0] - 0.1) <= 0.3 and v[1] == 0.0 and abs(v[2]) <= 0.1
    assert abs(w[0]) <= 0.1 and abs(w[1] - 0.2) <= 0.1 and w[2] == 0.0

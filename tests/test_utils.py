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

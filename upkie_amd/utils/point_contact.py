"""Contact point as PyBulletBackend.get_contact_points reports it
(upkie/utils/point_contact.py:10-55)."""

import numpy as np


class PointContact:
    """Contact information from the robot's perspective.

    Attributes:
        force_in_world: Total contact force (normal + friction) exerted by the
            floor on the link, in the world frame, in N.
        link_name: Name of the robot link in contact.
        position_contact_in_world: Position of the contact point in the world
            frame.
    """

    force_in_world: np.ndarray
    link_name: str
    position_contact_in_world: np.ndarray

    def __init__(self, link_name: str, position_contact_in_world: np.ndarray, force_in_world: np.ndarray):
        self.force_in_world = force_in_world
        self.link_name = link_name
        self.position_contact_in_world = position_contact_in_world

    def __repr__(self) -> str:
        return (
            f"PointContact(link_name='{self.link_name}', "
            f"position_contact_in_world={self.position_contact_in_world.tolist()}, "
            f"force_in_world={self.force_in_world.tolist()})"
        )


#: links that carry the floor contact geometry, in the order of the rows of
#: `BatchedSim.contact_points()` (the tires are fused into the wheel bodies)
CONTACT_LINKS = ("left_wheel_tire", "right_wheel_tire")


def point_contacts(points: np.ndarray, link_name=None) -> list:
    """List of `PointContact` from one env's ``[2, 8]`` rows of
    `BatchedSim.contact_points()`, filtered like
    PyBulletBackend.get_contact_points(link_name) (pybullet_backend.py:672-682:
    an unknown link gives an empty list)."""
    points = np.asarray(points, dtype=np.float64)
    result = []
    for tire, name in enumerate(CONTACT_LINKS):
        if link_name is not None and name != link_name:
            continue
        if points[tire, 0] == 0.0:
            continue
        result.append(
            PointContact(
                link_name=name,
                position_contact_in_world=points[tire, 1:4].copy(),
                force_in_world=points[tire, 4:7].copy(),
            )
        )
    return result

"""Robot model wrapper with the attributes of the reference's
``upkie.model.Model`` (upkie/model/model.py:19-110) plus the merged rigid-body
structure the kernels consume."""

import os
from typing import List, Optional

import numpy as np

from ..abi import UpkieModel
from .urdf import UrdfTree, load_urdf_model

_SYNTHETIC_URDF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "upkie_synthetic.urdf")


class JointLimit:
    """upkie/model/joint_limit.py"""

    def __init__(self, lower: float, upper: float, effort: float, velocity: float):
        self.lower, self.upper, self.effort, self.velocity = lower, upper, effort, velocity


class Joint:
    """upkie/model/joint.py: actuated joint with its URDF limits."""

    def __init__(self, index: int, idx_q: int, idx_v: int, name: str, limit: JointLimit):
        self.index, self.idx_q, self.idx_v, self.name, self.limit = index, idx_q, idx_v, name, limit


def default_urdf_path() -> str:
    """upkie_description's URDF when that package is installed (what the
    reference uses, model.py:70-72), else the synthetic URDF shipped here."""
    try:
        import upkie_description  # type: ignore

        return upkie_description.URDF_PATH
    except ImportError:
        return _SYNTHETIC_URDF


class Model:
    def __init__(self, urdf_path: Optional[str] = None):
        self.urdf_path = urdf_path if urdf_path is not None else default_urdf_path()
        ## Merged rigid-body structure handed to the HIP library.
        self.struct: UpkieModel = load_urdf_model(self.urdf_path)
        tree = UrdfTree(self.urdf_path)
        self.joints: List[Joint] = []
        for idx, j in enumerate(tree.actuated_joints):  # URDF order, kinematic_tree.py:105-127
            self.joints.append(
                Joint(idx + 1, idx, idx, j.name, JointLimit(j.limit["lower"], j.limit["upper"], j.limit["effort"], j.limit["velocity"]))
            )
        self.joint_names = {joint.name for joint in self.joints}
        self.upper_leg_joints = tuple(j for j in self.joints if "hip" in j.name or "knee" in j.name)
        self.wheel_joints = tuple(j for j in self.joints if "wheel" in j.name)
        self.wheel_base = float(self.struct.wheel_base)
        self.wheel_radius = float(self.struct.wheel_radius)
        self.left_wheeled = bool(self.struct.left_sign > 0)
        self.rotation_ars_to_world = np.diag([1.0, -1.0, -1.0])
        self.rotation_base_to_imu = np.array(self.struct.rot_base_to_imu[:]).reshape(3, 3)
        self.link_names = set(tree.links)
        self._tree = tree

    def link_position_in_base(self, link_name: str) -> np.ndarray:
        """Origin of a link frame in the base frame at the zero configuration."""
        return self._tree.p[link_name].copy()

    def link_attachment(self, link_name: str):
        """Where an external force on `link_name` acts: (composite body index,
        the link's centre of mass in that body's frame, rotation from the link
        frame to the body frame). Bullet applies forces at the link's inertial
        frame: `getLinkState(...)[0]` / `getBasePositionAndOrientation` are
        centre-of-mass positions and a LINK_FRAME position of [0, 0, 0] is the
        centre of mass (pybullet_backend.py:636-658)."""
        tree = self._tree
        body = self.body_of_link(link_name)
        root = tree.body_root_of(link_name)
        origin = np.zeros(3) if body == 0 else tree.p[root]
        com = tree.p[link_name] + tree.R[link_name] @ tree.links[link_name].com
        return body, com - origin, tree.R[link_name].copy()

    def body_of_link(self, link_name: str) -> int:
        """Index of the composite body a link is rigidly part of."""
        order = {"left_hip": 1, "left_knee": 2, "left_wheel": 3, "right_hip": 4, "right_knee": 5, "right_wheel": 6}
        root = self._tree.body_root_of(link_name)
        if root == self._tree.root:
            return 0
        return order[self._tree.parent_joint[root].name]

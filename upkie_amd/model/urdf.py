"""URDF -> merged rigid-body model.

The reference hands its URDF to Bullet (pybullet_backend.py:121-125), which
keeps every fixed-joint link as a separate body, and separately parses joint
limits and a few frames (upkie/model/kinematic_tree.py:48-146,
model.py:63-110). Here the URDF is reduced once, on the host, to what the
kernels integrate: 7 composite bodies (trunk + 2 x thigh/calf/wheel) whose
frames sit at their joint origin and are aligned with the base frame at the
zero configuration. Merging fixed links is dynamically equivalent.
"""

from collections import deque
from typing import Dict, List, Optional
from xml.etree import ElementTree

import numpy as np

from ..abi import JOINT_NAMES, MAX_LINKS, NJ, UpkieModel
from ..exceptions import ModelError
from ..utils.rotations import rotation_matrix_from_rpy


class _Link:
    def __init__(self, elem):
        self.name = elem.attrib["name"]
        self.mass = 0.0
        self.com = np.zeros(3)
        self.inertia = np.zeros((3, 3))
        inertial = elem.find("inertial")
        if inertial is not None:
            origin = inertial.find("origin")
            xyz, rpy = _origin(origin)
            R = rotation_matrix_from_rpy(rpy)
            self.com = xyz
            self.mass = float(inertial.find("mass").attrib["value"])
            i = inertial.find("inertia").attrib
            I = np.array(
                [
                    [float(i["ixx"]), float(i.get("ixy", 0)), float(i.get("ixz", 0))],
                    [float(i.get("ixy", 0)), float(i["iyy"]), float(i.get("iyz", 0))],
                    [float(i.get("ixz", 0)), float(i.get("iyz", 0)), float(i["izz"])],
                ]
            )
            self.inertia = R @ I @ R.T  # in the link frame, about the com
        self.collisions = []
        for col in elem.findall("collision"):
            xyz, rpy = _origin(col.find("origin"))
            geom = col.find("geometry")
            shape = list(geom)[0] if geom is not None and len(list(geom)) else None
            if shape is not None:
                self.collisions.append(
                    dict(shape=shape.tag, params={k: v for k, v in shape.attrib.items()}, xyz=xyz, rpy=rpy)
                )
        self.contact = {}
        contact = elem.find("contact")
        if contact is not None:
            for child in contact:
                if "value" in child.attrib:
                    self.contact[child.tag] = float(child.attrib["value"])


def _origin(elem):
    if elem is None:
        return np.zeros(3), (0.0, 0.0, 0.0)
    xyz = np.array([float(v) for v in elem.attrib.get("xyz", "0 0 0").split()])
    rpy = tuple(float(v) for v in elem.attrib.get("rpy", "0 0 0").split())
    return xyz, rpy


class _Joint:
    def __init__(self, elem):
        self.name = elem.attrib["name"]
        self.type = elem.attrib.get("type", "fixed")
        self.parent = elem.find("parent").attrib["link"]
        self.child = elem.find("child").attrib["link"]
        self.xyz, rpy = _origin(elem.find("origin"))
        self.R = rotation_matrix_from_rpy(rpy)
        axis = elem.find("axis")
        self.axis = (
            np.array([float(v) for v in axis.attrib["xyz"].split()]) if axis is not None else np.array([1.0, 0.0, 0.0])
        )
        limit = elem.find("limit")
        self.limit = None
        if limit is not None:  # actuated joints = joints with <limit>, kinematic_tree.py:105-127
            self.limit = dict(
                lower=float(limit.attrib.get("lower", -np.inf)),
                upper=float(limit.attrib.get("upper", +np.inf)),
                effort=float(limit.attrib["effort"]),
                velocity=float(limit.attrib["velocity"]),
            )
        dyn = elem.find("dynamics")
        self.damping = float(dyn.attrib.get("damping", 0.0)) if dyn is not None else 0.0


class UrdfTree:
    """Links, joints and zero-configuration transforms to the root link."""

    def __init__(self, urdf_path: str):
        root = ElementTree.parse(urdf_path).getroot()
        self.links: Dict[str, _Link] = {}
        self.joints: List[_Joint] = []
        first = None
        for elem in root:
            if elem.tag == "link":
                link = _Link(elem)
                self.links[link.name] = link
                first = first or link.name
            elif elem.tag == "joint" and elem.find("parent") is not None and elem.find("child") is not None:
                self.joints.append(_Joint(elem))
        if first is None:
            raise ModelError("No links found in URDF")
        self.root = first  # kinematic_tree.py:131-132
        self.children: Dict[str, List[_Joint]] = {}
        self.parent_joint: Dict[str, _Joint] = {}
        for joint in self.joints:
            self.children.setdefault(joint.parent, []).append(joint)
            self.parent_joint[joint.child] = joint
        # zero-configuration pose of every link in the root frame
        self.R = {self.root: np.eye(3)}
        self.p = {self.root: np.zeros(3)}
        queue = deque([self.root])
        while queue:
            parent = queue.popleft()
            for joint in self.children.get(parent, []):
                self.R[joint.child] = self.R[parent] @ joint.R
                self.p[joint.child] = self.p[parent] + self.R[parent] @ joint.xyz
                queue.append(joint.child)

    @property
    def actuated_joints(self) -> List[_Joint]:
        return [j for j in self.joints if j.limit is not None]

    def body_root_of(self, link: str) -> str:
        """Child link of the closest actuated joint above `link` (or the tree
        root): the composite body `link` is rigidly part of."""
        while link in self.parent_joint and self.parent_joint[link].limit is None:
            link = self.parent_joint[link].parent
        return link


def load_urdf_model(urdf_path: str, template: Optional[UpkieModel] = None) -> UpkieModel:
    """Reduce a wheeled-biped URDF to the 7-body `UpkieModel`.

    Solver parameters without a URDF counterpart (Bullet defaults: damping,
    solver iterations, breaking threshold) are taken from `template` (default:
    the values of `default_model()`).
    """
    from .default_model import default_model

    tree = UrdfTree(urdf_path)
    model = UpkieModel()
    base = template if template is not None else default_model()
    for field in (
        "gravity",
        "contact_breaking_threshold",
        "friction_cfm",
        "base_linear_damping",
        "base_angular_damping",
        "max_joint_velocity",
        "pgs_iterations",
        "pgs_tolerance",
        "enforce_joint_limits",
        "contact_stiffness",
        "contact_damping",
        "friction_mu",
    ):
        setattr(model, field, getattr(base, field))

    actuated = {j.name: j for j in tree.actuated_joints}
    missing = [n for n in JOINT_NAMES if n not in actuated]
    if missing or len(actuated) != NJ:
        raise ModelError(f"expected actuated joints {JOINT_NAMES}, URDF has {sorted(actuated)}")
    joints = [actuated[n] for n in JOINT_NAMES]
    body_roots = [tree.root] + [j.child for j in joints]  # link heading each composite body
    body_of_root = {name: i for i, name in enumerate(body_roots)}

    # topology check: trunk <- hip <- knee <- wheel on each side
    for side in range(2):
        expected_parent = 0
        for depth in range(3):
            j = joints[3 * side + depth]
            parent_body = body_of_root[tree.body_root_of(j.parent)]
            if parent_body != expected_parent:
                raise ModelError(f"joint {j.name} hangs off body {parent_body}, expected {expected_parent}")
            expected_parent = 1 + 3 * side + depth

    # composite inertials in base coordinates at the zero configuration
    origins = [np.zeros(3)] + [tree.p[j.child] for j in joints]
    mass = np.zeros(7)
    moment = np.zeros((7, 3))
    members: List[List[str]] = [[] for _ in range(7)]
    for name, link in tree.links.items():
        if name not in tree.R:
            continue  # not connected to the root
        b = body_of_root[tree.body_root_of(name)]
        members[b].append(name)
        c = tree.p[name] + tree.R[name] @ link.com
        mass[b] += link.mass
        moment[b] += link.mass * c
    for b in range(7):
        if mass[b] <= 0.0:
            raise ModelError(f"composite body {b} has no mass")
        com = moment[b] / mass[b]
        I = np.zeros((3, 3))
        for name in members[b]:
            link = tree.links[name]
            R = tree.R[name]
            d = tree.p[name] + R @ link.com - com
            I += R @ link.inertia @ R.T + link.mass * (d @ d * np.eye(3) - np.outer(d, d))
        model.mass[b] = mass[b]
        model.com[b][:] = list(com - origins[b])
        model.inertia[b][:] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    # the links behind each composite body: Bullet keeps them apart and
    # randomize_inertias scales them one by one (pybullet_backend.py:555-601)
    n = 0
    for b in range(7):
        for name in members[b]:
            link = tree.links[name]
            if link.mass <= 0.0:
                continue
            if n >= MAX_LINKS:
                raise ModelError(f"more than {MAX_LINKS} links with mass")
            R = tree.R[name]
            Il = R @ link.inertia @ R.T
            model.link_body[n] = b
            model.link_randomized[n] = 0 if name == tree.root else 1  # Bullet's base is not in range(getNumJoints)
            model.link_mass[n] = link.mass
            model.link_com[n][:] = list(tree.p[name] + R @ link.com - origins[b])
            model.link_inertia[n][:] = [Il[0, 0], Il[1, 1], Il[2, 2], Il[0, 1], Il[0, 2], Il[1, 2]]
            n += 1
    model.num_links = n

    for idx, j in enumerate(joints):
        parent_body = body_of_root[tree.body_root_of(j.parent)]
        model.joint_pos[idx][:] = list(tree.p[j.child] - origins[parent_body])
        axis = tree.R[j.child] @ (j.axis / np.linalg.norm(j.axis))
        model.joint_axis[idx][:] = list(axis)
        model.joint_lower[idx] = j.limit["lower"]
        model.joint_upper[idx] = j.limit["upper"]
        model.joint_effort[idx] = j.limit["effort"]
        model.joint_velocity[idx] = j.limit["velocity"]
        model.joint_damping[idx] = j.damping

    # tires: radius from the collision cylinder (model.py:122-144), contact block
    tire_pos = []
    for w, tire_name in enumerate(("left_wheel_tire", "right_wheel_tire")):
        if tire_name not in tree.links:
            raise ModelError(f"{tire_name} link not found in URDF")
        tire = tree.links[tire_name]
        if len(tire.collisions) != 1 or tire.collisions[0]["shape"] != "cylinder":
            raise ModelError(f"{tire_name} should have exactly one cylinder collision geometry")
        col = tire.collisions[0]
        if w == 0:
            model.wheel_radius = float(col["params"]["radius"])
        center = tree.p[tire_name] + tree.R[tire_name] @ col["xyz"]
        tire_pos.append(tree.p[tire_name])
        wheel_origin = origins[3 * w + 3]
        model.wheel_center[w][:] = list(center - wheel_origin)
        if "stiffness" in tire.contact:
            model.contact_stiffness = tire.contact["stiffness"]
        if "damping" in tire.contact:
            model.contact_damping = tire.contact["damping"]
        if "lateral_friction" in tire.contact:
            model.friction_mu = tire.contact["lateral_friction"]  # x plane friction 1
    model.wheel_base = float(np.linalg.norm(tire_pos[0] - tire_pos[1]))  # model.py:88

    if "left_wheel_hub" in tree.links:  # model.py:92-104
        z_axis = tree.R["left_wheel_hub"][:, 2]
        if abs(z_axis[0]) > 1e-4 or abs(z_axis[2]) > 1e-4:
            raise ModelError("cannot determine wheeledness: left-wheel hub z-axis is not lateral")
        model.left_sign = 1.0 if z_axis[1] < 0 else -1.0
    else:
        model.left_sign = 1.0
    if "imu" not in tree.links:  # pybullet_backend.py:153-154
        raise ModelError("Robot does not have a link named 'imu'")
    model.imu_pos[:] = list(tree.p["imu"])
    model.rot_base_to_imu[:] = list(tree.R["imu"].T.ravel())  # model.py:106
    return model

"""Synthetic default rigid-body model of Upkie.

The reference reads its robot model from the ``upkie_description`` package
(pyproject.toml:93), which is not part of the reference tree and cannot be
installed offline. Link masses, inertias and joint offsets are therefore
SYNTHETIC here, chosen so that every constant the reference's own tests pin
holds exactly:

- total mass 5.3382 kg (upkie/cpp/interfaces/tests/BulletInterfaceTest.cpp:328)
- centre of mass (-0.0059, 0, -0.2455) m in the base frame at the zero
  configuration (upkie/cpp/interfaces/bullet/tests/utils_test.cpp:93-98)
- wheel radius 0.05 m, wheel base 0.3048 m (tests/model/test_model.py:75-85)
- rotation_base_to_imu = diag(-1, 1, -1) (tests/model/test_model.py:67-73)
- torso frame at (0, 0, -0.1) (tests/model/test_kinematic_tree.py:31-36)
- joint limits of docs/kinematics.md:43-53
- left-wheeled, lateral joint axes, left/right mirror symmetry
- base at z = 0.6 m puts both tires on the floor (upkie_env.py:87-90)

When the real URDF is available, ``upkie_amd.model.urdf.load_urdf_model``
produces the same structure from it.
"""

import numpy as np

from ..abi import NB, NJ, UpkieModel

TOTAL_MASS = 5.3382
COM_IN_BASE = np.array([-0.0059, 0.0, -0.2455])
WHEEL_RADIUS = 0.05
WHEEL_BASE = 0.3048

# Synthetic geometry (metres, base frame at zero configuration)
HIP_POS = np.array([0.0, 0.09, -0.05])  # left hip joint origin in base
KNEE_IN_THIGH = np.array([0.0, 0.03, -0.25])
WHEEL_IN_CALF = np.array([0.0, 0.0324, -0.25])
TORSO_POS = np.array([0.0, 0.0, -0.1])
IMU_POS = np.array([0.0, 0.0, -0.02])

# Synthetic leg inertials: (mass, com in body frame, [xx yy zz xy xz yz])
THIGH = (0.60, np.array([0.0, 0.01, -0.15]), [4.2e-3, 4.0e-3, 6.0e-4, 0, 0, 0])
CALF = (0.45, np.array([0.0, 0.01, -0.18]), [2.6e-3, 2.5e-3, 3.5e-4, 0, 0, 0])
WHEEL = (0.25, np.array([0.0, 0.0, 0.0]), [1.6e-4, 2.8e-4, 1.6e-4, 0, 0, 0])
TRUNK_INERTIA = [2.3e-2, 1.6e-2, 1.4e-2, 0.0, 0.0, 0.0]


def mirror_y(v):
    return np.array([v[0], -v[1], v[2]])


def default_model() -> UpkieModel:
    """Build the synthetic default model structure."""
    m = UpkieModel()
    leg_bodies = (THIGH, CALF, WHEEL)
    leg_joint_pos = (HIP_POS, KNEE_IN_THIGH, WHEEL_IN_CALF)

    # Leg bodies and joints, left then right (mirrored in y)
    for side, sign in ((0, +1.0), (1, -1.0)):
        for depth in range(3):
            body = 1 + 3 * side + depth
            joint = 3 * side + depth
            mass, com, inertia = leg_bodies[depth]
            pos = leg_joint_pos[depth]
            m.mass[body] = mass
            m.com[body][:] = list(com if sign > 0 else mirror_y(com))
            m.inertia[body][:] = inertia  # diagonal: mirror-invariant
            m.joint_pos[joint][:] = list(pos if sign > 0 else mirror_y(pos))
            # moteus convention: motor axes along -z of joint frames whose z
            # points to -y (left) / +y (right) of the base: docs/kinematics.md
            # :73, model.py:92-104 => left joints turn about +y, right about -y
            m.joint_axis[joint][:] = [0.0, sign, 0.0]

    # Trunk: close the mass and centre-of-mass budget exactly
    legs_mass = 2.0 * sum(b[0] for b in leg_bodies)
    trunk_mass = TOTAL_MASS - legs_mass
    moment = np.zeros(3)
    for side, sign in ((0, +1.0), (1, -1.0)):
        origin = np.zeros(3)
        for depth in range(3):
            mass, com, _ = leg_bodies[depth]
            pos = leg_joint_pos[depth]
            origin = origin + (pos if sign > 0 else mirror_y(pos))
            c = origin + (com if sign > 0 else mirror_y(com))
            moment += mass * c
    trunk_com = (TOTAL_MASS * COM_IN_BASE - moment) / trunk_mass
    m.mass[0] = trunk_mass
    m.com[0][:] = list(trunk_com)
    m.inertia[0][:] = TRUNK_INERTIA

    hip_knee = dict(lower=None, effort=16.0, velocity=28.8)
    for joint in range(NJ):
        kind = joint % 3
        if kind == 0:  # hip, docs/kinematics.md:45-47
            m.joint_lower[joint], m.joint_upper[joint] = -1.26, 1.26
        elif kind == 1:  # knee, docs/kinematics.md:48-50
            m.joint_lower[joint], m.joint_upper[joint] = -2.51, 2.51
        else:  # wheel: no position limit, docs/kinematics.md:51
            m.joint_lower[joint], m.joint_upper[joint] = -np.inf, np.inf
        if kind < 2:
            m.joint_effort[joint] = hip_knee["effort"]
            m.joint_velocity[joint] = hip_knee["velocity"]
        else:  # docs/kinematics.md:52-53, static_config.h:61-62
            m.joint_effort[joint] = 1.7
            m.joint_velocity[joint] = 111.0
        m.joint_damping[joint] = 0.0

    m.wheel_radius = WHEEL_RADIUS
    for w in range(2):
        m.wheel_center[w][:] = [0.0, 0.0, 0.0]
    m.wheel_base = WHEEL_BASE
    m.left_sign = +1.0
    m.imu_pos[:] = list(IMU_POS)
    m.rot_base_to_imu[:] = [-1, 0, 0, 0, 1, 0, 0, 0, -1]
    m.gravity = 9.81  # pybullet_backend.py:110
    # Tire <contact> block: the reference only pins stiffness > 1000 and
    # damping > 100 (bullet/tests/utils_test.cpp:41-57)
    m.contact_stiffness = 30000.0
    m.contact_damping = 1000.0
    m.friction_mu = 1.0
    m.friction_cfm = 0.01
    m.contact_breaking_threshold = 0.02
    m.base_linear_damping = 0.04
    m.base_angular_damping = 0.04
    m.max_joint_velocity = 100.0
    m.pgs_iterations = 50
    m.pgs_tolerance = 1e-6
    m.enforce_joint_limits = 1  # Bullet enforces URDF revolute limits
    # one URDF link per body (the URDF loader records the real fused links)
    m.num_links = 7
    for b in range(7):
        m.link_body[b] = b
        m.link_randomized[b] = 0 if b == 0 else 1  # the trunk is the URDF root: Bullet's base is not in range(getNumJoints)
        m.link_mass[b] = m.mass[b]
        m.link_com[b][:] = list(m.com[b])
        m.link_inertia[b][:] = list(m.inertia[b])
    return m


def model_wheel_base(model: UpkieModel) -> float:
    """Distance between the two tire centres at the zero configuration."""
    pts = []
    for side in range(2):
        p = np.zeros(3)
        for depth in range(3):
            p = p + np.array(model.joint_pos[3 * side + depth][:])
        p = p + np.array(model.wheel_center[side][:])
        pts.append(p)
    return float(np.linalg.norm(pts[0] - pts[1]))


assert NB == 7 and NJ == 6

"""Writes the synthetic default model as a URDF with the link/joint naming and
frame conventions of upkie_description (docs/kinematics.md:55-77): rotated
joint frames whose z-axis is lateral, motor axes along -z, fixed "virtual"
links of 1 g, tire links with a collision cylinder and a <contact> block.

Used to exercise the URDF -> merged-model path end to end; the real
upkie_description URDF goes through exactly the same loader."""

import numpy as np

from . import default_model as dm

VIRTUAL_MASS = 1e-3  # docs/kinematics.md:77
VIRTUAL_INERTIA = 1e-9


def _fmt(v):
    return " ".join(repr(float(x)) for x in v)


def _inertial(mass, com_link, rpy, inertia6):
    ixx, iyy, izz, ixy, ixz, iyz = (float(x) for x in inertia6)
    mass = float(mass)
    return (
        f'    <inertial>\n      <origin xyz="{_fmt(com_link)}" rpy="{_fmt(rpy)}"/>\n'
        f'      <mass value="{mass!r}"/>\n'
        f'      <inertia ixx="{ixx!r}" ixy="{ixy!r}" ixz="{ixz!r}" iyy="{iyy!r}" iyz="{iyz!r}" izz="{izz!r}"/>\n'
        "    </inertial>\n"
    )


def _rx(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def synthetic_urdf(right_wheeled: bool = False) -> str:
    """URDF of the synthetic model. With `right_wheeled` the leg link frames
    are built the other way round (z-axis of the LEFT wheel hub pointing to +y,
    model.py:92-104), which is what distinguishes a Cookie from an Upkie
    (model.py:22): every joint then turns about -y instead of +y in the base
    frame and the wheel / odometry signs flip."""
    model = dm.default_model()
    flip = -1.0 if right_wheeled else 1.0
    name = "cookie_synthetic" if right_wheeled else "upkie_synthetic"
    out = [f'<?xml version="1.0"?>\n<robot name="{name}">\n']

    # virtual links fixed to the base
    virtual = [
        ("torso", dm.TORSO_POS, (0.0, 0.0, 0.0)),
        ("imu", dm.IMU_POS, (np.pi, 0.0, np.pi)),  # tests/utils/test_rotations.py:48-55
        ("left_hip_qdd100_stator", dm.HIP_POS, (flip * np.pi / 2, 0.0, 0.0)),
        ("right_hip_qdd100_stator", dm.mirror_y(dm.HIP_POS), (-flip * np.pi / 2, 0.0, 0.0)),
    ]
    # base link inertial = trunk minus the virtual point masses
    M = model.mass[0]
    c = np.array(model.com[0][:])
    I6 = model.inertia[0][:]
    I = np.array([[I6[0], I6[3], I6[4]], [I6[3], I6[1], I6[5]], [I6[4], I6[5], I6[2]]])
    mb = M - VIRTUAL_MASS * len(virtual)
    cb = (M * c - sum(VIRTUAL_MASS * np.array(p) for _, p, _ in virtual)) / mb
    Ib = I.copy()
    for _, p, _ in virtual:
        d = np.array(p) - c
        Ib -= VIRTUAL_INERTIA * np.eye(3) + VIRTUAL_MASS * (d @ d * np.eye(3) - np.outer(d, d))
    d = cb - c
    Ib -= mb * (d @ d * np.eye(3) - np.outer(d, d))
    out.append('  <link name="base">\n')
    out.append(_inertial(mb, cb, (0, 0, 0), [Ib[0, 0], Ib[1, 1], Ib[2, 2], Ib[0, 1], Ib[0, 2], Ib[1, 2]]))
    out.append("  </link>\n")
    for name, pos, rpy in virtual:
        out.append(f'  <link name="{name}">\n')
        out.append(_inertial(VIRTUAL_MASS, (0, 0, 0), (0, 0, 0), [VIRTUAL_INERTIA] * 3 + [0, 0, 0]))
        out.append("  </link>\n")
        out.append(
            f'  <joint name="{name}_fix" type="fixed">\n    <parent link="base"/>\n    <child link="{name}"/>\n'
            f'    <origin xyz="{_fmt(pos)}" rpy="{_fmt(rpy)}"/>\n  </joint>\n'
        )

    for side, (prefix, sign) in enumerate((("left", +1.0), ("right", -1.0))):
        R = _rx(flip * sign * np.pi / 2)  # leg link frames: z-axis lateral, pointing away from +-y
        inertial_rpy = (-flip * sign * np.pi / 2, 0.0, 0.0)  # brings the inertia axes back to base axes
        chain = [
            (f"{prefix}_hip", f"{prefix}_hip_qdd100_stator", f"{prefix}_thigh", "revolute"),
            (f"{prefix}_knee", f"{prefix}_thigh", f"{prefix}_calf", "revolute"),
            (f"{prefix}_wheel", f"{prefix}_calf", f"{prefix}_wheel_hub", "continuous"),
        ]
        for depth, (jname, parent, child, jtype) in enumerate(chain):
            body = 1 + 3 * side + depth
            joint = 3 * side + depth
            mass = model.mass[body]
            com = np.array(model.com[body][:])
            inertia = list(model.inertia[body][:])
            if depth == 2:  # split the wheel between hub and tire
                tire_fraction = 0.4
                hub_mass, tire_mass = mass * (1 - tire_fraction), mass * tire_fraction
                hub_inertia = [x * (1 - tire_fraction) for x in inertia]
                tire_inertia = [x * tire_fraction for x in inertia]
                mass, inertia = hub_mass, hub_inertia
            out.append(f'  <link name="{child}">\n')
            out.append(_inertial(mass, R.T @ com, inertial_rpy, inertia))
            out.append("  </link>\n")
            origin = np.zeros(3) if depth == 0 else R.T @ np.array(model.joint_pos[joint][:])
            limit = f'effort="{float(model.joint_effort[joint])!r}" velocity="{float(model.joint_velocity[joint])!r}"'
            if jtype == "revolute":
                limit = f'lower="{float(model.joint_lower[joint])!r}" upper="{float(model.joint_upper[joint])!r}" ' + limit
            out.append(
                f'  <joint name="{jname}" type="{jtype}">\n    <parent link="{parent}"/>\n    <child link="{child}"/>\n'
                f'    <origin xyz="{_fmt(origin)}" rpy="0 0 0"/>\n    <axis xyz="0 0 -1"/>\n'
                f"    <limit {limit}/>\n  </joint>\n"
            )
            if depth == 2:
                tire = f"{prefix}_wheel_tire"
                out.append(f'  <link name="{tire}">\n')
                out.append(_inertial(tire_mass, R.T @ com, inertial_rpy, tire_inertia))
                out.append(
                    f'    <collision>\n      <origin xyz="0 0 0" rpy="0 0 0"/>\n      <geometry>\n'
                    f'        <cylinder radius="{float(model.wheel_radius)!r}" length="0.03"/>\n      </geometry>\n    </collision>\n'
                    f'    <contact>\n      <stiffness value="{float(model.contact_stiffness)!r}"/>\n'
                    f'      <damping value="{float(model.contact_damping)!r}"/>\n'
                    f'      <lateral_friction value="{float(model.friction_mu)!r}"/>\n    </contact>\n'
                )
                out.append("  </link>\n")
                out.append(
                    f'  <joint name="{tire}_fix" type="fixed">\n    <parent link="{child}"/>\n    <child link="{tire}"/>\n'
                    f'    <origin xyz="0 0 0" rpy="0 0 0"/>\n  </joint>\n'
                )
    out.append("</robot>\n")
    return "".join(out)


if __name__ == "__main__":
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "upkie_synthetic.urdf"), "w") as f:
        f.write(synthetic_urdf())
    with open(os.path.join(here, "cookie_synthetic.urdf"), "w") as f:
        f.write(synthetic_urdf(right_wheeled=True))
    print("wrote upkie_synthetic.urdf and cookie_synthetic.urdf")

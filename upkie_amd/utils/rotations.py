"""Rotation helpers with the reference's conventions (upkie/utils/rotations.py).
Quaternions are ``[w, x, y, z]``."""

from typing import Tuple

import numpy as np


def rotation_matrix_from_quaternion(quat: Tuple[float, float, float, float]) -> np.ndarray:
    """rotations.py:36-71; raises ValueError on non-unit input (:50-51)."""
    if abs(np.dot(quat, quat) - 1.0) > 1e-5:
        raise ValueError(f"Quaternion {quat} is not normalized")
    qw, qx, qy, qz = quat
    return np.array(
        [
            [1 - 2 * (qy**2 + qz**2), 2 * (qx * qy - qz * qw), 2 * (qw * qy + qx * qz)],
            [2 * (qx * qy + qz * qw), 1 - 2 * (qx**2 + qz**2), 2 * (qy * qz - qx * qw)],
            [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx**2 + qy**2)],
        ]
    )


def quaternion_from_rotation_matrix(rotation_matrix: np.ndarray) -> np.ndarray:
    """Same branch choice and sign as scipy's Rotation.from_matrix().as_quat,
    which rotations.py:16-33 calls; output ``[w, x, y, z]``."""
    m = np.asarray(rotation_matrix, dtype=np.float64)
    if m.shape != (3, 3):
        raise ValueError(f"Expected 3x3 matrix, got {m.shape}")
    decision = np.array([m[0, 0], m[1, 1], m[2, 2], m[0, 0] + m[1, 1] + m[2, 2]])
    choice = int(np.argmax(decision))
    q = np.empty(4)  # x y z w
    if choice != 3:
        i = choice
        j = (i + 1) % 3
        k = (j + 1) % 3
        q[i] = 1 - decision[3] + 2 * m[i, i]
        q[j] = m[j, i] + m[i, j]
        q[k] = m[k, i] + m[i, k]
        q[3] = m[k, j] - m[j, k]
    else:
        q[0] = m[2, 1] - m[1, 2]
        q[1] = m[0, 2] - m[2, 0]
        q[2] = m[1, 0] - m[0, 1]
        q[3] = 1 + decision[3]
    q /= np.linalg.norm(q)
    return np.array([q[3], q[0], q[1], q[2]])


def rotation_matrix_from_rpy(rpy: Tuple[float, float, float]) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)
    (rotations.py:74-99)."""
    roll, pitch, yaw = rpy
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ]
    )


def quaternion_from_euler_zyx(yaw: float, pitch: float, roll: float) -> np.ndarray:
    """``[w, x, y, z]`` of Rz(yaw) Ry(pitch) Rx(roll), i.e. what
    ``ScipyRotation.from_euler("ZYX", [yaw, pitch, roll])`` builds
    (robot_state_randomization.py:146-152)."""
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    return np.array(
        [
            cy * cp * cr + sy * sp * sr,
            cy * cp * sr - sy * sp * cr,
            cy * sp * cr + sy * cp * sr,
            sy * cp * cr - cy * sp * sr,
        ]
    )

"""Initial robot state (upkie/utils/robot_state.py:52-120).

The reference stores the base orientation as a scipy Rotation; here it is a
unit quaternion ``[w, x, y, z]`` (a scipy Rotation is accepted and converted).
"""

from typing import Optional

import numpy as np

from .robot_state_randomization import RobotStateRandomization


def _as_quat_wxyz(orientation) -> np.ndarray:
    if orientation is None:
        return np.array([1.0, 0.0, 0.0, 0.0])
    if hasattr(orientation, "as_quat"):  # scipy Rotation: [x, y, z, w]
        x, y, z, w = orientation.as_quat()
        return np.array([w, x, y, z], dtype=np.float64)
    quat = np.array(orientation, dtype=np.float64)
    if quat.shape != (4,):
        raise ValueError("orientation must be a scipy Rotation or [w, x, y, z]")
    return quat


class RobotState:
    def __init__(
        self,
        angular_velocity_base_in_base: Optional[np.ndarray] = None,
        joint_configuration: Optional[np.ndarray] = None,
        joint_velocity: Optional[np.ndarray] = None,
        linear_velocity_base_to_world_in_world: Optional[np.ndarray] = None,
        orientation_base_in_world=None,
        position_base_in_world: Optional[np.ndarray] = None,
        randomization: Optional[RobotStateRandomization] = None,
    ):
        def vec(value, default):
            return np.array(value, dtype=np.float64) if value is not None else default

        self.angular_velocity_base_in_base = vec(angular_velocity_base_in_base, np.zeros(3))
        self.joint_configuration = vec(joint_configuration, np.zeros(6))
        self.joint_velocity = vec(joint_velocity, np.zeros(6))  # never used, :261-267
        self.linear_velocity_base_to_world_in_world = vec(linear_velocity_base_to_world_in_world, np.zeros(3))
        self.orientation_base_in_world = _as_quat_wxyz(orientation_base_in_world)
        # a scipy Rotation is kept as given so that sample_state composes exactly what the reference composes
        self._rotation = orientation_base_in_world if hasattr(orientation_base_in_world, "as_quat") else None
        # Upkie above the horizontal plane, robot_state.py:112
        self.position_base_in_world = vec(position_base_in_world, np.array([0.0, 0.0, 0.6]))
        self.randomization = randomization if randomization is not None else RobotStateRandomization()

    def write_to_config(self, cfg) -> None:
        """Fill the init-state and randomisation fields of an UpkieSimConfig."""
        cfg.init_pos[:] = list(self.position_base_in_world)
        cfg.init_quat[:] = list(self.orientation_base_in_world)
        cfg.init_linvel[:] = list(self.linear_velocity_base_to_world_in_world)
        cfg.init_angvel[:] = list(self.angular_velocity_base_in_base)
        cfg.init_joint[:] = list(self.joint_configuration)
        r = self.randomization
        cfg.rand_roll, cfg.rand_pitch = r.roll, r.pitch
        cfg.rand_x, cfg.rand_z = r.x, r.z
        cfg.rand_omega_x, cfg.rand_omega_y = r.omega_x, r.omega_y
        cfg.rand_linvel[:] = list(r.linear_velocity)

    def _base_rotation(self):
        """The base orientation as a scipy Rotation (kept as given when one was given)."""
        from scipy.spatial.transform import Rotation

        if self._rotation is not None:
            return self._rotation
        w, x, y, z = self.orientation_base_in_world
        return Rotation.from_quat([x, y, z, w])

    def sample_angular_velocity(self, np_random: np.random.Generator) -> np.ndarray:
        """Angular velocity around this state's (robot_state.py:109-125): offset added in the base frame."""
        return self.angular_velocity_base_in_base + self.randomization.sample_angular_velocity(np_random)

    def sample_linear_velocity(self, np_random: np.random.Generator) -> np.ndarray:
        """Linear velocity around this state's (robot_state.py:127-143), world frame."""
        return self.linear_velocity_base_to_world_in_world + self.randomization.sample_linear_velocity(np_random)

    def sample_orientation(self, np_random: np.random.Generator):
        """Orientation around this state's, a scipy Rotation (robot_state.py:145-160): the random offset is composed
        on the right of the base orientation."""
        return self._base_rotation() * self.randomization.sample_orientation(np_random)

    def sample_position(self, np_random: np.random.Generator) -> np.ndarray:
        """Position around this state's (robot_state.py:162-173)."""
        return self.position_base_in_world + self.randomization.sample_position(np_random)

    def sample_state(self, np_random: np.random.Generator) -> "RobotState":
        """A state drawn around this one as `RobotState.sample_state` draws it
        (robot_state.py:175-196): angular velocity, linear velocity,
        orientation, position, in that order, from the env's seeded generator.
        Used by the single-robot envs; batched envs draw on the device."""
        angular_velocity = self.sample_angular_velocity(np_random)
        linear_velocity = self.sample_linear_velocity(np_random)
        orientation = self.sample_orientation(np_random)
        position = self.sample_position(np_random)
        return RobotState(
            angular_velocity_base_in_base=angular_velocity,
            joint_configuration=self.joint_configuration,
            joint_velocity=self.joint_velocity,
            linear_velocity_base_to_world_in_world=linear_velocity,
            orientation_base_in_world=orientation,
            position_base_in_world=position,
            randomization=self.randomization,
        )

    def write_exact_to_config(self, cfg) -> None:
        """This very state as the config's initial state, randomisation off."""
        cfg.init_pos[:] = list(self.position_base_in_world)
        cfg.init_quat[:] = list(self.orientation_base_in_world)
        cfg.init_linvel[:] = list(self.linear_velocity_base_to_world_in_world)
        cfg.init_angvel[:] = list(self.angular_velocity_base_in_base)
        cfg.init_joint[:] = list(self.joint_configuration)
        cfg.rand_roll = cfg.rand_pitch = cfg.rand_x = cfg.rand_z = cfg.rand_omega_x = cfg.rand_omega_y = 0.0
        cfg.rand_linvel[:] = [0.0, 0.0, 0.0]

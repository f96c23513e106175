"""Initial-state randomisation magnitudes
(upkie/utils/robot_state_randomization.py:53-133). Batched envs sample on the
device (Philox, the reference's draw order robot_state.py:182-187); the
single-robot envs sample here, on the host, from gymnasium's seeded
``np_random`` exactly as the reference does (:135-193), so that
``reset(seed=s)`` starts from the reference's states."""

from typing import Optional

import numpy as np


class RobotStateRandomization:
    def __init__(
        self,
        roll: float = 0.0,
        pitch: float = 0.0,
        x: float = 0.0,
        z: float = 0.0,
        omega_x: float = 0.0,
        omega_y: float = 0.0,
        linear_velocity: Optional[np.ndarray] = None,
    ):
        self.roll = roll
        self.pitch = pitch
        self.x = x
        self.z = z
        self.omega_x = omega_x
        self.omega_y = omega_y
        self.linear_velocity = (
            np.array(linear_velocity, dtype=np.float64) if linear_velocity is not None else np.zeros(3)
        )

    def update(
        self,
        roll: Optional[float] = None,
        pitch: Optional[float] = None,
        x: Optional[float] = None,
        z: Optional[float] = None,
        omega_x: Optional[float] = None,
        omega_y: Optional[float] = None,
        v_x: Optional[float] = None,
        v_z: Optional[float] = None,
    ) -> None:
        """robot_state_randomization.py:92-133."""
        if roll is not None:
            self.roll = roll
        if pitch is not None:
            self.pitch = pitch
        if x is not None:
            self.x = x
        if z is not None:
            self.z = z
        if omega_x is not None:
            self.omega_x = omega_x
        if omega_y is not None:
            self.omega_y = omega_y
        if v_x is not None:
            self.linear_velocity[0] = v_x
        if v_z is not None:
            self.linear_velocity[2] = v_z

    # Host-side sampling for the single-robot envs
    # (robot_state_randomization.py:135-193): one vector draw of three uniform
    # numbers per quantity, bounds given as arrays (components with a zero
    # bound still consume their draw).
    def _draw3(self, np_random: np.random.Generator, low, high) -> np.ndarray:
        return np_random.uniform(low=np.array(low, dtype=np.float64), high=np.array(high, dtype=np.float64), size=3)

    def sample_angular_velocity(self, np_random: np.random.Generator) -> np.ndarray:
        """Body-frame angular velocity offset: (+-omega_x, +-omega_y, 0)."""
        return self._draw3(np_random, [-self.omega_x, -self.omega_y, 0.0], [self.omega_x, self.omega_y, 0.0])

    def sample_linear_velocity(self, np_random: np.random.Generator) -> np.ndarray:
        """World-frame linear velocity offset within +-linear_velocity."""
        return self._draw3(np_random, -self.linear_velocity, self.linear_velocity)

    def sample_orientation(self, np_random: np.random.Generator):
        """Rotation offset: intrinsic ZYX Euler angles (0, +-pitch, +-roll)."""
        from scipy.spatial.transform import Rotation

        bounds = np.array([0.0, self.pitch, self.roll])
        return Rotation.from_euler("ZYX", self._draw3(np_random, -bounds, bounds))

    def sample_position(self, np_random: np.random.Generator) -> np.ndarray:
        """Position offset: x within +-x, z within [0, z]."""
        return self._draw3(np_random, [-self.x, 0.0, 0.0], [self.x, 0.0, self.z])

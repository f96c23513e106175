"""Backend plugin interface: the same four abstract methods as the reference's
``upkie.envs.backends.Backend`` (upkie/envs/backends/backend.py:11-50)."""

from abc import ABC, abstractmethod

from ...utils.robot_state import RobotState


class Backend(ABC):
    @abstractmethod
    def get_spine_observation(self) -> dict:
        """Observation in spine format."""

    @abstractmethod
    def reset(self, init_state: RobotState) -> dict:
        """Reset to an initial state, return the initial spine observation."""

    @abstractmethod
    def step(self, action: dict) -> dict:
        """Apply a spine-format action, return the next spine observation."""

    @abstractmethod
    def close(self) -> None:
        """Release resources."""

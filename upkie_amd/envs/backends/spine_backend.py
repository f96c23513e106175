"""Backend connected to a spine through shared memory: the same plugin as the
reference's ``SpineBackend`` (upkie/envs/backends/spine_backend.py:108-260),
speaking to a `upkie_amd.spine.HipSpine` (or to any spine of the reference:
the wire protocol is the same)."""

import copy
from typing import Optional

from ...model.model import Model
from ...spine.spine_interface import SpineInterface
from ...utils.robot_state import RobotState
from .backend import Backend

## Default spine configuration, spine_backend.py:77-105.
DEFAULT_SPINE_CONFIG = {
    "bullet": {
        "gui": True,
        "reset": {
            "orientation_base_in_world": [1.0, 0.0, 0.0, 0.0],
            "position_base_in_world": [0.0, 0.0, 0.6],
        },
        "torque_control": {"kp": 20.0, "kd": 1.0},
    },
    "floor_contact": {"upper_leg_torque_threshold": 10.0},
    "wheel_contact": {
        "cutoff_period": 0.2,
        "liftoff_inertia": 0.001,
        "min_touchdown_acceleration": 2.0,
        "min_touchdown_torque": 0.015,
        "touchdown_inertia": 0.004,
    },
    "wheel_odometry": {"signed_radius": {"left_wheel": +0.05, "right_wheel": -0.05}},
}


def nested_update(target: dict, new: dict) -> None:
    """Recursive dict.update, upkie/utils/nested_update.py."""
    for key, value in new.items():
        if isinstance(value, dict) and isinstance(target.get(key), dict):
            nested_update(target[key], value)
        else:
            target[key] = value


class SpineBackend(Backend):
    def __init__(
        self,
        shm_name: str = "/upkie",
        model: Optional[Model] = None,
        spine_config: Optional[dict] = None,
        retries: int = 10,
        timeout_ns: int = 100_000_000,
    ):
        model = model if model is not None else Model()
        sign = +1.0 if model.left_wheeled else -1.0  # spine_backend.py:137-139
        signed_radius = sign * model.wheel_radius
        config = copy.deepcopy(DEFAULT_SPINE_CONFIG)
        nested_update(
            config,
            {
                "wheel_odometry": {"signed_radius": {"left_wheel": signed_radius, "right_wheel": -signed_radius}},
                "base_orientation": {"rotation_base_to_imu": [float(v) for v in model.rotation_base_to_imu.flatten()]},
            },
        )
        if spine_config is not None:
            nested_update(config, spine_config)
        self._spine = SpineInterface(shm_name, retries=retries, timeout_ns=timeout_ns)
        self._spine_config = config
        self._last_observation: dict = {}

    def close(self) -> None:
        if getattr(self, "_spine", None) is not None:
            try:
                self._spine.stop()
            finally:
                self._spine.close()
                self._spine = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def reset(self, init_state: Optional[RobotState] = None) -> dict:
        self._spine.stop()
        if init_state is not None:  # spine_backend.py:206-227
            reset = self._spine_config["bullet"]["reset"]
            reset["orientation_base_in_world"] = [float(v) for v in init_state.orientation_base_in_world]
            reset["position_base_in_world"] = [float(v) for v in init_state.position_base_in_world]
            reset["linear_velocity_base_to_world_in_world"] = [float(v) for v in init_state.linear_velocity_base_to_world_in_world]
            reset["angular_velocity_base_in_base"] = [float(v) for v in init_state.angular_velocity_base_in_base]
            reset["joint_configuration"] = [float(v) for v in init_state.joint_configuration]
        self._last_observation = self._spine.start(self._spine_config)
        return self._last_observation

    def step(self, action: dict) -> dict:
        self._last_observation = self._spine.set_action(dict(action))
        return self._last_observation

    def get_spine_observation(self) -> dict:
        return self._last_observation

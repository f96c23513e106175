"""Spine-format observation (pybullet_backend.py:313-331) for a batch."""

from typing import Dict

from ..abi import JOINT_NAMES, SERVO_OBS_KEYS


def spine_observation_dict(raw: Dict, env: int = None) -> dict:
    """Arrange the flat tensors of `BatchedSim.observe` into the nested spine
    dictionary. With `env` set, returns Python floats / lists for that env
    (B = 1 compatibility); otherwise ``[B, ...]`` tensors."""

    def pick(t):
        if env is None:
            return t
        v = t[env]
        return v.tolist() if v.dim() > 0 else v.item()

    servo = {}
    for j, name in enumerate(JOINT_NAMES):
        servo[name] = {key: pick(raw["servo"][:, j, k]) for k, key in enumerate(SERVO_OBS_KEYS)}
    rotation = raw["rotation_base_to_world"].reshape(-1, 3, 3)
    contact = raw["floor_contact"]
    return {
        "base_orientation": {
            "angular_velocity": pick(raw["angular_velocity"]),
            "linear_velocity": pick(raw["linear_velocity"]),
            "pitch": pick(raw["pitch"]),
            "rotation_base_to_world": pick(rotation),
        },
        "floor_contact": {"contact": bool(contact[env].item()) if env is not None else contact.bool()},
        "imu": {
            "orientation": pick(raw["imu_orientation"]),
            "angular_velocity": pick(raw["imu_angular_velocity"]),
            "linear_acceleration": pick(raw["imu_linear_acceleration"]),
            "raw_linear_acceleration": pick(raw["imu_raw_linear_acceleration"]),
        },
        "servo": servo,
        "wheel_odometry": {
            "position": pick(raw["wheel_odometry"][:, 0]),
            "velocity": pick(raw["wheel_odometry"][:, 1]),
        },
    }


class LazySpineObservation(dict):
    """``info["spine_observation"]`` of the vector envs: materialised by one
    extra kernel launch on first access after a step (keeps the per-step HBM
    traffic at the algorithmic minimum when nobody reads it).

    The IMU accelerometer is a finite difference against the velocity stored at
    the previous materialisation (pybullet_backend.py:405-408): read it every
    step (or pass ``eager_spine_observation=True`` to the env) for the
    reference's per-step semantics."""

    def __init__(self, sim):
        super().__init__()
        self._sim = sim
        self._fresh = False
        self._overrides: Dict = {}

    def invalidate(self) -> None:
        self._fresh = False

    def set_overrides(self, blocks: Dict) -> None:
        """Blocks written by the spine observer pipeline (upkie_amd.observers):
        merged over the backend's blocks of the same name, as the spine's
        observers write over / next to what the interface reported."""
        self._overrides = blocks
        self._fresh = False

    def materialize(self) -> "LazySpineObservation":
        if not self._fresh:
            super().clear()
            super().update(spine_observation_dict(self._sim.observe(update_imu=True)))
            for key, block in self._overrides.items():
                if super().__contains__(key):
                    super().__getitem__(key).update(block)
                else:
                    super().__setitem__(key, dict(block))
            self._fresh = True
        return self

    def __getitem__(self, key):
        self.materialize()
        return super().__getitem__(key)

    def __contains__(self, key):
        self.materialize()
        return super().__contains__(key)

    def keys(self):
        self.materialize()
        return super().keys()

    def items(self):
        self.materialize()
        return super().items()

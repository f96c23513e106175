"""Batched spine observers: what the C++ spine appends to every observation
(spines/common/observers.h:22-42), evaluated on the device for every env.

The Python hot path of the reference (PyBulletBackend) reports ground-truth
contact and position-based odometry (pybullet_backend.py:432-490); agents that
go to the real robot read the spine's estimators instead: BaseOrientation
(upkie/cpp/observers/BaseOrientation.h), FloorContact + WheelContact
(FloorContact.cpp, WheelContact.cpp) and the velocity-integrating
WheelOdometry (WheelOdometry.cpp). This class runs that pipeline on the
tensors `BatchedSim.observe()` returns, so a policy trained on the GPU sees
the observation semantics of the spine.
"""

import ctypes as C
from typing import Dict, Optional

import torch

from . import abi, lib
from .exceptions import UpkieRuntimeError


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


def observer_config_from_spine_config(num_envs: int, dt: float, spine_config: Optional[dict] = None) -> abi.UpkieObserverConfig:
    """`UpkieObserverConfig` from a spine configuration dictionary with the
    keys of ``_DEFAULT_SPINE_CONFIG`` (spine_backend.py:77-105): "floor_contact",
    "wheel_contact", "wheel_odometry" and "base_orientation"
    (BaseOrientation.h:157-176)."""
    cfg = abi.default_observer_config(num_envs, dt)
    spine_config = spine_config or {}
    fc = spine_config.get("floor_contact", {})
    if "upper_leg_torque_threshold" in fc:
        cfg.upper_leg_torque_threshold = float(fc["upper_leg_torque_threshold"])
    wc = spine_config.get("wheel_contact")
    if wc is not None:  # WheelContact::Parameters::configure reads all five keys, WheelContact.h:54-68
        cfg.liftoff_inertia = float(wc["liftoff_inertia"])
        cfg.min_touchdown_acceleration = float(wc["min_touchdown_acceleration"])
        cfg.min_touchdown_torque = float(wc["min_touchdown_torque"])
        cfg.wheel_cutoff_period = float(wc["cutoff_period"])
        cfg.touchdown_inertia = float(wc["touchdown_inertia"])
    radius = spine_config.get("wheel_odometry", {}).get("signed_radius")
    if radius is not None:
        cfg.signed_radius[0] = float(radius["left_wheel"])
        cfg.signed_radius[1] = float(radius["right_wheel"])
    bo = spine_config.get("base_orientation", {})
    if "rotation_base_to_imu" in bo:
        flat = [float(v) for row in bo["rotation_base_to_imu"] for v in (row if hasattr(row, "__len__") else [row])]
        if len(flat) != 9:
            raise ValueError("rotation_base_to_imu must be a 3x3 matrix")
        for i, v in enumerate(flat):
            cfg.rotation_base_to_imu[i] = v
    return cfg


def observer_blocks(tensors: Dict[str, Optional[torch.Tensor]]) -> Dict[str, dict]:
    """Arrange the flat output tensors of one pipeline run into the blocks the
    observers write into the observation dictionary."""
    wc = tensors["wheel_contact"]
    B = wc.shape[0]
    result = {
        "floor_contact": {  # FloorContact::write, FloorContact.cpp:93-104
            "contact": tensors["floor_contact"].bool(),
            "upper_leg_torque": tensors["upper_leg_torque"],
            "left_wheel": {"abs_acceleration": wc[:, 0, 0], "abs_torque": wc[:, 0, 1], "contact": wc[:, 0, 2] != 0, "inertia": wc[:, 0, 3]},
            "right_wheel": {"abs_acceleration": wc[:, 1, 0], "abs_torque": wc[:, 1, 1], "contact": wc[:, 1, 2] != 0, "inertia": wc[:, 1, 3]},
        },
        "wheel_odometry": {  # WheelOdometry::write, WheelOdometry.cpp:56-60
            "position": tensors["wheel_odometry"][:, 0],
            "velocity": tensors["wheel_odometry"][:, 1],
        },
    }
    if tensors.get("base_pitch") is not None:
        result["base_orientation"] = {  # BaseOrientation::write, BaseOrientation.cpp:36-40
            "pitch": tensors["base_pitch"],
            "angular_velocity": tensors["base_angular_velocity"],
            "rotation_base_to_world": tensors["rotation_base_to_world"].view(B, 3, 3),
        }
    return result


def observer_blocks_from_state(state: torch.Tensor) -> Dict[str, dict]:
    """The "floor_contact" and "wheel_odometry" blocks out of observer memory
    ``[16, B]`` (what `BatchedSim.attach_observers` maintains inside the step)."""
    W = abi.O_WHEEL
    wheel = lambda w: {  # noqa: E731
        "abs_acceleration": state[W + 5 * w + 1],
        "abs_torque": state[W + 5 * w + 2],
        "contact": state[W + 5 * w + 4] != 0,
        "inertia": state[W + 5 * w + 3],
    }
    return {
        "floor_contact": {
            "contact": state[abi.O_CONTACT] != 0,
            "upper_leg_torque": state[abi.O_UPPER_LEG_TORQUE],
            "left_wheel": wheel(0),
            "right_wheel": wheel(1),
        },
        "wheel_odometry": {"position": state[abi.O_ODOMETRY_POSITION], "velocity": state[abi.O_ODOMETRY_VELOCITY]},
    }


class BatchedObservers:
    """Observer memory `[16, B]` on the device + one launch per spine cycle."""

    def __init__(self, config: abi.UpkieObserverConfig, device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise UpkieRuntimeError("no HIP device visible (there is no CPU fallback)")
        self._lib = lib.load()
        self.device = torch.device(device)
        self.config = config
        self.num_envs = int(config.num_envs)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            status = self._lib.upkie_observers_create(C.byref(config), C.byref(self._handle))
        lib.check(status, None, what="observers")
        B = self.num_envs
        assert self._lib.upkie_observers_state_bytes(self._handle) == abi.OBSERVER_STATE_WORDS * B * 4
        self.state = torch.zeros((abi.OBSERVER_STATE_WORDS, B), dtype=torch.float32, device=self.device)

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.upkie_observers_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, mask: Optional[torch.Tensor] = None) -> None:
        """Observer::reset of the three observers for masked envs."""
        if mask is not None:
            mask = mask.to(self.device, torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            status = self._lib.upkie_observers_reset(self._handle, C.c_void_p(self.state.data_ptr()), _ptr(mask), self._stream())
        lib.check(status, self._handle, what="observers")

    def step(
        self,
        servo: torch.Tensor,
        imu_orientation: Optional[torch.Tensor] = None,
        imu_angular_velocity: Optional[torch.Tensor] = None,
        cross_button: Optional[torch.Tensor] = None,
    ) -> Dict[str, torch.Tensor]:
        """One ObserverPipeline::run. `servo` is `[B, 6, 5]` (position,
        velocity, torque, temperature, voltage), the IMU tensors are those of
        `BatchedSim.observe()`. Returns the blocks the observers write:
        "base_orientation" (when an IMU is given), "floor_contact",
        "wheel_odometry"."""
        B, dev = self.num_envs, self.device
        f32 = dict(dtype=torch.float32, device=dev)

        def f(t, shape):
            if t is None:
                return None
            t = t.to(dev, torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"expected shape {shape}, got {tuple(t.shape)}")
            return t

        servo = f(servo, (B, 6, 5))  # None: no "servo" block, only BaseOrientation runs (FloorContact.cpp:42-44)
        imu_orientation = f(imu_orientation, (B, 4))
        imu_angular_velocity = f(imu_angular_velocity, (B, 3))
        if cross_button is not None:
            cross_button = cross_button.to(dev, torch.uint8).contiguous()
        inp = abi.UpkieObserverInput(
            _ptr(servo), _ptr(imu_orientation), _ptr(imu_angular_velocity), _ptr(cross_button)
        )
        have_imu = imu_orientation is not None
        tensors = {
            "base_pitch": torch.empty(B, **f32) if have_imu else None,
            "base_angular_velocity": torch.empty((B, 3), **f32) if have_imu else None,
            "rotation_base_to_world": torch.empty((B, 9), **f32) if have_imu else None,
            "floor_contact": torch.empty(B, dtype=torch.uint8, device=dev),
            "upper_leg_torque": torch.empty(B, **f32),
            "wheel_contact": torch.empty((B, 2, 4), **f32),
            "wheel_odometry": torch.empty((B, 2), **f32),
        }
        out = abi.UpkieObserverOutput(*[_ptr(tensors[name]) for name, _ in abi.UpkieObserverOutput._fields_])
        with torch.cuda.device(dev):
            status = self._lib.upkie_observers_step(
                self._handle, C.c_void_p(self.state.data_ptr()), C.byref(inp), C.byref(out), self._stream()
            )
        lib.check(status, self._handle, what="observers")
        if servo is None:
            return {k: v for k, v in observer_blocks(tensors).items() if k == "base_orientation"}
        return observer_blocks(tensors)

    def step_from_sim(self, sim, update_imu: bool = False, cross_button: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Run the pipeline on the current spine observation of a `BatchedSim`."""
        obs = sim.observe(update_imu=update_imu)
        return self.step(obs["servo"], obs["imu_orientation"], obs["imu_angular_velocity"], cross_button)

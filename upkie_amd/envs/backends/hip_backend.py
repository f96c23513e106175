"""Single-robot backend on the HIP library with the reference's Backend
interface and spine dictionaries (upkie/envs/backends/pybullet_backend.py)."""

from typing import Dict, List, Optional

import numpy as np
import torch

from ... import abi
from ...exceptions import UpkieException
from ...model.joint_properties import JointProperties
from ...model.model import Model
from ...utils.external_force import ExternalForce
from ...utils.point_contact import PointContact, point_contacts
from ...utils.robot_state import RobotState
from ...utils.robot_state_randomization import RobotStateRandomization
from ..external_forces import ExternalForceSet
from ..spine_observation import spine_observation_dict
from .backend import Backend


class HipBackend(Backend):
    """Drop-in for ``PyBulletBackend(dt, gui=False, ...)`` with one robot.

    Same constructor keywords (pybullet_backend.py:55-66); ``gui=True`` is
    refused. Actions and observations are the reference's spine dictionaries.
    """

    def __init__(
        self,
        dt: float,
        gui: bool = False,
        inertia_variation: float = 0.0,
        joint_properties: Optional[Dict[str, JointProperties]] = None,
        js_path: str = "/dev/input/js0",
        model: Optional[Model] = None,
        nb_substeps: Optional[int] = None,
        torque_control_kd: float = 1.0,
        torque_control_kp: float = 20.0,
        device: str = "cuda:0",
        seed: int = 0,
        sim_factory=None,
    ) -> None:
        if gui:
            raise UpkieException("the HIP backend has no GUI")
        self._model = model if model is not None else Model()
        cfg = abi.default_sim_config(1, frequency=1.0 / dt, nb_substeps=nb_substeps, seed=seed)
        cfg.dt = dt
        if nb_substeps is None:
            cfg.nb_substeps = int(1000.0 * dt)  # pybullet_backend.py:85-87
        cfg.torque_control_kp = torque_control_kp
        cfg.torque_control_kd = torque_control_kd
        cfg.autoreset_mode = abi.AUTORESET_DISABLED
        for idx, name in enumerate(abi.JOINT_NAMES):
            props = (joint_properties or {}).get(name, JointProperties())
            cfg.joint_friction[idx] = props.friction
            cfg.torque_control_noise[idx] = props.torque_control_noise
            cfg.torque_measurement_noise[idx] = props.torque_measurement_noise
        if sim_factory is None:
            from ...sim import BatchedSim

            sim_factory = lambda c, m, d: BatchedSim(c, m, device=d)  # noqa: E731
        self.sim = sim_factory(cfg, self._model.struct, device)
        self._external_forces = ExternalForceSet(self._model, 1)
        self.config = cfg
        self.inertia_variation = inertia_variation
        self.torque_control_kd = torque_control_kd
        self.torque_control_kp = torque_control_kp
        self.joystick = None
        if abs(inertia_variation) > 1e-10:  # pybullet_backend.py:178-179
            self.randomize_inertias(inertia_variation)
        self._last_observation: Optional[dict] = None

    # Backend interface ---------------------------------------------------
    def close(self) -> None:
        self.sim.close()

    def reset(self, init_state: RobotState, sample: bool = False, seed: Optional[int] = None) -> dict:
        """Reset to `init_state` (pybullet_backend.py:220-232). With
        ``sample=True`` the state is drawn on the device around `init_state`
        with its randomisation magnitudes, as UpkieEnv.reset does on the host
        (upkie_env.py:189-190)."""
        nominal = init_state
        if not sample:
            nominal = RobotState(
                angular_velocity_base_in_base=init_state.angular_velocity_base_in_base,
                joint_configuration=init_state.joint_configuration,
                linear_velocity_base_to_world_in_world=init_state.linear_velocity_base_to_world_in_world,
                orientation_base_in_world=init_state.orientation_base_in_world,
                position_base_in_world=init_state.position_base_in_world,
                randomization=RobotStateRandomization(),
            )
        nominal.write_to_config(self.config)
        if seed is not None:
            self.config.seed = int(seed)
        self.sim.push_config()
        self.sim.reset()
        return self.get_spine_observation()

    def step(self, action: dict) -> dict:
        """Apply a spine action for one control period
        (pybullet_backend.py:269-311)."""
        act = np.zeros((1, 6, 6), dtype=np.float32)
        act[0, :, 0] = np.nan  # joints missing from the action get no torque
        servo_actions = action.get("servo", {})
        for idx, name in enumerate(abi.JOINT_NAMES):
            if name not in servo_actions:
                continue
            servo = servo_actions[name]
            velocity = servo["velocity"]
            assert not np.isnan(velocity)  # pybullet_backend.py:519
            act[0, idx] = [
                servo["position"],
                velocity,
                servo.get("feedforward_torque", 0.0),  # :284-291
                servo.get("kp_scale", 1.0),
                servo.get("kd_scale", 1.0),
                servo["maximum_torque"],
            ]
        self.sim.step_servos(torch.from_numpy(act))
        return self.get_spine_observation()

    def get_spine_observation(self) -> dict:
        """pybullet_backend.py:313-331, for the single robot."""
        raw = self.sim.observe(update_imu=True)
        self._last_observation = spine_observation_dict(raw, env=0)
        return self._last_observation

    # extras of PyBulletBackend ---------------------------------------------
    def randomize_inertias(self, inertia_variation: float) -> None:
        """pybullet_backend.py:571-601, per composite body."""
        self.sim.randomize_inertias(inertia_variation)

    def set_external_forces(self, external_forces: Dict[str, ExternalForce]) -> None:
        """pybullet_backend.py:603-623: forces on any link, world or link
        frame, persisting until the link is given another force."""
        self._external_forces.update(external_forces)
        self._external_forces.push(self.sim)

    def get_contact_points(self, link_name: Optional[str] = None) -> List[PointContact]:
        """pybullet_backend.py:660-716: contact points between the robot and
        the floor, optionally only those of one link. Only the tires collide
        with the floor here, one point each at most."""
        return point_contacts(self.sim.contact_points()[0].cpu().numpy(), link_name)

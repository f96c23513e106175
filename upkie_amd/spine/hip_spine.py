"""A spine whose actuation interface is ONE env of a GPU batch.

Mirrors ``Spine::simulate`` (upkie/cpp/spine/Spine.cpp:119-141) with the
Bullet interface replaced by a `UpkieServosVecEnv`: requests arrive through the
shared memory (`AgentInterface`), go through the reference's state machine
(`StateMachine`), a ``kStart`` resets the attached env with the "bullet.reset"
block of the configuration dictionary (spine_backend.py:206-227), a
``kAction`` steps the whole batch once (the attached env with the agent's
servo targets, every other env with `batch_policy`), and the observation
dictionary written back carries the spine's keys ("servo", "imu",
"base_orientation", "floor_contact", "wheel_odometry", "time").

The env period plays the role of ``nb_substeps`` spine cycles: a bullet spine
at 1 kHz with ``nb_substeps = 5`` (spines/bullet_spine.cpp) is an env at
200 Hz whose kernel runs 5 physics substeps.
"""

import time
from typing import Callable, Optional

import msgpack
import numpy as np
import torch

from ..abi import ACTION_KEYS, JOINT_NAMES
from ..exceptions import UpkieRuntimeError
from ..utils.robot_state import RobotState
from .agent_interface import AgentInterface
from .state_machine import Event, State, StateMachine


def _plain(value):
    """Tensors / arrays / numpy scalars -> lists and Python scalars (msgpack)."""
    if isinstance(value, dict):
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, torch.Tensor):
        return value.tolist() if value.dim() > 0 else value.item()
    if isinstance(value, np.ndarray):
        return value.tolist()
    if isinstance(value, np.generic):
        return value.item()
    return value


class HipSpine:
    def __init__(
        self,
        env,
        shm_name: str = "/upkie",
        shm_size: int = 1 << 20,
        env_index: int = 0,
        batch_policy: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
    ):
        if not hasattr(env, "get_neutral_action"):
            raise UpkieRuntimeError("HipSpine drives a servo-level env (UpkieServosVecEnv)")
        if not (0 <= env_index < env.num_envs):
            raise UpkieRuntimeError(f"env_index {env_index} outside the batch of {env.num_envs}")
        if env.autoreset_mode == "next_step":
            # the attached robot is only ever reset by its agent (kStart)
            raise UpkieRuntimeError('build the env with autoreset_mode="disabled" or "same_step": the agent owns resets')
        self.env = env
        self.env_index = int(env_index)
        self.batch_policy = batch_policy
        self.interface = AgentInterface(shm_name, shm_size)
        self.state_machine = StateMachine(self.interface)
        self._packer = msgpack.Packer(use_bin_type=True)
        self._observation: dict = {}
        self._obs = None  # last [B, 6, 5] servo observation, input of batch_policy
        self._time = 0.0
        self._interrupted = False
        self.cycles = 0

    def close(self) -> None:
        self.interface.close()

    def interrupt(self) -> None:
        """What SIGINT does to the C++ spine (handle_interrupts.h): shut down
        after `kNbStopCycles` more cycles."""
        self._interrupted = True

    # ------------------------------------------------------------------ cycles
    def _begin_cycle(self) -> None:  # Spine.cpp:143-168
        if self._interrupted:
            self.state_machine.process_event(Event.kInterrupt)
        self.state_machine.process_event(Event.kCycleBeginning)
        state = self.state_machine.state
        try:
            if state == State.kReset:
                config = msgpack.unpackb(self.interface.data(), raw=False)
                self._reset(config if isinstance(config, dict) else {})
            elif state == State.kStep:
                action = msgpack.unpackb(self.interface.data(), raw=False)
                self._step(action if isinstance(action, dict) else {})
        except (msgpack.exceptions.UnpackException, ValueError, UpkieRuntimeError) as exn:  # "Deserialization error", :163-166 (a payload the simulator refuses is treated alike)
            self.state_machine.process_event(Event.kInterrupt)
            self._last_error = exn

    def _end_cycle(self) -> None:  # Spine.cpp:170-181
        if self.state_machine.state in (State.kReset, State.kStep):
            self.interface.write(self._packer.pack(self._observation))
        self.state_machine.process_event(Event.kCycleEnd)

    def cycle(self) -> State:
        """One iteration of Spine::simulate's loop; returns the state after it."""
        self._begin_cycle()
        self._end_cycle()
        self.cycles += 1
        return self.state_machine.state

    def run(self, max_cycles: Optional[int] = None, idle_sleep: float = 0.0) -> None:
        """Serve requests until shut down (`interrupt()`) or `max_cycles`."""
        while self.state_machine.state != State.kOver:
            before = self.interface.request()
            self.cycle()
            if max_cycles is not None and self.cycles >= max_cycles:
                break
            if before == 0:  # nothing asked: let the agent (possibly a thread of this process) run
                time.sleep(idle_sleep)

    # --------------------------------------------------------------- actuation
    def _reset(self, config: dict) -> None:
        """Spine::reset -> BulletInterface::reset with config["bullet"]["reset"]
        (spine_backend.py:206-227 fills it from the agent's RobotState)."""
        env, i = self.env, self.env_index
        reset = (config.get("bullet") or {}).get("reset") or {}
        state = RobotState(
            orientation_base_in_world=reset.get("orientation_base_in_world"),
            position_base_in_world=reset.get("position_base_in_world"),
            linear_velocity_base_to_world_in_world=reset.get("linear_velocity_base_to_world_in_world"),
            angular_velocity_base_in_base=reset.get("angular_velocity_base_in_base"),
            joint_configuration=reset.get("joint_configuration"),
        )
        torque_control = (config.get("bullet") or {}).get("torque_control") or {}
        cfg = env.config
        saved = (list(cfg.init_pos), list(cfg.init_quat), list(cfg.init_linvel), list(cfg.init_angvel), list(cfg.init_joint),
                 cfg.rand_roll, cfg.rand_pitch, cfg.rand_x, cfg.rand_z, cfg.rand_omega_x, cfg.rand_omega_y, list(cfg.rand_linvel))
        if "kp" in torque_control:
            cfg.torque_control_kp = float(torque_control["kp"])
        if "kd" in torque_control:
            cfg.torque_control_kd = float(torque_control["kd"])
        try:
            state.write_to_config(cfg)  # no randomisation: the agent asked for this state
            env.sim.push_config()  # (raises on a state the library refuses, e.g. a quaternion that is not normalized)
            mask = torch.zeros(env.num_envs, dtype=torch.uint8)
            mask[i] = 1
            obs, info = env.reset(mask=mask)
        finally:
            # whatever happened, the rest of the batch keeps the env's own initial-state distribution
            (cfg.init_pos[:], cfg.init_quat[:], cfg.init_linvel[:], cfg.init_angvel[:], cfg.init_joint[:],
             cfg.rand_roll, cfg.rand_pitch, cfg.rand_x, cfg.rand_z, cfg.rand_omega_x, cfg.rand_omega_y, cfg.rand_linvel[:]) = saved
            env.sim.push_config()
        self._obs = obs
        self._time = 0.0
        self._observation = self._spine_observation(info)

    def _step(self, action: dict) -> None:
        env, i = self.env, self.env_index
        if self._obs is None:
            raise ValueError("action before start")
        actions = self.batch_policy(self._obs) if self.batch_policy is not None else env.get_neutral_action()
        actions = torch.as_tensor(actions, dtype=torch.float32).reshape(env.num_envs, 6, 6).clone()
        row = env.get_neutral_action()[i].cpu()  # missing joints / keys keep the neutral command, upkie_servos.py:255-262
        servo = action.get("servo") or {}
        for j, joint in enumerate(JOINT_NAMES):
            command = servo.get(joint) or {}
            for k, key in enumerate(ACTION_KEYS):
                if key in command and command[key] is not None:
                    row[j, k] = float(command[key])
        actions[i] = row.to(actions.device)
        obs, _, _, _, info = env.step(actions)
        self._obs = obs
        self._time += env.dt
        self._observation = self._spine_observation(info)

    def _spine_observation(self, info: dict) -> dict:
        spine = info["spine_observation"]
        spine.materialize()
        i = self.env_index
        out = {}
        for key, block in dict.items(spine):
            out[key] = _index(block, i)
        out["time"] = self._time  # observe_time, observers/observe_time.h
        return _plain(out)


def _index(block, i: int):
    """Env `i` of a nested dictionary of `[B, ...]` tensors."""
    if isinstance(block, dict):
        return {k: _index(v, i) for k, v in block.items()}
    if isinstance(block, torch.Tensor):
        return block[i]
    return block

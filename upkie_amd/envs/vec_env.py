"""Batched Upkie environments (gymnasium.vector semantics, torch tensors).

Same kwargs, observation/action layouts and arithmetic as the reference's
single-robot envs -- ``UpkieServos`` (upkie/envs/upkie_servos.py),
``UpkieGyropod`` (upkie_gyropod.py), ``UpkiePendulum`` (upkie_pendulum.py),
``UpkieBaseVelocity`` (upkie_base_velocity.py) on a ``PyBulletBackend`` --
but for ``num_envs`` independent robots stepped by one kernel launch:

    env = UpkiePendulumVecEnv(num_envs=4096, frequency=200.0)
    obs, info = env.reset(seed=0)                  # obs: [B, 4] on the GPU
    obs, reward, terminated, truncated, info = env.step(actions)   # [B, 1]

Terminated envs are reset by the next step() (gymnasium's NEXT_STEP autoreset),
which ignores their action and returns their reset observation.
"""

from typing import Dict, Optional

import numpy as np
import torch

from .. import abi
from ..exceptions import UpkieException, UpkieRuntimeError
from ..model.joint_properties import JointProperties
from ..model.model import Model
from ..utils.robot_state import RobotState
from .spaces import Box, batch_box
from .external_forces import ExternalForceSet
from .spine_observation import LazySpineObservation


def _default_sim_factory(config, model_struct, device):
    from ..sim import BatchedSim

    return BatchedSim(config, model_struct, device=device)


class _SpineObserverBlocks(dict):
    """Blocks the spine observers write, materialised together with the lazy
    spine observation: filters come out of the observer memory the step kernel
    maintains, BaseOrientation is computed from the IMU block."""

    def __init__(self, sim, observers):
        super().__init__()
        self._sim, self._observers = sim, observers

    def items(self):
        from ..observers import observer_blocks_from_state

        blocks = observer_blocks_from_state(self._sim.observer_state)
        raw = self._sim.observe(update_imu=False)
        blocks.update(self._observers.step(None, raw["imu_orientation"], raw["imu_angular_velocity"]))
        return blocks.items()


class _SameStepInfo(dict):
    """`info` of a SAME_STEP env whose step call completes the autoreset
    itself: one persistent dictionary (``spine_observation``, ``final_obs``);
    ``_final_obs`` = terminated | truncated is computed when it is read (a
    device op per step that most steps of a rollout never look at).

    The key behaves like a stored one under every dictionary protocol --
    iteration, ``len``, ``keys / values / items``, ``dict(info)``, ``{**info}``,
    ``copy.copy``, ``pickle`` (the last four give a plain ``dict`` holding the
    mask as a tensor of its own). Like ``obs``, ``terminated`` and
    ``truncated`` -- persistent buffers the next step rewrites in place -- the
    mask read from THIS object is that of the latest step: a rollout that keeps
    the info of step t takes ``dict(info)`` (a snapshot of the mask) at step t."""

    _LAZY = "_final_obs"

    def __init__(self, terminated, truncated, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._flags = (terminated, truncated)

    def __missing__(self, key):
        if key == self._LAZY:
            return self._flags[0] | self._flags[1]
        raise KeyError(key)

    def __contains__(self, key):
        return key == self._LAZY or super().__contains__(key)

    def __iter__(self):  # (overriding it also takes dict(info) / {**info} off CPython's exact-dict fast path: they go through keys())
        yield from super().__iter__()
        yield self._LAZY

    def __len__(self):
        return super().__len__() + 1

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(self)

    def values(self):
        return [self[k] for k in self]

    def items(self):
        return [(k, self[k]) for k in self]

    def copy(self):
        return dict(self.items())

    __copy__ = copy

    def __reduce__(self):
        return (dict, (self.items(),))

    def __repr__(self):
        return repr(dict(self.items()))


class UpkieVecEnv:
    """Common part of the batched envs: owns the simulation handle, the model
    and the configuration (the role of UpkieEnv + PyBulletBackend,
    upkie_env.py:19-251 / pybullet_backend.py:55-197)."""

    def __init__(
        self,
        num_envs: int = 1,
        device: str = "cuda:0",
        frequency: Optional[float] = 200.0,
        frequency_checks: bool = True,
        init_state: Optional[RobotState] = None,
        regulate_frequency: bool = False,
        max_gain_scale: float = 5.0,
        model: Optional[Model] = None,
        gui: bool = False,
        inertia_variation: float = 0.0,
        joint_properties: Optional[Dict[str, JointProperties]] = None,
        nb_substeps: Optional[int] = None,
        torque_control_kd: float = 1.0,
        torque_control_kp: float = 20.0,
        seed: int = 0,
        autoreset: bool = True,
        autoreset_mode: Optional[str] = None,
        max_episode_steps: Optional[int] = None,
        env_id_offset: int = 0,
        eager_spine_observation: bool = False,
        contact_model: str = "default",
        spine_observers=None,
        sim_factory=None,
        observers_factory=None,
    ):
        if frequency is None:  # upkie_gyropod.py:123-124, upkie_env.py:85-86
            raise UpkieException("This environment needs a loop frequency")
        if regulate_frequency:
            raise UpkieException(
                "regulate_frequency=True sleeps to real time (upkie_env.py:220-221); "
                "a batched simulation is stepped as fast as possible"
            )
        if gui:
            raise UpkieException("the batched simulation has no GUI")
        if not (0.0 < max_gain_scale < 10.0):  # upkie_servos.py:144-145
            raise UpkieRuntimeError(f"Invalid value {max_gain_scale=}")
        self.num_envs = int(num_envs)
        self.model = model if model is not None else Model()
        self.init_state = init_state if init_state is not None else RobotState(
            position_base_in_world=np.array([0.0, 0.0, 0.6])  # upkie_env.py:87-90
        )
        self.frequency = float(frequency)
        self.inertia_variation = float(inertia_variation)
        self.eager_spine_observation = eager_spine_observation
        cfg = abi.default_sim_config(self.num_envs, frequency=self.frequency, nb_substeps=nb_substeps, seed=seed)
        cfg.torque_control_kp = torque_control_kp
        cfg.torque_control_kd = torque_control_kd
        cfg.max_gain_scale = max_gain_scale
        # gymnasium.vector.AutoresetMode: "next_step" (in-kernel: an env flagged
        # done is re-initialised by its next step() and returns the reset
        # observation), "same_step" (the step that ends an episode returns the
        # first observation of the next one, the last observation of the old
        # one in info["final_obs"]), "disabled" (reset(mask=...) by hand)
        if autoreset_mode is None:
            autoreset_mode = "next_step" if autoreset else "disabled"
        if autoreset_mode not in ("next_step", "same_step", "disabled"):
            raise UpkieException(f"unknown autoreset_mode '{autoreset_mode}'")
        self.autoreset_mode = autoreset_mode
        # gymnasium's TimeLimit for a batch: `truncated` after this many steps
        # of an episode (the reference registers no limit, envs/__init__.py:38-44)
        self.max_episode_steps = None if max_episode_steps is None else int(max_episode_steps)
        cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP if autoreset_mode == "next_step" else abi.AUTORESET_DISABLED
        cfg.max_episode_steps = 0 if self.max_episode_steps is None else self.max_episode_steps
        cfg.env_id_offset = env_id_offset
        for idx, name in enumerate(abi.JOINT_NAMES):
            props = (joint_properties or {}).get(name, JointProperties())
            cfg.joint_friction[idx] = props.friction
            cfg.torque_control_noise[idx] = props.torque_control_noise
            cfg.torque_measurement_noise[idx] = props.torque_measurement_noise
        self.init_state.write_to_config(cfg)
        self.config = cfg
        self._configure(cfg)  # wrapper-specific fields
        factory = sim_factory if sim_factory is not None else _default_sim_factory
        self.sim = factory(cfg, self.model.struct, device)
        self.device = self.sim.device
        # "default": the product's contact specification; "bullet_like": what pybullet.stepSimulation() is published to
        # do (persistent 4-point manifolds, 50 fixed sweeps, cone friction: `BatchedSim.use_bullet_like_contacts`), slower
        if contact_model not in ("default", "bullet_like"):
            raise UpkieException(f"unknown contact_model '{contact_model}'")
        self.contact_model = contact_model
        if contact_model == "bullet_like":
            if spine_observers:
                raise UpkieException("the Bullet-like contact model does not run the in-step spine observers")
            self.sim.use_bullet_like_contacts()
        if abs(self.inertia_variation) > 1e-10:  # pybullet_backend.py:178-179
            self.sim.randomize_inertias(self.inertia_variation)
        self._spine = LazySpineObservation(self.sim)
        self._external_forces = ExternalForceSet(self.model, self.num_envs)
        # Optional spine observer pipeline (upkie/cpp/observers, order of
        # spines/common/observers.h:22-42); `spine_observers` is True or a spine
        # configuration dictionary (spine_backend.py:77-105). FloorContact, its
        # WheelContact estimators and WheelOdometry carry filters: they run
        # INSIDE the step kernel, one observer cycle per physics substep, which
        # is the spine's rate (1 kHz) under this env's agent rate, exactly as
        # Spine::simulate cycles nb_substeps times per action. BaseOrientation
        # is stateless and evaluated with the observation, on request.
        self._observers = None
        self._observer_config = None
        if spine_observers:
            from ..observers import BatchedObservers, observer_config_from_spine_config

            spine_config = spine_observers if isinstance(spine_observers, dict) else None
            substep = self.dt / int(cfg.nb_substeps)
            self._observer_config = observer_config_from_spine_config(self.num_envs, substep, spine_config)
            self.sim.attach_observers(self._observer_config)
            make = observers_factory if observers_factory is not None else BatchedObservers
            self._observers = make(self._observer_config, device)  # used for its stateless BaseOrientation stage

    # hooks ------------------------------------------------------------
    def _configure(self, cfg) -> None:
        pass

    # gymnasium.Env-like attributes -------------------------------------
    @property
    def dt(self) -> float:
        """Control period in seconds (upkie_env.py:113-119)."""
        return 1.0 / self.frequency

    @property
    def unwrapped(self):
        return self

    @property
    def observation(self) -> torch.Tensor:
        """The persistent device buffer `reset()` and `step()` return as `obs`
        (rewritten in place by every step): what a graph-captured policy reads
        (`upkie_amd.graphs.GraphedEnvStep`)."""
        sim = self.sim
        if self._stepper_kind == "servos" and sim.obs_servos is None:
            sim.obs_servos = torch.zeros((self.num_envs, 6, 5), dtype=torch.float32, device=self.device)
        return {"pendulum": sim.obs4, "gyropod": sim.obs6, "servos": sim.obs_servos}.get(self._stepper_kind, getattr(sim, "obs3", None))

    def close(self) -> None:
        if self._observers is not None:
            self._observers.close()
        self._disarm_same_step()
        self.sim.close()

    def _disarm_same_step(self) -> None:
        """Switch the in-step SAME_STEP autoreset off again (it is armed on `sim`, which other wrappers -- `HipSpine`,
        a `Backend` around the same handle, direct `sim.step_*` calls -- may keep using after this env)."""
        if self._final_obs is not None and hasattr(self.sim, "set_final_observation"):
            self.sim.set_final_observation(None)
        self._final_obs = None
        self._step_out = None

    def __enter__(self):
        return self

    def __exit__(self, *args) -> bool:
        self.close()
        return False

    #: `abi.OBSERVATION_*` layout of the step's observation buffer when a SAME_STEP
    #: autoreset can run as one launch on it (None: through `reset(mask=done)`)
    _same_step_layout = None
    _final_obs = None

    def _info(self) -> dict:
        self._spine.invalidate()
        if self._observers is not None:
            self._spine.set_overrides(_SpineObserverBlocks(self.sim, self._observers))
        if self.eager_spine_observation:
            self._spine.materialize()
        return {"spine_observation": self._spine}

    #: kind of `BatchedSim.stepper` behind `step()` (None: the wrapper composes its step from several calls)
    _stepper_kind = None
    _stepper = None
    _step_out = None

    def _fast_step(self, action, shape):
        """`step()` of the fused env kinds: one ctypes call on cached addresses
        (`BatchedSim.stepper`), the outputs returned as ONE cached tuple of the
        handle's persistent buffers -- `obs`, `reward`, `terminated`, `truncated`
        are rewritten in place by every step, `info` is one dictionary whose
        ``spine_observation`` materialises on first access (a rollout buffer
        copies what it keeps, as with any vector env that reuses its buffers)."""
        step = self._stepper
        if step is None:
            if not hasattr(self.sim, "stepper"):  # (test doubles)
                return None
            step = self._stepper = self.sim.stepper(self._stepper_kind)
        if not (type(action) is torch.Tensor and action.dtype is torch.float32 and action.device == self.device and action.is_contiguous()
                and action.numel() == self.num_envs * self._action_words):
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(shape).contiguous()
        step(action.data_ptr())
        out = self._step_out
        if out is None or self._observers is not None or self.eager_spine_observation or (self.autoreset_mode == "same_step" and self._final_obs is None):
            sim = self.sim
            obs = {"pendulum": sim.obs4, "gyropod": sim.obs6, "servos": sim.obs_servos}[self._stepper_kind]
            out = self._finish_step(obs, sim.reward, sim.terminated, sim.truncated)
            if self._observers is None and not self.eager_spine_observation and (self.autoreset_mode != "same_step" or self._final_obs is not None):
                if self.autoreset_mode == "same_step":
                    if self._same_step_layout is None or not hasattr(self.sim, "set_final_observation"):
                        return out  # (composed through reset(mask): nothing to cache)
                    info = _SameStepInfo(out[2], out[3], spine_observation=self._spine, final_obs=self._final_obs)
                    out = (out[0], out[1], out[2], out[3], info)
                self._step_out = out
            return out
        self._spine._fresh = False
        return out

    def _finish_step(self, obs, reward, terminated, truncated):
        # the kernels write 0/1 bytes: reinterpreting them as bool launches nothing
        terminated = terminated.view(torch.bool) if terminated.dtype == torch.uint8 else terminated.bool()
        truncated = truncated.view(torch.bool) if truncated.dtype == torch.uint8 else truncated.bool()
        # (the time limit lives in the kernel: `truncated` and the DONE word come from the step itself)
        if self.autoreset_mode != "same_step":
            return obs, reward, terminated, truncated, self._info()
        done = terminated | truncated
        if self._same_step_layout is not None and hasattr(self.sim, "set_final_observation"):
            # the step call itself restarted the envs whose DONE word it set and kept every env's last observation
            # aside (inside the same launch up to 8192 envs): armed on the first step, see below
            if self._final_obs is None:
                self._final_obs = obs.clone()
                self.sim.set_final_observation(self._final_obs)
                self.sim.autoreset_done(self._same_step_layout, obs, self._final_obs)  # this first step's own resets
            info = dict(self._info())
            # NOTE: one persistent buffer, rewritten by every step (the kernel stores into it): a rollout buffer that
            # keeps final observations must copy the rows it needs (`info["final_obs"][info["_final_obs"]]` does)
            info["final_obs"] = self._final_obs
            info["_final_obs"] = done
            return obs, reward, terminated, truncated, info
        final_obs = obs.clone()
        obs, info = self.reset(mask=done)  # untouched envs report their current observation
        info = dict(info)
        info["final_obs"] = final_obs
        info["_final_obs"] = done
        return obs, reward, terminated, truncated, info

    def update_init_rand(self, **kwargs) -> None:
        """upkie_env.py:244-251."""
        self.init_state.randomization.update(**kwargs)
        self.init_state.write_to_config(self.config)
        self.sim.push_config()

    def set_external_forces(self, external_forces, forces: Optional[torch.Tensor] = None) -> None:
        """``set_external_forces({link: ExternalForce | (force, local) | force})``
        as PyBulletBackend.set_external_forces (pybullet_backend.py:603-658):
        forces on any link, world or link frame, applied at the link's centre
        of mass at every substep until the link is given another force; a
        force is a 3-vector (all envs) or ``[B, 3]`` (one per env). The
        two-argument form ``set_external_forces(link_name, forces)`` sets one
        world-frame force (``None``: remove every force)."""
        if isinstance(external_forces, str):
            if forces is None:
                self._external_forces.clear()
            else:
                self._external_forces.update({external_forces: (forces, False)})
        else:
            self._external_forces.update(external_forces)
        self._external_forces.push(self.sim)

    def contact_points(self) -> torch.Tensor:
        """Tire/floor contact points of every env, ``[B, 2, 8]`` (left, right
        tire): ``[exists, position in world (3), force in world (3), 0]`` --
        PyBulletBackend.get_contact_points (pybullet_backend.py:660-716) for
        the batch; `sim.get_contact_points(link_name, env)` gives the
        reference's list of `PointContact` for one env."""
        return self.sim.contact_points()

    #: True for the single-robot envs: the initial state is drawn on the host
    #: from gymnasium's seeded generator exactly as the reference draws it
    #: (upkie_env.py:180-190), so `reset(seed=s)` starts from the reference's
    #: state; batched envs draw on the device (Philox keyed by seed, env, episode)
    host_sampling = False
    _np_random = None

    def _reset_sim(self, seed: Optional[int], mask: Optional[torch.Tensor]) -> torch.Tensor:
        if seed is not None:
            self.config.seed = int(seed)
            self.sim.push_config()
            if mask is None:
                # gymnasium's reset(seed) contract: the same seed replays the same
                # episodes, so the counters that key the random streams restart too
                self.sim.restart_random_streams()
        if self.host_sampling and mask is None:
            if seed is not None or self._np_random is None:
                # gymnasium.utils.seeding.np_random: Generator(PCG64(SeedSequence(seed))). A first reset() without a
                # seed falls back to the constructor's `seed=` (which already keys the device's noise streams), so that
                # envs.make(..., seed=s) followed by reset() is reproducible as it was with the device-side sampler;
                # gymnasium itself would draw OS entropy there.
                self._np_random = np.random.default_rng(seed if seed is not None else int(self.config.seed))
            self.sampled_init_state = self.init_state.sample_state(self._np_random)
            self.sampled_init_state.write_exact_to_config(self.config)
            self.sim.push_config()
            try:
                return self.sim.reset(mask)
            finally:
                self.init_state.write_to_config(self.config)
                self.sim.push_config()
        return self.sim.reset(mask)


class UpkiePendulumVecEnv(UpkieVecEnv):
    """Batched ``UpkiePendulum``: action [ground velocity], observation
    [pitch, ground position, pitch rate, ground velocity]
    (upkie_pendulum.py:20-60)."""

    _same_step_layout = abi.OBSERVATION_PENDULUM

    def __init__(self, num_envs: int = 1, fall_pitch: float = 1.0, max_ground_velocity: float = 3.0, **kwargs):
        self.fall_pitch = fall_pitch
        self.max_ground_velocity = max_ground_velocity
        super().__init__(num_envs=num_envs, **kwargs)
        self._pendulum_obs_indices = torch.tensor([1, 0, 4, 3], device=self.device)  # _PENDULUM_OBS_INDICES, upkie_pendulum.py:17
        # upkie_gyropod.py:126-142 limits, permuted by _PENDULUM_OBS_INDICES = [1, 0, 4, 3]
        obs_limit = np.array([np.pi, np.inf, 1000.0, max_ground_velocity], dtype=np.float32)
        act_limit = np.array([max_ground_velocity], dtype=np.float32)
        self.single_observation_space = Box(-obs_limit, +obs_limit, shape=obs_limit.shape, dtype=np.float32)
        self.single_action_space = Box(-act_limit, +act_limit, shape=act_limit.shape, dtype=np.float32)
        self.observation_space = batch_box(self.single_observation_space, self.num_envs)
        self.action_space = batch_box(self.single_action_space, self.num_envs)

    def _configure(self, cfg) -> None:
        cfg.fall_pitch = self.fall_pitch
        cfg.max_ground_velocity = self.max_ground_velocity

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None, mask: Optional[torch.Tensor] = None):
        obs6 = self._reset_sim(seed, mask)
        torch.index_select(obs6, 1, self._pendulum_obs_indices, out=self.sim.obs4)  # upkie_pendulum.py:17,122
        return self.sim.obs4, self._info()  # (the buffer step() rewrites: `obs = env.step(policy(obs))[0]` stays on one tensor)

    _stepper_kind, _action_words = "pendulum", 1
    _linear_policy_gains = None
    _agent_stepper = None

    def step(self, action):
        out = self._fast_step(action, (self.num_envs,))
        if out is not None:
            return out
        obs, reward, terminated, truncated = self.sim.step_pendulum(action)  # (converted / reshaped to [B] there)
        return self._finish_step(obs, reward, terminated, truncated)

    def step_linear_policy(self, gains=None, clip: Optional[float] = None):
        """``env.step(clamp(gains . obs, -clip, clip))`` with the policy evaluated INSIDE the step's launch, on the
        observation the previous step (or `reset`) left in the env's observation buffer: the README agent
        (README.md:60-67, examples/pybullet/pd_balancing.py) costs no launch of its own between two steps
        (`upkie_sim_step_pendulum_agent`). ``gains`` (four numbers, over [pitch, position, pitch rate, velocity]) and
        ``clip`` default to what the config holds (README's [10, 1, 0, 0.1], 0.99); they are handed to the library
        when they change: host sequences (lists, tuples, numpy arrays) are compared VALUE BY VALUE on every call
        (four floats; mutating the same list in place is seen); a DEVICE tensor is read back -- a synchronisation --
        only when it is a different object from the previous call's (the identity shortcut: write new gains into a new
        tensor, or hand over host numbers). Same return value as `step`. NEXT_STEP or disabled autoreset (a SAME_STEP
        env steps with `step(policy(obs))`)."""
        if self.autoreset_mode == "same_step" or not hasattr(self.sim, "step_pendulum_agent"):
            raise UpkieException("step_linear_policy runs under NEXT_STEP or disabled autoreset; use step(policy(obs))")
        cfg, changed = self.sim.config, False
        on_device = isinstance(gains, torch.Tensor) and gains.device.type != "cpu"
        if gains is not None and not (on_device and gains is self._linear_policy_gains):
            values = [float(g) for g in (gains.tolist() if hasattr(gains, "tolist") else gains)]
            if len(values) != 4:
                raise UpkieException("a linear policy over the Pendulum observation has four gains")
            if list(cfg.agent_gains) != values:
                cfg.agent_gains[:] = values
                changed = True
            remember = gains if on_device else None
        else:
            remember = self._linear_policy_gains
        if clip is not None and float(cfg.agent_clip) != float(clip):
            cfg.agent_clip = float(clip)
            changed = True
        if changed:
            self.sim.push_config()
        self._linear_policy_gains = remember  # (only once the gains were validated and handed over)
        agent_step = self._agent_stepper
        if agent_step is None and hasattr(self.sim, "stepper"):
            try:
                agent_step = self._agent_stepper = self.sim.stepper("pendulum_agent")  # one ctypes call on cached addresses
            except KeyError:  # (test doubles without this kind)
                agent_step = self._agent_stepper = False
        if agent_step:
            agent_step()
            sim = self.sim
            obs, reward, terminated, truncated = sim.obs4, sim.reward, sim.terminated, sim.truncated
        else:
            obs, reward, terminated, truncated = self.sim.step_pendulum_agent()
        out = self._step_out
        if out is None or self._observers is not None or self.eager_spine_observation:
            out = self._finish_step(obs, reward, terminated, truncated)
            if self._observers is None and not self.eager_spine_observation:
                self._step_out = out  # (the same persistent buffers `step` returns)
            return out
        self._spine._fresh = False
        return out


class UpkieGyropodVecEnv(UpkieVecEnv):
    """Batched ``UpkieGyropod``: action [ground velocity, yaw velocity],
    observation [ground position, pitch, yaw, ground velocity, pitch rate,
    yaw velocity] (upkie_gyropod.py:20-98)."""

    _same_step_layout = abi.OBSERVATION_GYROPOD

    def __init__(
        self,
        num_envs: int = 1,
        fall_pitch: float = 1.0,
        leg_gain_scale: float = 1.0,
        max_ground_velocity: float = 3.0,
        max_yaw_velocity: float = 1.0,
        **kwargs,
    ):
        self.fall_pitch = fall_pitch
        self._leg_gain_scale = leg_gain_scale
        self.max_ground_velocity = max_ground_velocity
        self.max_yaw_velocity = max_yaw_velocity
        super().__init__(num_envs=num_envs, **kwargs)
        obs_limit = np.array(  # upkie_gyropod.py:126-142
            [np.inf, np.pi, np.inf, max_ground_velocity, 1000.0, max_yaw_velocity], dtype=np.float32
        )
        act_limit = np.array([max_ground_velocity, max_yaw_velocity], dtype=np.float32)
        self.single_observation_space = Box(-obs_limit, +obs_limit, shape=obs_limit.shape, dtype=np.float32)
        self.single_action_space = Box(-act_limit, +act_limit, shape=act_limit.shape, dtype=np.float32)
        self.observation_space = batch_box(self.single_observation_space, self.num_envs)
        self.action_space = batch_box(self.single_action_space, self.num_envs)

    def _configure(self, cfg) -> None:
        cfg.fall_pitch = self.fall_pitch
        cfg.leg_gain_scale = self._leg_gain_scale
        cfg.max_ground_velocity = self.max_ground_velocity
        cfg.max_yaw_velocity = self.max_yaw_velocity

    @property
    def leg_gain_scale(self) -> float:
        return self._leg_gain_scale

    def set_leg_gain_scale(self, leg_gain_scale: float) -> None:
        """upkie_gyropod.py:178-184."""
        self._leg_gain_scale = leg_gain_scale
        self.config.leg_gain_scale = leg_gain_scale
        self.sim.push_config()

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None, mask: Optional[torch.Tensor] = None):
        obs6 = self._reset_sim(seed, mask)
        return obs6, self._info()

    def _gyropod_step(self, action):
        act = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(self.num_envs, 2)
        return self.sim.step_gyropod(act)

    _stepper_kind, _action_words = "gyropod", 2

    def step(self, action):
        out = self._fast_step(action, (self.num_envs, 2)) if self._stepper_kind is not None else None
        if out is not None:
            return out
        return self._finish_step(*self._gyropod_step(action))


class UpkieServosVecEnv(UpkieVecEnv):
    """Batched ``UpkieServos``: action ``[B, 6, 6]`` = joints x ACTION_KEYS
    (position, velocity, feedforward_torque, kp_scale, kd_scale,
    maximum_torque), observation ``[B, 6, 5]`` = joints x (position, velocity,
    torque, temperature, voltage) (upkie_servos.py:20-306). Never terminates
    on its own (upkie_env.py:231-238)."""

    _same_step_layout = abi.OBSERVATION_SERVOS

    def __init__(self, num_envs: int = 1, **kwargs):
        super().__init__(num_envs=num_envs, **kwargs)
        m = self.model.struct
        lo = np.zeros((6, 6), dtype=np.float32)
        hi = np.zeros((6, 6), dtype=np.float32)
        olo = np.zeros((6, 5), dtype=np.float32)
        ohi = np.zeros((6, 5), dtype=np.float32)
        neutral = np.zeros((6, 6), dtype=np.float32)
        for j in range(6):  # upkie_servos.py:173-278
            eff, vel = m.joint_effort[j], m.joint_velocity[j]
            lo[j] = [m.joint_lower[j], -vel, -eff, 0.0, 0.0, 0.0]
            hi[j] = [m.joint_upper[j], vel, eff, self.config.max_gain_scale, self.config.max_gain_scale, eff]
            olo[j] = [m.joint_lower[j], -vel, -eff, 0.0, 10.0]
            ohi[j] = [m.joint_upper[j], vel, eff, 100.0, 44.0]
            neutral[j] = [np.nan, 0.0, 0.0, 1.0, 1.0, eff]  # upkie_servos.py:255-262
        self._neutral = neutral
        self.single_action_space = Box(lo, hi, shape=(6, 6), dtype=np.float32)
        self.single_observation_space = Box(olo, ohi, shape=(6, 5), dtype=np.float32)
        self.observation_space = batch_box(self.single_observation_space, self.num_envs)
        self.action_space = batch_box(self.single_action_space, self.num_envs)

    def get_neutral_action(self) -> torch.Tensor:
        """``[B, 6, 6]`` action where servos do not move (upkie_servos.py:308-314)."""
        return torch.from_numpy(self._neutral).to(self.device).expand(self.num_envs, 6, 6).clone()

    def _servo_obs(self) -> torch.Tensor:
        obs = torch.empty((self.num_envs, 6, 5), dtype=torch.float32, device=self.device)
        st = self.sim.state
        obs[:, :, 0] = st[abi.S_Q : abi.S_Q + 6].t()
        obs[:, :, 1] = st[abi.S_QD : abi.S_QD + 6].t()
        obs[:, :, 2] = st[abi.S_TORQUE : abi.S_TORQUE + 6].t()
        obs[:, :, 3] = 42.0
        obs[:, :, 4] = 18.0
        return obs

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None, mask: Optional[torch.Tensor] = None):
        self._reset_sim(seed, mask)
        obs = self._servo_obs()
        if hasattr(self.sim, "obs_servos"):
            if self.sim.obs_servos is None:  # (allocated here, not by the first step: `env.observation` holds the reset observation from the start)
                self.sim.obs_servos = torch.zeros((self.num_envs, 6, 5), dtype=torch.float32, device=self.device)
            if self.sim.obs_servos.shape == obs.shape:
                self.sim.obs_servos.copy_(obs)
                obs = self.sim.obs_servos  # (the buffer step() rewrites)
        return obs, self._info()

    _stepper_kind, _action_words = "servos", 36

    def step(self, action):
        out = self._fast_step(action, (self.num_envs, 6, 6))
        if out is not None:
            return out
        act = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(self.num_envs, 6, 6)
        obs, reward, terminated, truncated = self.sim.step_servos(act)
        return self._finish_step(obs, reward, terminated, truncated)

    def step_servo_policy(self, policy: "abi.UpkieServoPolicy"):
        """`step` with the action computed on the device by a servo-level law (`abi.torque_balancing_policy`:
        examples/pybullet/torque_balancing.py:15-37; `abi.velocity_balancing_policy`: the README balancer through the
        wheels' velocity loop) from the state the step starts from -- inside the step's own launch on eight lanes per
        env (`upkie_sim_step_servos_policy`), a launch in front of it otherwise. The law flags robots that fell past
        its `fall_pitch` for the NEXT_STEP autoreset. Same return value as `step`; NEXT_STEP or disabled autoreset."""
        if self.autoreset_mode == "same_step" or not hasattr(self.sim, "step_servos_policy"):
            raise UpkieException("step_servo_policy runs under NEXT_STEP or disabled autoreset; use step(action)")
        obs, reward, terminated, truncated = self.sim.step_servos_policy(policy)
        out = self._step_out
        if out is None or out[0] is not obs or self._observers is not None or self.eager_spine_observation:
            out = self._finish_step(obs, reward, terminated, truncated)
            if self._observers is None and not self.eager_spine_observation:
                self._step_out = out
            return out
        self._spine._fresh = False
        return out


class UpkieBaseVelocityVecEnv(UpkieGyropodVecEnv):
    """Batched ``UpkieBaseVelocity``: action [linear velocity, yaw velocity]
    goes through the MPC balancer, observation is the dead-reckoned SE(2) pose
    [x, y, yaw] (upkie_base_velocity.py:21-202)."""

    _same_step_layout = None  # the MPC workspace and the dead-reckoned pose restart with the env: through reset(mask)
    _stepper_kind = None  # (its step is the balancer + the Gyropod step: composed below)

    def __init__(
        self,
        num_envs: int = 1,
        fall_pitch: float = 1.0,
        max_ground_velocity: float = 3.0,
        max_yaw_velocity: float = 1.0,
        leg_length: float = 0.58,
        max_ground_accel: float = 10.0,
        nb_timesteps: int = 50,
        mpc_factory=None,
        **kwargs,
    ):
        super().__init__(
            num_envs=num_envs,
            fall_pitch=fall_pitch,
            max_ground_velocity=max_ground_velocity,
            max_yaw_velocity=max_yaw_velocity,
            **kwargs,
        )
        obs_limit = np.full(3, np.inf, dtype=np.float32)
        self.single_observation_space = Box(-obs_limit, +obs_limit, shape=(3,), dtype=np.float32)
        self.observation_space = batch_box(self.single_observation_space, self.num_envs)
        mpc_cfg = abi.default_mpc_config(self.num_envs, nb_timesteps)
        mpc_cfg.fall_pitch = fall_pitch
        mpc_cfg.leg_length = leg_length
        mpc_cfg.max_ground_accel = max_ground_accel
        mpc_cfg.max_ground_velocity = max_ground_velocity
        if mpc_factory is None:
            from ..mpc import BatchedMpc

            mpc_factory = BatchedMpc
        self.mpc_balancer = mpc_factory(mpc_cfg, device=str(self.device))
        B = self.num_envs
        self._x0 = torch.zeros((B, 4), dtype=torch.float32, device=self.device)
        self._contact = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self._xy = torch.zeros((B, 2), dtype=torch.float32, device=self.device)
        # balancer and step in ONE launch (upkie_sim_step_base_velocity_mpc) instead of two, same bits: on by default where
        # the eight-lane kernel steps the batch (up to 16384 envs: a wavefront solves the QPs of its eight envs as one
        # half-filled MFMA tile in front of their step); off on the two-lane mapping, where it measured SLOWER (37.9 vs
        # 35.6 us at 16384 envs: a wavefront that steps 32 envs solves two tiles one after the other, the separate kernel
        # spreads them over twice as many wavefronts; profiles/r02_secondary_configs_c.jsonl)
        self.fuse_mpc = getattr(self.sim, "lanes_per_env", 0) == 8

    def _remember(self, obs6: torch.Tensor) -> None:
        # MPC state [ground position, pitch, ground velocity, pitch rate] of the
        # LAST observation (upkie_base_velocity.py:185-187,195)
        self._x0.copy_(obs6[:, [0, 1, 3, 4]])
        self._contact.copy_((self.sim.state[abi.S_CONTACT] != 0).to(torch.uint8))

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None, mask: Optional[torch.Tensor] = None):
        obs6, info = super().reset(seed=seed, options=options, mask=mask)
        self._remember(obs6)
        self.mpc_balancer.reset(mask)  # upkie_base_velocity.py:158
        if hasattr(self.sim, "step_base_velocity") and hasattr(self.mpc_balancer, "step_env"):
            # fused path (the condition step() takes it on): the dead-reckoned pose lives in the state words (the
            # reset branch of the kernel has just zeroed them for the reset envs); the generic composition below
            # dead-reckons in `_xy` and never touches those words
            self._xy.copy_(self.sim.state[abi.S_SE2_X : abi.S_SE2_Y + 1].t())
        elif mask is None:
            self._xy.zero_()
        else:
            self._xy.masked_fill_(mask.to(self.device).bool()[:, None], 0.0)
        # reset envs start at the origin (upkie_base_velocity.py:160-162);
        # untouched ones report their current dead-reckoned pose
        obs = torch.cat([self._xy, obs6[:, 2:3].to(self._xy.dtype)], dim=1)
        return obs, info

    def step(self, action):
        act = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(self.num_envs, 2).contiguous()
        if self.fuse_mpc and hasattr(self.sim, "step_base_velocity_mpc") and hasattr(self.mpc_balancer, "_handle"):
            # balancer and step in one call (one launch with two lanes per env and a horizon <= 16), everything on device
            obs, reward, terminated, truncated = self.sim.step_base_velocity_mpc(self.mpc_balancer, act, self._x0, self._contact)
            return self._finish_step(obs, reward, terminated, truncated)
        if hasattr(self.sim, "step_base_velocity") and hasattr(self.mpc_balancer, "step_env"):
            # fused path: two launches per env.step(), everything stays on device
            done = self.sim.state[abi.S_DONE] if self.config.autoreset_mode else None
            commanded = self.mpc_balancer.step_env(self._x0, act, self._contact, done, self.dt)
            obs, reward, terminated, truncated = self.sim.step_base_velocity(act, commanded, self._x0, self._contact)
            return self._finish_step(obs, reward, terminated, truncated)
        # generic composition (used with the CPU test doubles)
        linear_velocity, yaw_velocity = act[:, 0].contiguous(), act[:, 1]
        autoreset = (self.sim.state[abi.S_DONE] != 0) if self.config.autoreset_mode else None
        ground_velocity, _ = self.mpc_balancer.step(self._x0, linear_velocity, self._contact, self.dt)
        gyropod_action = torch.stack([ground_velocity, yaw_velocity], dim=1)
        obs6, reward, terminated, truncated = self._gyropod_step(gyropod_action)
        self._remember(obs6)
        yaw = obs6[:, 2]
        if autoreset is not None and bool(autoreset.any()):
            self.mpc_balancer.reset(autoreset.to(torch.uint8))
            self._xy[autoreset] = 0.0
            live = ~autoreset
            self._xy[live, 0] += (linear_velocity * torch.cos(yaw) * self.dt)[live]  # :197-199
            self._xy[live, 1] += (linear_velocity * torch.sin(yaw) * self.dt)[live]
        else:
            self._xy[:, 0] += linear_velocity * torch.cos(yaw) * self.dt  # :197-199
            self._xy[:, 1] += linear_velocity * torch.sin(yaw) * self.dt
        obs = torch.cat([self._xy, yaw[:, None]], dim=1)
        return self._finish_step(obs, reward, terminated, truncated)

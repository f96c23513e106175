"""Batched simulation handle: PyTorch tensors for device memory and streams,
the HIP library for every bit of arithmetic.

This is the batched counterpart of the reference's ``PyBulletBackend``
(upkie/envs/backends/pybullet_backend.py:31): one instance owns B
independent robots living in one ``[STATE_WORDS, B]`` fp32 tensor.
"""

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import abi, lib
from .exceptions import UpkieRuntimeError
from .model.default_model import default_model


def _ptr(t: Optional[torch.Tensor]):
    # (a plain integer: ctypes turns it into the c_void_p the argtypes ask for, without an object per argument)
    return t.data_ptr() if t is not None else None


# the raw handle of torch's current stream on a device without building a torch.cuda.Stream object per call
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class BatchedSim:
    """B robots stepped by one HIP kernel launch per ``env.step()``."""

    def __init__(
        self,
        config: abi.UpkieSimConfig,
        model: Optional[abi.UpkieModel] = None,
        device: str = "cuda:0",
    ):
        if not torch.cuda.is_available():
            raise UpkieRuntimeError(
                "no HIP device visible: the batched simulation only runs on "
                "a GPU (there is no CPU fallback)"
            )
        self._lib = lib.load()
        self.device = torch.device(device)
        self._device_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.config = config
        self.model = model if model is not None else default_model()
        self.num_envs = int(config.num_envs)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            status = self._lib.upkie_sim_create(
                C.byref(self.config), C.byref(self.model), C.byref(self._handle)
            )
        lib.check(status, None)
        B = self.num_envs
        nbytes = self._lib.upkie_sim_state_bytes(self._handle)
        assert nbytes == abi.STATE_WORDS * B * 4
        self.state = torch.zeros(
            (abi.STATE_WORDS, B), dtype=torch.float32, device=self.device
        )
        self.state[abi.S_QUAT] = 1.0
        self.reward = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.terminated = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self.truncated = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self.obs4 = torch.zeros((B, 4), dtype=torch.float32, device=self.device)
        self.obs6 = torch.zeros((B, 6), dtype=torch.float32, device=self.device)
        self.obs_servos = None
        self.body_inertials = None  # [70, B] per-env inertial records (randomize_inertias)
        self.link_scale = None  # [MAX_LINKS, B] the factors drawn per URDF link
        self.ext_force = None
        self._ext_point = (C.c_double * 3)(0.0, 0.0, 0.0)

    # ------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle:
            self._lib.upkie_sim_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        if _raw_stream is not None:
            return _raw_stream(self._device_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check(self, status: int) -> None:
        if status < 0:
            lib.check(status, self._handle)

    def _launch(self, fn, *args) -> None:
        """One call into the library on torch's current stream of this handle's device. A Python-level RL loop pays
        this once per `env.step()`: no device context manager when the device is already current, no stream object."""
        if torch.cuda.current_device() == self._device_index:
            status = fn(self._handle, *args, self._stream())
        else:
            with torch.cuda.device(self.device):
                status = fn(self._handle, *args, self._stream())
        if status < 0:
            lib.check(status, self._handle)

    def push_config(self) -> None:
        """Hand the (mutated) ``self.config`` to the library."""
        self._check(self._lib.upkie_sim_set_config(self._handle, C.byref(self.config)))

    # ---------------------------------------------------------- randomise
    def randomize_inertias(self, inertia_variation: float) -> torch.Tensor:
        """`PyBulletBackend.randomize_inertias` (pybullet_backend.py:571-601)
        for every env: one factor 1 + U(-v, v) per URDF link (kept in
        ``self.link_scale``) scales that link's mass and inertia; returns the
        per-env inertial records of the 7 composite bodies the links are fused
        into, ``[NB * INERTIAL_WORDS, B]`` (row ``10 * body + word``: mass,
        centre of mass, inertia about it)."""
        if self.body_inertials is None:
            self.body_inertials = torch.zeros((abi.NB * abi.INERTIAL_WORDS, self.num_envs), dtype=torch.float32, device=self.device)
            self.link_scale = torch.ones((abi.MAX_LINKS, self.num_envs), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(
                self._lib.upkie_sim_sample_body_inertials(
                    self._handle,
                    _ptr(self.body_inertials),
                    _ptr(self.link_scale),
                    float(inertia_variation),
                    self._stream(),
                )
            )
        self._push_randomization()
        return self.body_inertials

    def set_body_inertials(self, body_inertials: Optional[torch.Tensor]) -> None:
        """Install caller-made inertial records ``[NB * INERTIAL_WORDS, B]``
        (None: back to the model's)."""
        if body_inertials is not None:
            body_inertials = body_inertials.to(self.device, torch.float32).contiguous()
            if tuple(body_inertials.shape) != (abi.NB * abi.INERTIAL_WORDS, self.num_envs):
                raise ValueError(f"body_inertials must be [{abi.NB * abi.INERTIAL_WORDS}, {self.num_envs}]")
        self.body_inertials = body_inertials
        self._push_randomization()

    def set_external_force(self, force: Optional[torch.Tensor], point=(0.0, 0.0, 0.0)):
        """World-frame force ``[3, B]`` on the trunk at base-frame ``point``,
        held until overwritten (pybullet_backend.py:603-658)."""
        if force is None:
            self.set_external_forces(None)
        else:
            if tuple(force.shape) != (3, self.num_envs):
                raise ValueError(  # external_force.py:38-41
                    f"force must have shape (3, {self.num_envs})"
                )
            self.set_external_forces(force[None], bodies=[0], points=[point], local=[False])

    def set_external_forces(self, forces: Optional[torch.Tensor], bodies=(), points=(), local=()):
        """Forces ``[count, 3, B]`` acting at the same time, force `i` on
        composite body ``bodies[i]`` (0 trunk, 1-3 left thigh/calf/wheel, 4-6
        right) at ``points[i]`` of that body's frame, in the world frame or,
        with ``local[i]``, in the body frame; re-applied at every substep until
        replaced (pybullet_backend.py:603-658). None removes them."""
        slots = abi.UpkieExternalForces()
        if forces is not None:
            count = len(bodies)
            if count > abi.MAX_EXTERNAL_FORCES:
                raise ValueError(f"at most {abi.MAX_EXTERNAL_FORCES} external forces at a time")
            if tuple(forces.shape) != (count, 3, self.num_envs) or len(points) != count or len(local) != count:
                raise ValueError(f"forces must have shape ({count}, 3, {self.num_envs})")
            forces = forces.to(self.device, torch.float32).contiguous()
            slots.count = count
            for i in range(count):
                slots.body[i] = int(bodies[i])
                slots.local[i] = 1 if local[i] else 0
                for k in range(3):
                    slots.point[i][k] = float(points[i][k])
        self.ext_force = forces
        self._ext_slots = slots
        self._push_randomization()

    def contact_sweeps(self, A: torch.Tensor, rhs: torch.Tensor, lam: torch.Tensor, both_tires: torch.Tensor):
        """The step kernels' projected Gauss-Seidel sweeps on given contact systems
        (`upkie_sim_contact_sweeps`): ``A [n, 21]`` packed lower by rows, ``rhs
        [n, 6]``, ``lam [n, 6]`` the warm start; returns (impulses, sweeps run)."""
        n = int(A.shape[0])
        A = A.to(self.device, torch.float32).contiguous()
        rhs = rhs.to(self.device, torch.float32).contiguous()
        lam = lam.to(self.device, torch.float32).contiguous().clone()
        both = both_tires.to(self.device, torch.uint8).contiguous()
        assert A.shape == (n, 21) and rhs.shape == (n, 6) and lam.shape == (n, 6) and both.shape == (n,)
        sweeps = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._launch(self._lib.upkie_sim_contact_sweeps, n, A.data_ptr(), rhs.data_ptr(), lam.data_ptr(), both.data_ptr(), sweeps.data_ptr())
        return lam, sweeps

    def sample_pushes(self, push_index: int, max_norm: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Push number `push_index` of every env (`upkie_sim_sample_pushes`): a
        world-frame force ``[3, B]`` with norm ~ U(0, max_norm) in a uniformly
        random horizontal direction, drawn on the device (Philox keyed by seed,
        global env id, push number). Written into `out` (a new tensor if None);
        hand it to `set_external_force` once -- the step kernels re-read the
        buffer at every substep -- and `zero_()` it to end the push."""
        if out is None:
            out = torch.zeros((3, self.num_envs), dtype=torch.float32, device=self.device)
        assert out.shape == (3, self.num_envs) and out.is_contiguous() and out.dtype == torch.float32
        self._launch(self._lib.upkie_sim_sample_pushes, out.data_ptr(), int(push_index) & 0xFFFFFFFF, float(max_norm))
        return out

    def _push_randomization(self):
        self._check(self._lib.upkie_sim_set_randomization(self._handle, _ptr(self.body_inertials), None, None))
        slots = getattr(self, "_ext_slots", None)
        if self.ext_force is not None and slots is not None:
            self._check(self._lib.upkie_sim_set_external_forces(self._handle, _ptr(self.ext_force), C.byref(slots)))

    # ---------------------------------------------------------------- API
    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reset masked envs (all if None); returns the Gyropod obs [B, 6]."""
        if mask is not None:
            mask = mask.to(self.device, torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            self._check(
                self._lib.upkie_sim_reset(
                    self._handle,
                    _ptr(self.state),
                    _ptr(mask),
                    _ptr(self.obs6),
                    self._stream(),
                )
            )
        return self.obs6

    def _step(self, fn, act, obs):
        self._launch(fn, self.state.data_ptr(), act.data_ptr(), obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                     self.truncated.data_ptr())
        return obs, self.reward, self.terminated, self.truncated

    def _as_action(self, act, shape):
        if not (type(act) is torch.Tensor and act.dtype is torch.float32 and act.device == self.device):
            act = torch.as_tensor(act, dtype=torch.float32, device=self.device)
        if act.shape != shape:
            act = act.reshape(shape)
        return act if act.is_contiguous() else act.contiguous()

    def stepper(self, kind: str):
        """The host side of one `env.step()` as ONE ctypes call on addresses
        fetched once: ``stepper("pendulum" | "gyropod" | "servos")`` returns
        ``step(action_address)`` that launches the step of that env kind on
        torch's current stream, reading a contiguous fp32 action buffer of the
        kind's shape at `action_address` and writing this handle's persistent
        output buffers (``obs4`` / ``obs6`` / ``obs_servos``, ``reward``,
        ``terminated``, ``truncated``). What `UpkieVecEnv.step` runs (a Python RL
        loop pays this per step: 2-3 us instead of 8)."""
        if kind == "servos" and self.obs_servos is None:
            self.obs_servos = torch.zeros((self.num_envs, 6, 5), dtype=torch.float32, device=self.device)
        fn, obs = {"pendulum": (self._lib.upkie_sim_step_pendulum, self.obs4), "gyropod": (self._lib.upkie_sim_step_gyropod, self.obs6),
                   "servos": (self._lib.upkie_sim_step_servos, self.obs_servos),
                   "pendulum_agent": (self._lib.upkie_sim_step_pendulum_agent, self.obs4)}[kind]
        handle, index, raw_stream = self._handle, self._device_index, _raw_stream
        state, obs_p, rew, term, trunc = self.state.data_ptr(), obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(), self.truncated.data_ptr()
        current_device = torch.cuda.current_device
        slow = self._launch
        if kind == "pendulum_agent":  # (no action buffer: the linear agent acts on the observation held in `obs4`)

            def step_agent() -> None:
                if raw_stream is not None and current_device() == index:
                    status = fn(handle, state, obs_p, rew, term, trunc, raw_stream(index))
                    if status < 0:
                        lib.check(status, handle)
                else:
                    slow(fn, state, obs_p, rew, term, trunc)

            return step_agent

        def step(action_address: int) -> None:
            if raw_stream is not None and current_device() == index:
                status = fn(handle, state, action_address, obs_p, rew, term, trunc, raw_stream(index))
                if status < 0:
                    lib.check(status, handle)
            else:
                slow(fn, state, action_address, obs_p, rew, term, trunc)

        return step

    def step_into_fn(self, kind: str, policy: Optional["abi.UpkieServoPolicy"] = None, mpc=None, mpc_x0: Optional[torch.Tensor] = None,
                     mpc_contact: Optional[torch.Tensor] = None):
        """`stepper` with caller-chosen OUTPUT buffers: returns
        ``step(action_address, obs_address, reward_address, terminated_address, truncated_address)``
        for env kind "pendulum" | "gyropod" | "servos" | "servos_policy" (the
        servo-level `policy` evaluated inside the launch; `action_address` is
        ignored) | "base_velocity" (the balancer `mpc` -- a `BatchedMpc` -- in
        front of the step, one launch where the mapping allows; `mpc_x0`
        ``[B, 4]`` / `mpc_contact` ``[B]`` u8 carry the balancer's inputs from
        step to step). The step writes straight into the addresses it is given
        -- e.g. a slot of the staging buffer a collective ships
        (`upkie_amd.distributed.ShardedVecEnv`) -- no copy, no packing launch."""
        lib_, handle, index, raw_stream = self._lib, self._handle, self._device_index, _raw_stream
        state = self.state.data_ptr()
        current_device = torch.cuda.current_device
        device = self.device

        def launch(call):
            if raw_stream is not None and current_device() == index:
                status = call(raw_stream(index))
            else:
                with torch.cuda.device(device):
                    status = call(self._stream())
            if status < 0:
                lib.check(status, handle)

        if kind in ("pendulum", "gyropod", "servos"):
            fn = {"pendulum": lib_.upkie_sim_step_pendulum, "gyropod": lib_.upkie_sim_step_gyropod, "servos": lib_.upkie_sim_step_servos}[kind]
            return lambda act, obs, rew, term, trunc: launch(lambda st: fn(handle, state, act, obs, rew, term, trunc, st))
        if kind == "servos_policy":
            assert policy is not None
            if getattr(self, "_policy_act", None) is None:
                self._policy_act = torch.zeros((self.num_envs, 6, 6), dtype=torch.float32, device=self.device)
            policy_act, policy_ref = self._policy_act.data_ptr(), C.byref(policy)
            fn = lib_.upkie_sim_step_servos_policy
            return lambda act, obs, rew, term, trunc: launch(lambda st: fn(handle, state, policy_ref, policy_act, obs, rew, term, trunc, st))
        if kind == "base_velocity":
            assert mpc is not None and mpc_x0 is not None and mpc_contact is not None
            ws, commanded, x0, contact = mpc.workspace.data_ptr(), mpc.commanded_velocity.data_ptr(), mpc_x0.data_ptr(), mpc_contact.data_ptr()
            fn, mpc_handle = lib_.upkie_sim_step_base_velocity_mpc, mpc._handle
            return lambda act, obs, rew, term, trunc: launch(lambda st: fn(handle, mpc_handle, state, ws, act, commanded, obs, x0, contact, rew, term, trunc, st))
        raise ValueError(f"unknown env kind '{kind}'")

    def step_pendulum(self, act):
        act = self._as_action(act, (self.num_envs,))
        return self._step(self._lib.upkie_sim_step_pendulum, act, self.obs4)

    def step_gyropod(self, act):
        act = self._as_action(act, (self.num_envs, 2))
        return self._step(self._lib.upkie_sim_step_gyropod, act, self.obs6)

    def step_servos(self, act):
        act = self._as_action(act, (self.num_envs, 6, 6))
        if self.obs_servos is None:
            self.obs_servos = torch.zeros(
                (self.num_envs, 6, 5), dtype=torch.float32, device=self.device
            )
        return self._step(self._lib.upkie_sim_step_servos, act, self.obs_servos)

    def step_base_velocity_mpc(self, mpc, act, mpc_x0, mpc_contact):
        """UpkieBaseVelocity's whole step in one call (`upkie_sim_step_base_velocity_mpc`):
        the MPC balancer `mpc` (a `BatchedMpc`) on the previous observation, then
        the step; one launch where the lane mapping and the horizon allow it."""
        act = self._as_action(act, (self.num_envs, 2))
        if getattr(self, "obs3", None) is None:
            self.obs3 = torch.zeros((self.num_envs, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(
                self._lib.upkie_sim_step_base_velocity_mpc(
                    self._handle, mpc._handle, _ptr(self.state), _ptr(mpc.workspace), _ptr(act), _ptr(mpc.commanded_velocity),
                    _ptr(self.obs3), _ptr(mpc_x0), _ptr(mpc_contact), _ptr(self.reward), _ptr(self.terminated), _ptr(self.truncated),
                    self._stream(),
                )
            )
        return self.obs3, self.reward, self.terminated, self.truncated

    def servo_policy(self, policy: "abi.UpkieServoPolicy", act: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Write the next `step_servos` action ``[B, 6, 6]`` from the state with
        the on-device servo-level policy (`upkie_sim_servo_policy`: one small
        launch, no host round trip); fallen envs are flagged for the NEXT_STEP
        autoreset."""
        if act is None:
            if getattr(self, "_policy_act", None) is None:
                self._policy_act = torch.zeros((self.num_envs, 6, 6), dtype=torch.float32, device=self.device)
            act = self._policy_act
        self._launch(self._lib.upkie_sim_servo_policy, self.state.data_ptr(), C.byref(policy), act.data_ptr())
        return act

    def step_servos_policy(self, policy: "abi.UpkieServoPolicy"):
        """`servo_policy` + `step_servos` as one call (`upkie_sim_step_servos_policy`):
        one launch up to 8192 envs (the policy evaluated by the step's own lanes),
        two otherwise."""
        if getattr(self, "_policy_act", None) is None:
            self._policy_act = torch.zeros((self.num_envs, 6, 6), dtype=torch.float32, device=self.device)
        if self.obs_servos is None:
            self.obs_servos = torch.zeros((self.num_envs, 6, 5), dtype=torch.float32, device=self.device)
        self._launch(self._lib.upkie_sim_step_servos_policy, self.state.data_ptr(), C.byref(policy), self._policy_act.data_ptr(),
                     self.obs_servos.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(), self.truncated.data_ptr())
        return self.obs_servos, self.reward, self.terminated, self.truncated

    def step_base_velocity(self, act, commanded_velocity, mpc_x0, mpc_contact):
        """Second half of the fused UpkieBaseVelocity step: ``act[B, 2]`` =
        [linear velocity, yaw velocity], ground velocity from the MPC
        balancer; returns the SE(2) observation ``[B, 3]`` and refreshes the
        balancer's next inputs in place."""
        act = self._as_action(act, (self.num_envs, 2))
        if getattr(self, "obs3", None) is None:
            self.obs3 = torch.zeros((self.num_envs, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(
                self._lib.upkie_sim_step_base_velocity(
                    self._handle,
                    _ptr(self.state),
                    _ptr(act),
                    _ptr(commanded_velocity),
                    _ptr(self.obs3),
                    _ptr(mpc_x0),
                    _ptr(mpc_contact),
                    _ptr(self.reward),
                    _ptr(self.terminated),
                    _ptr(self.truncated),
                    self._stream(),
                )
            )
        return self.obs3, self.reward, self.terminated, self.truncated

    def step_pendulum_agent(self):
        """Pendulum step with the README's linear agent evaluated on-device
        from the observation currently held in ``self.obs4``."""
        self._launch(self._lib.upkie_sim_step_pendulum_agent, self.state.data_ptr(), self.obs4.data_ptr(), self.reward.data_ptr(),
                     self.terminated.data_ptr(), self.truncated.data_ptr())
        return self.obs4, self.reward, self.terminated, self.truncated

    def step_pendulum_records(self, prev_records: torch.Tensor, records: torch.Tensor) -> torch.Tensor:
        """On-device agent step reading the previous observation from
        ``prev_records`` and writing this step's records to ``records``."""
        assert records.shape == (self.num_envs, 8) and prev_records.shape == (self.num_envs, 8)
        assert records.is_contiguous() and prev_records.is_contiguous()
        with torch.cuda.device(self.device):
            self._check(
                self._lib.upkie_sim_step_pendulum_agent_records(
                    self._handle, _ptr(self.state), _ptr(prev_records), _ptr(records), self._stream()
                )
            )
        return records

    def step_pendulum_records_raw(self, prev_records_ptr: int, records_ptr: int) -> None:
        """`step_pendulum_records` on device addresses of two ``[B, 8]`` fp32 record
        buffers the caller keeps alive (the rollout ring of `RolloutGather`): the
        host side of one env.step() is then one ctypes call."""
        self._launch(self._lib.upkie_sim_step_pendulum_agent_records, self.state.data_ptr(), prev_records_ptr, records_ptr)

    def rollout_pendulum_records(self, prev_records: torch.Tensor, records: torch.Tensor) -> torch.Tensor:
        """``records.shape[0]`` consecutive on-device-agent steps, records
        ``[K, B, 8]``: the same results as K `step_pendulum_records` calls
        chained through their records, in one launch up to 32768 envs (the
        state stays in registers between the steps)."""
        assert records.dim() == 3 and records.shape[1:] == (self.num_envs, 8) and prev_records.shape == (self.num_envs, 8)
        assert records.is_contiguous() and prev_records.is_contiguous()
        with torch.cuda.device(self.device):
            self._check(
                self._lib.upkie_sim_step_pendulum_agent_rollout(
                    self._handle, _ptr(self.state), _ptr(prev_records), _ptr(records), int(records.shape[0]), self._stream()
                )
            )
        return records

    def step_pendulum_packed(self, records: torch.Tensor, act=None) -> torch.Tensor:
        """Pendulum step writing one ``[obs(4) | reward, terminated,
        truncated, 0]`` record per env into ``records[B, 8]``; with
        ``act=None`` the on-device linear agent acts on the previous record."""
        assert records.shape == (self.num_envs, 8) and records.is_contiguous()
        with torch.cuda.device(self.device):
            if act is None:
                status = self._lib.upkie_sim_step_pendulum_agent_packed(
                    self._handle, _ptr(self.state), _ptr(records), self._stream()
                )
            else:
                act = self._as_action(act, (self.num_envs,))
                status = self._lib.upkie_sim_step_pendulum_packed(
                    self._handle, _ptr(self.state), _ptr(act), _ptr(records), self._stream()
                )
        self._check(status)
        return records

    def observe(self, update_imu: bool = True) -> dict:
        """Full spine observation as ``[B, ...]`` tensors
        (pybullet_backend.py:313-331), materialised on request only."""
        B, dev = self.num_envs, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        out = {
            "pitch": torch.empty(B, **f32),
            "angular_velocity": torch.empty((B, 3), **f32),
            "linear_velocity": torch.empty((B, 3), **f32),
            "rotation_base_to_world": torch.empty((B, 9), **f32),
            "floor_contact": torch.empty(B, dtype=torch.uint8, device=dev),
            "imu_orientation": torch.empty((B, 4), **f32),
            "imu_angular_velocity": torch.empty((B, 3), **f32),
            "imu_linear_acceleration": torch.empty((B, 3), **f32),
            "imu_raw_linear_acceleration": torch.empty((B, 3), **f32),
            "servo": torch.empty((B, 6, 5), **f32),
            "wheel_odometry": torch.empty((B, 2), **f32),
        }
        so = abi.UpkieSpineObservation()
        for name, t in out.items():
            setattr(so, name, t.data_ptr())
        with torch.cuda.device(dev):
            self._check(
                self._lib.upkie_sim_observe(
                    self._handle,
                    _ptr(self.state),
                    C.byref(so),
                    1 if update_imu else 0,
                    self._stream(),
                )
            )
        return out

    def use_bullet_like_contacts(self, on: bool = True) -> Optional[torch.Tensor]:
        """Contact model of this handle's steps (`upkie_sim_set_contact_manifold`):
        True = the Bullet-like specification (persistent 4-point manifolds per
        tire, 50 fixed warm-started sequential-impulse sweeps, cone friction
        along the sliding direction, no friction CFM: what
        `pybullet.stepSimulation()` is published to do, pybullet_backend.py:306)
        on a zeroed per-env manifold this handle keeps ``[64, B]``; False = the
        product's default specification. Eight lanes per env up to 16384 envs
        (Servos steps: up to 8192; one cached point per tire -- a robot lying
        flat on its side keeps the deepest one --, joint stops inside the same
        50 sweeps since round 6), one env per lane otherwise (every case;
        `set_lanes_per_env(1)`); about 2 x the default model's step: a fidelity
        option, not the fast path."""
        self.contact_manifold = torch.zeros((abi.CONTACT_MANIFOLD_WORDS, self.num_envs), dtype=torch.float32, device=self.device) if on else None
        self._check(self._lib.upkie_sim_set_contact_manifold(self._handle, _ptr(self.contact_manifold)))
        return self.contact_manifold

    def set_final_observation(self, final_obs: Optional[torch.Tensor]) -> None:
        """SAME_STEP autoreset completed by the step calls themselves
        (`upkie_sim_set_final_observation`): `final_obs`, shaped like the step's
        observation output, receives every env's last observation; finished
        envs come back re-initialised. ``None`` switches it off."""
        if final_obs is not None:
            assert final_obs.is_contiguous() and final_obs.dtype == torch.float32 and final_obs.device == self.device
        self.final_obs = final_obs  # (kept alive here: the library holds the raw pointer)
        self._check(self._lib.upkie_sim_set_final_observation(self._handle, _ptr(final_obs)))

    def autoreset_done(self, layout: int, obs: torch.Tensor, final_obs: Optional[torch.Tensor]) -> torch.Tensor:
        """gymnasium SAME_STEP autoreset in one launch: envs whose DONE word is
        set are re-initialised; their rows of ``obs`` (what the step of layout
        `abi.OBSERVATION_*` just wrote) go to ``final_obs`` and are replaced
        by the reset observation. Returns ``obs``."""
        self._launch(self._lib.upkie_sim_autoreset_done, int(layout), self.state.data_ptr(), obs.data_ptr(), _ptr(final_obs))
        return obs

    def contact_points(self) -> torch.Tensor:
        """Tire/floor contact points of every env, ``[B, 2, 8]``: per tire (left,
        right) ``[exists, position in world (3), force in world (3), 0]``
        (PyBulletBackend.get_contact_points, pybullet_backend.py:660-716). A
        query, not part of the step: one extra small launch."""
        out = torch.empty((self.num_envs, 2, abi.CONTACT_POINT_WORDS), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.upkie_sim_contact_points(self._handle, _ptr(self.state), _ptr(out), self._stream()))
        return out

    def get_contact_points(self, link_name: Optional[str] = None, env: int = 0) -> list:
        """`PyBulletBackend.get_contact_points(link_name)` for one env of the
        batch (what ``env.unwrapped.backend.get_contact_points(...)`` of
        examples/pybullet/count_wheel_contacts.py:34-48 calls)."""
        from .utils.point_contact import point_contacts

        return point_contacts(self.contact_points()[env].cpu().numpy(), link_name)

    def attach_observers(self, config: Optional[abi.UpkieObserverConfig]) -> Optional[torch.Tensor]:
        """Run the spine's FloorContact / WheelContact / WheelOdometry observers
        inside every step, one observer cycle per physics substep (the spine's
        rate under a slower agent). Returns the observer memory ``[16, B]``
        (words `abi.O_*`), which holds their outputs; None detaches."""
        if config is None:
            self._check(self._lib.upkie_sim_attach_observers(self._handle, None, None))
            self.observer_state = None
            return None
        self.observer_state = torch.zeros((abi.OBSERVER_STATE_WORDS, self.num_envs), dtype=torch.float32, device=self.device)
        self._observer_config = config
        self._check(self._lib.upkie_sim_attach_observers(self._handle, C.byref(config), _ptr(self.observer_state)))
        return self.observer_state

    def flag_done(self, done: torch.Tensor) -> None:
        """Overwrite the per-env `done` word the NEXT_STEP autoreset reads: envs
        flagged here are re-initialised by their next step (used by the
        envs' time limit, which the kernel knows nothing about)."""
        self.state[abi.S_DONE] = done.to(self.device, torch.float32)

    @property
    def pgs_tolerance(self) -> float:
        """Relative tolerance the kernels' Gauss-Seidel sweeps actually stop at
        (`upkie_sim_pgs_tolerance`: the model's, floored at 1e-5 for fp32)."""
        return float(self._lib.upkie_sim_pgs_tolerance(self._handle))

    @property
    def lanes_per_env(self) -> int:
        """Lanes of a wavefront sharing one env in this handle's step kernels."""
        return int(self._lib.upkie_sim_lanes_per_env(self._handle))

    def lanes_per_env_of(self, observation_layout: int) -> int:
        """... in the step kernel of one entry point, named by its observation
        layout (`abi.OBSERVATION_*`): the Servos kernels leave the eight-lane
        mapping at 8192 envs, the others at 16384."""
        return int(self._lib.upkie_sim_lanes_per_env_of(self._handle, int(observation_layout)))

    def release_graph_captures(self) -> None:
        """`upkie_sim_release_graph_captures`: the hipGraphs recorded from this
        handle so far will not be replayed again (destroyed or about to be
        re-captured); their settings blocks (8 per handle) are free again."""
        self._check(self._lib.upkie_sim_release_graph_captures(self._handle))

    def set_lanes_per_env(self, lanes: int) -> None:
        """Force the lane mapping of this handle's later launches
        (`upkie_sim_set_lanes_per_env`): 1, 2 or 8 lanes per env, 0 = by batch
        size again."""
        self._check(self._lib.upkie_sim_set_lanes_per_env(self._handle, int(lanes)))

    def guard_counts(self, reset: bool = False) -> dict:
        """Non-finite guard of the step kernels (`upkie_sim_guard_counts`):
        command words replaced by the neutral action's value and env states
        replaced by the initial state (those envs reported `terminated`), since
        creation or the last ``reset=True``. Waits for the current stream."""
        counts = (C.c_uint32 * 2)()
        with torch.cuda.device(self.device):
            self._check(self._lib.upkie_sim_guard_counts(self._handle, counts, 1 if reset else 0, self._stream()))
        return {"commands_replaced": int(counts[0]), "states_replaced": int(counts[1])}

    CENSUS_FIELDS = ("joint_limit", "sweep_cap_hits", "friction_cone", "active_set_solves", "wavefront_substeps_limit", "wavefront_substeps_sweeps", "sweeps_total", "sweeps_max")

    def enable_census(self, on: bool = True) -> Optional[torch.Tensor]:
        """Rare-path census of the eight-lane step kernel (`upkie_sim_set_census`):
        a zeroed device buffer the kernels count into, or None when switched off."""
        self.census = torch.zeros(abi.CENSUS_WORDS, dtype=torch.int32, device=self.device) if on else None
        self._check(self._lib.upkie_sim_set_census(self._handle, _ptr(self.census)))
        return self.census

    def census_counts(self) -> dict:
        """Counters of `enable_census` so far: env-substeps on the eight-lane
        kernel's rare paths, and wavefront-substeps that ran them."""
        values = self.census.cpu().tolist()
        counts = dict(zip(self.CENSUS_FIELDS, values))
        counts["wavefront_max_sweeps_histogram"] = values[8:abi.CENSUS_WORDS]
        return counts

    def restart_random_streams(self) -> None:
        """Zero the per-env episode and noise-step counters that key the Philox
        streams: (seed, env, episode = 0) is replayed by the next reset, as
        gymnasium's `reset(seed=s)` contract asks."""
        self.state[abi.S_EPISODE].zero_()
        self.state[abi.S_STEP].zero_()

    def state_numpy(self) -> np.ndarray:
        return self.state.detach().cpu().numpy()

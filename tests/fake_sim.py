"""Test doubles with the interface of `BatchedSim` / `BatchedMpc` but driven by
the fp64 oracle on the CPU, so the env wrappers (`upkie_amd.envs`) can be
tested without a GPU -- the counterpart of the reference's mock-the-simulator
tests (tests/envs/backends/test_pybullet_backend_mock.py, upkie/envs/
testing.py). TEST INFRASTRUCTURE ONLY: the product never falls back to this."""

import ctypes as C

import numpy as np
import torch

from oracle import oracle as O
from upkie_amd import abi


class OracleSim:
    def __init__(self, config, model_struct, device="cpu"):
        self.config = config
        self.model = model_struct
        self.device = torch.device("cpu")
        self.num_envs = int(config.num_envs)
        self._o = O.Oracle(model_struct, config)
        self._o.state[abi.S_QUAT] = 1.0
        B = self.num_envs
        self.obs4 = torch.zeros((B, 4))
        self.obs6 = torch.zeros((B, 6))
        self.obs_servos = torch.zeros((B, 6, 5))
        self.reward = torch.zeros(B)
        self.terminated = torch.zeros(B, dtype=torch.uint8)
        self.truncated = torch.zeros(B, dtype=torch.uint8)
        self.body_inertials = None
        self.ext_force = None

    @property
    def state(self):
        return torch.from_numpy(self._o.state.astype(np.float32))

    def close(self):
        pass

    def use_bullet_like_contacts(self, on=True):
        self._o.use_bullet_like_contacts(on)

    def restart_random_streams(self):
        self._o.state[abi.S_EPISODE] = 0.0
        self._o.state[abi.S_STEP] = 0.0

    def flag_done(self, done):
        self._o.state[abi.S_DONE] = done.double().numpy()

    def attach_observers(self, config):
        if config is None:
            self._o.observer_config = self._o.observer_state = None
            return None
        self._o.attach_observers(config)
        return self.observer_state

    @property
    def observer_state(self):
        return None if self._o.observer_state is None else torch.from_numpy(self._o.observer_state.astype(np.float32))

    def push_config(self):
        # the library refuses what the reference's rotation helpers refuse (rotations.py:50-51)
        if abs(sum(q * q for q in self.config.init_quat) - 1.0) > 1e-5:
            from upkie_amd.lib import UpkieHipError

            raise UpkieHipError(abi.ERR_INVALID_ARGUMENT, "init_quat is not normalized")
        self._o.config = self.config

    def randomize_inertias(self, variation):
        self._o.body_inertials = self._o.sample_body_inertials(variation)
        self.body_inertials = torch.from_numpy(self._o.body_inertials.astype(np.float32))
        self.link_scale = torch.from_numpy(self._o.link_scale.astype(np.float32))
        return self.body_inertials

    def sample_pushes(self, push_index, max_norm, out=None):
        force = torch.from_numpy(self._o.sample_pushes(push_index, max_norm).astype(np.float32))
        if out is None:
            return force
        out.copy_(force)
        return out

    def set_external_force(self, force, point=(0.0, 0.0, 0.0)):
        self.ext_force = force
        self._o.ext_slots = None
        self._o.ext_force = None if force is None else np.ascontiguousarray(force.double().numpy())
        self._o.ext_point = np.array(point, dtype=np.float64)

    def set_external_forces(self, forces, bodies=(), points=(), local=()):
        self.ext_force = forces
        if forces is None:
            self._o.ext_force, self._o.ext_slots = None, None
            return
        slots = abi.UpkieExternalForces()
        slots.count = len(bodies)
        for i in range(len(bodies)):
            slots.body[i] = int(bodies[i])
            slots.local[i] = 1 if local[i] else 0
            for k in range(3):
                slots.point[i][k] = float(points[i][k])
        self._o.ext_slots = slots
        self._o.ext_force = np.ascontiguousarray(forces.double().numpy())

    def reset(self, mask=None):
        m = None if mask is None else mask.to(torch.uint8).numpy()
        self.obs6.copy_(torch.from_numpy(self._o.reset(m).astype(np.float32)))  # (the same buffer every time, like the handle's)
        return self.obs6

    def _out(self, obs, rew, term, trunc, holder):
        obs = torch.from_numpy(np.asarray(obs, dtype=np.float32))
        getattr(self, holder).copy_(obs)
        # (persistent buffers, rewritten in place, like the handle's: `UpkieVecEnv._fast_step` returns them as one cached tuple)
        self.reward.copy_(torch.from_numpy(rew.astype(np.float32)))
        self.terminated.copy_(torch.from_numpy(term))
        self.truncated.copy_(torch.from_numpy(trunc))
        return getattr(self, holder), self.reward, self.terminated, self.truncated

    final_obs = None

    def set_final_observation(self, final_obs):
        """`BatchedSim.set_final_observation`: the step calls complete a SAME_STEP autoreset themselves."""
        self.final_obs = final_obs

    def _same_step(self, out, layout):
        if self.final_obs is not None and self.config.autoreset_mode == abi.AUTORESET_DISABLED:
            self.autoreset_done(layout, out[0], self.final_obs)
        return out

    def step_pendulum(self, act):
        return self._same_step(self._out(*self._o.step_pendulum(torch.as_tensor(act).double().numpy().reshape(-1)), "obs4"), abi.OBSERVATION_PENDULUM)

    def step_gyropod(self, act):
        return self._same_step(self._out(*self._o.step_gyropod(torch.as_tensor(act).double().numpy().reshape(-1, 2)), "obs6"), abi.OBSERVATION_GYROPOD)

    def step_servos(self, act):
        return self._same_step(self._out(*self._o.step_servos(torch.as_tensor(act).double().numpy().reshape(-1, 6, 6)), "obs_servos"), abi.OBSERVATION_SERVOS)

    def step_pendulum_agent(self):
        return self._out(*self._o.step_pendulum_agent(self.obs4.double().numpy()), "obs4")

    def step_pendulum_records(self, prev_records, records):
        """The on-device linear agent acting on the observation held in the
        previous step's records; this step's packed records written in place."""
        obs, rew, term, trunc = self._o.step_pendulum_agent(prev_records[:, :4].double().numpy())
        records[:, :4] = torch.from_numpy(np.asarray(obs, dtype=np.float32))
        records[:, 4] = torch.from_numpy(np.asarray(rew, dtype=np.float32))
        records[:, 5] = torch.from_numpy(np.asarray(term, dtype=np.float32))
        records[:, 6] = torch.from_numpy(np.asarray(trunc, dtype=np.float32))
        records[:, 7] = 0.0
        return records

    def autoreset_done(self, layout, obs, final_obs):
        """SAME_STEP autoreset of the envs whose DONE word is set (what
        upkie_sim_autoreset_done does in one launch)."""
        done = self._o.state[abi.S_DONE] != 0
        if final_obs is not None:
            final_obs.copy_(obs[:, :4] if layout == abi.OBSERVATION_PENDULUM_RECORDS else obs)
        if not done.any():
            return obs
        obs6 = torch.from_numpy(self._o.reset(done.astype(np.uint8)).astype(np.float32))
        sel = torch.from_numpy(done)
        if layout in (abi.OBSERVATION_PENDULUM, abi.OBSERVATION_PENDULUM_RECORDS):
            obs[sel, :4] = obs6[sel][:, [1, 0, 4, 3]]
        elif layout == abi.OBSERVATION_GYROPOD:
            obs[sel] = obs6[sel]
        else:
            servo = torch.from_numpy(self._o.observe(False)["servo"].astype(np.float32))
            obs[sel] = servo[sel]
        return obs

    def contact_points(self):
        return torch.from_numpy(self._o.contact_points().astype(np.float32))

    def get_contact_points(self, link_name=None, env=0):
        from upkie_amd.utils.point_contact import point_contacts

        return point_contacts(self.contact_points()[env].numpy(), link_name)

    def rollout_pendulum_records(self, prev_records, records):
        prev = prev_records
        for k in range(records.shape[0]):
            self.step_pendulum_records(prev, records[k])
            prev = records[k]
        return records

    def observe(self, update_imu=True):
        out = self._o.observe(update_imu)
        return {k: torch.from_numpy(v if v.dtype == np.uint8 else v.astype(np.float32)) for k, v in out.items()}


def servo_policy_action(policy, state, radius_signed):
    """`servo_policy_kernel` / the in-launch policy of the eight-lane Servos
    kernel in numpy on an oracle state: action [B, 6, 6] and who fell."""
    q = state[abi.S_QUAT:abi.S_QUAT + 4]
    pitch = np.arcsin(np.clip(2.0 * (q[0] * q[2] - q[3] * q[1]), -1.0, 1.0))
    p = 0.5 * (state[abi.S_Q + 2] - state[abi.S_Q + 5]) * radius_signed
    pd = 0.5 * (state[abi.S_QD + 2] - state[abi.S_QD + 5]) * radius_signed
    B = state.shape[1]
    act = np.zeros((B, 6, 6))
    for j in range(6):
        for i in range(6):
            act[:, j, i] = policy.action[j][i]
        fb = policy.pitch_to_velocity[j] * pitch + policy.position_to_velocity[j] * p + policy.velocity_to_velocity[j] * pd
        clip = policy.velocity_feedback_clip[j]
        if clip > 0.0:
            fb = np.clip(fb, -clip, clip)
        act[:, j, 1] += fb
        act[:, j, 2] += policy.pitch_to_torque[j] * pitch
    fallen = np.abs(pitch) > policy.fall_pitch if policy.fall_pitch > 0.0 else np.zeros(B, dtype=bool)
    return act, fallen


def _from_address(address, count, ctype, dtype):
    return np.frombuffer((ctype * count).from_address(address), dtype=dtype)


def _step_into_fn(self, kind, policy=None, mpc=None, mpc_x0=None, mpc_contact=None):
    """`BatchedSim.step_into_fn` on the oracle: the same raw-address contract
    (CPU tensors have addresses too), so `ShardedVecEnv` runs unchanged."""
    B = self.num_envs
    obs_words = {"pendulum": 4, "gyropod": 6, "servos": 30, "servos_policy": 30, "base_velocity": 3}[kind]
    act_words = {"pendulum": 1, "gyropod": 2, "servos": 36, "servos_policy": 0, "base_velocity": 2}[kind]
    xy = np.zeros((B, 2))

    def step(act, obs, rew, term, trunc):
        o = self._o
        a = _from_address(act, B * act_words, C.c_float, np.float32).astype(np.float64) if act_words else None
        if kind == "pendulum":
            out = o.step_pendulum(a)
        elif kind == "gyropod":
            out = o.step_gyropod(a.reshape(B, 2))
        elif kind == "servos":
            out = o.step_servos(a.reshape(B, 6, 6))
        elif kind == "servos_policy":
            action, fallen = servo_policy_action(policy, o.state, float(self.model.left_sign) * float(self.model.wheel_radius))
            o.state[abi.S_DONE] = np.where(fallen, 1.0, o.state[abi.S_DONE])
            if self.ext_force is not None:
                o.ext_force = np.ascontiguousarray(self.ext_force.double().numpy())
            out = o.step_servos(action.astype(np.float32).astype(np.float64))
        else:  # UpkieBaseVelocity.step, upkie_base_velocity.py:164-202 (the generic composition of UpkieBaseVelocityVecEnv)
            a = a.reshape(B, 2)
            autoreset = (o.state[abi.S_DONE] != 0) if self.config.autoreset_mode else np.zeros(B, dtype=bool)
            v, _ = mpc.step(mpc_x0, torch.from_numpy(a[:, 0].copy()), mpc_contact, float(self.config.dt))
            obs6, r, t, tr = o.step_gyropod(np.stack([v.double().numpy(), a[:, 1]], axis=1))
            mpc_x0.copy_(torch.from_numpy(obs6[:, [0, 1, 3, 4]].astype(np.float32)))
            mpc_contact.copy_(torch.from_numpy((o.state[abi.S_CONTACT] != 0).astype(np.uint8)))
            if autoreset.any():
                mpc.reset(torch.from_numpy(autoreset.astype(np.uint8)))
                xy[autoreset] = 0.0
            live = ~autoreset
            dt = float(self.config.dt)
            xy[live, 0] += (a[:, 0] * np.cos(obs6[:, 2]) * dt)[live]
            xy[live, 1] += (a[:, 0] * np.sin(obs6[:, 2]) * dt)[live]
            out = (np.concatenate([xy, obs6[:, 2:3]], axis=1), r, t, tr)
        o_, r_, t_, tr_ = out
        _from_address(obs, B * obs_words, C.c_float, np.float32)[:] = np.asarray(o_, dtype=np.float32).reshape(-1)
        _from_address(rew, B, C.c_float, np.float32)[:] = r_.astype(np.float32)
        _from_address(term, B, C.c_uint8, np.uint8)[:] = t_
        _from_address(trunc, B, C.c_uint8, np.uint8)[:] = tr_

    return step


OracleSim.step_into_fn = _step_into_fn


def _stepper(self, kind):
    """`BatchedSim.stepper` on the oracle: ``step(action_address)`` writing the double's persistent output buffers."""
    B = self.num_envs
    words = {"pendulum": 1, "gyropod": 2, "servos": 36}[kind]
    call = {"pendulum": self.step_pendulum, "gyropod": self.step_gyropod, "servos": self.step_servos}[kind]

    def step(action_address):
        call(torch.from_numpy(_from_address(action_address, B * words, C.c_float, np.float32).copy()))

    return step


OracleSim.stepper = _stepper


def oracle_sim_factory(config, model_struct, device):
    return OracleSim(config, model_struct, device)


class OracleMpc:
    def __init__(self, config, device="cpu"):
        self.config = config
        self.num_envs = int(config.num_envs)
        N, B = config.nb_timesteps, self.num_envs
        self._ws = np.zeros((2 * N, B))
        self._v = np.zeros(B)
        self._first = np.zeros(B)

    @property
    def commanded_velocity(self):
        return torch.from_numpy(self._v.astype(np.float32))

    def reset(self, mask=None):
        sel = slice(None) if mask is None else mask.bool().numpy()
        self._ws[:, sel] = 0.0
        self._v[sel] = 0.0

    def step(self, x0, target_velocity, contact, dt):
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        x = np.ascontiguousarray(x0.double().numpy())
        vt = np.ascontiguousarray(target_velocity.double().numpy())
        ct = np.ascontiguousarray(contact.to(torch.uint8).numpy())
        O.lib().oracle_mpc_step(C.byref(self.config), p(self._ws), p(x), p(vt), p(ct), C.c_double(dt), p(self._v), p(self._first))
        return torch.from_numpy(self._v.astype(np.float32)), torch.from_numpy(self._first.astype(np.float32))

    def close(self):
        pass


class OracleObservers:
    """`BatchedObservers` double on the fp64 observer oracle."""

    def __init__(self, config, device="cpu"):
        self.config = config
        self.num_envs = int(config.num_envs)
        self.device = torch.device("cpu")
        self._o = O.ObserverOracle(config)

    @property
    def state(self):
        return torch.from_numpy(self._o.state.astype(np.float32))

    def reset(self, mask=None):
        self._o.reset(None if mask is None else mask.bool().numpy())

    def step(self, servo, imu_orientation=None, imu_angular_velocity=None, cross_button=None):
        from upkie_amd.observers import observer_blocks

        n = lambda t: None if t is None else t.double().numpy()
        only_base = servo is None
        if only_base:
            servo = torch.zeros((self.num_envs, 6, 5))
            saved = self._o.state.copy()
        out = self._o.step(n(servo), n(imu_orientation), n(imu_angular_velocity), None if cross_button is None else cross_button.numpy())
        if only_base:
            self._o.state[:] = saved
        tensors = {k: torch.from_numpy(np.asarray(v, dtype=np.uint8 if v.dtype == np.uint8 else np.float32)) for k, v in out.items()}
        blocks = observer_blocks(tensors)
        return {"base_orientation": blocks["base_orientation"]} if only_base else blocks

    def step_from_sim(self, sim, update_imu=False, cross_button=None):
        obs = sim.observe(update_imu=update_imu)
        return self.step(obs["servo"], obs["imu_orientation"], obs["imu_angular_velocity"], cross_button)

    def close(self):
        pass

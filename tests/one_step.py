"""One-step ("teacher-forced") comparison of a step implementation with the
fp64 oracle, where a closed loop is too chaotic to compare trajectories.

TEST INFRASTRUCTURE ONLY (it drives the oracle).

The closed-loop windows of `test_timed_windows_gpu.py` compare what an RL run
sees; where tires skid (BASELINE configs[4] under
examples/pybullet/torque_balancing.py:15-37's law) two correct integrators part
within a few hundred steps, so those windows can only compare populations.
Here the ORACLE ALONE runs the window. Before each of its steps its state is
rounded to fp32 in place (so that both sides start from the same bits), handed
to every implementation under test together with the same fp32 action, push and
inertial records, each takes ONE `env.step()` from it, and the resulting state,
the six joint torques and the observation are compared with the oracle's next
state. The defect of ONE step is measured on exactly the states the chaotic
window visits, and binned by what the robot was doing:

  reset          the step re-initialises the env (NEXT_STEP autoreset)
  stop_impact    a hip or knee within 1e-3 rad of its stop, moving at more than
                 0.5 rad/s before or after the step (arriving, bouncing)
  joint_at_stop  a hip or knee within 1e-3 rad of its stop, resting
  airborne       a tire has no contact point before or after the step (taking
                 off, landing, lying on one side)
  sliding        a tire's friction force at >= 98 % of mu x its normal force
  saturated      a joint torque at its limit (the wheel torque law clipped)
  push           the external push of SURVEY 8d's schedule is being held
  rolling        none of the above

(first match wins, in that order). What is compared, per env and step, every
difference relative to max(1, |the oracle's value|) -- absolute for what is of
order one, in units of the value where fp32 resolves no better (a robot that ran
40 m away):

  position   max |.| over base position (m), quaternion, joint angles (rad)
  velocity   max |.| over base linear (m/s), angular (rad/s), joint rates (rad/s)
  wheel_rate max |.| over the two wheel rates alone (rad/s): the stiff one
  torque     max |.| over the six joint torques of the last substep (N.m)
  obs        max |.| over the step's observation
"""

import numpy as np

from upkie_amd import abi

REGIMES = ("reset", "stop_impact", "joint_at_stop", "airborne", "sliding", "saturated", "push", "rolling")
METRICS = ("position", "velocity", "wheel_rate", "torque", "obs")


def pitch_of(s):
    return np.arcsin(np.clip(2.0 * (s[abi.S_QUAT] * s[abi.S_QUAT + 2] - s[abi.S_QUAT + 3] * s[abi.S_QUAT + 1]), -1, 1))


def to_f32_in_place(a):
    """Round an fp64 array to the nearest fp32 values, in place."""
    a[...] = a.astype(np.float32).astype(np.float64)
    return a


def contact_summary(points, mu):
    """`Oracle.contact_points()` [B, 2, 8] -> (a tire has no contact point [B],
    a tire's friction force at >= 98 % of mu x normal force [B])."""
    exists = points[:, :, 0] != 0
    normal = points[:, :, 6]
    tangential = np.hypot(points[:, :, 4], points[:, :, 5])
    sliding = exists & (tangential >= 0.98 * mu * np.maximum(normal, 0.0)) & (normal > 0)
    return ~exists.all(axis=1), sliding.any(axis=1)


def classify(model, before, after, resetting, pushed, torque_limit):
    """Exclusive regime index per env (into REGIMES). `before` / `after`:
    (state [48, B], no_contact [B], sliding [B]) around the oracle's step;
    `torque_limit` [6, B]."""
    (s0, air0, slide0), (s1, air1, slide1) = before, after
    B = s0.shape[1]
    lower, upper = np.array(model.joint_lower[:]), np.array(model.joint_upper[:])
    at_stop = np.zeros(B, dtype=bool)
    impact = np.zeros(B, dtype=bool)
    for j in (0, 1, 3, 4):
        here = np.zeros(B, dtype=bool)
        for s in (s0, s1):
            q = s[abi.S_Q + j]
            here |= (q <= lower[j] + 1e-3) | (q >= upper[j] - 1e-3)
        at_stop |= here
        impact |= here & ((np.abs(s0[abi.S_QD + j]) > 0.5) | (np.abs(s1[abi.S_QD + j]) > 0.5))
    saturated = ((torque_limit > 0) & (np.abs(s1[abi.S_TORQUE:abi.S_TORQUE + 6]) >= 0.999 * torque_limit)).any(axis=0)
    flags = [resetting, impact, at_stop, air0 | air1, slide0 | slide1, saturated, pushed, np.ones(B, dtype=bool)]
    regime = np.full(B, -1)
    for i, f in enumerate(flags):
        regime = np.where((regime < 0) & f, i, regime)
    return regime


def defects(state_dev, obs_dev, state_ref, obs_ref):
    """Per env: the METRICS of one step, device (fp32 arrays) against oracle."""
    # relative to max(1, |value|): a robot that ran 40 m away has an fp32 resolution of 4e-6 m, a wheel at 100 rad/s of 8e-6 rad/s
    scale = np.maximum(1.0, np.abs(state_ref))
    d = np.abs(state_dev.astype(np.float64) - state_ref) / scale
    # q and -q are the same rotation
    qd_alt = np.abs(state_dev[abi.S_QUAT:abi.S_QUAT + 4].astype(np.float64) + state_ref[abi.S_QUAT:abi.S_QUAT + 4]).max(axis=0)
    pos = np.maximum(d[abi.S_POS:abi.S_POS + 3].max(axis=0), np.minimum(d[abi.S_QUAT:abi.S_QUAT + 4].max(axis=0), qd_alt))
    pos = np.maximum(pos, d[abi.S_Q:abi.S_Q + 6].max(axis=0))
    vel = np.maximum(d[abi.S_LINVEL:abi.S_LINVEL + 6].max(axis=0), d[abi.S_QD:abi.S_QD + 6].max(axis=0))
    wheel = np.maximum(d[abi.S_QD + 2], d[abi.S_QD + 5])
    torque = d[abi.S_TORQUE:abi.S_TORQUE + 6].max(axis=0)
    B = state_ref.shape[1]
    o_ref = obs_ref.reshape(B, -1)
    obs = (np.abs(obs_dev.astype(np.float64).reshape(B, -1) - o_ref) / np.maximum(1.0, np.abs(o_ref))).max(axis=1)
    return np.stack([pos, vel, wheel, torque, obs])


class Bins:
    """Per implementation and regime: every env-step's defects, kept as
    quantile sketches (all samples: a window is ~5 M env-steps x 5 metrics of
    fp32, which fits)."""

    def __init__(self, names):
        self.samples = {n: {r: [] for r in REGIMES} for n in names}

    def add(self, name, regime, d):
        for i, r in enumerate(REGIMES):
            sel = regime == i
            if sel.any():
                self.samples[name][r].append(d[:, sel].astype(np.float32))

    def table(self, qs=(0.5, 0.99, 0.9999, 1.0)):
        out = {}
        for name, by_regime in self.samples.items():
            out[name] = {}
            for r, chunks in by_regime.items():
                if not chunks:
                    out[name][r] = {"env_steps": 0}
                    continue
                a = np.concatenate(chunks, axis=1).astype(np.float64)
                row = {"env_steps": int(a.shape[1])}
                for m, metric in enumerate(METRICS):
                    row[metric] = {f"q{q:g}": float(np.quantile(a[m], q)) for q in qs}
                out[name][r] = row
        return out


def window(ref, model, devices, policy_action, steps, kind="servos", push_schedule=None, bullet_like=False, first_obs=None):
    """Run the oracle `ref` (an `oracle.Oracle`, already reset, randomisation
    installed) alone through `steps` steps of env kind `kind` ("servos":
    action [B, 6, 6], observation [B, 6, 5]; "pendulum": action [B],
    observation [B, 4]) under ``policy_action(state, last_obs) -> (action,
    fallen [B] or None)``; every implementation in `devices` (name -> adapter
    with ``step(kind, state32, manifold32, action32, force32) -> (state [48,
    B], obs, terminated [B], manifold)``) takes each step ONCE from the
    oracle's state. `push_schedule(k) -> force [3, B] or None` is asked at
    every step (None: keep). Returns (`Bins`, env-steps per regime, manifold
    agreement counts, terminated-flag mismatches per implementation)."""
    B = ref.B
    mu = float(model.friction_mu)
    bins = Bins(list(devices))
    effort = np.array(model.joint_effort[:])
    census = {r: 0 for r in REGIMES}
    manifold_points = {n: [0, 0] for n in devices}  # envs-steps compared, with the same live points
    flag_mismatch = {n: 0 for n in devices}
    to_f32_in_place(ref.state)
    if ref.body_inertials is not None:
        to_f32_in_place(ref.body_inertials)
    air, slide = contact_summary(ref.contact_points(), mu)
    obs_ref = first_obs
    ref_step = {"servos": ref.step_servos, "pendulum": ref.step_pendulum}[kind]
    for k in range(steps):
        if push_schedule is not None:
            force = push_schedule(k)
            if force is not None:
                ref.ext_force = to_f32_in_place(np.ascontiguousarray(force, dtype=np.float64))
        to_f32_in_place(ref.state)
        if bullet_like:
            to_f32_in_place(ref.bullet_manifold)
        act, fallen = policy_action(ref.state, None if obs_ref is None else to_f32_in_place(obs_ref))
        act = to_f32_in_place(np.ascontiguousarray(act, dtype=np.float64))
        if fallen is not None:
            ref.state[abi.S_DONE] = np.where(fallen, 1.0, ref.state[abi.S_DONE])
        resetting = ref.state[abi.S_DONE] != 0
        state32 = ref.state.astype(np.float32)
        manifold32 = ref.bullet_manifold.astype(np.float32) if bullet_like else None
        force32 = None if ref.ext_force is None else ref.ext_force.astype(np.float32)
        pushed = np.zeros(B, dtype=bool) if ref.ext_force is None else (np.abs(ref.ext_force).max(axis=0) > 0)
        before = (ref.state.copy(), air, slide)
        # [6, B]: what the step clips each joint's torque to
        torque_limit = np.minimum(act[:, :, 5], effort[None, :]).T if kind == "servos" else np.repeat(effort[:, None], B, axis=1)
        obs_ref, _, term_ref, _ = ref_step(act)
        air, slide = contact_summary(ref.contact_points(), mu)
        regime = classify(model, before, (ref.state, air, slide), resetting, pushed, torque_limit)
        for i, r in enumerate(REGIMES):
            census[r] += int((regime == i).sum())
        for name, dev in devices.items():
            state_dev, obs_dev, term_dev, manifold_dev = dev.step(kind, state32, manifold32, act.astype(np.float32), force32)
            d = defects(state_dev, obs_dev, ref.state, obs_ref)
            bins.add(name, regime, d)
            flag_mismatch[name] += int((np.asarray(term_dev) != term_ref).sum())
            if bullet_like and manifold_dev is not None:
                live_dev = manifold_dev.reshape(2, 4, 8, B)[:, :, 7] != 0
                live_ref = ref.bullet_manifold.reshape(2, 4, 8, B)[:, :, 7] != 0
                if getattr(dev, "points_per_tire", 4) == 1:
                    same = (live_dev.any(axis=1) == live_ref.any(axis=1)).all(axis=0)
                else:
                    same = (live_dev == live_ref).all(axis=(0, 1))
                manifold_points[name][0] += B
                manifold_points[name][1] += int(same.sum())
    return bins, census, manifold_points, flag_mismatch


class OracleTwin:
    """The adapter protocol on a second oracle: a zero-defect implementation,
    for running the machinery without a GPU (tests/test_one_step_machinery.py)."""

    points_per_tire = 4

    def __init__(self, model, config, body_inertials=None, bullet_like=False, perturb=0.0, seed=0):
        from oracle import oracle as O

        self.o = O.Oracle(model, config)
        if bullet_like:
            self.o.use_bullet_like_contacts()
        self.o.body_inertials = body_inertials
        self.perturb = perturb
        self.rng = np.random.default_rng(seed)

    def step(self, kind, state32, manifold32, act32, force32):
        self.o.state[...] = state32.astype(np.float64)
        if manifold32 is not None:
            self.o.bullet_manifold[...] = manifold32.astype(np.float64)
        if force32 is not None:
            self.o.ext_force = np.ascontiguousarray(force32.astype(np.float64))
            self.o.ext_point = np.zeros(3)
        obs, _, term, _ = {"servos": self.o.step_servos, "pendulum": self.o.step_pendulum}[kind](act32.astype(np.float64))
        state = self.o.state.astype(np.float32)
        if self.perturb:
            state = state * (1.0 + self.perturb * self.rng.standard_normal(state.shape)).astype(np.float32)
        return state, obs.astype(np.float32), term, None if manifold32 is None else self.o.bullet_manifold.astype(np.float32)

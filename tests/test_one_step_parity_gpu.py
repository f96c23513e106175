"""One-step ("teacher-forced") parity of the HIP step kernels with the fp64
oracle on the states the CHAOTIC windows visit (north_star:
"per-step joint torques/observations match the reference CPU ... path within a
stated float tolerance").

`test_timed_windows_gpu.py` runs device and oracle side by side; where tires
skid (BASELINE configs[4] under examples/pybullet/torque_balancing.py:15-37's
law) the two part within a few hundred steps and only populations can be
compared. Here the oracle alone runs the window (tests/one_step.py): before
each step its state -- rounded to fp32, so both sides start from the same bits
-- is uploaded to the device, every lane mapping of the step kernel (eight, two
and one lane per env; UPKIE_LANES_PER_ENV) takes ONE `env.step()` through the
C-ABI with the same action, push force, inertial records (and contact
manifold), and the new state, the six joint torques and the observation are
held to the oracle's, binned by regime, with a tolerance STATED PER REGIME.

Reports: gpurun_out/parity_windows/one_step_*.json.
"""

import json
import os

import numpy as np
import pytest
import torch

from upkie_amd import abi
from upkie_amd.sim import BatchedSim

from .fake_sim import servo_policy_action
from .one_step import METRICS, REGIMES, window
from .test_one_step_machinery import c5_oracle, c5_push_schedule

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORTS = os.path.join(ROOT, "gpurun_out", "parity_windows")


def write_report(name: str, report: dict) -> None:
    os.makedirs(REPORTS, exist_ok=True)
    with open(os.path.join(REPORTS, name + ".json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(name, json.dumps(report, sort_keys=True))


class HipStep:
    """tests/one_step.py's adapter on `libupkie_hip.so`: one `BatchedSim` on a
    forced lane mapping; `step` uploads the oracle's state and takes one step."""

    def __init__(self, cfg, model, lanes, body_inertials, bullet_like, monkeypatch, pushes=True):
        monkeypatch.setenv("UPKIE_LANES_PER_ENV", str(lanes))
        self.sim = BatchedSim(cfg, model)
        monkeypatch.delenv("UPKIE_LANES_PER_ENV")
        self.lanes = lanes
        B = self.sim.num_envs
        if body_inertials is not None:
            self.sim.set_body_inertials(torch.from_numpy(body_inertials.astype(np.float32)))
        self.push = None
        if pushes:
            self.push = torch.zeros((3, B), dtype=torch.float32, device=self.sim.device)
            self.sim.set_external_force(self.push)
        self.bullet_like = bullet_like
        if bullet_like:
            self.sim.use_bullet_like_contacts()
        # the eight-lane Bullet-like kernel keeps ONE cached point per tire
        self.points_per_tire = 1 if (bullet_like and lanes == 8) else 4
        self.sim.reset()

    def step(self, kind, state32, manifold32, act32, force32):
        sim = self.sim
        sim.state.copy_(torch.from_numpy(state32))
        if manifold32 is not None:
            sim.contact_manifold.copy_(torch.from_numpy(manifold32))
        if force32 is not None and self.push is not None:
            self.push.copy_(torch.from_numpy(force32))
        act = torch.from_numpy(act32).to(sim.device)
        if kind == "servos":
            assert sim.lanes_per_env_of(abi.OBSERVATION_SERVOS) == self.lanes
            obs, _, term, _ = sim.step_servos(act)
        else:
            assert sim.lanes_per_env == self.lanes
            obs, _, term, _ = sim.step_pendulum(act)
        manifold = sim.contact_manifold.cpu().numpy() if manifold32 is not None else None
        return sim.state_numpy(), obs.cpu().numpy(), term.cpu().numpy(), manifold

    def close(self):
        self.sim.close()


# Stated tolerances of ONE env.step() (5 substeps of 1 ms), fp32 kernels against the fp64 oracle from the same fp32 state, per
# regime: (position, velocity, wheel_rate, torque) at the median / at 99 % of the env-steps of the regime / worst. Differences are
# relative to max(1, |value|) (tests/one_step.py). Measured (round 6, MI355X, the same on eight, two and one lane per env) in the
# comments; the bounds sit 3-5 x above. Why the regimes differ:
#  - rolling / push: everything is smooth; what is left is fp32 rounding through five substeps. The contact rows divide a gap by
#    h = 1 ms, so one ulp of height (6e-8 m at z ~ 0.6 m) is a 6e-5 m/s floor on velocities. The WORST env-step of several million
#    is one whose tire starts to slip INSIDE the step (the regime is told from the states before and after it): see sliding.
#  - saturated: the torque law clips; an env whose unclipped torque is within rounding of the limit clips on one side only, and
#    the explicit 1 kHz wheel loop (kd dt / I_wheel = 3.6 per substep) turns that into 1e-3 rad/s of wheel rate.
#  - sliding: the friction rows sit on their bounds; WHICH rows do is decided by comparisons of fp32 numbers, and a row that is on
#    the bound on one side and just inside on the other changes the tire force by a fraction of mu x the normal force for one 1 ms
#    substep: the bulk agrees like rolling (median 1e-5 rad/s), the last per cent by 1e-3 rad/s of wheel rate (a 0.6 kg wheel of
#    5 cm radius is the light end of the tire force), the worst by a substep's worth of it (0.5 rad/s).
#  - airborne / landing: a tire that touches down in a substep on one side and in the next on the other (gap at fp32 resolution)
#    differs by one substep of contact impulse.
#  - reset: the state is DRAWN (Philox: same integers on both sides), then one torque-free substep; a tire that starts exactly on
#    the floor is in contact or not by the last bit of z: g h = 9.8e-3 m/s on 1 % of the resets.
#  - joint_at_stop (resting on a stop) / stop_impact (arriving at one): the limit row is listed while the joint is within reach of
#    its stop and lets it close the gap within the substep: nothing hangs on the last bit of q any more (see the table below for
#    what the rule of rounds 1-5 measured); what is left is the rounding of a ten-row solve.
T = lambda pos, vel, wheel, torque: {"position": pos, "velocity": vel, "wheel_rate": wheel, "torque": torque}  # noqa: E731
TOLERANCES = {
    # regime: {metric: (median, q0.99, worst)}
    "rolling": T((5e-7, 3e-6, 2e-3), (1e-5, 2e-4, 1.0), (5e-6, 2e-4, 1.0), (1e-5, 2e-4, 1.0)),  # 8e-8/5e-7/3e-4, 1.8e-6/3e-5/0.13, 8e-7/3e-5/0.13, 2.4e-6/2.5e-5/0.24
    "push": T((5e-7, 2e-5, 2e-3), (1e-5, 5e-4, 1.0), (5e-6, 5e-4, 1.0), (1e-5, 1e-3, 1.0)),  # 7.5e-8/4.6e-6/1.9e-4, 1.7e-6/9e-5/0.18, 8e-7/5e-5/0.18, 2.7e-6/1.4e-4/0.2
    "saturated": T((1e-6, 5e-5, 1e-3), (1e-4, 5e-3, 0.2), (1e-4, 5e-3, 0.2), (1e-5, 5e-3, 0.2)),  # 2.3e-7/1.2e-5/1.2e-4, 2.7e-5/1.7e-3/1.5e-2, .., 2.2e-6/1.1e-3/3e-2
    "sliding": T((1e-6, 5e-5, 5e-3), (5e-5, 5e-3, 2.0), (2e-5, 5e-3, 2.0), (5e-5, 5e-3, 1.0)),  # 2.3e-7/1.3e-5/9e-4, 9.6e-6/1.5e-3/0.46, 5.9e-6/1.4e-3/0.46, 1.1e-5/1.4e-3/0.2
    "airborne": T((5e-7, 5e-6, 2e-2), (2e-5, 5e-4, 5.0), (5e-6, 5e-4, 5.0), (1e-5, 2e-4, 1.0)),  # 1.3e-7/6e-7/3.5e-3, 3.3e-6/8e-5/1.0, 5e-7/7e-5/1.0, 1.4e-6/2.6e-5/0.2
    "reset": T((1e-7, 1e-5, 3e-5), (1e-6, 1.2e-2, 2e-2), (1e-6, 1.2e-2, 2e-2), (1e-6, 1e-6, 1e-6)),  # 1.7e-8/2.9e-6/9.9e-6, 1e-8/2.8e-3/9.9e-3, torque 0
    # Measured with the gap-aware limit rows (joint_limit_row, dynamics.hpp; every mapping, both contact models on one lane): resting
    # 1.4e-7/1.2e-6/4.7e-5, 2.3e-6/1.4e-4/0.14, torque 8e-8/1e-5/6e-2; arriving 1.2e-7/1.6e-6/1.1e-5, 6.4e-6/1.9e-4/5e-3, 6e-7/2e-5/8e-5.
    # Under the rule of rounds 1-5 (a row only AT or beyond the stop, ERP bias alone) the same test measured position 1.6e-7 / 2.4e-3 /
    # 1.8e-2 and velocity 1.5e-5 / 4.6e-2 / 4.0: the fp64 checker alternated between substeps with and without the row of a joint
    # RESTING on its stop, the fp32 kernels kept it -- which is what made the rule change.
    "joint_at_stop": T((1e-6, 1e-5, 5e-4), (2e-5, 1e-3, 1.0), (1e-5, 1e-3, 1.0), (5e-6, 1e-4, 0.5)),
    "stop_impact": T((1e-6, 1e-5, 1e-4), (5e-5, 1e-3, 5e-2), (2e-5, 1e-3, 5e-2), (5e-6, 2e-4, 1e-3)),
}


def check(table, census, min_env_steps=0):
    failures = []
    for name, by_regime in table.items():
        for regime, row in by_regime.items():
            if row["env_steps"] <= min_env_steps:
                continue
            for metric, (median, q99, worst) in TOLERANCES[regime].items():
                got = row[metric]
                if not (got["q0.5"] <= median and got["q0.99"] <= q99 and got["q1"] <= worst):
                    failures.append((name, regime, metric, got, (median, q99, worst)))
    return failures


@pytest.mark.parametrize("law, contact_model", [("torque", "default"), ("torque", "bullet_like"), ("velocity", "default")])
def test_c5_window_one_step_from_the_oracle_state(law, contact_model, monkeypatch):
    """BASELINE configs[4]'s share of one GPU (4096 UpkieServos envs, per-link
    inertia randomisation, SURVEY 8d's push schedule, NEXT_STEP autoreset) for
    1200 steps under examples/pybullet/torque_balancing.py's law (robots run
    away and skid: 30 % of the env-steps slide, 16 % have a tire in the air) and
    under the README law through the wheels' velocity loop."""
    B, steps = 4096, 1200
    bullet_like = contact_model == "bullet_like"
    ref, model, cfg = c5_oracle(B, bullet_like=bullet_like)
    policy = (abi.torque_balancing_policy(10.0, 1.0, float(model.left_sign)) if law == "torque"
              else abi.velocity_balancing_policy(float(model.wheel_radius), 1.0, float(model.left_sign)))
    rs = float(model.left_sign) * float(model.wheel_radius)
    lanes = (8, 1) if bullet_like else (8, 2, 1)
    devices = {f"{n}_lanes": HipStep(cfg, model, n, ref.body_inertials, bullet_like, monkeypatch) for n in lanes}
    bins, census, points, flags = window(ref, model, devices, lambda s, _: servo_policy_action(policy, s, rs), steps, "servos", c5_push_schedule(ref), bullet_like)
    table = bins.table()
    report = {"law": law, "contact_model": contact_model, "envs": B, "steps": steps, "env_steps_per_regime": census, "one_step_defect": table,
              "metrics": METRICS, "regimes": REGIMES, "terminated_flag_mismatches": flags, "tolerances": TOLERANCES}
    if bullet_like:
        report["env_steps_with_the_oracles_live_points"] = {n: v[1] / max(v[0], 1) for n, v in points.items()}
    write_report(f"one_step_c5_{law}_law" + ("_bullet_like" if bullet_like else ""), report)
    for d in devices.values():
        d.close()
    assert census["sliding"] > 0.1 * B * steps or law != "torque", census  # the window IS the chaotic one
    failures = check(table, census)
    assert not failures, failures
    assert all(v == 0 for v in flags.values()), flags


def test_c2_window_one_step_from_the_oracle_state(monkeypatch):
    """The headline workload (BASELINE configs[1]: 4096 Upkie-Pendulum envs,
    README gains) over its 2200-step window, through the falls and the
    NEXT_STEP autoresets: `upkie_sim_step_pendulum` on every lane mapping, one
    step at a time from the oracle's state."""
    import bench
    from oracle import oracle as O
    from upkie_amd.model.default_model import default_model

    B, steps = bench.ENVS_PER_GPU, 2200
    cfg, model = bench.make_config(B), default_model()
    ref = O.Oracle(model, cfg)
    first = ref.reset()[:, [1, 0, 4, 3]]
    gains = np.array([10.0, 1.0, 0.0, 0.1])
    policy = lambda s, obs: (np.clip(obs.astype(np.float32).astype(np.float64) @ gains, -0.99, 0.99), None)  # noqa: E731
    devices = {f"{n}_lanes": HipStep(cfg, model, n, None, False, monkeypatch, pushes=False) for n in (8, 2, 1)}
    bins, census, _, flags = window(ref, model, devices, policy, steps, "pendulum", first_obs=first)
    table = bins.table()
    report = {"envs": B, "steps": steps, "env_steps_per_regime": census, "one_step_defect": table, "metrics": METRICS, "terminated_flag_mismatches": flags,
              "tolerances": TOLERANCES}
    write_report("one_step_c2", report)
    for d in devices.values():
        d.close()
    assert census["reset"] >= 0.5 * B, census  # the window reaches the falls
    failures = check(table, census)
    assert not failures, failures
    # an env whose pitch crosses fall_pitch within fp32 resolution of the threshold may be flagged a step apart
    assert all(v <= 4 for v in flags.values()), flags


def stops_policy(model):
    """A servo-level law that HOLDS hips and knees against their stops (ADVICE
    r5: nothing compared that path with the oracle): feedforward torques push
    every leg joint into a stop, no position target; wheels as
    torque_balancing.py."""
    policy = abi.torque_balancing_policy(10.0, 1.0, float(model.left_sign))
    for j, push in ((0, 3.0), (1, -3.0), (3, -3.0), (4, 3.0)):
        policy.action[j][0] = float("nan")
        policy.action[j][2] = push
        policy.action[j][3] = 0.0
        policy.action[j][4] = 0.2
    return policy


@pytest.mark.parametrize("contact_model", ["default", "bullet_like"])
def test_joints_held_at_their_stops_one_step_from_the_oracle_state(contact_model, monkeypatch):
    """UpkieServos agents may rest on a joint stop for most of an episode: 1024
    envs under a law that pushes every hip and knee into a stop, 300 steps, all
    lane mappings, both contact models (under the Bullet-like model the limit
    rows share the 50 sweeps with the contact rows)."""
    B, steps = 1024, 300
    bullet_like = contact_model == "bullet_like"
    ref, model, cfg = c5_oracle(B, bullet_like=bullet_like)
    policy = stops_policy(model)
    rs = float(model.left_sign) * float(model.wheel_radius)
    lanes = (8, 1) if bullet_like else (8, 2, 1)
    devices = {f"{n}_lanes": HipStep(cfg, model, n, ref.body_inertials, bullet_like, monkeypatch) for n in lanes}
    bins, census, points, flags = window(ref, model, devices, lambda s, _: servo_policy_action(policy, s, rs), steps, "servos", None, bullet_like)
    table = bins.table()
    report = {"contact_model": contact_model, "envs": B, "steps": steps, "env_steps_per_regime": census, "one_step_defect": table, "metrics": METRICS,
              "tolerances": TOLERANCES}
    write_report("one_step_joint_stops" + ("_bullet_like" if bullet_like else ""), report)
    for d in devices.values():
        d.close()
    assert census["joint_at_stop"] + census["stop_impact"] > 0.3 * B * steps, census
    if bullet_like:
        # Round 6: on eight lanes too the limit rows are rows of the specification's own sweeps now (until then that
        # mapping answered a joint at its stop with the DEFAULT model's solve: the first run of this test measured position 3e-4 /
        # 2e-2 / 0.1 and velocity 5e-2 / 1.7 / 100 there): both mappings are held to the same tolerances
        monkeypatch.delenv("UPKIE_LANES_PER_ENV", raising=False)
        probe = BatchedSim(cfg, model)
        probe.use_bullet_like_contacts()
        assert probe.lanes_per_env_of(abi.OBSERVATION_SERVOS) == 8
        probe.set_lanes_per_env(1)
        assert probe.lanes_per_env_of(abi.OBSERVATION_SERVOS) == 1
        probe.close()
    failures = check(table, census)
    assert not failures, failures

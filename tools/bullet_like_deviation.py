"""How far is the product's contact specification from what Bullet's multibody
solver is published to do? Runs BASELINE's C2 (Upkie-Pendulum, README agent)
and one GPU's share of C5 (UpkieServos, inertia and push randomisation, both
servo-level laws) on the fp64 oracle under BOTH contact specifications from
identical initial states -- the product's (one point per tire, exact solve +
sweeps to convergence, box friction in the rolling / lateral directions,
friction CFM 0.01) and the Bullet-like one of oracle/upkie_oracle.c
(persistent 4-point manifolds, 50 fixed warm-started sweeps, cone friction
along the sliding direction, no friction CFM; restated from SURVEY.md
Appendix B.1 / B.2, unverified: Bullet is not available here) -- and prints
the per-quantity deviation table of DESIGN.md section 4. CPU only.
Usage: python tools/bullet_like_deviation.py [envs]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from oracle import oracle as O
from tests.helpers import randomized_config
from upkie_amd import abi
from upkie_amd.model.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MARKS = (1, 10, 50, 200)


def pair(cfg, model):
    a, b = O.Oracle(model, cfg), O.Oracle(model, cfg)
    b.use_bullet_like_contacts()
    return a, b


def row(name, delta):
    delta = np.abs(np.asarray(delta))
    return f"  {name:34s} median {np.median(delta):9.2e}   p99 {np.percentile(delta, 99):9.2e}   max {delta.max():9.2e}"


def c2():
    print(f"C2: Upkie-Pendulum, README agent on the observation, {B} envs, bench.py's initial-state randomisation, no autoreset")
    model = Model().struct
    cfg = randomized_config(B, seed=0)
    ours, bullet = pair(cfg, model)
    oa = ours.reset()[:, [1, 0, 4, 3]]
    ob = bullet.reset()[:, [1, 0, 4, 3]]
    print(" after reset (one torque-free substep):")
    print(row("pitch [rad]", oa[:, 0] - ob[:, 0]))
    for k in range(1, max(MARKS) + 1):
        oa, _, ta, _ = ours.step_pendulum_agent(oa)
        ob, _, tb, _ = bullet.step_pendulum_agent(ob)
        if k in MARKS:
            live = (ta == 0) & (tb == 0)
            print(f" step {k} ({int(live.sum())} envs standing under both):")
            for i, name in enumerate(("pitch [rad]", "ground position [m]", "pitch rate [rad/s]", "ground velocity [m/s]")):
                print(row(name, (oa - ob)[live, i]))
            print(row("wheel torques [N m]", (ours.state - bullet.state)[abi.S_TORQUE + 2][live]))
            print(row("hip / knee torques [N m]", (ours.state - bullet.state)[[abi.S_TORQUE, abi.S_TORQUE + 1, abi.S_TORQUE + 3, abi.S_TORQUE + 4]][:, live]))
            print(row("base height [m]", (ours.state - bullet.state)[abi.S_POS + 2][live]))


def c5(law):
    print(f"C5 share: UpkieServos, {B} envs, inertia_variation 0.2, wheel friction 0.1, torso push every 400 steps (U(0, 20) N, random heading, 20 steps), law = {law}")
    model = Model().struct
    cfg = randomized_config(B, seed=0)
    cfg.joint_friction[2] = cfg.joint_friction[5] = 0.1
    ours, bullet = pair(cfg, model)
    r, sign = float(model.wheel_radius), float(model.left_sign)
    falls = [0, 0]
    first_fall = [np.full(B, -1), np.full(B, -1)]
    for o in (ours, bullet):
        o.body_inertials = o.sample_body_inertials(0.2)
        o.ext_force = np.zeros((3, B))
        o.ext_point = np.zeros(3)
        o.reset()
    act = np.zeros((B, 6, 6))
    act[:, :, 3] = 1.0
    act[:, :, 4] = 1.0
    act[:, :, 5] = 16.0
    act[:, [2, 5], 0] = np.nan
    steps = 800
    for k in range(steps):
        for idx, o in enumerate((ours, bullet)):
            if k % 400 == 0:
                o.ext_force = o.sample_pushes(k // 400, 20.0)
            elif k % 400 == 20:
                o.ext_force = np.zeros((3, B))
            st = o.state
            pitch = np.arcsin(np.clip(2.0 * (st[abi.S_QUAT] * st[abi.S_QUAT + 2] - st[abi.S_QUAT + 3] * st[abi.S_QUAT + 1]), -1, 1))
            pos = 0.5 * (st[abi.S_Q + 2] - st[abi.S_Q + 5]) * r * sign
            vel = 0.5 * (st[abi.S_QD + 2] - st[abi.S_QD + 5]) * r * sign
            a = act.copy()
            if law == "torque":
                a[:, [2, 5], 4] = 0.0
                a[:, 2, 2] = sign * 10.0 * pitch
                a[:, 5, 2] = -sign * 10.0 * pitch
            else:
                v = np.clip(10.0 * pitch + pos + 0.1 * vel, -0.99, 0.99) / r
                a[:, 2, 1] = sign * v
                a[:, 5, 1] = -sign * v
            o.step_servos(a)
            pitch = np.arcsin(np.clip(2.0 * (o.state[abi.S_QUAT] * o.state[abi.S_QUAT + 2] - o.state[abi.S_QUAT + 3] * o.state[abi.S_QUAT + 1]), -1, 1))
            fallen = np.abs(pitch) > 1.0
            if fallen.any():
                falls[idx] += int(fallen.sum())
                first_fall[idx] = np.where((first_fall[idx] < 0) & fallen, k, first_fall[idx])
                o.reset(mask=fallen.astype(np.uint8))
        if k + 1 in MARKS:
            never = (first_fall[0] < 0) & (first_fall[1] < 0)
            d = ours.state - bullet.state
            qa, qb = ours.state[abi.S_QUAT : abi.S_QUAT + 4], bullet.state[abi.S_QUAT : abi.S_QUAT + 4]
            pa = np.arcsin(np.clip(2.0 * (qa[0] * qa[2] - qa[3] * qa[1]), -1, 1))
            pb = np.arcsin(np.clip(2.0 * (qb[0] * qb[2] - qb[3] * qb[1]), -1, 1))
            print(f" step {k + 1} ({int(never.sum())} envs that fell under neither so far):")
            print(row("pitch [rad]", (pa - pb)[never]))
            print(row("base position x, y [m]", d[abi.S_POS : abi.S_POS + 2][:, never]))
            print(row("wheel velocities [rad/s]", d[[abi.S_QD + 2, abi.S_QD + 5]][:, never]))
            print(row("wheel torques [N m]", d[[abi.S_TORQUE + 2, abi.S_TORQUE + 5]][:, never]))
    for name, idx in (("product spec", 0), ("Bullet-like", 1)):
        ff = first_fall[idx]
        print(f" {name:13s}: {falls[idx]} falls in {steps} steps x {B} envs; {int((ff >= 0).sum())} envs fell at least once, median step of the first fall {int(np.median(ff[ff >= 0])) if (ff >= 0).any() else -1}")


c2()
c5("velocity")
c5("torque")

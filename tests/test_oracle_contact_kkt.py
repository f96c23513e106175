"""The oracle's contact solve against the DEFINITION of the problem it solves,
not against another solver: every system the sweeps were needed for, captured
while one GPU's share of BASELINE's C5 runs on the oracle (robots pushed,
skidding, tipping over, landing), must satisfy the complementarity conditions
of the box-friction contact problem Bullet's solver iterates on
(btSequentialImpulseConstraintSolver: lower/upper limits of a friction row =
-/+ mu x the normal impulse of its contact):

  normal rows     lam >= 0,  w >= 0,  lam w = 0            (w = (W + CFM) lam - rhs)
  friction rows   |lam| <= mu lam_n;  strictly inside: w = 0;
                  on the upper bound: w <= 0;  on the lower bound: w >= 0

Test infrastructure only (oracle/upkie_oracle.c, oracle_debug_capture)."""

import ctypes as C

import numpy as np

from oracle import oracle as O
from tests.helpers import randomized_config
from upkie_amd import abi
from upkie_amd.model.model import Model


def run_c5_share_on_the_oracle(B, steps, threshold, law="velocity", rows=0):
    cfg = randomized_config(B, seed=0)
    cfg.rand_pitch = 0.1
    cfg.joint_friction[2] = cfg.joint_friction[5] = 0.1
    model = Model().struct
    oracle = O.Oracle(model, cfg)
    oracle.body_inertials = oracle.sample_body_inertials(0.2)
    rng = np.random.default_rng(0)
    force = np.zeros((3, B))
    force[0] = rng.uniform(-5, 5, B)
    oracle.ext_force = force
    oracle.ext_point = np.array([0.0, 0.0, -0.1])
    oracle.reset()
    lib = O.lib()
    C.c_long.in_dll(lib, "oracle_debug_captured").value = 0
    C.c_long.in_dll(lib, "oracle_debug_capture_threshold").value = threshold
    C.c_long.in_dll(lib, "oracle_debug_capture_rows").value = rows  # (0: systems of any size)
    r, sign = float(model.wheel_radius), float(model.left_sign)
    act = np.zeros((B, 6, 6))
    act[:, :, 3] = 1.0
    act[:, :, 4] = 1.0
    act[:, :, 5] = 16.0
    act[:, [2, 5], 0] = np.nan
    for _ in range(steps):
        st = oracle.state
        pitch = 2.0 * st[abi.S_QUAT + 2]
        pos = 0.5 * (st[abi.S_Q + 2] - st[abi.S_Q + 5]) * r * sign
        v = np.clip(10.0 * pitch + pos, -0.99, 0.99) / r
        if law == "torque":  # examples/pybullet/torque_balancing.py:15-37: wheel torques +-10 x pitch, kd_scale 0
            act[:, [2, 5], 4] = 0.0
            act[:, 2, 2] = sign * 10.0 * pitch
            act[:, 5, 2] = -sign * 10.0 * pitch
        else:  # the README balancer through the wheels' velocity loop
            act[:, 2, 1] = sign * v
            act[:, 5, 1] = -sign * v
        oracle.step_servos(act)
        fallen = np.abs(2.0 * oracle.state[abi.S_QUAT + 2]) > 1.0
        if fallen.any():
            oracle.reset(mask=fallen.astype(np.uint8))
    captured = max(0, min(C.c_long.in_dll(lib, "oracle_debug_captured").value, 4096))
    cases = np.ctypeslib.as_array((C.c_double * (4096 * 55)).in_dll(lib, "oracle_debug_capture")).reshape(4096, 55)[:captured].copy()
    C.c_long.in_dll(lib, "oracle_debug_capture_threshold").value = 0
    C.c_long.in_dll(lib, "oracle_debug_capture_rows").value = 0
    return cases, float(model.friction_mu)


def check_complementarity(cases, mu):
    checked = {"sticking": 0, "sliding": 0, "lifted": 0, "lateral_on_a_bound": 0}
    for c in cases:
        n = int(c[0])
        assert n in (3, 6)
        A = c[1:37].reshape(6, 6)[:n, :n]
        rhs, lam = c[37 : 37 + n], c[49 : 49 + n]
        w = A @ lam - rhs
        scale = max(np.abs(lam).max(), np.abs(rhs / np.diag(A)).max(), 1e-9)
        tol = 2e-5 * scale * np.diag(A).max()  # the sweeps stop at 1e-6 of the largest impulse
        for tire in range(n // 3):
            k = 3 * tire
            assert lam[k] >= 0.0 and w[k] >= -tol and abs(lam[k] * w[k]) <= tol * scale, (lam, w)
            if lam[k] == 0.0:
                assert lam[k + 1] == 0.0 and lam[k + 2] == 0.0
                checked["lifted"] += 1
            bound = mu * lam[k]
            # EVERY friction row on its own, the lateral rows of two touching tires included (solved as a pair, exactly,
            # since round 3: until then only their sum could be held to the definition)
            for t in (k + 1, k + 2):
                assert abs(lam[t]) <= bound * (1 + 1e-12) + 1e-300
                if abs(lam[t]) < bound * (1 - 1e-9):
                    assert abs(w[t]) <= tol, (t, lam, w)
                    checked["sticking"] += 1
                elif bound > 0.0:
                    assert (w[t] <= tol) if lam[t] > 0.0 else (w[t] >= -tol), (t, lam, w)
                    checked["sliding"] += 1
                    checked["lateral_on_a_bound"] += 1 if t == k + 2 else 0
    return checked


def test_contact_impulses_satisfy_the_complementarity_conditions():
    cases, mu = run_c5_share_on_the_oracle(B=256, steps=250, threshold=2)  # the systems that needed the sweeps at all
    assert len(cases) > 300, len(cases)
    checked = check_complementarity(cases, mu)
    assert checked["sticking"] > 100 and checked["sliding"] > 100 and checked["lifted"] > 10, checked


def test_torque_law_systems():
    """The reference example's own servo law (examples/pybullet/torque_balancing.py:15-37) makes the robots run away,
    skid and tumble sideways: tires lifted, lateral rows on their bounds -- the systems whose lateral pair the
    round-2 sweeps got wrong (clamped free-pair solution, up to O(1) violation of the conditions below, 0.1 % of the
    systems at the 50-sweep cap). No system reaches the cap any more."""
    lib = O.lib()
    hist = (C.c_long * 64).in_dll(lib, "oracle_debug_sweep_hist")
    for i in range(64):
        hist[i] = 0
    cases, mu = run_c5_share_on_the_oracle(B=128, steps=300, threshold=2, law="torque")
    assert len(cases) > 2000, len(cases)
    checked = check_complementarity(cases, mu)
    assert checked["sliding"] > 1000 and checked["lifted"] > 300 and checked["lateral_on_a_bound"] > 200, checked
    sweeps = np.array(list(hist))
    assert sweeps[50:].sum() == 0 and sweeps.sum() > 20000, sweeps.tolist()  # nothing at the cap

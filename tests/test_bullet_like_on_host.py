"""The device's Bullet-like contact specification (upkie_amd/csrc/bullet_like.hpp:
persistent 4-point manifolds, 50 fixed warm-started sequential-impulse sweeps,
cone friction, no friction CFM) compiled for the HOST and run substep by
substep against its fp64 twin in the oracle (`bullet_like_contacts`,
oracle/upkie_oracle.c: dense Delassus matrix, world frame): same manifold
bookkeeping (which points are cached, replaced, dropped), same impulses, same
motion -- without a GPU. The GPU tests (tests/test_bullet_like_gpu.py) hold
the kernels to the oracle over whole env.step() rollouts."""

import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from upkie_amd import abi
from upkie_amd.model.default_model import default_model

from .test_device_arithmetic_on_host import harness, random_state  # noqa: F401 (fixture)

WORDS = 64


def both(harness, model, s64, manifold64, tau, h=1e-3):  # noqa: F811
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    so, mo = s64.copy(), manifold64.copy()
    co = O.lib().oracle_substep_bullet_like(C.byref(model), p(so), p(np.ascontiguousarray(tau, dtype=np.float64)), C.c_double(h), p(mo))
    return so, mo, co


def device(harness, model, s32, manifold32, tau, h=1e-3):  # noqa: F811
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    harness.harness_substep_bullet_like.restype = C.c_int
    t32 = np.ascontiguousarray(tau, dtype=np.float32)
    rc = harness.harness_substep_bullet_like(C.byref(model), p(s32), p(t32), C.c_float(h), p(manifold32))
    assert rc >= 0
    return rc


def run_sequence(harness, model, s0, taus, h=1e-3):  # noqa: F811
    """The same substeps on both sides, each carrying its own state and manifold."""
    so, mo = s0.copy(), np.zeros(WORDS)
    sh, mh = s0.astype(np.float32), np.zeros(WORDS, dtype=np.float32)
    trace = []
    for tau in taus:
        so, mo, co = both(harness, model, so, mo, tau, h)
        ch = device(harness, model, sh, mh, tau, h)
        live_o, live_h = mo.reshape(2, 4, 8)[:, :, 7], mh.reshape(2, 4, 8)[:, :, 7]
        trace.append((so.copy(), sh.astype(np.float64), mo.copy(), mh.astype(np.float64), co, ch, live_o.copy(), live_h.copy()))
    return trace


def test_standing_robot_keeps_one_point_per_tire_and_matches_the_oracle(harness):  # noqa: F811
    rng = np.random.default_rng(11)
    model = default_model()
    worst = np.zeros(25)
    worst_impulse = 0.0
    for _ in range(20):
        s = random_state(rng, True)
        s[abi.S_QUAT:abi.S_QUAT + 4] = [1, 0, 0, 0]
        s[abi.S_Q:abi.S_Q + 6] = 0
        s[abi.S_QD:abi.S_QD + 6] = rng.uniform(-0.5, 0.5, 6)
        s[abi.S_LINVEL:abi.S_LINVEL + 3] = rng.uniform(-0.05, 0.05, 3)
        s[abi.S_ANGVEL:abi.S_ANGVEL + 3] = rng.uniform(-0.1, 0.1, 3)
        s[abi.S_POS + 2] = 0.6
        # let it land first (the oracle's own substeps put the tires on the floor), then compare 40 substeps
        probe, manifold = s.copy(), np.zeros(WORDS)
        for _ in range(400):
            probe, manifold, contact = both(harness, model, probe, manifold, np.zeros(6))
        assert contact == 1
        taus = rng.uniform(-0.5, 0.5, (40, 6))
        so, mo = probe.copy(), manifold.copy()
        sh, mh = probe.astype(np.float32), manifold.astype(np.float32)
        for tau in taus:
            so, mo, co = both(harness, model, so, mo, tau)
            ch = device(harness, model, sh, mh, tau)
            assert co == ch == 1
            live_o, live_h = mo.reshape(2, 4, 8)[:, :, 7], mh.reshape(2, 4, 8)[:, :, 7]
            assert np.array_equal(live_o, live_h)
            assert live_o.sum(axis=1).tolist() == [1.0, 1.0]  # a wheel that rolls keeps ONE cached point
            worst = np.maximum(worst, np.abs(so[:25] - sh[:25].astype(np.float64)))
            worst_impulse = max(worst_impulse, float(np.abs(mo.reshape(2, 4, 8)[:, :, 6] - mh.reshape(2, 4, 8)[:, :, 6]).max()))
    assert worst[0:3].max() < 2e-6 and worst[3:7].max() < 2e-6, worst
    assert worst[7:10].max() < 5e-4 and worst[10:13].max() < 2e-3, worst
    # (round 6: the robots land with free legs, which fold onto their stops; a joint within reach of its stop now lists its limit
    # row -- joint_limit_row, dynamics.hpp -- so these substeps go through the general row list, whose running velocity change
    # carries more fp32 rounding than the six-row fast path they used to take: wheel rates within 5e-2 rad/s as before, which
    # over 40 substeps is up to 1e-4 rad of wheel angle; hips and knees as before)
    assert worst[[13, 14, 16, 17]].max() < 5e-6 and worst[[15, 18]].max() < 1e-4 and worst[19:25].max() < 5e-2, worst
    assert worst_impulse < 2e-4, worst_impulse  # normal impulses ~ 0.026 N.s per tire and substep


def test_random_states_on_the_floor_sequences_match_the_oracle(harness):  # noqa: F811
    """Pitched / rolled / yawed robots with spinning wheels dropped within reach
    of the floor: sliding contacts (friction along the sliding direction), points
    appearing and disappearing, 25 substeps each."""
    rng = np.random.default_rng(12)
    model = default_model()
    same_cache = total = 0
    worst = np.zeros(25)
    for _ in range(40):
        s = random_state(rng, True)
        s[abi.S_POS + 2] = rng.uniform(0.52, 0.62)
        taus = rng.uniform(-1.0, 1.0, (25, 6))
        for so, sh, mo, mh, co, ch, live_o, live_h in run_sequence(harness, model, s, taus):
            total += 1
            if np.array_equal(live_o, live_h) and co == ch:
                same_cache += 1
                worst = np.maximum(worst, np.abs(so[:25] - sh[:25]))
            else:
                break  # (a point cached on one side only: the sequences part there -- counted, not compared further)
    assert same_cache >= 0.98 * total, (same_cache, total)
    assert worst[0:3].max() < 5e-6 and worst[3:7].max() < 5e-6, worst
    assert worst[7:10].max() < 5e-3 and worst[10:13].max() < 2e-2, worst  # sliding tires: friction at its cone, 25 substeps
    assert worst[13:19].max() < 1e-4, worst


def test_a_knee_at_its_stop_is_a_row_of_the_same_solve(harness):  # noqa: F811
    rng = np.random.default_rng(13)
    model = default_model()
    hits = 0
    for _ in range(10):
        s = random_state(rng, True)
        s[abi.S_QUAT:abi.S_QUAT + 4] = [1, 0, 0, 0]
        s[abi.S_Q:abi.S_Q + 6] = 0
        j = int(rng.integers(0, 2)) * 3 + 1  # a knee
        s[abi.S_Q + j] = model.joint_upper[j] + 0.01 if rng.uniform() < 0.5 else model.joint_lower[j] - 0.01
        s[abi.S_QD + j] = rng.uniform(-1, 1)
        taus = np.zeros((10, 6))
        taus[:, j] = rng.uniform(-2, 2)
        for so, sh, mo, mh, co, ch, live_o, live_h in run_sequence(harness, model, s, taus):
            assert np.array_equal(live_o, live_h)
            assert np.abs(so[abi.S_Q:abi.S_Q + 6] - sh[abi.S_Q:abi.S_Q + 6]).max() < 2e-5
            assert np.abs(so[abi.S_QD:abi.S_QD + 6] - sh[abi.S_QD:abi.S_QD + 6]).max() < 5e-2
            hits += 1
    assert hits == 100


def test_free_flight_has_no_rows_and_matches_the_default_substep(harness):  # noqa: F811
    rng = np.random.default_rng(14)
    model = default_model()
    for _ in range(10):
        s = random_state(rng, False)
        tau = rng.uniform(-1, 1, 6)
        trace = run_sequence(harness, model, s, [tau])
        so, sh, mo, mh, co, ch, live_o, live_h = trace[0]
        assert co == ch == 0 and not live_o.any() and not live_h.any()
        assert np.abs(so[:13] - sh[:13]).max() < 2e-5


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rot_quat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (w * y + x * z)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_full_manifolds_four_points_per_tire_match_the_oracle(harness):  # noqa: F811
    """A rolling wheel never holds more than one point (its deepest point
    replaces the cached one), so the 4-point bookkeeping is exercised with
    manifolds filled by hand: three more points per tire on the tire circle,
    0.25 / -0.25 / 0.5 rad from the cached one, live, with applied impulses --
    24 contact rows in the first substep; then replacement of the nearest
    point, eviction of the shallowest when full, dropping beyond the threshold
    as the robot moves on, on both sides alike."""
    rng = np.random.default_rng(15)
    model = default_model()
    most_rows = 0
    worst = np.zeros(25)
    for trial in range(12):
        s = random_state(rng, True)
        s[abi.S_QUAT:abi.S_QUAT + 4] = [1, 0, 0, 0]
        s[abi.S_Q:abi.S_Q + 6] = rng.uniform(-0.2, 0.2, 6)
        s[abi.S_QD:abi.S_QD + 6] = 0
        s[abi.S_LINVEL:abi.S_LINVEL + 3] = 0
        s[abi.S_ANGVEL:abi.S_ANGVEL + 3] = 0
        s[abi.S_POS + 2] = 0.6
        manifold = np.zeros(WORDS)
        for _ in range(300):  # land
            s, manifold, contact = both(harness, model, s, manifold, np.zeros(6))
        assert contact == 1
        R = _rot_quat(s[abi.S_QUAT:abi.S_QUAT + 4])
        m = manifold.reshape(2, 4, 8)
        for w in range(2):
            live = np.nonzero(m[w, :, 7])[0]
            assert len(live) == 1
            p0 = m[w, live[0]].copy()
            psi = sum(float(np.sign(model.joint_axis[3 * w + j][1])) * s[abi.S_Q + 3 * w + j] for j in range(3))
            centre = np.array(model.wheel_center[w])
            free = [p for p in range(4) if p != live[0]]
            for slot, angle in zip(free, (0.25, -0.25, 0.5)):
                local = centre + _rot_y(angle) @ (p0[:3] - centre)
                world_delta = R @ _rot_y(psi) @ (local - p0[:3])
                m[w, slot, :3] = local
                m[w, slot, 3:5] = p0[3:5] + world_delta[:2]
                m[w, slot, 5] = 0.0
                m[w, slot, 6] = 0.005
                m[w, slot, 7] = 1.0
        manifold = m.reshape(-1)
        so, mo = s.copy(), manifold.copy()
        sh, mh = s.astype(np.float32), manifold.astype(np.float32)
        taus = np.zeros((12, 6))
        taus[:, 2] = rng.uniform(-0.5, 0.5)
        taus[:, 5] = rng.uniform(-0.5, 0.5)
        for k, tau in enumerate(taus):
            so, mo, co = both(harness, model, so, mo, tau)
            ch = device(harness, model, sh, mh, tau)
            live_o, live_h = mo.reshape(2, 4, 8)[:, :, 7], mh.reshape(2, 4, 8)[:, :, 7]
            assert co == ch and np.array_equal(live_o, live_h), (trial, k, live_o, live_h)
            most_rows = max(most_rows, int(3 * live_o.sum()))
            worst = np.maximum(worst, np.abs(so[:25] - sh[:25].astype(np.float64)))
            assert np.abs(mo.reshape(2, 4, 8)[:, :, :6] - mh.reshape(2, 4, 8)[:, :, :6]).max() < 1e-5  # the same points cached
    assert most_rows >= 21, most_rows  # (full or nearly full manifolds were solved)
    assert worst[0:3].max() < 5e-6 and worst[3:7].max() < 5e-6 and worst[13:19].max() < 5e-5, worst
    assert worst[7:10].max() < 5e-3 and worst[10:13].max() < 2e-2, worst


def test_eight_lane_bullet_like_substep_matches_the_oracle(harness):  # noqa: F811
    """The eight-lane kernel's Bullet-like substep (octet.hpp, oct_bullet_like_solve: the default rows rotated into the
    sliding direction, dense 6 x 6 sweeps in every lane) run as eight lockstep host threads, substep by substep against
    the oracle's twin: landed robots rolling and sliding, 30 substeps each; applied normal impulses carried along."""
    rng = np.random.default_rng(21)
    model = default_model()
    harness.harness_substep_octet_bullet_like.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    worst = np.zeros(25)
    worst_applied = 0.0
    compared = 0
    for trial in range(16):
        s = random_state(rng, True)
        if trial % 2 == 0:  # upright, rolling
            s[abi.S_QUAT:abi.S_QUAT + 4] = [1, 0, 0, 0]
            s[abi.S_Q:abi.S_Q + 6] = 0
            s[abi.S_LINVEL:abi.S_LINVEL + 3] = rng.uniform(-0.05, 0.05, 3)
            s[abi.S_ANGVEL:abi.S_ANGVEL + 3] = rng.uniform(-0.1, 0.1, 3)
        manifold = np.zeros(WORDS)
        for _ in range(300):  # land on the oracle, legs held at zero by a PD law
            hold = np.zeros(6)
            for j in (0, 1, 3, 4):
                hold[j] = np.clip(20.0 * (0.0 - s[abi.S_Q + j]) - 1.0 * s[abi.S_QD + j], -10.0, 10.0)
            s, manifold, contact = both(harness, model, s, manifold, hold)
        if contact != 1:
            continue
        so, mo = s.copy(), manifold.copy()
        sh = s.astype(np.float32)
        m = manifold.reshape(2, 4, 8)
        applied = np.array([(m[w, :, 6] * m[w, :, 7]).sum() for w in range(2)], dtype=np.float32)
        taus = rng.uniform(-0.8, 0.8, (30, 6))
        status = np.zeros(1, dtype=np.int32)
        lower, upper = np.array(model.joint_lower[:6]), np.array(model.joint_upper[:6])
        for tau in taus:
            q = so[abi.S_Q:abi.S_Q + 6]
            # legs held by their servos, as in the envs the eight-lane variant serves (a limp leg folds onto its stops,
            # and a joint at its stop is outside this variant: it takes the default model's joint-stop path there)
            tau = tau.copy()
            for j in (0, 1, 3, 4):
                tau[j] = np.clip(20.0 * (0.0 - q[j]) - 1.0 * so[abi.S_QD + j], -10.0, 10.0)
            if np.any((q <= lower + 1e-3) | (q >= upper - 1e-3)):
                break
            so, mo, co = both(harness, model, so, mo, tau)
            t32 = np.ascontiguousarray(tau, dtype=np.float32)
            ok = harness.harness_substep_octet_bullet_like(C.byref(model), p(sh), p(t32), C.c_float(1e-3), 1, p(status), p(applied))
            assert ok == 1  # the eight lanes agree on the base
            live = mo.reshape(2, 4, 8)[:, :, 7]
            if live.sum(axis=1).max() > 1 or (status[0] == 1) != (co == 1):
                break  # (more than one cached point on a tire: outside the eight-lane variant's case -- not expected)
            compared += 1
            worst = np.maximum(worst, np.abs(so[:25] - sh[:25].astype(np.float64)))
            ref_applied = (mo.reshape(2, 4, 8)[:, :, 6] * live).sum(axis=1)
            worst_applied = max(worst_applied, float(np.abs(ref_applied - applied).max()))
    assert compared >= 400, compared
    assert worst[0:3].max() < 5e-6 and worst[3:7].max() < 5e-6 and worst[13:19].max() < 5e-5, worst
    assert worst[7:10].max() < 5e-3 and worst[10:13].max() < 2e-2, worst
    assert worst_applied < 5e-4, worst_applied


def test_eight_lane_substep_with_joints_at_their_stops_matches_the_oracle(harness):  # noqa: F811
    """Round 6: on the eight-lane mapping a Bullet-like substep with a hip or knee within
    reach of its stop used to take the DEFAULT model's joint-stop solve. It now gathers its rows -- the tires' and the
    limit rows -- into the general row list and runs the specification's own 50 sweeps on them
    (general_constraint_solve_bullet_like): landed robots whose leg joints are PUSHED into their stops by constant
    torques, every substep taken by the eight lockstep host threads from the oracle twin's state (one substep at a time:
    the same rows on both sides), applied normal impulses carried along."""
    rng = np.random.default_rng(31)
    model = default_model()
    harness.harness_substep_octet_bullet_like.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    lower, upper = np.array(model.joint_lower[:6]), np.array(model.joint_upper[:6])
    worst = np.zeros(25)
    worst_applied = 0.0
    compared = at_stop = 0
    for trial in range(12):
        s = random_state(rng, True)
        s[abi.S_QUAT:abi.S_QUAT + 4] = [1, 0, 0, 0]
        s[abi.S_Q:abi.S_Q + 6] = 0
        s[abi.S_LINVEL:abi.S_LINVEL + 3] = rng.uniform(-0.05, 0.05, 3)
        s[abi.S_ANGVEL:abi.S_ANGVEL + 3] = rng.uniform(-0.1, 0.1, 3)
        push = rng.choice([-3.0, 3.0], 6)
        push[[2, 5]] = rng.uniform(-0.3, 0.3, 2)
        so, mo = s.copy(), np.zeros(WORDS)
        status = np.zeros(1, dtype=np.int32)
        for k in range(500):
            q, qd = so[abi.S_Q:abi.S_Q + 6], so[abi.S_QD:abi.S_QD + 6]
            tau = push - 0.2 * qd
            near = np.any((q[[0, 1, 3, 4]] <= lower[[0, 1, 3, 4]] + 0.02) | (q[[0, 1, 3, 4]] >= upper[[0, 1, 3, 4]] - 0.02))
            live = mo.reshape(2, 4, 8)[:, :, 7]
            if k >= 100 and near and live.sum(axis=1).max() <= 1:
                sh = so.astype(np.float32)
                applied = (mo.reshape(2, 4, 8)[:, :, 6] * live).sum(axis=1).astype(np.float32)
                t32 = np.ascontiguousarray(tau, dtype=np.float32)
                s_ref, m_ref, c_ref = both(harness, model, so.astype(np.float32).astype(np.float64), mo.astype(np.float32).astype(np.float64), t32.astype(np.float64))
                ok = harness.harness_substep_octet_bullet_like(C.byref(model), p(sh), p(t32), C.c_float(1e-3), 1, p(status), p(applied))
                assert ok == 1
                live_ref = m_ref.reshape(2, 4, 8)[:, :, 7]
                if live_ref.sum(axis=1).max() <= 1 and (status[0] == 1) == (c_ref == 1):
                    compared += 1
                    at_stop += int(np.any((q[[0, 1, 3, 4]] <= lower[[0, 1, 3, 4]]) | (q[[0, 1, 3, 4]] >= upper[[0, 1, 3, 4]])))
                    worst = np.maximum(worst, np.abs(s_ref[:25] - sh[:25].astype(np.float64)))
                    ref_applied = (m_ref.reshape(2, 4, 8)[:, :, 6] * live_ref).sum(axis=1)
                    worst_applied = max(worst_applied, float(np.abs(ref_applied - applied).max()))
            so, mo, _ = both(harness, model, so, mo, tau)
    assert compared >= 1500 and at_stop >= 300, (compared, at_stop)
    assert worst[0:3].max() < 2e-6 and worst[3:7].max() < 2e-6 and worst[[13, 14, 16, 17]].max() < 5e-6, worst
    assert worst[7:10].max() < 2e-3 and worst[10:13].max() < 1e-2 and worst[19:25].max() < 5e-2, worst
    assert worst_applied < 5e-4, worst_applied


class BulletLikeProbe(C.Structure):
    """`BulletLikeProbe` of upkie_amd/csrc/bullet_like.hpp (host build only)."""

    _fields_ = [("system", C.c_float * 50), ("change", C.c_float * 64), ("lam", (C.c_float * 6) * 64), ("sweeps", C.c_int)]


def bullet_like_sweep_statistics(harness, trials=10, substeps=240, seed=33):  # noqa: F811
    """Rolling robots (legs held by their servos, small random wheel torques: the regime of the bench workloads) on
    the eight-lane Bullet-like substep of the HOST build, every solve probed: per solve the first sweep that (a) changed
    no bit of any impulse, (b) reproduced the impulses of 1-8 sweeps earlier (a limit cycle: the result of all 50 sweeps
    follows from the phase), (c) moved no impulse by more than 2.4e-7 / 1e-5 of the largest one (fp32 resolution / the
    default model's sweep tolerance); 51 = never within the 50. Also used by tools/archive/bullet_like_sweeps.py."""
    assert harness.harness_bullet_like_probe_bytes() == C.sizeof(BulletLikeProbe)
    harness.harness_substep_octet_bullet_like.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    model = default_model()
    rng = np.random.default_rng(seed)
    probe = BulletLikeProbe()
    rows = []
    harness.harness_bullet_like_probe(C.byref(probe))
    try:
        for trial in range(trials):
            s = random_state(rng, True)
            s[abi.S_QUAT:abi.S_QUAT + 4] = [1, 0, 0, 0]
            s[abi.S_Q:abi.S_Q + 6] = 0
            s[abi.S_LINVEL:abi.S_LINVEL + 3] = rng.uniform(-0.05, 0.05, 3)
            s[abi.S_ANGVEL:abi.S_ANGVEL + 3] = rng.uniform(-0.1, 0.1, 3)
            s8, applied, status = s.astype(np.float32), np.zeros(2, dtype=np.float32), np.zeros(1, dtype=np.int32)
            for _ in range(substeps):
                tau = rng.uniform(-0.8, 0.8, 6).astype(np.float32)
                for j in (0, 1, 3, 4):
                    tau[j] = np.clip(20.0 * (0.0 - s8[abi.S_Q + j]) - 1.0 * s8[abi.S_QD + j], -10.0, 10.0)
                probe.sweeps = 0
                assert harness.harness_substep_octet_bullet_like(C.byref(model), p(s8), p(tau), C.c_float(1e-3), 1, p(status), p(applied)) == 1
                if status[0] != 1 or probe.sweeps == 0:
                    continue
                n = probe.sweeps
                lam = np.array([list(probe.lam[i]) for i in range(n)], dtype=np.float32)
                change = np.array(probe.change[:n])
                scale = np.abs(lam).max(axis=1)
                first = lambda mask: int(np.argmax(mask)) + 1 if mask.any() else n + 1  # noqa: E731
                cycle = n + 1
                for it in range(1, n):
                    if any(it >= per and np.array_equal(lam[it].view(np.uint32), lam[it - per].view(np.uint32)) for per in range(1, 9)):
                        cycle = it + 1
                        break
                rows.append((first(change == 0.0), cycle, first(change <= 2.4e-7 * scale), first(change <= 1e-5 * scale),
                             float(np.abs(lam[min(first(change <= 2.4e-7 * scale), n) - 1] - lam[n - 1]).max() / max(scale[n - 1], 1e-30))))
    finally:
        harness.harness_bullet_like_probe(None)
    return np.array(rows), int(model.pgs_iterations)


def test_why_the_fixed_sweeps_of_the_bullet_like_model_stay_fixed(harness):  # noqa: F811
    """VERDICT r4 asked to leave the 50-sweep loop 'as soon as a full sweep changes no impulse bit' and to measure how
    many sweeps that is. Measured on the host arithmetic (the device's, sweep for sweep): the exact fixed point is reached
    inside the 50 sweeps by a minority of the solves (fp32 settles into 1-2 ulp limit cycles, and the two tires'
    nearly parallel friction rows converge like 0.74^sweep); with cycles of period <= 8 detected, or stopping at fp32
    resolution instead, a single env would save sweeps -- but a wavefront sweeps eight envs in lockstep and leaves with
    its slowest: the mean over wavefronts stays above 45 of 50. So the kernels keep the fixed count (and their results
    stay those of the published algorithm, sweep for sweep); profiles/r05_bullet_like_sweeps.txt has the table."""
    rows, cap = bullet_like_sweep_statistics(harness, trials=6, substeps=200)
    assert len(rows) >= 1000 and cap == 50
    exact, cycle, resolution, loose, deviation = rows.T
    rng = np.random.default_rng(0)
    wave = lambda counts: float(np.mean([np.minimum(counts[rng.integers(0, len(counts), 8)], cap).max() for _ in range(2000)]))  # noqa: E731
    assert np.mean(exact > cap) >= 0.3          # the exact fixed point: not reached in 50 sweeps by a third or more of the solves
    assert wave(exact) >= 48 and wave(cycle) >= 45 and wave(resolution) >= 45  # what a wavefront of eight envs would run
    assert np.quantile(deviation, 0.99) <= 5e-6  # (stopping at fp32 resolution would have been accurate -- it just does not pay)

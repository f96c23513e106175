"""The product's device functions (upkie_amd/csrc/dynamics.hpp) are
__host__ __device__: compile them for the host and compare single physics
substeps with the oracle on random states. This checks the kernel arithmetic
on CPU-only machines; the GPU tests then check the launches themselves."""

import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from upkie_amd import abi
from upkie_amd.model.default_model import default_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_harness.hip")
LIB = os.path.join(ROOT, "tests", "_host_harness.so")


@pytest.fixture(scope="module")
def harness():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "upkie_amd", "csrc")
    deps = [SRC, os.path.join(ROOT, "include", "upkie_hip.h")] + [os.path.join(csrc, n) for n in sorted(os.listdir(csrc))]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run(
            ["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", LIB],
            check=True, capture_output=True,
        )
    return C.CDLL(LIB)


def random_state(rng, on_floor: bool):
    s = np.zeros(abi.STATE_WORDS)
    pitch = rng.uniform(-0.3, 0.3)
    roll = rng.uniform(-0.05, 0.05)
    yaw = rng.uniform(-3, 3)
    cy, sy, cp, sp, cr, sr = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(roll / 2), np.sin(roll / 2)
    s[abi.S_QUAT : abi.S_QUAT + 4] = [
        cy * cp * cr + sy * sp * sr, cy * cp * sr - sy * sp * cr, cy * sp * cr + sy * cp * sr, sy * cp * cr - cy * sp * sr]
    s[abi.S_Q : abi.S_Q + 6] = rng.uniform(-0.4, 0.4, 6)
    s[abi.S_QD : abi.S_QD + 6] = rng.uniform(-2, 2, 6)
    s[abi.S_LINVEL : abi.S_LINVEL + 3] = rng.uniform(-0.5, 0.5, 3)
    s[abi.S_ANGVEL : abi.S_ANGVEL + 3] = rng.uniform(-1, 1, 3)
    s[abi.S_POS : abi.S_POS + 3] = [rng.uniform(-1, 1), rng.uniform(-1, 1), 0.6 if on_floor else 2.0]
    return s


def make_slots(bodies, points, local):
    slots = abi.UpkieExternalForces()
    slots.count = len(bodies)
    for i in range(len(bodies)):
        slots.body[i] = bodies[i]
        slots.local[i] = 1 if local[i] else 0
        for k in range(3):
            slots.point[i][k] = points[i][k]
    return slots


def run_both(harness, model, s64, tau, h=1e-3, scale=None, force=None, point=None, slots=None):
    """`force` [3] with `point` (trunk, world frame) or `force` [count, 3] with `slots`."""
    so = s64.copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    if force is not None and slots is None:
        slots = make_slots([0], [point], [False])
        force = np.asarray(force, dtype=np.float64).reshape(1, 3)
    force = None if force is None else np.ascontiguousarray(force, dtype=np.float64)
    O.lib().oracle_substep_ext(C.byref(model), p(so), p(tau), C.c_double(h), p(scale), p(force), C.byref(slots) if slots is not None else None)
    s32 = s64.astype(np.float32)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32) if a is not None else None
    t32, sc32, fo32 = f32(tau), f32(scale), f32(force)
    harness.harness_substep.restype = C.c_int
    rc = harness.harness_substep(C.byref(model), p(s32), p(t32), C.c_float(h), p(sc32), p(fo32), C.byref(slots) if slots is not None else None)
    assert rc >= 0
    return so, s32.astype(np.float64)


@pytest.mark.parametrize("on_floor", [False, True])
def test_substep_matches_oracle(harness, on_floor):
    rng = np.random.default_rng(5)
    model = default_model()
    worst = np.zeros(25)
    for _ in range(200):
        s = random_state(rng, on_floor)
        if on_floor:  # put the lower tire within the contact range of the floor
            probe = s.copy()
            O.lib().oracle_substep(C.byref(model), probe.ctypes.data_as(C.c_void_p), np.zeros(6).ctypes.data_as(C.c_void_p), C.c_double(0.0), None, None, None)
        tau = rng.uniform(-1.5, 1.5, 6)
        so, sh = run_both(harness, model, s, tau)
        worst = np.maximum(worst, np.abs(so[:25] - sh[:25]))
    assert worst[0:3].max() < 5e-7  # position
    assert worst[3:7].max() < 5e-7  # quaternion
    assert worst[7:10].max() < 2e-4  # linear velocity (gap / h amplifies fp32 z)
    assert worst[10:13].max() < 1e-3  # angular velocity
    assert worst[13:19].max() < 1e-6  # joint angles
    assert worst[19:25].max() < 2e-2  # joint velocities (wheel inertia 2.8e-4)


def records_of_model(model, scale=None):
    """Per-env inertial records [70] of the model's bodies, each scaled as a whole."""
    rec = np.zeros(abi.NB * abi.INERTIAL_WORDS)
    for b in range(abi.NB):
        f = 1.0 if scale is None else scale[b]
        rec[10 * b] = f * model.mass[b]
        rec[10 * b + 1 : 10 * b + 4] = list(model.com[b])
        rec[10 * b + 4 : 10 * b + 10] = [f * x for x in model.inertia[b]]
    return rec


def test_substep_with_body_inertials_and_external_force(harness):
    rng = np.random.default_rng(6)
    model = default_model()
    for trial in range(50):
        s = random_state(rng, True)
        rec = records_of_model(model, rng.uniform(0.8, 1.2, 7))
        if trial % 2:  # shifted centres of mass and full inertia tensors, as fused links give
            for b in range(abi.NB):
                if b in (3, 6):  # wheels stay axisymmetric (the model says so: their rotation is skipped)
                    rec[10 * b + 2] += rng.uniform(-0.01, 0.01)
                    continue
                rec[10 * b + 1 : 10 * b + 4] += rng.uniform(-0.01, 0.01, 3)
                rec[10 * b + 7 : 10 * b + 10] += rng.uniform(-0.1, 0.1, 3) * rec[10 * b + 4 : 10 * b + 7].min()
        force = rng.uniform(-20, 20, 3)
        point = np.array([0.0, 0.0, -0.1])
        so, sh = run_both(harness, model, s, rng.uniform(-1, 1, 6), scale=rec, force=force, point=point)
        assert np.abs(so[7:10] - sh[7:10]).max() < 2e-4
        assert np.abs(so[10:13] - sh[10:13]).max() < 1e-3
        assert np.abs(so[0:7] - sh[0:7]).max() < 5e-7


def test_fuse_links_matches_oracle_and_link_semantics(harness):
    """randomize_inertias (pybullet_backend.py:571-601) per URDF link: the
    device's fuse_links() against the oracle's, on the URDF-derived model whose
    trunk and wheels are fused from several links; and against first
    principles (total mass, first moment, unit factors give the model back)."""
    from upkie_amd.model.model import Model

    model = Model().struct
    assert model.num_links > abi.NB and model.link_randomized[0] == 0  # root link: not in range(getNumJoints), :563
    cfg = abi.default_sim_config(16, seed=9)
    o = O.Oracle(model, cfg)
    rec = o.sample_body_inertials(0.3)  # [70, B]
    f = o.link_scale  # [MAX_LINKS, B]
    n = model.num_links
    assert np.all(f[0] == 1.0) and np.all(f[n:] == 1.0)
    assert np.all(np.abs(f[1:n] - 1.0) <= 0.3) and np.std(f[1:n]) > 0.1
    harness.harness_fuse_links.restype = C.c_int
    for e in range(16):
        got = np.zeros(70, dtype=np.float32)
        fe = np.ascontiguousarray(f[:, e], dtype=np.float32)
        assert harness.harness_fuse_links(C.byref(model), fe.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p)) == n
        np.testing.assert_allclose(got, rec[:, e], rtol=2e-5, atol=1e-9)
        for b in range(abi.NB):
            links = [l for l in range(n) if model.link_body[l] == b]
            m = sum(f[l, e] * model.link_mass[l] for l in links)
            first = sum(f[l, e] * model.link_mass[l] * np.array(model.link_com[l][:]) for l in links)
            assert rec[10 * b, e] == pytest.approx(m, rel=1e-12)
            np.testing.assert_allclose(rec[10 * b + 1 : 10 * b + 4, e], first / m, atol=1e-12)
    ones = np.ones(abi.MAX_LINKS, dtype=np.float32)
    got = np.zeros(70, dtype=np.float32)
    harness.harness_fuse_links(C.byref(model), ones.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(got, records_of_model(model), rtol=2e-5, atol=2e-9)
    # variation 0 -> the oracle gives the model back, too
    np.testing.assert_allclose(o.sample_body_inertials(0.0)[:, 3], records_of_model(model), rtol=1e-12, atol=1e-15)


def test_free_fall_semi_implicit_euler_in_fp32(harness):
    """BulletInterfaceTest.cpp:263-285 on the device arithmetic."""
    model = default_model()
    model.base_linear_damping = 0.0
    model.base_angular_damping = 0.0
    s = np.zeros(abi.STATE_WORDS, dtype=np.float32)
    s[abi.S_QUAT] = 1.0
    s[abi.S_POS + 2] = 1.0
    tau = np.zeros(6, dtype=np.float32)
    for _ in range(2):
        harness.harness_substep(C.byref(model), s.ctypes.data_as(C.c_void_p), tau.ctypes.data_as(C.c_void_p), C.c_float(1e-3), None, None, None)
    assert float(s[abi.S_POS + 2]) - 1.0 == pytest.approx(-3 * 9.81e-6, abs=2e-7)
    assert float(s[abi.S_LINVEL + 2]) == pytest.approx(-2 * 9.81e-3, abs=1e-7)


def test_substep_with_joint_limits(harness):
    """Hip/knee limit rows (general constraint path) vs the oracle's coupled
    solve: states with joints at or beyond their limits, in the air and on the
    floor."""
    rng = np.random.default_rng(8)
    model = default_model()
    model.enforce_joint_limits = 1
    hits = 0
    for trial in range(120):
        s = random_state(rng, on_floor=trial % 2 == 0)
        for j, lim in ((0, 1.26), (1, 2.51), (3, 1.26), (4, 2.51)):
            if rng.uniform() < 0.5:
                s[abi.S_Q + j] = rng.choice([-1, 1]) * (lim + rng.uniform(0.0, 0.01))
                s[abi.S_QD + j] = rng.uniform(-3, 3)
                hits += 1
        so, sh = run_both(harness, model, s, rng.uniform(-3.0, 3.0, 6))
        assert np.abs(so[0:7] - sh[0:7]).max() < 1e-6
        assert np.abs(so[7:10] - sh[7:10]).max() < 5e-4
        assert np.abs(so[10:13] - sh[10:13]).max() < 3e-3
        assert np.abs(so[19:25] - sh[19:25]).max() < 5e-2
    assert hits > 100


def test_substep_with_external_forces_on_any_link(harness):
    """World- and link-frame forces on trunk, thigh, calf and wheel links
    (pybullet_backend.py:603-658), several at a time: device arithmetic vs the
    oracle's Jacobian-transpose formulation."""
    rng = np.random.default_rng(8)
    model = default_model()
    for trial in range(120):
        s = random_state(rng, trial % 2 == 0)
        count = 1 + trial % 4
        bodies = [int(b) for b in rng.integers(0, 7, count)]
        local = [bool(v) for v in rng.integers(0, 2, count)]
        points = rng.uniform(-0.05, 0.05, (count, 3))
        force = rng.uniform(-15, 15, (count, 3))
        so, sh = run_both(harness, model, s, rng.uniform(-1, 1, 6), force=force, slots=make_slots(bodies, points, local))
        assert np.abs(so[0:7] - sh[0:7]).max() < 5e-7
        assert np.abs(so[7:10] - sh[7:10]).max() < 3e-4
        assert np.abs(so[10:13] - sh[10:13]).max() < 2e-3
        assert np.abs(so[13:19] - sh[13:19]).max() < 1e-6
        # a force really does something to the joints of the link it acts on
    s = random_state(rng, False)
    free, _ = run_both(harness, model, s, np.zeros(6))
    pushed, pushed32 = run_both(harness, model, s, np.zeros(6), force=np.array([[0.0, 0.0, 5.0]]), slots=make_slots([5], [[0.0, 0.0, -0.1]], [False]))
    assert np.abs(pushed[19 + 3 : 19 + 5] - free[19 + 3 : 19 + 5]).max() > 1e-3  # right hip / knee accelerate
    assert np.abs(pushed[19:25] - pushed32[19:25]).max() < 2e-2


# ---------------------------------------------------------------------------
# Eight lanes per env (octet.hpp): the lanes of one env run as eight host
# threads in lockstep, lane exchanges go through a slot array.
OCT_NO_CONTACT, OCT_CONTACT = 0, 1


def run_octet(harness, model, s64, tau, h=1e-3, records=None, wrench=None, substeps=1, limits_in_registers=False):
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    s32 = s64.astype(np.float32)
    t32 = np.ascontiguousarray(tau, dtype=np.float32)
    r32 = None if records is None else np.ascontiguousarray(records, dtype=np.float32)
    w32 = None if wrench is None else np.ascontiguousarray(wrench, dtype=np.float32)
    status = np.zeros(64, dtype=np.int32)
    harness.harness_substep_octet.restype = C.c_int
    ok = harness.harness_substep_octet(C.byref(model), p(s32), p(t32), C.c_float(h), p(r32), p(w32), C.c_int(substeps), p(status), C.c_int(1 if limits_in_registers else 0))
    assert ok == 1, "the eight lanes disagree on the base state or on the status"
    return s32.astype(np.float64), status[:substeps]


def one_lane(harness, model, s64, tau, h=1e-3, records=None, force=None, slots=None):
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    s32 = s64.astype(np.float32)
    t32 = np.ascontiguousarray(tau, dtype=np.float32)
    r32 = None if records is None else np.ascontiguousarray(records, dtype=np.float32)
    f32 = None if force is None else np.ascontiguousarray(force, dtype=np.float32)
    harness.harness_substep.restype = C.c_int
    rc = harness.harness_substep(C.byref(model), p(s32), p(t32), C.c_float(h), p(r32), p(f32), C.byref(slots) if slots is not None else None)
    assert rc >= 0
    return s32.astype(np.float64), rc


def settle_on_floor(model, s):
    """Lower the base until both tires are within the contact range (zero-time oracle probe of the tire heights is not
    exposed: use the default standing height, legs near straight)."""
    s[abi.S_Q : abi.S_Q + 6] *= 0.2
    s[abi.S_QUAT : abi.S_QUAT + 4] = [np.cos(0.05), 0.0, np.sin(0.05), 0.0]
    s[abi.S_POS + 2] = 0.6
    return s


@pytest.mark.parametrize("on_floor", [False, True])
def test_octet_substep_matches_one_lane_and_oracle(harness, on_floor):
    """Same substep, reassociated over eight lanes: against the one-lane device
    arithmetic (fp32 rounding apart) and against the fp64 oracle with the
    tolerances of test_substep_matches_oracle."""
    rng = np.random.default_rng(11)
    model = default_model()
    worst_1 = np.zeros(25)
    worst_o = np.zeros(25)
    mine = 0
    for _ in range(150):
        s = random_state(rng, on_floor)
        tau = rng.uniform(-1.5, 1.5, 6)
        s8, status = run_octet(harness, model, s, tau)
        s1, contact = one_lane(harness, model, s, tau)
        so, _ = run_both(harness, model, s, tau)
        mine += 1
        assert (status[0] == OCT_CONTACT) == bool(contact)
        worst_1 = np.maximum(worst_1, np.abs(s8[:25] - s1[:25]))
        worst_o = np.maximum(worst_o, np.abs(s8[:25] - so[:25]))
    assert mine == 150
    assert worst_o[0:3].max() < 5e-7 and worst_o[3:7].max() < 5e-7
    assert worst_o[7:10].max() < 2e-4 and worst_o[10:13].max() < 1e-3
    assert worst_o[13:19].max() < 1e-6 and worst_o[19:25].max() < 2e-2
    # the two device mappings agree more closely than either does with fp64
    assert worst_1[0:7].max() < 5e-7 and worst_1[7:10].max() < 2e-4 and worst_1[10:13].max() < 1e-3
    assert worst_1[13:19].max() < 1e-6 and worst_1[19:25].max() < 2e-2


def test_octet_standing_robot_takes_the_fast_path(harness):
    """A robot standing on both tires, legs held by the servos' PD law, wheels
    driven by a balancing feedback: once it has landed every substep is the
    eight-lane path's (contact, admissible solution), and 100 substeps of the
    closed loop track the one-lane arithmetic."""
    rng = np.random.default_rng(12)
    model = default_model()

    def torques(s):
        tau = np.zeros(6)
        for j in (0, 1, 3, 4):
            tau[j] = np.clip(20.0 * (0.0 - s[abi.S_Q + j]) - 1.0 * s[abi.S_QD + j], -16.0, 16.0)
        pitch = 2.0 * s[abi.S_QUAT + 2]
        v = (10.0 * pitch) / 0.05  # wheel velocity target of the README agent
        tau[2] = np.clip(1.0 * (v - s[abi.S_QD + 2]), -1.7, 1.7)
        tau[5] = np.clip(1.0 * (-v - s[abi.S_QD + 5]), -1.7, 1.7)
        return tau

    fast = 0
    for _ in range(4):
        s = np.zeros(abi.STATE_WORDS)
        pitch = rng.uniform(-0.03, 0.03)
        s[abi.S_QUAT] = np.cos(pitch / 2)
        s[abi.S_QUAT + 2] = np.sin(pitch / 2)
        s[abi.S_POS + 2] = 0.6
        for _ in range(300):  # land and settle (one-lane arithmetic)
            s, _ = one_lane(harness, model, s, torques(s))
        s8, s1 = s.copy(), s.copy()
        for _ in range(100):
            s8, status = run_octet(harness, model, s8, torques(s8))
            assert status[0] == OCT_CONTACT
            fast += 1
            s1, contact = one_lane(harness, model, s1, torques(s1))
            assert contact
        assert np.abs(s8[0:7] - s1[0:7]).max() < 5e-6
        assert np.abs(s8[7:13] - s1[7:13]).max() < 5e-3
        assert np.abs(s8[13:19] - s1[13:19]).max() < 2e-4
    assert fast >= 390, fast


def test_octet_per_env_inertials_and_trunk_wrench(harness):
    rng = np.random.default_rng(13)
    model = default_model()
    checked = 0
    for trial in range(40):
        s = random_state(rng, trial % 2 == 0)
        rec = records_of_model(model, rng.uniform(0.8, 1.2, 7))
        for b in range(abi.NB):
            if b in (3, 6):
                rec[10 * b + 2] += rng.uniform(-0.01, 0.01)
                continue
            rec[10 * b + 1 : 10 * b + 4] += rng.uniform(-0.01, 0.01, 3)
            rec[10 * b + 7 : 10 * b + 10] += rng.uniform(-0.1, 0.1, 3) * rec[10 * b + 4 : 10 * b + 7].min()
        force = rng.uniform(-20, 20, 3)  # base frame (a "local" force on the trunk)
        point = np.array([0.02, -0.01, -0.1])
        wrench = np.concatenate([force, np.cross(point, force)])
        tau = rng.uniform(-1, 1, 6)
        s8, status = run_octet(harness, model, s, tau, records=rec, wrench=wrench)
        s1, _ = one_lane(harness, model, s, tau, records=rec, force=force.reshape(1, 3), slots=make_slots([0], [point], [True]))
        checked += 1
        assert np.abs(s8[0:7] - s1[0:7]).max() < 5e-7
        assert np.abs(s8[7:10] - s1[7:10]).max() < 2e-4 and np.abs(s8[10:13] - s1[10:13]).max() < 1e-3
        assert np.abs(s8[19:25] - s1[19:25]).max() < 2e-2
    assert checked == 40


def test_octet_rare_cases(harness):
    """The rare cases stay inside the eight-lane substep, against the one-lane
    arithmetic: a hip / knee at its stop (the general solver over scratch memory
    on the system gathered from the lanes, as limit_path_scratch builds it),
    one tire in the air while the other touches (identity rows), contact
    impulses outside the friction cone (projected Gauss-Seidel sweeps,
    contact_pgs6 on the gathered system)."""
    rng = np.random.default_rng(14)
    model = default_model()
    model.enforce_joint_limits = 1
    hits = 0
    for trial in range(60):
        s = random_state(rng, on_floor=trial % 2 == 0)
        for j, lim in ((0, 1.26), (1, 2.51), (3, 1.26), (4, 2.51)):
            if rng.uniform() < 0.5:
                s[abi.S_Q + j] = rng.choice([-1, 1]) * (lim + rng.uniform(0.0, 0.01))
                s[abi.S_QD + j] = rng.uniform(-3, 3)
                hits += 1
        tau = rng.uniform(-3.0, 3.0, 6)
        s1, contact = one_lane(harness, model, s, tau)
        for in_registers in (False, True):  # the scratch-memory solve (Pendulum / Gyropod kernels) and the register one (Servos kernels)
            s8, status = run_octet(harness, model, s, tau, limits_in_registers=in_registers)
            assert (status[0] == OCT_CONTACT) == bool(contact)
            assert np.abs(s8[0:7] - s1[0:7]).max() < 1e-6
            assert np.abs(s8[7:10] - s1[7:10]).max() < 5e-4 and np.abs(s8[10:13] - s1[10:13]).max() < 3e-3
            assert np.abs(s8[13:19] - s1[13:19]).max() < 1e-5 and np.abs(s8[19:25] - s1[19:25]).max() < 5e-2
    assert hits > 80
    model.enforce_joint_limits = 0
    one_tire = 0
    for roll in (0.2, 0.3, -0.3):
        s = np.zeros(abi.STATE_WORDS)
        s[abi.S_QUAT] = np.cos(roll / 2)
        s[abi.S_QUAT + 1] = np.sin(roll / 2)
        s[abi.S_POS + 2] = 0.66
        for _ in range(200):  # falls onto one tire
            nxt, contact = one_lane(harness, model, s, np.zeros(6))
            if contact:
                break
            s = nxt
        assert contact
        s1, _ = one_lane(harness, model, s, np.zeros(6))
        s8, status = run_octet(harness, model, s, np.zeros(6))
        assert status[0] == OCT_CONTACT
        one_tire += 1
        assert np.abs(s8[0:7] - s1[0:7]).max() < 5e-7 and np.abs(s8[7:13] - s1[7:13]).max() < 1e-3
        assert np.abs(s8[19:25] - s1[19:25]).max() < 2e-2
    assert one_tire == 3
    # saturated wheel torques on a standing robot: the tires slip (friction cone active)
    s = np.zeros(abi.STATE_WORDS)
    s[abi.S_QUAT] = 1.0
    s[abi.S_POS + 2] = 0.6

    def hold(s, wheel):
        tau = np.zeros(6)
        for j in (0, 1, 3, 4):
            tau[j] = np.clip(20.0 * (0.0 - s[abi.S_Q + j]) - 1.0 * s[abi.S_QD + j], -16.0, 16.0)
        tau[2], tau[5] = wheel, -wheel
        return tau

    for _ in range(300):
        s, _ = one_lane(harness, model, s, hold(s, 0.0))
    model.friction_mu = 0.2  # 1.7 N.m / 0.05 m = 34 N at the rim against 0.2 x 26 N of grip
    worst = np.zeros(25)
    s1, s8 = s.copy(), s.copy()
    for _ in range(20):
        s1, contact = one_lane(harness, model, s1, hold(s1, 1.7))
        s8, status = run_octet(harness, model, s8, hold(s8, 1.7))
        assert contact and status[0] == OCT_CONTACT
        worst = np.maximum(worst, np.abs(s8[:25] - s1[:25]))
    assert abs(s1[abi.S_QD + 2]) > 20.0  # the wheels do spin up: the cone was active
    assert worst[0:7].max() < 2e-6 and worst[7:13].max() < 5e-3 and worst[13:19].max() < 1e-4

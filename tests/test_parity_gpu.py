"""HIP path (through the C-ABI) vs the fp64 oracle on the same seeded inputs.

Tolerances follow SURVEY.md Appendix A.9: fp32 kernel vs fp64 oracle, single
step state relative 1e-5, 200-step closed loop |dtheta| <= 1e-3 rad,
|dp| <= 1e-3 m; torques 1e-5 * max(1, |tau|) per substep (looser after
accumulation over a step).
"""

import numpy as np
import pytest
import torch

from upkie_amd import abi

from .helpers import make_pair, randomized_config, state_errors

pytestmark = pytest.mark.gpu


def assert_mostly_close(actual, desired, atol, fraction=0.99, hard_atol=None):
    """Per-env comparison for regimes where a few envs sit on a physical
    discontinuity (tire slip onset, torque saturation): `fraction` of the envs
    must agree within `atol`, all of them within `hard_atol` if given."""
    actual = np.asarray(actual, dtype=np.float64).reshape(len(actual), -1)
    desired = np.asarray(desired, dtype=np.float64).reshape(len(desired), -1)
    err = np.max(np.abs(actual - desired), axis=1)
    ok = float(np.mean(err <= atol))
    assert ok >= fraction, f"only {ok:.4f} of envs within {atol} (worst {err.max():.3e})"
    if hard_atol is not None:
        assert err.max() <= hard_atol, f"worst env error {err.max():.3e} > {hard_atol}"


def test_library_sees_gpu():
    from upkie_amd import lib

    assert lib.load().upkie_hip_device_count() >= 1


def test_reset_matches_oracle():
    oracle, sim = make_pair(256, seed=3)
    obs_o = oracle.reset()
    obs_h = sim.reset().cpu().numpy()
    err = state_errors(oracle.state, sim.state_numpy())
    # fp32 floor: the state's z is quantised at 6e-8 m and the contact row
    # divides the gap by h = 1 ms, so velocities can differ by ~6e-5 m/s
    assert err["pos"] < 2e-6 and err["quat"] < 2e-6, err
    assert err["linvel"] < 2e-4 and err["angvel"] < 1e-3, err
    # wheel speed = rim speed / 0.05 m: 20x the linear-velocity floor
    assert err["q"] < 2e-5 and err["qd"] < 2e-2, err
    assert err["episode"] == 0 and err["done"] == 0 and err["contact"] == 0, err
    np.testing.assert_allclose(obs_h[:, :3], obs_o[:, :3], atol=2e-5)
    np.testing.assert_allclose(obs_h[:, 3:], obs_o[:, 3:], atol=2e-3)
    # randomisation bounds of the config hold on device too
    pitch = obs_h[:, 1]
    assert np.all(np.abs(pitch) <= 0.1 + 1e-3) and np.std(pitch) > 0.03


def test_single_pendulum_step_matches_oracle():
    """Tight single-step parity in the traction regime: commanded ground
    velocity within 0.02 m/s of the current one, so the wheel torque
    kd * (omega* - omega) stays below its 1.7 N.m cap."""
    oracle, sim = make_pair(512, seed=1)
    obs6 = oracle.reset()
    sim.reset()
    rng = np.random.default_rng(0)
    act = (obs6[:, 3] + rng.uniform(-0.02, 0.02, 512)).astype(np.float32)
    obs_o, rew_o, term_o, trunc_o = oracle.step_pendulum(act.astype(np.float64))
    obs_h, rew_h, term_h, trunc_h = sim.step_pendulum(torch.from_numpy(act))
    err = state_errors(oracle.state, sim.state_numpy())
    assert err["pos"] < 5e-6 and err["quat"] < 5e-6, err
    assert err["linvel"] < 5e-4 and err["angvel"] < 2e-3, err
    assert err["q"] < 5e-5 and err["qd"] < 3e-2, err
    assert err["torque"] < 3e-2 and err["legref"] < 1e-6, err
    np.testing.assert_allclose(obs_h.cpu().numpy()[:, :2], obs_o[:, :2], atol=2e-5)
    np.testing.assert_allclose(obs_h.cpu().numpy()[:, 2:], obs_o[:, 2:], atol=2e-3)
    assert np.array_equal(term_h.cpu().numpy(), term_o)
    assert float(rew_h.abs().max()) == 0.0 and int(trunc_h.max()) == 0


def test_single_pendulum_step_saturated_actions():
    """Actions far from the current velocity saturate the wheel torque and
    break traction; the 1 kHz explicit velocity loop on the bare wheel
    inertia (kd dt / I_wheel = 3.6 > 2) then amplifies rounding differences
    by ~2.6x per substep, so fp32 and fp64 can only agree loosely there."""
    oracle, sim = make_pair(512, seed=1)
    oracle.reset()
    sim.reset()
    rng = np.random.default_rng(0)
    act = rng.uniform(-1.0, 1.0, 512).astype(np.float32)
    obs_o, _, term_o, _ = oracle.step_pendulum(act.astype(np.float64))
    obs_h, _, term_h, _ = sim.step_pendulum(torch.from_numpy(act))
    err = state_errors(oracle.state, sim.state_numpy())
    assert err["pos"] < 1e-4 and err["quat"] < 1e-4, err
    assert err["linvel"] < 2e-2 and err["angvel"] < 5e-2, err
    obs_h = obs_h.cpu().numpy()
    assert np.max(np.abs(obs_h[:, :2] - obs_o[:, :2])) < 1e-3
    assert np.array_equal(term_h.cpu().numpy(), term_o)


def test_closed_loop_200_steps_matches_oracle():
    oracle, sim = make_pair(128, seed=7)
    obs_o = oracle.reset()[:, [1, 0, 4, 3]]
    sim.reset()
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    for _ in range(200):
        obs_o, _, term_o, _ = oracle.step_pendulum_agent(obs_o)
        obs_h, _, term_h, _ = sim.step_pendulum_agent()
    obs_h = obs_h.cpu().numpy()
    # measured in round 3 (tools/parity_margins.py, profiles/r03_parity_margins.txt): 9e-7 rad, 6e-6 m, 1.1e-5 rad/s, 8e-6 m/s
    assert np.max(np.abs(obs_h[:, 0] - obs_o[:, 0])) <= 1e-5  # pitch, rad
    assert np.max(np.abs(obs_h[:, 1] - obs_o[:, 1])) <= 3e-5  # position, m
    assert np.max(np.abs(obs_h[:, 2] - obs_o[:, 2])) <= 1e-4  # pitch rate
    assert np.max(np.abs(obs_h[:, 3] - obs_o[:, 3])) <= 1e-4  # velocity
    assert np.array_equal(term_h.cpu().numpy(), term_o)


def test_gyropod_step_matches_oracle():
    oracle, sim = make_pair(256, seed=5)
    oracle.reset()
    sim.reset()
    rng = np.random.default_rng(1)
    for _ in range(5):
        act = rng.uniform(-1.5, 1.5, (256, 2)).astype(np.float32)
        obs_o, _, term_o, _ = oracle.step_gyropod(act.astype(np.float64))
        obs_h, _, term_h, _ = sim.step_gyropod(torch.from_numpy(act))
    # random +-1.5 m/s commands saturate the wheel torque: see the saturated
    # actions test for why a few envs can only agree loosely
    # (measured in round 3: median 1.2e-5, p97 3e-4, worst env 9.9e-4 -- profiles/r03_parity_margins.txt)
    assert_mostly_close(obs_h.cpu().numpy(), obs_o, atol=1e-3, fraction=0.99, hard_atol=4e-3)
    err = state_errors(oracle.state, sim.state_numpy())
    assert err["yaw"] < 1e-6 and err["pos"] < 1e-3, err


def test_servos_step_matches_oracle():
    """Torque-balancing style Servos actions (examples/pybullet/
    torque_balancing.py:15-37): hips/knees hold position, wheels get a
    feedforward torque and a small velocity target."""
    oracle, sim = make_pair(256, seed=9)
    oracle.reset()
    sim.reset()
    rng = np.random.default_rng(2)
    act = np.zeros((256, 6, 6), dtype=np.float32)
    act[:, :, 0] = rng.uniform(-0.05, 0.05, (256, 6))  # position
    act[:, [2, 5], 0] = np.nan  # wheels: no position feedback
    act[:, [2, 5], 1] = rng.uniform(-0.5, 0.5, (256, 2))  # wheel velocity
    act[:, [2, 5], 2] = rng.uniform(-0.2, 0.2, (256, 2))  # feedforward torque
    act[:, :, 3] = rng.uniform(0.5, 1.5, (256, 6))  # kp_scale
    act[:, :, 4] = rng.uniform(0.5, 1.5, (256, 6))  # kd_scale
    act[:, :, 5] = 16.0  # clamped to the wheel effort 1.7 on wheels
    for _ in range(3):
        obs_o, _, _, _ = oracle.step_servos(act.astype(np.float64))
        obs_h, _, term_h, _ = sim.step_servos(torch.from_numpy(act))
    obs_h = obs_h.cpu().numpy()
    np.testing.assert_allclose(obs_h[:, :, 0], obs_o[:, :, 0], atol=2e-5)  # position
    # measured in round 3: velocity median 2e-5, p99 2.1e-3, worst 9e-3 rad/s; torque median 4e-6, p99 4.3e-3, worst 2.7e-2 N m
    assert_mostly_close(obs_h[:, :, 1], obs_o[:, :, 1], atol=5e-3, fraction=0.99, hard_atol=3e-2)  # velocity
    assert_mostly_close(obs_h[:, :, 2], obs_o[:, :, 2], atol=1e-2, fraction=0.99, hard_atol=6e-2)  # torque
    assert np.all(obs_h[:, :, 3] == 42.0) and np.all(obs_h[:, :, 4] == 18.0)
    assert int(term_h.max()) == 0


def test_servos_clamping_matches_oracle():
    """Every one of the 36 action scalars is clamped to the servo limits
    (upkie_servos.py:316-344) before the torque law. With one 1 ms substep
    per step the reported torque only depends on the pre-step state, so wild
    out-of-range actions can be compared tightly."""
    cfg = randomized_config(256, seed=6)
    cfg.dt = 1e-3
    cfg.nb_substeps = 1
    oracle, sim = make_pair(256, cfg=cfg)
    oracle.reset()
    sim.reset()
    rng = np.random.default_rng(3)
    act = np.zeros((256, 6, 6), dtype=np.float32)
    act[:, :, 0] = rng.uniform(-4.0, 4.0, (256, 6))  # beyond +-1.26 / +-2.51
    act[:, [2, 5], 0] = np.nan
    act[:, :, 1] = rng.uniform(-200.0, 200.0, (256, 6))  # beyond 28.8 / 111
    act[:, :, 2] = rng.uniform(-30.0, 30.0, (256, 6))  # beyond 16 / 1.7
    act[:, :, 3] = rng.uniform(-1.0, 8.0, (256, 6))  # beyond [0, 5]
    act[:, :, 4] = rng.uniform(-1.0, 8.0, (256, 6))
    act[:, :, 5] = rng.uniform(-2.0, 30.0, (256, 6))  # beyond [0, effort]
    obs_o, _, _, _ = oracle.step_servos(act.astype(np.float64))
    obs_h, _, _, _ = sim.step_servos(torch.from_numpy(act))
    np.testing.assert_allclose(obs_h.cpu().numpy()[:, :, 2], obs_o[:, :, 2], atol=2e-3)
    assert np.abs(obs_h.cpu().numpy()[:, [2, 5], 2]).max() <= 1.7 + 1e-6
    assert np.abs(obs_h.cpu().numpy()[:, :, 2]).max() <= 16.0 + 1e-6


def test_spine_observation_matches_oracle():
    oracle, sim = make_pair(128, seed=11)
    oracle.reset()
    sim.reset()
    act = np.full(128, 0.3, dtype=np.float32)
    for _ in range(3):
        oracle.step_pendulum(act.astype(np.float64))
        sim.step_pendulum(torch.from_numpy(act))
        obs_o = oracle.observe(update_imu=True)
        obs_h = sim.observe(update_imu=True)
    tol = {
        "pitch": 1e-4,
        "angular_velocity": 2e-3,
        "linear_velocity": 1e-3,
        "rotation_base_to_world": 1e-4,
        "imu_orientation": 1e-4,
        "imu_angular_velocity": 2e-3,
        "imu_linear_acceleration": 0.25,  # finite difference over dt: 1e-3 / 5e-3
        "imu_raw_linear_acceleration": 0.25,
        "wheel_odometry": 5e-3,
    }
    for key, atol in tol.items():
        a = obs_o[key].reshape(128, -1)
        b = obs_h[key].cpu().numpy().reshape(128, -1)
        assert np.max(np.abs(a - b)) <= atol, key
    assert np.array_equal(obs_h["floor_contact"].cpu().numpy(), obs_o["floor_contact"])


def test_autoreset_next_step():
    cfg = randomized_config(64, seed=2, autoreset=True)
    cfg.fall_pitch = 0.12  # just above the +-0.1 rad reset range: frequent falls
    oracle, sim = make_pair(64, cfg=cfg)
    oracle.reset()
    sim.reset()
    act = np.zeros(64, dtype=np.float32)  # wheels hold still: everyone tips over
    resets = 0
    in_sync = np.ones(64, dtype=bool)  # envs whose episodes ended on the same steps so far
    for _ in range(300):
        obs_o, _, term_o, _ = oracle.step_pendulum(act.astype(np.float64))
        obs_h, rew_h, term_h, trunc_h = sim.step_pendulum(torch.from_numpy(act))
        in_sync &= term_h.cpu().numpy() == term_o
        resets += int(term_o.sum())
    assert resets > 64  # every env fell and was reset at least once
    # a fall threshold crossed within rounding can shift one env by a step
    assert in_sync.mean() >= 0.96  # (measured in round 3: every env in step with the oracle)
    episodes_o = oracle.state[abi.S_EPISODE]
    episodes_h = sim.state_numpy()[abi.S_EPISODE]
    assert np.array_equal(episodes_o[in_sync], episodes_h[in_sync])
    assert episodes_h.min() >= 2
    assert_mostly_close(obs_h.cpu().numpy()[in_sync], obs_o[in_sync], atol=5e-3, fraction=0.9)


def test_randomization_buffers():
    oracle, sim = make_pair(128, seed=4)
    rec_h = sim.randomize_inertias(0.2).cpu().numpy()
    rec_o = oracle.sample_body_inertials(0.2)
    np.testing.assert_allclose(rec_h, rec_o, rtol=2e-5, atol=1e-9)
    link_scale = sim.link_scale.cpu().numpy()
    np.testing.assert_allclose(link_scale, oracle.link_scale, atol=1e-6)
    assert link_scale.min() >= 0.8 - 1e-6 and link_scale.max() <= 1.2 + 1e-6
    oracle.body_inertials = rec_o
    force = np.zeros((3, 128))
    force[0] = np.linspace(-10, 10, 128)
    force[1] = 3.0
    oracle.ext_force = force
    oracle.ext_point = np.array([0.0, 0.0, -0.1])  # "torso" frame
    sim.set_external_force(torch.from_numpy(force).float(), point=(0.0, 0.0, -0.1))
    oracle.reset()
    sim.reset()
    act = np.zeros(128, dtype=np.float32)
    for _ in range(10):
        obs_o, _, _, _ = oracle.step_pendulum(act.astype(np.float64))
        obs_h, _, _, _ = sim.step_pendulum(torch.from_numpy(act))
    np.testing.assert_allclose(obs_h.cpu().numpy(), obs_o, atol=3e-3)
    err = state_errors(oracle.state, sim.state_numpy())
    assert err["pos"] < 2e-4, err


@pytest.mark.parametrize("N", [16, 17, 32, 33, 48, 49, 50, 64])  # (one .. four row tiles; 17 / 33 / 49: a tile that is mostly padding)
def test_mpc_step_matches_oracle(N):
    """MFMA ADMM kernel vs the fp64 oracle ADMM (same recurrences) and vs the
    exact QP solution. Tolerance on plan.first_input: 1e-4 * a_max = 1e-3 m/s2 (SURVEY.md A.9 asks 2e-3 * a_max;
    ProxQP itself only guarantees eps_abs = 1e-3). Measured, round 6 (fp16 matrix path, the constant part of the product
    out of the loop): 2e-6 m/s2 on the first two steps at every horizon, 3e-4 .. 5e-4 on the saturating ones at
    N = 49 / 50; the fp32 kernels of rounds 2-6 were 2e-5 at N = 16, 5e-3 .. 1e-2 at N = 50, 5e-2 at N = 49."""
    import ctypes as C

    from oracle import oracle as O
    from upkie_amd.mpc import BatchedMpc

    B = 500  # not a multiple of 16: exercises the tail wave
    cfg = abi.default_mpc_config(B, N)
    mpc = BatchedMpc(cfg)
    rng = np.random.default_rng(0)
    ws = np.zeros((2 * N, B))
    v_o = np.zeros(B)
    first_o = np.zeros(B)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    P, Kx, kv = np.zeros((N, N)), np.zeros((N, 4)), np.zeros(N)
    O.lib().oracle_mpc_build(C.byref(cfg), p(P), p(Kx), p(kv))
    for step in range(4):
        scale = 1.0 if step < 2 else 5.0  # later steps saturate the bounds
        x0 = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.15, 0.15, B) * scale, rng.uniform(-0.5, 0.5, B) * scale, rng.uniform(-0.5, 0.5, B) * scale], axis=1)
        vt = rng.uniform(-0.5, 0.5, B)
        contact = (rng.uniform(size=B) > 0.1).astype(np.uint8)
        O.lib().oracle_mpc_step(C.byref(cfg), p(ws), p(np.ascontiguousarray(x0)), p(vt), p(contact), C.c_double(0.005), p(v_o), p(first_o))
        v_h, first_h = mpc.step(torch.from_numpy(x0).float(), torch.from_numpy(vt).float(), torch.from_numpy(contact), dt=0.005)
        assert np.max(np.abs(first_h.cpu().numpy() - first_o)) <= 1e-4 * cfg.max_ground_accel
        assert np.max(np.abs(v_h.cpu().numpy() - v_o)) <= 5e-6
        # ... and the EXACT solution of the QP (projected Newton on the fp64 problem), every tenth env: what the fixed
        # number of iterations leaves open from an unrelated warm start stays inside SURVEY A.9's 2e-3 a_max
        # (profiles/r04_mpc_iterations.txt: 4e-4 m/s2 at N = 16, 6e-3 at N = 50)
        q = x0 @ Kx.T + np.outer(vt, kv)
        for e in range(0, B, 10):
            u = np.zeros(N)
            assert O.lib().oracle_mpc_solve_exact(N, p(P), p(np.ascontiguousarray(q[e])), C.c_double(cfg.max_ground_accel), p(u)) >= 0
            assert abs(float(first_h[e]) - u[0]) <= 2e-3 * cfg.max_ground_accel, (N, step, e)
    # the whole plan and its duals, not only the first input
    ws_h = mpc.workspace.cpu().numpy()
    np.testing.assert_allclose(ws_h[:N], ws[:N], atol=2e-3)  # measured: 5e-4 at most (N = 49, the saturating steps; 3e-6 before them)
    np.testing.assert_allclose(ws_h[N:], ws[N:], atol=4e-3)  # duals, measured: 1e-3 at most (N = 49 / 64)
    mask = torch.zeros(B, dtype=torch.uint8)
    mask[::2] = 1
    mpc.reset(mask)
    v = mpc.commanded_velocity.cpu().numpy()
    assert np.all(v[::2] == 0.0) and np.any(v[1::2] != 0.0)
    assert float(mpc.workspace[:, ::2].abs().max()) == 0.0


@pytest.mark.parametrize("N, rho, relaxation, leg_length", [(50, 1e-4, 1.5, 0.58), (50, 1e-2, 1.0, 0.58), (16, 1e-5, 1.6, 0.4), (32, 1e-1, 1.2, 0.7)])
def test_mpc_other_solver_settings_and_models(N, rho, relaxation, leg_length):
    """The fp16 operands are scaled by a power of two chosen from the matrix itself (csrc/mpc.hpp: the largest entry of
    Minv / scale lies in [8, 16)): other ADMM penalties, relaxations and pendulum lengths -- Minv's entries range over
    six decades here -- stay as close to the fp64 ADMM of the same settings as the defaults do."""
    import ctypes as C

    from oracle import oracle as O
    from upkie_amd.mpc import BatchedMpc

    B = 256
    cfg = abi.default_mpc_config(B, N)
    cfg.admm_rho, cfg.admm_relaxation, cfg.leg_length = rho, relaxation, leg_length
    mpc = BatchedMpc(cfg)
    rng = np.random.default_rng(1)
    ws, v_o, first_o = np.zeros((2 * N, B)), np.zeros(B), np.zeros(B)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for step in range(3):
        scale = 1.0 if step < 2 else 5.0
        x0 = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.15, 0.15, B) * scale, rng.uniform(-0.5, 0.5, B) * scale, rng.uniform(-0.5, 0.5, B) * scale], axis=1)
        vt = rng.uniform(-0.5, 0.5, B)
        contact = np.ones(B, dtype=np.uint8)
        O.lib().oracle_mpc_step(C.byref(cfg), p(ws), p(np.ascontiguousarray(x0)), p(vt), p(contact), C.c_double(0.005), p(v_o), p(first_o))
        v_h, first_h = mpc.step(torch.from_numpy(x0).float(), torch.from_numpy(vt).float(), torch.from_numpy(contact), dt=0.005)
        err = float(np.max(np.abs(first_h.cpu().numpy() - first_o)))
        print(f"N={N} rho={rho} alpha={relaxation} l={leg_length} step {step}: |first input - checker| {err:.2e}")
        assert err <= 2e-4 * cfg.max_ground_accel, (step, err)
    assert torch.isfinite(mpc.workspace).all()


@pytest.mark.parametrize("four_tiles", ["0", "1"])
def test_mpc_fp32_kernels_of_the_long_horizons(four_tiles):
    """`UPKIE_MPC_FP32=1` (read once per process): the fp32 MFMA kernels for horizons > 16, kept as the A/B partners of
    the fp16 matrix path -- three row tiles + vector tail at N = 49 / 50, or round 5's four tiles."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, UPKIE_MPC_FP32="1", UPKIE_MPC_FOUR_TILES=four_tiles)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpc_fp32_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]


def test_packed_records_match_unpacked_step():
    """The 32-byte record path used by the multi-GPU gather computes exactly
    what the four-array path computes (same kernel, different stores)."""
    from upkie_amd.sim import BatchedSim

    cfg = randomized_config(300, seed=8, autoreset=True)
    cfg.fall_pitch = 0.12
    cfg.agent_gains[:] = [0.0, 0.0, 0.0, 0.0]  # passive agent: everyone tips over
    a, b = BatchedSim(cfg), BatchedSim(cfg)
    a.reset()
    obs6 = b.reset()
    a.obs4.copy_(a.obs6[:, [1, 0, 4, 3]])
    rec = torch.zeros((300, 8), dtype=torch.float32, device="cuda:0")
    rec[:, :4] = obs6[:, [1, 0, 4, 3]]
    falls = 0
    for _ in range(120):
        obs, rew, term, trunc = a.step_pendulum_agent()
        b.step_pendulum_packed(rec)
        assert torch.equal(rec[:, :4], obs)
        assert torch.equal(rec[:, 5], term.float()) and float(rec[:, [4, 6, 7]].abs().max()) == 0.0
        falls += int(term.sum())
    assert falls > 0
    assert torch.equal(a.state, b.state)


def test_sharded_results_do_not_depend_on_the_shard():
    """Random streams are keyed by the global env index: a shard hosting envs
    [100, 164) reproduces rows 100..163 of the full batch bit for bit."""
    from upkie_amd.sim import BatchedSim

    full_cfg = randomized_config(256, seed=12, autoreset=True)
    full = BatchedSim(full_cfg)
    part_cfg = randomized_config(64, seed=12, autoreset=True)
    part_cfg.env_id_offset = 100
    part = BatchedSim(part_cfg)
    full.reset()
    part.reset()
    assert torch.equal(full.state[:, 100:164], part.state)
    act_full = torch.linspace(-0.3, 0.3, 256, device="cuda:0")
    for _ in range(10):
        o_full, *_ = full.step_pendulum(act_full)
        o_part, *_ = part.step_pendulum(act_full[100:164])
    assert torch.equal(o_full[100:164], o_part)


def test_torque_noise_matches_oracle():
    """Same Philox streams on both sides: the noisy torques agree draw by
    draw (fp32 Box-Muller vs fp64), and the statistics match the reference's
    checks (test_pybullet_backend_mock.py:404-417)."""
    cfg = abi.default_sim_config(128, frequency=1000.0, nb_substeps=1, seed=4)
    cfg.init_pos[2] = 3.0
    for j in range(6):
        cfg.torque_control_noise[j] = 0.1
    cfg.torque_measurement_noise[2] = 0.05
    oracle, sim = make_pair(128, cfg=cfg)
    oracle.reset()
    sim.reset()
    act = np.zeros((128, 6, 6), dtype=np.float32)
    act[:, :, 0] = np.nan
    act[:, :, 2] = 1.0
    act[:, :, 5] = 16.0
    samples = []
    for _ in range(60):
        obs_o, *_ = oracle.step_servos(act.astype(np.float64))
        obs_h, *_ = sim.step_servos(torch.from_numpy(act))
        obs_h = obs_h.cpu().numpy()
        np.testing.assert_allclose(obs_h[:, :, 2], obs_o[:, :, 2], atol=2e-5)
        samples.append(obs_h[:, :, 2] - 1.0)
    x = np.stack(samples)
    assert abs(x[:, :, 0].mean()) < 0.01 and 0.09 < x[:, :, 0].std() < 0.11
    assert 0.105 < x[:, :, 2].std() < 0.12  # control 0.1 and measurement 0.05 add in quadrature
    spine = sim.observe(update_imu=False)["servo"].cpu().numpy()
    np.testing.assert_allclose(spine[:, :, 2], obs_h[:, :, 2], atol=1e-7)  # same measurement draw
    err = state_errors(oracle.state, sim.state_numpy())
    assert err["pos"] < 1e-5 and err["q"] < 1e-4, err


def test_joint_limits_match_oracle():
    """enforce_joint_limits: a feedforward torque drives the knees into their
    +-2.51 rad stop; the limit rows hold them there on both sides."""
    from upkie_amd.model.default_model import default_model

    model = default_model()
    model.enforce_joint_limits = 1
    cfg = abi.default_sim_config(64, seed=2)
    cfg.init_pos[2] = 3.0  # airborne: only the limit rows act
    oracle, sim = make_pair(64, cfg=cfg, model=model)
    oracle.reset()
    sim.reset()
    act = np.zeros((64, 6, 6), dtype=np.float32)
    act[:, :, 0] = np.nan
    act[:, [1, 4], 2] = np.linspace(2.0, 8.0, 64)[:, None]  # knee feedforward torque
    act[:, :, 5] = 16.0
    for _ in range(60):
        obs_o, *_ = oracle.step_servos(act.astype(np.float64))
        obs_h, *_ = sim.step_servos(torch.from_numpy(act))
    obs_h = obs_h.cpu().numpy()
    knees_o, knees_h = obs_o[:, [1, 4], 0], obs_h[:, [1, 4], 0]
    assert np.all(knees_o > 2.5) and np.all(knees_o < 2.56)  # resting on the stop (ERP 0.2)
    np.testing.assert_allclose(knees_h, knees_o, atol=2e-4)
    np.testing.assert_allclose(obs_h[:, :, 0], obs_o[:, :, 0], atol=2e-3)
    # without the flag the knees run through the limit
    free = default_model()
    free.enforce_joint_limits = 0
    oracle2, sim2 = make_pair(64, cfg=cfg, model=free)
    oracle2.reset(); sim2.reset()
    for _ in range(60):
        obs_h2, *_ = sim2.step_servos(torch.from_numpy(act))
    assert float(obs_h2[:, 1, 0].min()) > 3.0


def test_dense_batch_variant_matches_small_batch_variant():
    """Batches >= 131072 run the register-capped (2 waves/SIMD) build of the
    same kernel: env i of a big batch must equal env i of a small one."""
    from upkie_amd.sim import BatchedSim

    big = BatchedSim(randomized_config(131072, seed=21, autoreset=True))
    small = BatchedSim(randomized_config(40000, seed=21, autoreset=True))  # one lane per env, 512-register build
    big.reset()
    small.reset()
    big.obs4.copy_(big.obs6[:, [1, 0, 4, 3]])
    small.obs4.copy_(small.obs6[:, [1, 0, 4, 3]])
    for _ in range(30):
        ob, *_ = big.step_pendulum_agent()
        os_, *_ = small.step_pendulum_agent()
    # different register allocation may reorder fp32 operations: tolerance, not bits
    assert float((ob[:40000] - os_).abs().max()) < 1e-3
    assert float((big.state[:25, :40000] - small.state[:25]).abs().max()) < 5e-2


def test_sharded_pendulum_single_rank_rollout():
    """The bench's driver object: records land in the rollout ring buffer and
    equal what the four-array agent step produces."""
    from upkie_amd.distributed import ShardedPendulum
    from upkie_amd.sim import BatchedSim

    cfg = randomized_config(200, seed=14, autoreset=True)
    env = ShardedPendulum(cfg, device="cuda:0", horizon=8)
    ref = BatchedSim(cfg)
    env.reset()
    ref.reset()
    ref.obs4.copy_(ref.obs6[:, [1, 0, 4, 3]])
    history = []
    for _ in range(20):
        env.step_agent()
        obs, rew, term, trunc = ref.step_pendulum_agent()
        history.append((obs.clone(), term.clone()))
    env.flush()
    for back in range(8):  # the ring keeps the last 8 steps
        rec = env.gather.last(back)[0]
        obs, term = history[-1 - back]
        assert torch.equal(rec[:, :4], obs) and torch.equal(rec[:, 5], term.float())
    assert torch.equal(env.sim.state, ref.state)
    env.shutdown()


@pytest.mark.parametrize("wide", ["2", "8"])
@pytest.mark.parametrize("mode", ["servos", "gyropod"])
def test_two_lanes_per_env_equals_one_lane_per_env(mode, wide, monkeypatch):
    """The pair mapping (one leg per lane, DPP exchanges), the eight-lane
    mapping (one body per lane) and the one-lane mapping are schedules of the
    same arithmetic: same results up to fp32 summation order, with inertia
    randomisation, pushes, noise and joint limits all active."""
    from upkie_amd.sim import BatchedSim

    cfg = randomized_config(333, seed=17, autoreset=True)  # odd size: a half-filled last wave
    cfg.fall_pitch = 0.3
    for j in range(6):
        cfg.torque_control_noise[j] = 0.02
        cfg.joint_friction[j] = 0.05
    cfg.torque_measurement_noise[4] = 0.03
    sims = []
    for lanes in ("1", wide):
        monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
        sim = BatchedSim(cfg)
        assert sim.lanes_per_env == int(lanes)
        sim.randomize_inertias(0.2)
        force = torch.zeros(3, 333)
        force[0] = torch.linspace(-8, 8, 333)
        sim.set_external_force(force, point=(0.0, 0.0, -0.1))
        sim.reset()
        sims.append(sim)
    def compare(outs, hard_q):
        a, b = sims[0].state, sims[1].state
        assert_mostly_close(a[:7].t().cpu().numpy(), b[:7].t().cpu().numpy(), atol=2e-4, fraction=0.9, hard_atol=2e-2)
        # (knees resting on their stops under noise: a stiff, rounding-sensitive regime)
        assert_mostly_close(a[abi.S_Q : abi.S_Q + 6].t().cpu().numpy(), b[abi.S_Q : abi.S_Q + 6].t().cpu().numpy(), atol=5e-3, fraction=0.95, hard_atol=hard_q)
        if mode == "servos":  # velocities / torques chatter on the stops: compare the reported positions
            assert_mostly_close(outs[0][0][:, :, 0].cpu().numpy(), outs[1][0][:, :, 0].cpu().numpy(), atol=5e-3, fraction=0.95)
        else:
            assert_mostly_close(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), atol=2e-2, fraction=0.9)

    rng = np.random.default_rng(0)
    for step in range(40):
        if mode == "servos":
            act = np.zeros((333, 6, 6), dtype=np.float32)
            act[:, :, 0] = rng.uniform(-0.1, 0.1, (333, 6))
            act[:, [2, 5], 0] = np.nan
            if step > 10:  # later: free the knees and drive them into their stops
                act[:, [1, 4], 0] = np.nan
                act[:, [1, 4], 2] = 6.0
            act[:, :, 3:5] = 1.0
            act[:, :, 5] = 16.0
            outs = [s.step_servos(torch.from_numpy(act)) for s in sims]
        else:
            act = rng.uniform(-0.3, 0.3, (333, 2)).astype(np.float32)
            outs = [s.step_gyropod(torch.from_numpy(act)) for s in sims]
        assert torch.equal(outs[0][2], outs[1][2])  # terminated flags
        if step == 25:
            compare(outs, hard_q=0.05)
    a, b = sims[0].state, sims[1].state
    assert torch.equal(a[abi.S_EPISODE], b[abi.S_EPISODE]) and torch.equal(a[abi.S_STEP], b[abi.S_STEP])
    # Knees pushed against their stops by 6 N.m under torque noise: a stiff regime in which two schedules of the same
    # arithmetic drift apart tenfold every five steps (1e-8 -> 1e-3 between steps 10 and 39 for the pair mapping, from
    # 5e-7 for the eight-lane one, whose sums associate differently from the start). Held tightly 14 steps into that
    # regime (above) and loosely at the end.
    assert_mostly_close(a[:7].t().cpu().numpy(), b[:7].t().cpu().numpy(), atol=2e-3, fraction=0.9, hard_atol=5e-2)
    assert_mostly_close(a[abi.S_Q : abi.S_Q + 6].t().cpu().numpy(), b[abi.S_Q : abi.S_Q + 6].t().cpu().numpy(), atol=1e-2, fraction=0.9, hard_atol=0.2)


@pytest.mark.parametrize("lanes", ["1", "2", "8"])
def test_external_forces_on_any_link_match_oracle(lanes, monkeypatch):
    """Four simultaneous forces (pybullet_backend.py:603-658): world-frame on
    the trunk, link-frame on a calf, world-frame on the other leg's thigh and
    on a wheel, per env, through both lane mappings."""
    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 200
    cfg = randomized_config(B, seed=21)
    cfg.init_pos[2] = 0.9  # hanging above the floor: no contact masks the effect of the forces
    oracle, sim = make_pair(B, cfg=cfg)
    rng = np.random.default_rng(4)
    bodies, local = [0, 2, 4, 6], [False, True, False, False]
    points = [[0.0, 0.0, 0.05], [0.01, -0.02, -0.1], [0.0, 0.03, -0.08], [0.0, 0.0, 0.0]]
    forces = rng.uniform(-6.0, 6.0, (4, 3, B))
    forces[0, 2] += 30.0  # most of the weight is carried
    slots = abi.UpkieExternalForces()
    slots.count = 4
    for i in range(4):
        slots.body[i], slots.local[i] = bodies[i], int(local[i])
        for k in range(3):
            slots.point[i][k] = points[i][k]
    oracle.ext_force, oracle.ext_slots = forces, slots
    sim.set_external_forces(torch.from_numpy(forces).float(), bodies=bodies, points=points, local=local)
    oracle.reset()
    sim.reset()
    act = np.zeros((B, 6, 6))
    act[:, :, 0] = np.nan  # no position feedback: the legs swing under the forces
    act[:, :, 4] = 1.0
    act[:, :, 5] = 16.0
    for _ in range(8):
        oracle.step_servos(act)
        sim.step_servos(torch.from_numpy(act).float())
    err = state_errors(oracle.state, sim.state_numpy())
    assert err["pos"] < 2e-5 and err["quat"] < 2e-5, err
    assert err["q"] < 2e-4, err
    # joint velocities per env: a few legs reach a joint stop during the swing (a discontinuity)
    assert_mostly_close(sim.state_numpy()[abi.S_QD : abi.S_QD + 6].T, oracle.state[abi.S_QD : abi.S_QD + 6].T, atol=2e-2, fraction=0.95, hard_atol=0.2)
    moved = np.abs(sim.state_numpy()[abi.S_Q : abi.S_Q + 6]).max(axis=1)
    assert moved[1] > 1e-2 and moved[3] > 1e-2  # left knee (calf force) and right hip (thigh force) did swing
    # removing the forces: free fall again
    sim.set_external_forces(None)
    oracle.ext_force, oracle.ext_slots = None, None
    vz0 = sim.state_numpy()[abi.S_LINVEL + 2].copy()
    sim.step_servos(torch.from_numpy(act).float())
    oracle.step_servos(act)
    dv = sim.state_numpy()[abi.S_LINVEL + 2] - vz0
    assert np.allclose(dv, -9.81 * 0.005, atol=8e-3)  # the base is not the centre of mass: swinging legs shift it
    assert state_errors(oracle.state, sim.state_numpy())["pos"] < 3e-5


@pytest.mark.parametrize("lanes", ["1", "2", "8"])
def test_joint_limit_on_one_leg_only(lanes, monkeypatch):
    """Asymmetric joint stops: only the RIGHT knee is driven into its stop (the
    left leg is held). In the two-lanes-per-env mapping the lane owning the left
    leg learns about the limit from its partner: regression test for an
    exchange that was skipped by a short-circuit `||` (the left lane then took
    the contact-only branch while the right one took the limit branch)."""
    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 96
    cfg = randomized_config(B, seed=33)
    cfg.init_pos[2] = 3.0  # falling freely for the whole test: the limit rows are the only constraints
    oracle, sim = make_pair(B, cfg=cfg)
    oracle.reset()
    sim.reset()
    act = np.zeros((B, 6, 6))
    act[:, :, 3:5] = 1.0
    act[:, :, 5] = 16.0
    act[:, [2, 5], 0] = np.nan
    act[:, 4, 0] = np.nan  # right knee: no position feedback, pushed by a feedforward torque
    act[:, 4, 2] = 10.0
    for _ in range(120):  # 10 N m against kd = 1: about 10 rad/s, the stop at 2.51 rad is reached after ~0.3 s
        oracle.step_servos(act)
        sim.step_servos(torch.from_numpy(act).float())
    s = sim.state_numpy()
    assert np.isfinite(s[:25]).all()
    limit = float(sim.model.joint_upper[4])
    assert np.all(np.abs(s[abi.S_Q + 4]) > limit - 2e-3) and np.all(np.abs(s[abi.S_Q + 4]) < limit + 2e-2)  # resting on the stop
    assert np.all(np.abs(s[abi.S_Q + 1]) < 0.2)  # the left knee stayed where it was held
    err = state_errors(oracle.state, s)
    legs = [abi.S_Q + j for j in (0, 1, 3, 4)]  # (free wheels drift apart in fp32: 2.8e-4 kg m^2 of inertia)
    # (round 6, gap-aware limit rows: the knee arrives ON its stop in both, legs within 3e-5 rad -- 2e-3 under the old rule, which
    # let it overshoot and pushed it back --; the free right wheel takes the arrival's kick: 3e-2 rad after 120 steps)
    assert np.abs(oracle.state[legs] - s[legs]).max() < 2e-4 and err["pos"] < 1e-3 and err["q"] < 6e-2, err


@pytest.mark.parametrize("lanes", ["1", "2", "8"])
def test_collapse_with_limp_joints_stays_finite(lanes, monkeypatch):
    """Every servo limp (no position feedback, zero damping gain): the robots
    fall, fold into hip and knee stops on one side or both and roll on their
    tires. 400 steps through the rare paths (limit rows + contacts, PGS
    fallback) in either mapping: finite, joints within their stops."""
    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 192
    cfg = randomized_config(B, seed=5)
    _, sim = make_pair(B, cfg=cfg)
    sim.reset()
    act = torch.zeros((B, 6, 6))
    act[:, :, 0] = float("nan")
    act[:, :, 3] = 1.0
    act[:, :, 5] = 16.0
    for _ in range(400):
        sim.step_servos(act)
    s = sim.state_numpy()
    assert np.isfinite(s[:25]).all()
    lower, upper = np.array(sim.model.joint_lower[:]), np.array(sim.model.joint_upper[:])
    for j in (0, 1, 3, 4):
        assert np.all(s[abi.S_Q + j] > lower[j] - 0.05) and np.all(s[abi.S_Q + j] < upper[j] + 0.05)
    assert np.abs(s[abi.S_LINVEL : abi.S_LINVEL + 3]).max() < 20.0


def test_flailing_robots_both_mappings_step_alike():
    """Random full-scale feedforward torques on every joint, no position
    feedback: legs flail into their stops (one leg, the other, both), tires
    slip, robots tumble. Each step is taken by BOTH lane mappings from the same
    state and compared, then the trajectory continues from the one-lane result:
    per-step equivalence of the two mappings over every regime the trajectory
    visits, without chaos getting in the way."""
    from upkie_amd.sim import BatchedSim

    B = 256
    cfg = randomized_config(B, seed=77)
    sims = {}
    import os

    for lanes in ("1", "2"):
        os.environ["UPKIE_LANES_PER_ENV"] = lanes
        sims[lanes] = BatchedSim(cfg)
    os.environ.pop("UPKIE_LANES_PER_ENV")
    a, b = sims["1"], sims["2"]
    a.reset()
    # third schedule: the 256-register build that very large batches run (its joint-limit solve
    # works in scratch memory): the first B envs of a 131072-env batch mirror the trajectory
    BIG = 131072
    dense = BatchedSim(randomized_config(BIG, seed=77))
    dense.reset()
    act_big = torch.zeros((BIG, 6, 6), device=dense.device)
    act_big[:, :, 0] = float("nan")
    act_big[:, :, 4] = 0.2
    act_big[:, :, 5] = 16.0
    act_big[:, :, 2] = (torch.rand((BIG, 6), device=dense.device) * 2 - 1) * torch.tensor([16, 16, 1.7, 16, 16, 1.7], device=dense.device)
    from oracle import oracle as O

    oracle = O.Oracle(a.model, cfg)
    rng = np.random.default_rng(3)
    worst = 0.0
    oracle_far = []
    limit_steps = 0
    lower, upper = np.array(a.model.joint_lower[:]), np.array(a.model.joint_upper[:])
    for step in range(250):
        act = np.zeros((B, 6, 6), dtype=np.float32)
        act[:, :, 0] = np.nan
        act[:, :, 2] = rng.uniform(-1.0, 1.0, (B, 6)) * np.array([16, 16, 1.7, 16, 16, 1.7])
        act[:, :, 4] = 0.2
        act[:, :, 5] = 16.0
        b.state.copy_(a.state)
        oracle.state[:] = a.state_numpy().astype(np.float64)
        t = torch.from_numpy(act)
        dense.state[:, :B].copy_(a.state)
        act_big[:B].copy_(t)
        a.step_servos(t)
        b.step_servos(t)
        dense.step_servos(act_big)
        oracle.step_servos(act.astype(np.float64))
        sa, sb = a.state_numpy(), b.state_numpy()
        sd = dense.state[:, :B].cpu().numpy()
        assert np.isfinite(sd[:25]).all(), step
        assert np.mean(np.abs(sa[:19] - sd[:19]).max(axis=0) > 2e-3) <= 0.05, step
        # ... and the fp64 oracle takes the same step from the same state
        eo = np.abs(sa[:19] - oracle.state[:19]).max(axis=0)
        oracle_far.append(float(np.mean(eo > 5e-3)))
        assert np.isfinite(sa[:25]).all() and np.isfinite(sb[:25]).all(), step
        q = sa[abi.S_Q : abi.S_Q + 6]
        # (round 6: a joint ARRIVES on its stop -- the gap-aware limit row lets it close the gap within the substep -- instead of
        # overshooting it and being pushed back: "at the stop" is within 1e-4 rad of it, either side)
        limit_steps += int(((q[[0, 1, 3, 4]] <= lower[[0, 1, 3, 4], None] + 1e-4) | (q[[0, 1, 3, 4]] >= upper[[0, 1, 3, 4], None] - 1e-4)).any(axis=0).sum())
        err = np.abs(sa[:19] - sb[:19]).max(axis=0)  # positions, orientation, velocities, joint angles per env
        # a step amplifies fp32 rounding differently in the two schedules only where a decision flips
        # (stick/slip, a row switching on): allow a small fraction of envs per step
        assert np.mean(err > 2e-3) <= 0.05, (step, float(err.max()))
        worst = max(worst, float(np.median(err)))
    assert worst < 2e-4
    assert limit_steps > 2000  # the stops were visited a lot (by one leg or both)
    # fp32 kernel vs fp64 oracle, one step at a time through the same regimes: a few percent of the
    # env-steps land on the other side of a discontinuity (slip onset, a stop switching on)
    assert np.mean(oracle_far) < 0.05 and max(oracle_far) < 0.25, (np.mean(oracle_far), max(oracle_far))


def test_contact_points_match_oracle():
    """upkie_sim_contact_points vs the oracle's restatement of
    PyBulletBackend.get_contact_points (pybullet_backend.py:660-716): same
    tires in contact, same points, same forces -- rolling, pushed and with
    randomised inertias; and the query leaves the state untouched."""
    B = 256
    oracle, sim = make_pair(B, seed=21)
    records = sim.randomize_inertias(0.2)
    oracle.body_inertials = records.cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(5)
    force = rng.uniform(-3.0, 3.0, size=(3, B))
    sim.set_external_force(torch.from_numpy(force.astype(np.float32)), point=(0.0, 0.0, 0.1))
    oracle.ext_force = np.ascontiguousarray(force.astype(np.float32).astype(np.float64))
    oracle.ext_point = np.array([0.0, 0.0, 0.1])
    oracle.reset()
    sim.reset()
    act = rng.uniform(-0.4, 0.4, size=B).astype(np.float32)
    for _ in range(5):
        oracle.step_pendulum(act.astype(np.float64))
        sim.step_pendulum(torch.from_numpy(act))
    # compare the query itself: same state on both sides
    oracle.state[:] = sim.state_numpy().astype(np.float64)
    before = sim.state_numpy().copy()
    ref = oracle.contact_points()
    got = sim.contact_points().cpu().numpy().astype(np.float64)
    assert np.array_equal(sim.state_numpy(), before)
    assert np.array_equal(got[:, :, 0], ref[:, :, 0]) and ref[:, :, 0].sum() == 2 * B
    np.testing.assert_allclose(got[:, :, 1:4], ref[:, :, 1:4], atol=2e-6)
    weight = 9.81 * float(sum(sim.model.mass[:]))
    # forces: fp32 rows with a 1/h = 1000 gain on velocities -> relative to the weight
    assert_mostly_close(got[:, :, 4:7].reshape(B, -1), ref[:, :, 4:7].reshape(B, -1), atol=2e-3 * weight, hard_atol=2e-2 * weight)
    assert np.all(got[:, :, 7] == 0.0)
    total = got[:, :, 6].sum(axis=1)
    # robots were dropped on the contact spring a few steps ago: the floor carries the weight on the median only
    assert np.all(total > 0.5 * weight) and abs(np.median(total) - weight) < 0.25 * weight


def test_contact_points_in_the_air_and_on_one_wheel():
    from upkie_amd.utils.robot_state import RobotState  # noqa: F401

    B = 64
    oracle, sim = make_pair(B, seed=22)
    oracle.reset()
    sim.reset()
    state = sim.state_numpy().astype(np.float64)
    state[abi.S_POS + 2, : B // 2] += 1.0  # lifted
    # rolled about x: right wheel off the floor, left wheel pressed in
    roll = 0.2  # tires at +-0.03 m: beyond the 0.02 m manifold breaking threshold once the base is lifted
    q = state[abi.S_QUAT : abi.S_QUAT + 4, B // 2 :]
    qr = np.array([np.cos(roll / 2), np.sin(roll / 2), 0.0, 0.0])
    w0, x0, y0, z0 = q
    state[abi.S_QUAT : abi.S_QUAT + 4, B // 2 :] = np.stack([
        qr[0] * w0 - qr[1] * x0, qr[0] * x0 + qr[1] * w0, qr[0] * y0 - qr[1] * z0, qr[0] * z0 + qr[1] * y0])
    state[abi.S_POS + 2, B // 2 :] += 0.015
    oracle.state[:] = state
    sim.state.copy_(torch.from_numpy(state.astype(np.float32)))
    oracle.state[:] = sim.state_numpy().astype(np.float64)
    ref = oracle.contact_points()
    got = sim.contact_points().cpu().numpy().astype(np.float64)
    assert np.array_equal(got[:, :, 0], ref[:, :, 0])
    assert got[: B // 2].max() == 0.0 and np.abs(got[: B // 2]).max() == 0.0
    assert np.all(got[B // 2 :, 0, 0] + got[B // 2 :, 1, 0] == 1.0)  # exactly one tire down
    np.testing.assert_allclose(got[:, :, 1:4], ref[:, :, 1:4], atol=2e-6)
    weight = 9.81 * float(sum(sim.model.mass[:]))
    np.testing.assert_allclose(got[:, :, 4:7], ref[:, :, 4:7], atol=2e-2 * weight)
    lists = sim.get_contact_points(env=B - 1)
    assert len(lists) == 1 and lists[0].link_name in ("left_wheel_tire", "right_wheel_tire")
    assert sim.get_contact_points("torso", env=B - 1) == []


@pytest.mark.parametrize("lanes", ["1", "2", "8"])
def test_per_link_inertia_randomisation_on_the_urdf_model(lanes, monkeypatch):
    """randomize_inertias on the URDF-derived model (13 links behind 7 bodies,
    pybullet_backend.py:555-601): link factors and fused body records equal the
    oracle's, and both step alike with them (shifted centres of mass, wheels
    whose hub and tire got different factors)."""
    from upkie_amd.model.model import Model

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 192
    oracle, sim = make_pair(B, seed=31, model=Model().struct)
    rec_h = sim.randomize_inertias(0.3).cpu().numpy().astype(np.float64)
    rec_o = oracle.sample_body_inertials(0.3)
    np.testing.assert_allclose(sim.link_scale.cpu().numpy(), oracle.link_scale, atol=1e-6)
    np.testing.assert_allclose(rec_h, rec_o, rtol=3e-5, atol=1e-9)
    assert np.all(oracle.link_scale[0] == 1.0) and np.std(oracle.link_scale[1:13]) > 0.1
    assert np.std(rec_o[0]) > 0 and np.std(rec_o[30]) > 0.01  # trunk barely (light links only), wheels a lot
    oracle.body_inertials = rec_h  # same records on both sides: compare the dynamics, not the fp32 fusion
    oracle.reset()
    sim.reset()
    rng = np.random.default_rng(2)
    act = rng.uniform(-0.3, 0.3, size=B).astype(np.float32)
    for _ in range(4):
        obs_o, *_ = oracle.step_pendulum(act.astype(np.float64))
        obs_h, *_ = sim.step_pendulum(torch.from_numpy(act))
    assert_mostly_close(obs_h.cpu().numpy(), obs_o, atol=2e-3, hard_atol=2e-2)
    err = state_errors(oracle.state, sim.state_numpy())
    assert err["pos"] < 2e-4 and err["quat"] < 2e-4, err
    # and the randomisation matters: the same actions without it end elsewhere
    plain_o, plain = make_pair(B, seed=31, model=Model().struct)
    plain.reset()
    for _ in range(4):
        obs_p, *_ = plain.step_pendulum(torch.from_numpy(act))
    assert float((obs_p - obs_h).abs().max()) > 1e-3


@pytest.mark.parametrize("lanes", ["1", "2", "8"])
def test_time_limit_in_the_kernel_matches_oracle(lanes, monkeypatch):
    """UpkieSimConfig.max_episode_steps (gymnasium's TimeLimit for the batch):
    `truncated` on the step that reaches the limit unless the robot fell in it,
    the DONE word set, NEXT_STEP autoreset restarting the count; the packed
    record carries the flag in word 6."""
    from upkie_amd.sim import BatchedSim

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 130
    cfg = randomized_config(B, seed=8, autoreset=True)
    cfg.max_episode_steps = 7
    cfg.fall_pitch = 0.09  # some envs (initial pitch up to 0.1) fall before the limit
    oracle, sim = make_pair(B, cfg=cfg)
    oracle.reset()
    sim.reset()
    packed = BatchedSim(cfg)
    packed.reset()
    act = np.linspace(-0.6, 0.6, B).astype(np.float32)
    records = torch.zeros((B, 8), device=sim.device)
    seen_trunc = seen_term = 0
    for step in range(30):
        _, _, term_o, trunc_o = oracle.step_pendulum(act.astype(np.float64))
        _, _, term_h, trunc_h = sim.step_pendulum(torch.from_numpy(act))
        packed.step_pendulum_packed(records, torch.from_numpy(act))
        oracle.state[:] = sim.state_numpy().astype(np.float64)  # keep the closed loops together
        assert np.array_equal(term_h.cpu().numpy(), term_o) and np.array_equal(trunc_h.cpu().numpy(), trunc_o), step
        assert np.array_equal(records[:, 5].cpu().numpy(), term_o.astype(np.float32))
        assert np.array_equal(records[:, 6].cpu().numpy(), trunc_o.astype(np.float32))
        assert not np.any(term_o & trunc_o)
        seen_trunc += int(trunc_o.sum())
        seen_term += int(term_o.sum())
    assert seen_trunc > B and seen_term > 0
    assert sim.state_numpy()[abi.S_ELAPSED].max() <= 7


@pytest.mark.parametrize("B", [16381, 8])
def test_rollout_in_one_launch_equals_step_by_step_at_the_ends_of_the_eight_lane_range(B):
    """The multi-step kernel reaches the state through a buffer descriptor with 32-bit offsets (state_words.hpp): the
    same bit-for-bit equality with chained launches at the largest batch the eight-lane mapping serves (two wavefronts
    per SIMD, a ragged last wavefront) and at a single wavefront's worth of envs."""
    from upkie_amd.sim import BatchedSim

    K = 16
    cfg = randomized_config(B, seed=3, autoreset=True)
    cfg.fall_pitch = 0.11
    cfg.max_episode_steps = 11
    a, b = BatchedSim(cfg), BatchedSim(cfg)
    assert a.lanes_per_env == 8
    o6 = a.reset()
    b.reset()
    prev = torch.zeros((B, 8), device=a.device)
    prev[:, :4] = o6[:, [1, 0, 4, 3]]
    for window in range(2):
        fused = torch.zeros((K, B, 8), device=a.device)
        a.rollout_pendulum_records(prev, fused)
        chained = torch.zeros((K, B, 8), device=a.device)
        p = prev
        for k in range(K):
            b.step_pendulum_records(p, chained[k])
            p = chained[k]
        assert torch.equal(fused, chained), window
        assert torch.equal(a.state[:48], b.state[:48]), window
        prev = fused[K - 1].clone()
    assert float(fused[:, :, 5].sum()) + float(fused[:, :, 6].sum()) > 0  # falls / time limits happened


@pytest.mark.parametrize("lanes", ["8", "2", "1"])
def test_rollout_in_one_launch_equals_step_by_step(lanes, monkeypatch):
    """upkie_sim_step_pendulum_agent_rollout: K fused-agent steps in one launch
    (two lanes per env: state carried in registers; one lane per env: K
    launches) are bit-identical to K chained upkie_sim_step_pendulum_agent_records
    calls -- with falls, NEXT_STEP autoresets, a time limit and torque noise in
    the window."""
    from upkie_amd.sim import BatchedSim

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B, K = 333, 24
    cfg = randomized_config(B, seed=12, autoreset=True)
    cfg.fall_pitch = 0.11
    cfg.max_episode_steps = 9
    for j in range(6):
        cfg.torque_control_noise[j] = 0.05
    a, b = BatchedSim(cfg), BatchedSim(cfg)
    o6 = a.reset()
    b.reset()
    prev = torch.zeros((B, 8), device=a.device)
    prev[:, :4] = o6[:, [1, 0, 4, 3]]
    for window in range(3):  # carried counters must survive the launch boundary too
        fused = torch.zeros((K, B, 8), device=a.device)
        a.rollout_pendulum_records(prev, fused)
        chained = torch.zeros((K, B, 8), device=a.device)
        p = prev
        for k in range(K):
            b.step_pendulum_records(p, chained[k])
            p = chained[k]
        assert torch.equal(fused, chained), window
        assert torch.equal(a.state[:48], b.state[:48]), window
        prev = fused[K - 1].clone()
    assert float(fused[:, :, 5].sum()) + float(chained[:, :, 6].sum()) > 0  # falls / time limits happened
    assert float(a.state[abi.S_EPISODE].max()) > 3

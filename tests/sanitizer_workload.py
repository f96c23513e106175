"""Run by tests/test_sanitizers.py in a python started under AddressSanitizer
(LD_PRELOAD) with the oracle (`oracle`) or the host build of the device
arithmetic (`harness`) compiled with -fsanitize=address,undefined: the paths the
CPU suite exercises -- every env kind of the oracle through resets, falls,
pushes, randomised inertias, joints at their stops, both contact models, the
balancer and the observers; the captured contact systems replayed through the
device's sweeps and active-set solve, one-, and eight-lane substeps."""

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_workload():
    import bench
    from oracle import oracle as O
    from tests.fake_sim import OracleMpc, servo_policy_action
    from tests.test_one_step_parity_gpu import stops_policy
    from upkie_amd import abi
    from upkie_amd.model.default_model import default_model

    B = 64
    model = default_model()
    for bullet_like in (False, True):
        cfg = bench.make_config(B)
        cfg.joint_friction[2] = cfg.joint_friction[5] = 0.1
        cfg.torque_control_noise[2] = 0.01
        cfg.torque_measurement_noise[2] = 0.01
        ref = O.Oracle(model, cfg)
        if bullet_like:
            ref.use_bullet_like_contacts()
        ref.body_inertials = ref.sample_body_inertials(0.2)
        ref.ext_force = ref.sample_pushes(0, 20.0)
        ref.ext_point = np.zeros(3)
        obs = ref.reset()[:, [1, 0, 4, 3]]
        for _ in range(150):
            obs, _, _, _ = ref.step_pendulum_agent(obs)
        rs = float(model.left_sign) * float(model.wheel_radius)
        for policy in (abi.torque_balancing_policy(10.0, 1.0, float(model.left_sign)), stops_policy(model)):
            ref.reset()
            for _ in range(150):
                act, fallen = servo_policy_action(policy, ref.state, rs)
                ref.state[abi.S_DONE] = np.where(fallen, 1.0, ref.state[abi.S_DONE])
                ref.step_servos(act)
        ref.step_gyropod(np.full((B, 2), np.nan))  # the non-finite guard
        ref.observe()
        ref.contact_points()
        assert np.isfinite(ref.state).all()
    import torch

    for horizon in (16, 50):
        mpc = OracleMpc(abi.default_mpc_config(B, nb_timesteps=horizon))
        x0 = torch.zeros((B, 4))
        x0[:, 1] = 0.05
        for _ in range(5):
            mpc.step(x0, torch.full((B,), 0.3), torch.ones(B, dtype=torch.uint8), 0.005)
    obs_cfg = abi.default_observer_config(B, 1e-3)
    oo = O.ObserverOracle(obs_cfg)
    rng = np.random.default_rng(0)
    for _ in range(50):
        oo.step(rng.normal(size=(B, 6, 5)), np.tile([1.0, 0, 0, 0], (B, 1)), rng.normal(size=(B, 3)))
    print("oracle workload done")


def harness_workload(path):
    from oracle import oracle as O
    from tests import test_contact_sweeps_replay as replay
    from tests import test_device_arithmetic_on_host as host
    from upkie_amd import abi
    from upkie_amd.model.default_model import default_model

    harness = C.CDLL(path)
    model = default_model()
    rng = np.random.default_rng(5)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    for on_floor in (False, True):
        for _ in range(60):
            s = host.random_state(rng, on_floor)
            host.run_both(harness, model, s, rng.uniform(-1.5, 1.5, 6))
            # the same state through the eight-lane substep (eight lockstep threads) and the Bullet-like substeps
            s32 = s.astype(np.float32)
            tau = rng.uniform(-1.5, 1.5, 6).astype(np.float32)
            status = np.zeros(4, dtype=np.int32)
            harness.harness_substep_octet(C.byref(model), p(s32), p(tau), C.c_float(1e-3), None, None, 4, p(status), 0)
            s32 = s.astype(np.float32)
            harness.harness_substep_octet(C.byref(model), p(s32), p(tau), C.c_float(1e-3), None, None, 4, p(status), 1)
            s32 = s.astype(np.float32)
            manifold = np.zeros(64, dtype=np.float32)
            for _k in range(4):
                harness.harness_substep_bullet_like(C.byref(model), p(s32), p(tau), C.c_float(1e-3), p(manifold))
            s32 = s.astype(np.float32)
            applied = np.zeros(2, dtype=np.float32)
            harness.harness_substep_octet_bullet_like(C.byref(model), p(s32), p(tau), C.c_float(1e-3), 4, p(status), p(applied))
    # joints at their stops (limit paths)
    for _ in range(40):
        s = host.random_state(rng, True)
        s[abi.S_Q + 1] = model.joint_upper[1] + 1e-3
        s[abi.S_Q + 3] = model.joint_lower[3] - 1e-3
        host.run_both(harness, model, s, rng.uniform(-1.5, 1.5, 6))
        s32 = s.astype(np.float32)
        tau = rng.uniform(-1.5, 1.5, 6).astype(np.float32)
        status = np.zeros(4, dtype=np.int32)
        for in_registers in (0, 1):
            t = s32.copy()
            harness.harness_substep_octet(C.byref(model), p(t), p(tau), C.c_float(1e-3), None, None, 4, p(status), in_registers)
        manifold = np.zeros(64, dtype=np.float32)
        t = s32.copy()
        harness.harness_substep_bullet_like(C.byref(model), p(t), p(tau), C.c_float(1e-3), p(manifold))
    # captured contact systems through the sweeps and the active-set solve
    _, packed, rhs, warm, _, _ = replay.captured_systems(envs=32, steps=200)
    harness.harness_contact_pgs6.restype = C.c_int
    harness.harness_contact_solve6.restype = C.c_int
    for i in range(len(packed)):
        a32, r32 = np.ascontiguousarray(packed[i], dtype=np.float32), np.ascontiguousarray(rhs[i], dtype=np.float32)
        lam = np.ascontiguousarray(warm[i], dtype=np.float32)
        harness.harness_contact_pgs6(C.byref(model), p(a32), p(r32), p(lam), 1)
        lam = np.ascontiguousarray(warm[i], dtype=np.float32)
        harness.harness_contact_solve6(C.byref(model), p(a32), p(r32), p(lam))
    print(f"harness workload done ({len(packed)} captured systems)")


if __name__ == "__main__":
    if sys.argv[1] == "oracle":
        oracle_workload()
    else:
        harness_workload(sys.argv[2])

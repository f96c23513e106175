"""HIP observer pipeline (through the C-ABI) vs the fp64 observer oracle on
the same seeded input sequences; the oracle itself is pinned by the
reference's observer tests (tests/test_oracle_observers.py).

Tolerances: the kernel filters in fp32; a 200-step low-pass recursion on
O(10) signals accumulates O(1e-5) absolute error. Contact flags are decisions
on thresholds: they must agree except for envs whose estimate sits within
fp32 noise of a threshold (checked: <= 0.5 % of the env-steps)."""

import numpy as np
import pytest
import torch

from oracle import oracle as O
from upkie_amd import abi
from upkie_amd.observers import BatchedObservers

pytestmark = pytest.mark.gpu


def random_sequence(B, steps, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(steps)[:, None]
    phase = rng.uniform(0, 2 * np.pi, (1, B))
    loaded = (np.sin(0.01 * t + phase) > -0.3)  # envs touch down and lift off at their own times
    servo = np.zeros((steps, B, 6, 5))
    servo[:, :, :, 0] = rng.uniform(-1, 1, (steps, B, 6))
    for w in (2, 5):
        amp = np.where(loaded, 0.5, 4.0)
        servo[:, :, w, 1] = amp * np.sin(0.05 * t + rng.uniform(0, 6, (1, B))) * 5.0 + 0.1 * rng.standard_normal((steps, B))
        servo[:, :, w, 2] = np.where(loaded, 0.8 + 0.3 * rng.standard_normal((steps, B)), 0.002 * rng.standard_normal((steps, B)))
    for j in (0, 1, 3, 4):
        servo[:, :, j, 1] = rng.standard_normal((steps, B))
        servo[:, :, j, 2] = np.where(loaded, 4.0, 0.3) * rng.standard_normal((steps, B)) + np.where(rng.uniform(size=(1, B)) < 0.1, 8.0, 0.0)
    q = rng.standard_normal((steps, B, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w = rng.standard_normal((steps, B, 3))
    cross = (rng.uniform(size=(steps, B)) < 0.002).astype(np.uint8)
    return servo, q, w, cross


@pytest.mark.parametrize("dt", [1e-3, 1.0 / 250.0])
def test_observer_pipeline_matches_oracle(dt):
    B, steps = 1000, 300
    cfg = abi.default_observer_config(B, dt)
    servo, q, w, cross = random_sequence(B, steps, seed=3)
    oracle = O.ObserverOracle(cfg)
    dev = BatchedObservers(cfg)
    dev.reset()
    flag_mismatch = 0
    seen = set()
    for k in range(steps):
        ref = oracle.step(servo[k], q[k], w[k], cross[k])
        out = dev.step(
            torch.from_numpy(servo[k]).float(),
            torch.from_numpy(q[k]).float(),
            torch.from_numpy(w[k]).float(),
            torch.from_numpy(cross[k]),
        )
        fc = out["floor_contact"]["contact"].cpu().numpy()
        mism = fc != ref["floor_contact"]
        flag_mismatch += int(mism.sum())
        seen.update(np.unique(fc).tolist())
        if mism.any():  # a threshold decision flipped: resynchronise those envs so one flip is counted once
            idx = np.nonzero(mism)[0]
            dev.state[:, idx] = torch.from_numpy(oracle.state[:, idx]).float().to(dev.device)
            continue
        np.testing.assert_allclose(out["floor_contact"]["upper_leg_torque"].cpu().numpy(), ref["upper_leg_torque"], rtol=2e-5, atol=2e-5)
        wc = torch.stack(
            [
                torch.stack([out["floor_contact"][n][k2].float() for k2 in ("abs_acceleration", "abs_torque", "contact", "inertia")], -1)
                for n in ("left_wheel", "right_wheel")
            ],
            1,
        ).cpu().numpy()
        np.testing.assert_array_equal(wc[:, :, 2], ref["wheel_contact"][:, :, 2])
        np.testing.assert_allclose(wc[:, :, :2], ref["wheel_contact"][:, :, :2], rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(wc[:, :, 3], ref["wheel_contact"][:, :, 3], rtol=2e-3, atol=1e-6)
        odo = torch.stack([out["wheel_odometry"]["position"], out["wheel_odometry"]["velocity"]], -1).cpu().numpy()
        np.testing.assert_allclose(odo, ref["wheel_odometry"], rtol=1e-5, atol=2e-6)
        bo = out["base_orientation"]
        np.testing.assert_allclose(bo["pitch"].cpu().numpy(), ref["base_pitch"], atol=2e-6)
        np.testing.assert_allclose(bo["angular_velocity"].cpu().numpy(), ref["base_angular_velocity"], atol=1e-6)
        np.testing.assert_allclose(bo["rotation_base_to_world"].reshape(B, 9).cpu().numpy(), ref["rotation_base_to_world"], atol=1e-6)
    assert seen == {0, 1}
    assert flag_mismatch <= 0.005 * B * steps, flag_mismatch


def test_reference_golden_cases_on_device():
    """FloorContactTest.cpp:81-108 and BaseOrientationTest.cpp:49-56,70-86 on
    the HIP path itself."""
    dt = 1.0 / 250.0
    cfg = abi.default_observer_config(1, dt)
    cfg.wheel_cutoff_period = 3 * dt
    dev = BatchedObservers(cfg)
    servo = torch.zeros((1, 6, 5))
    servo[0, 2, 1] = servo[0, 5, 1] = 10.0
    servo[0, 2, 2] = servo[0, 5, 2] = 10.0
    dev.step(servo)
    servo[0, 2, 1] = servo[0, 5, 1] = 10.0 + 5.0 * dt
    out = dev.step(servo)
    assert bool(out["floor_contact"]["contact"][0])
    out = dev.step(servo, cross_button=torch.ones(1, dtype=torch.uint8))
    assert not bool(out["floor_contact"]["contact"][0])

    # pitch of a pure rotation about y by 1e-3 rad, through a quaternion
    cfg = abi.default_observer_config(1, 1e-3)
    for i in range(9):
        cfg.rotation_base_to_imu[i] = cfg.rotation_ars_to_world[i] = 1.0 if i % 4 == 0 else 0.0
    dev = BatchedObservers(cfg)
    theta = 1e-3
    q = torch.tensor([[np.cos(theta / 2), 0.0, np.sin(theta / 2), 0.0]])
    out = dev.step(torch.zeros((1, 6, 5)), q, torch.zeros((1, 3)))
    assert abs(float(out["base_orientation"]["pitch"][0]) - theta) < 1e-6

    cfg = abi.default_observer_config(1, 1e-3)
    for i, v in enumerate([0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]):
        cfg.rotation_base_to_imu[i] = v
    dev = BatchedObservers(cfg)
    q = torch.tensor([[0.008472769239730098, -0.9953038144146671, -0.09639792825405252, -0.002443076206500708]])
    out = dev.step(torch.zeros((1, 6, 5)), q, torch.zeros((1, 3)))
    assert abs(float(out["base_orientation"]["pitch"][0]) - (-0.016)) < 1e-3


def test_observers_on_simulated_robots():
    """The pipeline on top of the simulator at the spine rate (1 kHz): robots
    balancing on the floor are reported in contact by the estimators and the
    integrated odometry follows the ground-truth wheel odometry."""
    from upkie_amd.model.default_model import default_model
    from upkie_amd.sim import BatchedSim

    B = 256
    cfg = abi.default_sim_config(B, frequency=1000.0)  # one physics substep per env step = one spine cycle
    cfg.init_pos[2] = 0.58
    sim = BatchedSim(cfg, default_model())
    sim.reset()
    obs_cfg = abi.default_observer_config(B, 1e-3)
    obs_cfg.signed_radius[0] = +sim.model.wheel_radius
    obs_cfg.signed_radius[1] = -sim.model.wheel_radius
    observers = BatchedObservers(obs_cfg)
    observers.reset()
    obs = torch.zeros((B, 4), device=sim.device)
    out = None
    truth0 = None
    for k in range(1500):
        act = (10.0 * obs[:, 0] + 1.0 * obs[:, 1] + 0.1 * obs[:, 3]).clamp(-0.99, 0.99).reshape(B)
        act = act + 0.3 * float(np.sin(0.01 * k))  # drive back and forth so the wheels accelerate
        obs = sim.step_pendulum(act)[0]
        out = observers.step_from_sim(sim)
        if k == 999:
            truth0 = sim.observe(update_imu=False)["wheel_odometry"][:, 0].clone()
            est0 = out["wheel_odometry"]["position"].clone()
    alive = sim.state[abi.S_EPISODE] == sim.state[abi.S_EPISODE].min()  # never fell and reset
    assert alive.float().mean() > 0.9
    contact = out["floor_contact"]["contact"].bool() & alive
    assert contact.float().mean() > 0.8
    truth = sim.observe(update_imu=False)["wheel_odometry"][:, 0]
    moved_true = (truth - truth0)[contact]
    moved_est = (out["wheel_odometry"]["position"] - est0)[contact]
    assert torch.allclose(moved_est, moved_true, atol=5e-3)


@pytest.mark.parametrize("lanes", ["1", "2"])
def test_in_step_spine_observers_match_oracle(lanes, monkeypatch):
    """FloorContact / WheelContact / WheelOdometry inside the step kernel (one
    observer cycle per 1 ms substep, `upkie_sim_attach_observers`) against the
    oracle doing the same, both lane mappings, with autoresets: observer memory
    word by word."""
    from tests.helpers import make_pair, randomized_config

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 300
    cfg = randomized_config(B, seed=9, autoreset=True)
    cfg.fall_pitch = 0.35
    cfg.init_pos[2] = 0.58
    oracle, sim = make_pair(B, cfg=cfg)
    obs_cfg = abi.default_observer_config(B, 1e-3)
    oracle.attach_observers(obs_cfg)
    mem = sim.attach_observers(obs_cfg)
    obs_o = oracle.reset()[:, [1, 0, 4, 3]]
    sim.reset()
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    rng = np.random.default_rng(0)
    flips = 0
    for k in range(260):
        drive = 0.5 * np.sin(0.03 * k) + (2.5 if k > 150 else 0.0) * (np.arange(B) < 40)  # the first 40 envs get driven into a fall
        act = np.clip(10.0 * obs_o[:, 0] + obs_o[:, 1] + 0.1 * obs_o[:, 3], -0.99, 0.99) + drive
        obs_o, _, _, _ = oracle.step_pendulum(act)
        sim.step_pendulum(torch.from_numpy(act).float())
        # follow the oracle's trajectory: the comparison is about the observers, not about chaos
        sim.state.copy_(torch.from_numpy(oracle.state).float())
        a, b = mem.cpu().numpy().astype(np.float64), oracle.observer_state
        flags = [abi.O_WHEEL + 4, abi.O_WHEEL + 9, abi.O_CONTACT]
        mism = (a[flags] != b[flags]).any(axis=0)
        flips += int(mism.sum())
        ok = ~mism
        np.testing.assert_allclose(a[abi.O_UPPER_LEG_TORQUE, ok], b[abi.O_UPPER_LEG_TORQUE, ok], rtol=1e-3, atol=2e-3)
        np.testing.assert_allclose(a[abi.O_ODOMETRY_POSITION, ok], b[abi.O_ODOMETRY_POSITION, ok], rtol=1e-3, atol=2e-4)
        for w in (0, 1):
            np.testing.assert_allclose(a[abi.O_WHEEL + 5 * w + 2, ok], b[abi.O_WHEEL + 5 * w + 2, ok], rtol=2e-3, atol=2e-4)  # abs_torque
        if mism.any():  # a threshold decision flipped in fp32: resynchronise those envs
            mem[:, torch.from_numpy(mism)] = torch.from_numpy(b[:, mism]).float().to(mem.device)
    assert flips <= 0.01 * B * 260, flips
    assert (oracle.state[abi.S_EPISODE] > 1).sum() >= 30  # resets happened, observers restarted with them
    assert oracle.observer_state[abi.O_CONTACT].mean() > 0.5

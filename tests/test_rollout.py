"""Rollout consumer: GAE oracle against closed forms on
the CPU, the HIP kernel against the oracle on the GPU."""

import ctypes as C

import numpy as np
import pytest
import torch

from oracle.rollout_oracle import gae as oracle_gae
from upkie_amd import abi, lib


def test_oracle_gae_closed_forms():
    T, gamma = 12, 0.9
    r = np.ones((T, 1))
    v = np.zeros((T, 1))
    starts = np.zeros((T, 1))
    # lambda = 1, zero values: advantage = discounted reward-to-go (+ bootstrap)
    adv, ret = oracle_gae(r, v, starts, [2.0], [0], gamma, 1.0)
    for t in range(T):
        expected = sum(gamma**k for k in range(T - t)) + gamma ** (T - t) * 2.0
        assert adv[t, 0] == pytest.approx(expected, rel=1e-12)
    assert np.array_equal(ret, adv)
    # lambda = 0: one-step TD error
    v = np.linspace(0.0, 1.1, T)[:, None]
    adv, ret = oracle_gae(r, v, starts, [0.5], [0], gamma, 0.0)
    nxt = np.append(v[1:, 0], 0.5)
    assert np.allclose(adv[:, 0], 1.0 + gamma * nxt - v[:, 0])
    assert np.allclose(ret, adv + v)
    # an episode boundary cuts the recursion and the bootstrap
    starts[6] = 1.0
    adv2, _ = oracle_gae(r, v, starts, [0.5], [1], gamma, 0.95)
    assert adv2[5, 0] == pytest.approx(1.0 - v[5, 0])
    assert adv2[T - 1, 0] == pytest.approx(1.0 - v[T - 1, 0])


def test_gae_entry_point_fails_loudly_without_a_gpu():
    library = lib.load()
    assert library.upkie_rollout_gae(0, 4, None, None, None, None, None, 0.99, 0.95, None, None, None) == abi.ERR_INVALID_ARGUMENT
    assert library.upkie_rollout_gae(4, 4, None, None, None, None, None, 0.99, 0.95, None, None, None) == abi.ERR_INVALID_ARGUMENT
    assert b"null argument" in library.upkie_sim_last_error(None)
    if not torch.cuda.is_available():
        from upkie_amd.exceptions import UpkieRuntimeError
        from upkie_amd.rollout import compute_gae

        z = torch.zeros((3, 2))
        with pytest.raises(UpkieRuntimeError):
            compute_gae(z, z, z, torch.zeros(2), torch.zeros(2), 0.99, 0.95)


@pytest.mark.gpu
def test_gae_kernel_matches_oracle():
    from upkie_amd.rollout import compute_gae

    rng = np.random.default_rng(0)
    T, N = 64, 5000
    r = rng.standard_normal((T, N))
    v = rng.standard_normal((T, N))
    starts = (rng.uniform(size=(T, N)) < 0.05).astype(np.uint8)
    lv = rng.standard_normal(N)
    ld = (rng.uniform(size=N) < 0.1).astype(np.uint8)
    ref_adv, ref_ret = oracle_gae(r, v, starts, lv, ld, 0.99, 0.95)
    dev = "cuda:0"
    adv, ret = compute_gae(
        torch.from_numpy(r).float().to(dev), torch.from_numpy(v).float().to(dev), torch.from_numpy(starts).to(dev),
        torch.from_numpy(lv).float().to(dev), torch.from_numpy(ld).to(dev), 0.99, 0.95,
    )
    # fp32 recursion with |gamma * lambda| < 1: errors do not accumulate beyond a few ulp of the running sum
    np.testing.assert_allclose(adv.cpu().numpy(), ref_adv, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), ref_ret, rtol=1e-5, atol=2e-5)


@pytest.mark.gpu
def test_rollout_buffer_collects_a_pendulum_rollout():
    """End to end on the device: vector env -> rollout buffer -> GAE ->
    shuffled minibatches; nothing leaves HBM."""
    import upkie_amd.envs as envs
    from upkie_amd.rollout import RolloutBuffer
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    N, T = 256, 64
    env = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=N, frequency=200.0, fall_pitch=0.22, autoreset_mode="same_step",
                    init_state=RobotState(randomization=RobotStateRandomization(pitch=0.2)))
    buf = RolloutBuffer(T, N, (4,), (1,), device=env.device, gamma=0.99, gae_lambda=0.95)
    obs, _ = env.reset(seed=0)
    starts = torch.ones(N, dtype=torch.uint8, device=env.device)
    for t in range(T):
        action = torch.zeros((N, 1), device=env.device)
        value = -obs[:, 0].abs()
        next_obs, reward, terminated, truncated, info = env.step(action)
        reward = 1.0 - next_obs[:, 0].abs()  # the reference returns a constant reward; shape one here
        buf.add(obs, action, reward, starts, value, torch.zeros(N, device=env.device))
        starts = (terminated | truncated).to(torch.uint8)
        obs = next_obs
    assert buf.full
    buf.compute_returns_and_advantage(-obs[:, 0].abs(), starts)
    ref_adv, ref_ret = oracle_gae(buf.rewards.cpu().numpy(), buf.values.cpu().numpy(), buf.episode_starts.cpu().numpy(),
                                  (-obs[:, 0].abs()).cpu().numpy(), starts.cpu().numpy(), 0.99, 0.95)
    np.testing.assert_allclose(buf.advantages.cpu().numpy(), ref_adv, rtol=1e-5, atol=2e-5)
    assert buf.episode_starts[1:].any()  # some robots fell and restarted inside the rollout
    seen = 0
    for batch in buf.get(batch_size=1024):
        assert batch["observations"].shape[1:] == (4,) and batch["advantages"].is_cuda
        seen += batch["returns"].shape[0]
    assert seen == N * T
    env.close()

"""`upkie_amd.policies.LinearPolicy` (one launch for README.md:60-67's agent): argument checks without a GPU, numerics
against the torch expression on the device."""

import numpy as np
import pytest
import torch

from upkie_amd.exceptions import UpkieRuntimeError
from upkie_amd.policies import LinearPolicy


def test_linear_policy_refuses_host_tensors_and_bad_shapes():
    policy = LinearPolicy([10.0, 1.0, 0.0, 0.1], clip=0.99, device="cpu")
    assert (policy.obs_dim, policy.act_dim) == (4, 1)
    with pytest.raises(UpkieRuntimeError):  # no CPU fallback
        policy(torch.zeros(8, 4))
    with pytest.raises(ValueError):
        LinearPolicy(np.zeros((2, 3, 4)), device="cpu")
    with pytest.raises(ValueError):
        LinearPolicy([1.0], clip=0.0, device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4096, 4, 1), (333, 6, 2), (1, 30, 6)])
def test_linear_policy_matches_the_torch_expression(shape):
    n, d, a = shape
    g = torch.Generator().manual_seed(n)
    obs = torch.randn((n, d), generator=g).cuda()
    W, b = torch.randn((d, a), generator=g), torch.randn((a,), generator=g)
    policy = LinearPolicy(W, bias=b, clip=0.8)
    got = policy(obs)
    want = (obs.double() @ W.double().cuda() + b.double().cuda()).clamp(-0.8, 0.8)
    assert got.shape == (n, a) and torch.allclose(got.double(), want, atol=2e-6)
    unclamped = LinearPolicy(W)(obs)
    assert torch.allclose(unclamped.double(), obs.double() @ W.double().cuda(), atol=1e-5)
    assert policy(obs) is got  # one persistent action buffer


@pytest.mark.gpu
def test_linear_policy_drives_the_public_loop_like_the_torch_policy():
    import upkie_amd.envs as envs

    outs = []
    for fused in (False, True):
        env = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=256, frequency=200.0, autoreset_mode="next_step", seed=5)
        obs, _ = env.reset(seed=5)
        gains = torch.tensor([10.0, 1.0, 0.0, 0.1], device=env.device)
        policy = LinearPolicy(gains, clip=0.99) if fused else (lambda o: (o @ gains).clamp(-0.99, 0.99).unsqueeze(1))
        for _ in range(50):
            obs, reward, terminated, truncated, info = env.step(policy(obs))
        outs.append(obs.clone())
        env.close()
    assert torch.allclose(outs[0], outs[1], atol=1e-4)  # (the two sums associate differently: last-bit differences in the action)


@pytest.mark.gpu
def test_step_linear_policy_on_the_device_matches_step_of_the_torch_policy():
    """`UpkiePendulumVecEnv.step_linear_policy` (the policy inside the step's launch, `upkie_sim_step_pendulum_agent`)
    against `step(policy(obs))` on the device, through falls and NEXT_STEP autoresets."""
    import upkie_amd.envs as envs

    gains = [10.0, 1.0, 0.0, 0.1]
    a = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=512, frequency=200.0, autoreset_mode="next_step", seed=11, fall_pitch=0.2)
    b = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=512, frequency=200.0, autoreset_mode="next_step", seed=11, fall_pitch=0.2)
    oa, _ = a.reset(seed=11)
    ob, _ = b.reset(seed=11)
    g = torch.tensor(gains, device=a.device)
    ends = 0
    for _ in range(200):
        oa, ra, ta, ua, ia = a.step_linear_policy(gains, clip=0.99)
        ob, rb, tb, ub, ib = b.step((ob @ g).clamp(-0.99, 0.99).unsqueeze(1))
        assert torch.equal(ta, tb)
        ends += int(ta.sum())
        assert torch.allclose(oa, ob, atol=2e-4)
    assert a.step_linear_policy()[0] is oa  # the env's persistent observation buffer, as `step` returns it
    a.close()
    b.close()


@pytest.mark.gpu
def test_step_servo_policy_of_the_vector_env_is_the_library_call():
    """`UpkieServosVecEnv.step_servo_policy(policy)` returns the env's step outputs for `upkie_sim_step_servos_policy`."""
    import upkie_amd.envs as envs
    from upkie_amd import abi

    a = envs.make("Upkie-HIP-Servos-Vec", num_envs=256, frequency=200.0, autoreset_mode="next_step", seed=4)
    b = envs.make("Upkie-HIP-Servos-Vec", num_envs=256, frequency=200.0, autoreset_mode="next_step", seed=4)
    a.reset(seed=4)
    b.reset(seed=4)
    m = a.model.struct
    policy = abi.velocity_balancing_policy(float(m.wheel_radius), 1.0, float(m.left_sign))
    for _ in range(60):
        obs, reward, terminated, truncated, info = a.step_servo_policy(policy)
        want = b.sim.step_servos_policy(policy)
        assert torch.equal(obs, want[0]) and obs.shape == (256, 6, 5) and terminated.dtype == torch.bool
    assert "spine_observation" in info
    a.close()
    b.close()

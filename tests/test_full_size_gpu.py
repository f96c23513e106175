"""BASELINE.json's full batch sizes (4096, 16384, 65536 envs per GPU) through
size-independent properties (invariants, determinism, symmetry, equality of a
large batch's prefix with a small batch). The direct oracle comparisons AT
the BASELINE sizes are in tests/test_baseline_configs_gpu.py."""

import numpy as np
import pytest
import torch

from upkie_amd import abi
from upkie_amd.sim import BatchedSim

from .helpers import randomized_config

pytestmark = pytest.mark.gpu


def run_agent(sim, steps):
    sim.reset()
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    falls = torch.zeros(sim.num_envs, dtype=torch.int32, device=sim.device)
    for _ in range(steps):
        _, _, term, _ = sim.step_pendulum_agent()
        falls += term.int()
    return falls


@pytest.mark.parametrize("B", [4096, 16384, 65536])
def test_invariants_at_full_size(B):
    sim = BatchedSim(randomized_config(B, seed=1, autoreset=True))
    falls = run_agent(sim, 300)
    st = sim.state
    assert bool(torch.isfinite(st).all())
    quat_norm = st[abi.S_QUAT : abi.S_QUAT + 4].square().sum(0).sqrt()
    assert float((quat_norm - 1).abs().max()) < 1e-5  # unit quaternions
    z = st[abi.S_POS + 2]
    assert float(z.min()) > 0.55 and float(z.max()) < 0.61  # standing on the floor, ~1 mm into the tire spring
    assert bool((st[abi.S_CONTACT] == 1).all())
    assert int(falls.sum()) == 0  # nobody falls in the first 1.5 s of the README agent
    assert float(sim.obs4[:, 0].abs().max()) < 0.2
    assert float(sim.reward.abs().max()) == 0.0 and int(sim.truncated.max()) == 0
    # the episode counter is exactly one reset per env
    assert bool((st[abi.S_EPISODE] == 1).all())


@pytest.mark.parametrize("big_b,small_b", [(65536, 40000), (32768, 20000), (8192, 1000)])
def test_prefix_of_a_large_batch_equals_a_small_batch(big_b, small_b):
    """Env i does not depend on how many envs share the launch. (Batches up to
    8192 envs use the eight-lanes-per-env mapping, up to 32768 two lanes per
    env, larger ones one lane per env: bit equality holds within a mapping,
    tolerance across, see test_two_lanes_per_env_equals_one_lane_per_env.)"""
    big = BatchedSim(randomized_config(big_b, seed=9, autoreset=True))
    small = BatchedSim(randomized_config(small_b, seed=9, autoreset=True))
    run_agent(big, 100)
    run_agent(small, 100)
    assert torch.equal(big.state[:, :small_b], small.state)
    assert torch.equal(big.obs4[:small_b], small.obs4)


def test_determinism_same_seed_same_bits_and_seed_matters():
    a = BatchedSim(randomized_config(16384, seed=3, autoreset=True))
    b = BatchedSim(randomized_config(16384, seed=3, autoreset=True))
    c = BatchedSim(randomized_config(16384, seed=4, autoreset=True))
    for sim in (a, b, c):
        run_agent(sim, 60)
    assert torch.equal(a.state, b.state)
    assert not torch.equal(a.state, c.state)


def test_left_right_mirror_symmetry():
    """The robot is mirror symmetric about its sagittal plane: mirroring the
    initial roll / lateral state must mirror the trajectory (pitch, ground
    position and their rates unchanged; y, roll, yaw flipped)."""
    B = 4096
    cfg = randomized_config(B, seed=5)
    cfg.rand_roll = 0.05
    sim = BatchedSim(cfg)
    sim.reset()
    mirrored = BatchedSim(cfg)
    m = sim.state.clone()
    # reflection y -> -y: position y, velocity y flip; rotations about x and z flip
    m[abi.S_POS + 1] *= -1
    m[abi.S_LINVEL + 1] *= -1
    m[abi.S_QUAT + 1] *= -1  # qx
    m[abi.S_QUAT + 3] *= -1  # qz
    m[abi.S_ANGVEL + 0] *= -1
    m[abi.S_ANGVEL + 2] *= -1
    # left and right legs swap; the right joint axes are -y where the left are
    # +y, so the same physical (pitch-plane) rotation has the opposite joint
    # angle on the other side: swap AND negate
    for word in (abi.S_Q, abi.S_QD):
        left = m[word : word + 3].clone()
        m[word : word + 3] = -m[word + 3 : word + 6]
        m[word + 3 : word + 6] = -left
    legl = m[abi.S_LEGREF : abi.S_LEGREF + 2].clone()
    m[abi.S_LEGREF : abi.S_LEGREF + 2] = -m[abi.S_LEGREF + 2 : abi.S_LEGREF + 4]
    m[abi.S_LEGREF + 2 : abi.S_LEGREF + 4] = -legl
    mirrored.state.copy_(m)
    act = torch.linspace(-0.2, 0.2, B, device="cuda:0")
    for _ in range(20):
        o1, *_ = sim.step_pendulum(act)
        o2, *_ = mirrored.step_pendulum(act)
    # pitch, ground position and their rates are invariant under the mirror map
    # (a handful of envs sit at a slip onset where rounding order matters)
    err = (o1 - o2).abs().max(dim=1).values
    assert float((err < 2e-4).float().mean()) > 0.95
    assert float(err.max()) < 5e-2
    dy = (sim.state[abi.S_POS + 1] + mirrored.state[abi.S_POS + 1]).abs()
    dx = (sim.state[abi.S_POS] - mirrored.state[abi.S_POS]).abs()
    dz = (sim.state[abi.S_POS + 2] - mirrored.state[abi.S_POS + 2]).abs()
    for d in (dx, dy, dz):
        assert float((d < 2e-6).float().mean()) > 0.95 and float(d.max()) < 1e-3


def test_free_fall_is_exact_for_every_env():
    """Semi-implicit Euler in the air (BulletInterfaceTest.cpp:263-326) for a
    full batch: z(t_k) = z0 - g h^2 k (k + 1) / 2 up to the fp32 resolution of z."""
    from upkie_amd.model.default_model import default_model

    model = default_model()
    model.base_linear_damping = 0.0
    model.base_angular_damping = 0.0
    cfg = abi.default_sim_config(16384, seed=0)
    cfg.init_pos[2] = 5.0
    sim = BatchedSim(cfg, model)
    sim.reset()  # one substep
    act = torch.zeros(16384, device="cuda:0")
    for _ in range(4):
        sim.step_pendulum(act)  # 20 more substeps
    k = 21
    z = sim.state[abi.S_POS + 2]
    assert float((z - (5.0 - 9.81 * 1e-6 * k * (k + 1) / 2)).abs().max()) < 2e-5
    assert float((sim.state[abi.S_LINVEL + 2] + 9.81e-3 * k).abs().max()) < 1e-5
    assert float(z.max() - z.min()) == 0.0  # identical envs stay identical


@pytest.mark.parametrize("B", [4096, 32768, 65536])
def test_bench_workload_fused_launches_equal_single_launches(B):
    """bench.py's workload at full size: 2 x 32 fused steps (one launch each up
    to 32768 envs; step by step inside the library beyond) give the records and
    the state of 64 single-step launches, bit for bit, across the window in
    which the README agent's robots start falling and restarting."""
    import bench

    a, b = BatchedSim(bench.make_config(B)), BatchedSim(bench.make_config(B))
    o6 = a.reset()
    b.reset()
    prev = torch.zeros((B, 8), device=a.device)
    prev[:, :4] = o6[:, [1, 0, 4, 3]]
    # run both well into the regime with falls first (same path on both: not what is compared)
    warm = torch.zeros((32, B, 8), device=a.device)
    for _ in range(56):  # ~9 s of simulated time: the README agent's slow unstable mode has started toppling robots
        a.rollout_pendulum_records(prev, warm)
        b.rollout_pendulum_records(prev, warm.clone())
        prev = warm[31].clone()
    assert torch.equal(a.state, b.state)
    fused = torch.zeros((2, 32, B, 8), device=a.device)
    chained = torch.zeros((64, B, 8), device=a.device)
    p = prev
    for w in range(2):
        a.rollout_pendulum_records(p, fused[w])
        p = fused[w, 31]
    p = prev
    for k in range(64):
        b.step_pendulum_records(p, chained[k])
        p = chained[k]
    assert torch.equal(fused.reshape(64, B, 8), chained)
    assert torch.equal(a.state, b.state)
    assert float(a.state[abi.S_EPISODE].max()) >= 2  # episodes ended and restarted along the way

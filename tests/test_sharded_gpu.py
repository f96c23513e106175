"""`ShardedVecEnv` on the GPU: every env kind stepped through the staged
message (the kernel stores observation / reward / flags straight into the
`StepBlob` slot) equals the same handle stepped the plain way; the rank-0-policy
path and `bench.py --config c5` run through RCCL on a one-rank group (a one-GPU
box cannot host two ranks; 2 / 3 / 8 ranks: tests/test_distributed.py, gloo)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from upkie_amd import abi
from upkie_amd.distributed import ShardedVecEnv
from upkie_amd.sim import BatchedSim

from .helpers import randomized_config
from .test_distributed import _kind_policy

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", ["pendulum", "gyropod", "servos"])
def test_steps_through_the_staged_message_equal_plain_steps(kind):
    B, steps = 1000, 25
    cfg = randomized_config(B, seed=4, autoreset=True)
    cfg.fall_pitch = 0.2
    env = ShardedVecEnv(kind, cfg, "cuda:0", horizon=32, chunk=8)
    sim = BatchedSim(randomized_config(B, seed=4, autoreset=True))
    sim.config.fall_pitch = 0.2
    sim.push_config()
    policy = lambda o: _kind_policy(kind)(o.cpu()).to("cuda:0")  # noqa: E731
    obs = env.reset()
    sim.reset()
    plain = {"pendulum": sim.step_pendulum, "gyropod": sim.step_gyropod, "servos": sim.step_servos}[kind]
    for k in range(steps):
        act = policy(obs)
        obs, rew, term, trunc = env.step(act)
        o2, r2, t2, tr2 = plain(act.reshape(B, -1) if kind != "servos" else act)
        assert torch.equal(torch.nan_to_num(obs), torch.nan_to_num(o2.reshape(obs.shape))) and torch.equal(term, t2) and torch.equal(rew, r2)
    env.flush()
    got = env.records(steps - 3)
    assert got[0].shape == (1, B) + env.obs_shape and got[2].dtype == torch.uint8
    assert torch.equal(env.sim.state, sim.state)


def test_servos_with_the_law_inside_the_launch_and_base_velocity():
    B = 512
    cfg = randomized_config(B, seed=6, autoreset=True)
    law = abi.velocity_balancing_policy(0.05, 0.3, 1.0)
    env = ShardedVecEnv("servos", cfg, "cuda:0", servo_policy=law, chunk=8)
    sim = BatchedSim(randomized_config(B, seed=6, autoreset=True))
    env.reset()
    sim.reset()
    for _ in range(12):
        obs = env.step(None)[0]
        o2 = sim.step_servos_policy(law)[0]
        assert torch.equal(obs, o2)
    bv = ShardedVecEnv("base_velocity", randomized_config(B, seed=6, autoreset=True), "cuda:0", chunk=8)
    obs = bv.reset()
    act = torch.zeros((B, 2), device="cuda:0")
    act[:, 0] = 0.3
    for _ in range(30):
        obs, rew, term, trunc = bv.step(act)
    torch.cuda.synchronize()
    assert not bool(term.any()) and float(obs[:, 0].min()) > 0.04  # dead-reckoned x = 30 steps x 0.3 m/s x 5 ms
    assert float(bv.mpc.commanded_velocity.abs().max()) > 0.0  # the balancer in the launch did command something


def test_rank_zero_policy_path_through_rccl_on_one_rank():
    env = dict(os.environ, UPKIE_FORCE_PROCESS_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "tests", "sharded_root_policy_gpu.py")]
    result = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert result.returncode == 0, result.stderr[-3000:]
    out = json.loads([line for line in result.stdout.splitlines() if line.startswith("{")][-1])
    for kind in ("pendulum", "gyropod", "servos", "base_velocity"):
        assert out[kind]["bit_equal"], out
        assert out[kind]["resets"] == out[kind]["resets_plain"], out
    assert out["pendulum"]["stale_bit_equal"] and out["servos"]["stale_bit_equal"], out  # step_from_root(policy, stale=1)
    print("rank-0-policy loop, one-rank RCCL group:", out["root_policy_loop_4096_envs_one_rank"])
    assert out["root_policy_loop_4096_envs_one_rank"]["stale_1_us_per_step"] > 0


def test_bench_c5_under_torchrun_through_rccl_on_one_rank():
    env = dict(os.environ, UPKIE_FORCE_PROCESS_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29743", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "c5", "--steps", "100", "--warmup", "20", "--no-steady-state"]
    result = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert result.returncode == 0, result.stderr[-3000:]
    out = json.loads([line for line in result.stdout.splitlines() if line.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["gather"].startswith("RCCL gather") and out["value"] > 1e6
    assert out["config"]["envs_per_gpu"] == 4096 and out["roofline"]["algorithmic_bytes_per_env_step"] == 630

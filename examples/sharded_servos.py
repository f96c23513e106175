"""BASELINE.json configs[4] in miniature: UpkieServos envs sharded over the
GPUs of a node (`upkie_amd.distributed.ShardedVecEnv`), the servo-level law
evaluated inside the step's launch, every step's outputs gathered to rank 0's
rollout ring by one asynchronous RCCL collective per chunk of steps.

    python examples/sharded_servos.py                       # one GPU, no collective
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_servos.py
"""
import os

import torch

from _common import steps

from upkie_amd import abi
from upkie_amd.distributed import ShardedVecEnv, init_distributed

if __name__ == "__main__":
    rank, world, local_rank = init_distributed()
    B = 4096  # envs per GPU
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    cfg = abi.default_sim_config(B, frequency=200.0, seed=0)
    cfg.rand_pitch = 0.1
    cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
    cfg.env_id_offset = rank * B  # random streams are keyed by the global env id: results do not depend on the number of ranks
    law = abi.velocity_balancing_policy(0.05, 1.0, 1.0)  # the README balancer through the wheels' velocity loop; robots beyond 1 rad restart
    env = ShardedVecEnv("servos", cfg, device, rank=rank, world_size=world, chunk=32, horizon=128, servo_policy=law)
    env.reset()
    n = steps(256)
    for _ in range(n):
        env.step(None)  # the law runs inside the launch; this step's [B, 6, 5] observations land in the staged message
    env.flush()
    if rank == 0:
        obs, reward, terminated, truncated = env.records(n - 1)  # [world, B, 6, 5], [world, B], ...
        print(f"{world} rank(s) x {B} envs, {n} steps; last step on rank 0: observations {tuple(obs.shape)}, "
              f"mean |wheel velocity| {float(obs[:, :, [2, 5], 1].abs().mean()):.2f} rad/s; episodes restarted so far:", end=" ")
    resets = env.total_resets()
    if rank == 0:
        print(resets - world * B)
    env.shutdown()

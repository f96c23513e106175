"""Multi-GPU sharding of the env batch: one process per GPU, envs partitioned
by index, no data-path collective inside the step; one RCCL gather per step
ships the packed (obs, reward, terminated, truncated) records to rank 0
(BASELINE.json north_star).

The reference has nothing to mirror here: every PyBulletBackend is its own
world (pybullet_backend.py:100-125), so envs are independent units and the
random streams are keyed by the GLOBAL env index (results do not depend on the
number of ranks).
"""

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

RECORD_WORDS = 8  # [obs(4) | reward, terminated, truncated, 0]


def shard_range(rank: int, world_size: int, total_envs: int) -> Tuple[int, int]:
    """Global env index range [lo, hi) hosted by `rank` (contiguous blocks,
    remainder spread over the first ranks)."""
    base, rem = divmod(total_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(expected_world: Optional[int] = None, backend: Optional[str] = None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them). Backend
    "nccl" is RCCL on ROCm; "gloo" is used for CPU tests. Returns
    (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if expected_world is not None and expected_world != world:
        raise RuntimeError(
            f"--gpus {expected_world} but WORLD_SIZE={world}: launch with "
            "python -m torch.distributed.run --nproc-per-node N"
        )
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device(f"cuda:{local_rank}")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


class RecordGather:
    """Gathers fixed-size per-env records from every rank to rank 0.

    Works on any device / backend (RCCL for GPU tensors, gloo for CPU tests);
    buffers are allocated once, so the per-step cost is one collective."""

    def __init__(self, local_envs: int, rank: int, world_size: int, device, words: int = RECORD_WORDS):
        self.rank, self.world_size = rank, world_size
        self.local = torch.zeros((local_envs, words), dtype=torch.float32, device=device)
        self.gathered: Optional[torch.Tensor] = None
        self._gather_list: Optional[List[torch.Tensor]] = None
        if rank == 0:
            self.gathered = torch.zeros((world_size, local_envs, words), dtype=torch.float32, device=device)
            self._gather_list = list(self.gathered.unbind(0))

    def gather(self) -> Optional[torch.Tensor]:
        """Rank 0 returns ``[world, local_envs, words]`` (rank-major = global
        env order), other ranks return None."""
        if self.world_size == 1:
            self.gathered[0].copy_(self.local)
            return self.gathered
        dist.gather(self.local, self._gather_list, dst=0)
        return self.gathered


class ShardedPendulum:
    """This rank's shard of a batch of Upkie-Pendulum envs plus the per-step
    gather of records to rank 0."""

    def __init__(self, config, device: str, rank: int = 0, world_size: int = 1, model=None):
        from .sim import BatchedSim

        self.rank, self.world_size = rank, world_size
        self.sim = BatchedSim(config, model, device=device)
        self.records = RecordGather(self.sim.num_envs, rank, world_size, self.sim.device)
        self._device = self.sim.device

    def reset(self) -> None:
        obs6 = self.sim.reset()
        self.records.local.zero_()
        self.records.local[:, :4] = obs6[:, [1, 0, 4, 3]]  # upkie_pendulum.py:17

    def step_agent(self) -> Optional[torch.Tensor]:
        """One env.step() of every local env with the on-device linear agent,
        then (world_size > 1) the gather to rank 0."""
        self.sim.step_pendulum_packed(self.records.local)
        if self.world_size > 1:
            return self.records.gather()
        return self.records.local

    def step(self, act: torch.Tensor) -> Optional[torch.Tensor]:
        self.sim.step_pendulum_packed(self.records.local, act)
        if self.world_size > 1:
            return self.records.gather()
        return self.records.local

    def barrier(self) -> None:
        if self.world_size > 1:
            dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.world_size == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self._device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def total_resets(self) -> int:
        """Episodes started so far, summed over ranks."""
        from . import abi

        n = self.sim.state[abi.S_EPISODE].sum().to(torch.float64).reshape(1)
        if self.world_size > 1:
            dist.all_reduce(n, op=dist.ReduceOp.SUM)
        return int(n.item())

    def shutdown(self) -> None:
        self.sim.close()
        if self.world_size > 1 and dist.is_initialized():
            dist.destroy_process_group()

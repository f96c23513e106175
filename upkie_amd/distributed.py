"""Multi-GPU sharding of the env batch: one process per GPU, envs partitioned
by index, no data-path collective inside the step; one RCCL gather per step
ships the packed (obs, reward, terminated, truncated) records to rank 0
(BASELINE.json north_star).

The reference has nothing to mirror here: every PyBulletBackend is its own
world (pybullet_backend.py:100-125), so envs are independent units and the
random streams are keyed by the GLOBAL env index (results do not depend on the
number of ranks).

xGMI is point-to-point and the message is tiny (32 B/env: 128 KB per rank at
4096 envs), so the gather is latency bound, not bandwidth bound. It is
therefore taken off the critical path: records are double-buffered, the
gather of step t is issued asynchronously and overlaps the kernel of step
t + 1, and rank 0 receives straight into the slot of a rollout ring buffer
``[T, world, B, 8]`` (the "PPO rollout consumer" of BASELINE.json configs[3]),
so there is no extra copy on the consumer side either.
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


class RolloutGather:
    """Double-buffered per-env records on every rank + a rollout ring buffer on
    rank 0 that the gathers write into directly.

    Usage per step ``t`` on every rank::

        out = g.begin_step()          # buffer this step's kernel writes
        prev = g.previous             # buffer holding step t-1 (agent input)
        ... launch the step writing `out` ...
        g.end_step()                  # async gather of `out` to rank 0

    `begin_step` first waits (stream-level) for the gather that last used the
    buffer about to be overwritten, i.e. the one issued two steps earlier: the
    gather of step t-1 is still free to run during step t's kernel.
    Works on any device / backend (RCCL for GPU tensors, gloo for CPU tests).
    """

    def __init__(self, local_envs: int, rank: int, world_size: int, device, horizon: int = 128, words: int = RECORD_WORDS):
        self.rank, self.world_size, self.horizon = rank, world_size, horizon
        self.buffers = [torch.zeros((local_envs, words), dtype=torch.float32, device=device) for _ in range(2)]
        self._work = [None, None]
        self._step = 0  # index of the step being produced
        self.rollout: Optional[torch.Tensor] = None
        if rank == 0:
            # [T, world, B, words]: slot t % T holds step t of every env (rank-major = global env order)
            self.rollout = torch.zeros((horizon, world_size, local_envs, words), dtype=torch.float32, device=device)

    @property
    def current(self) -> torch.Tensor:
        if self.world_size == 1:  # single rank: produce straight into the ring slot
            return self.rollout[self._step % self.horizon, 0]
        return self.buffers[self._step % 2]

    @property
    def previous(self) -> torch.Tensor:
        if self.world_size == 1:
            return self.rollout[(self._step - 1) % self.horizon, 0]
        return self.buffers[(self._step + 1) % 2]

    def begin_step(self) -> torch.Tensor:
        work = self._work[self._step % 2]
        if work is not None:
            work.wait()  # the gather issued two steps ago has read this buffer
            self._work[self._step % 2] = None
        return self.current

    def end_step(self) -> None:
        slot = self._step % self.horizon
        out = self.current
        if self.world_size > 1:
            gather_list: Optional[List[torch.Tensor]] = None
            if self.rank == 0:
                gather_list = list(self.rollout[slot].unbind(0))
            self._work[self._step % 2] = dist.gather(out, gather_list, dst=0, async_op=True)
        self._step += 1

    def flush(self) -> None:
        """Wait for every gather in flight (end of a rollout / of the bench)."""
        for i, work in enumerate(self._work):
            if work is not None:
                work.wait()
                self._work[i] = None

    def last(self, steps_back: int = 0) -> Optional[torch.Tensor]:
        """Rank 0: records ``[world, B, words]`` of the step issued
        `steps_back` steps before the latest one (after `flush`)."""
        if self.rollout is None:
            return None
        return self.rollout[(self._step - 1 - steps_back) % self.horizon]


class ShardedPendulum:
    """This rank's shard of a batch of Upkie-Pendulum envs plus the pipelined
    gather of per-step records into rank 0's rollout buffer."""

    def __init__(self, config, device: str, rank: int = 0, world_size: int = 1, model=None, horizon: int = 128):
        from .sim import BatchedSim

        self.rank, self.world_size = rank, world_size
        self.sim = BatchedSim(config, model, device=device)
        self.gather = RolloutGather(self.sim.num_envs, rank, world_size, self.sim.device, horizon=horizon)
        self._device = self.sim.device

    def reset(self) -> None:
        obs6 = self.sim.reset()
        self.gather.flush()
        for buf in self.gather.buffers:
            buf.zero_()
        self.gather.previous.zero_()
        # the agent's first input: the reset observation, upkie_pendulum.py:17
        self.gather.previous[:, :4] = obs6[:, [1, 0, 4, 3]]

    def step_agent(self) -> None:
        """One env.step() of every local env with the on-device linear agent;
        the records of this step travel to rank 0 while the next step runs."""
        out = self.gather.begin_step()
        self.sim.step_pendulum_records(self.gather.previous, out)
        self.gather.end_step()

    def step(self, act: torch.Tensor) -> None:
        out = self.gather.begin_step()
        self.sim.step_pendulum_packed(out, act)
        self.gather.end_step()

    def flush(self) -> None:
        self.gather.flush()

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
        self.gather.flush()
        self.sim.close()
        if self.world_size > 1 and dist.is_initialized():
            dist.destroy_process_group()

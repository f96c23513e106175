"""Multi-GPU sharding of the env batch: one process per GPU, envs partitioned
by index, no data-path collective inside the step; one RCCL gather per step
ships the packed (obs, reward, terminated, truncated) records to rank 0
(BASELINE.json north_star).

The reference has nothing to mirror here: every PyBulletBackend is its own
world (pybullet_backend.py:100-125), so envs are independent units and the
random streams are keyed by the GLOBAL env index (results do not depend on the
number of ranks).

xGMI is point-to-point and the message is tiny (32 B/env: 128 KB per rank and
step at 4096 envs), so the gather is latency bound, not bandwidth bound. It is
therefore taken off the critical path: records are staged in double-buffered
chunks of `chunk` consecutive steps, a full chunk is gathered in one
asynchronous collective that overlaps the kernels of the next chunk, and rank
0 receives straight into a rollout ring buffer ``[chunks, world, chunk, B, 8]``
(the "PPO rollout consumer" of BASELINE.json configs[3]), so there is no extra
copy on the consumer side either.
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
    "nccl" is RCCL on ROCm; "gloo" is used for CPU tests. A single process
    joins no group unless UPKIE_FORCE_PROCESS_GROUP=1 (a one-rank RCCL group:
    how the collective path is exercised on a one-GPU box). Returns
    (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if expected_world is not None and expected_world != world:
        raise RuntimeError(
            f"--gpus {expected_world} but WORLD_SIZE={world}: launch with "
            "python -m torch.distributed.run --nproc-per-node N"
        )
    if (world > 1 or os.environ.get("UPKIE_FORCE_PROCESS_GROUP") == "1") and not dist.is_initialized():
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
    """Per-env records of every step on each rank + a rollout ring buffer on
    rank 0 that the gathers write into directly.

    Usage per step ``t`` on every rank::

        out = g.begin_step()          # buffer this step's kernel writes
        prev = g.previous             # buffer holding step t-1 (agent input)
        ... launch the step writing `out` ...
        g.end_step()                  # every `chunk` steps: async gather to rank 0

    Records are staged in chunks of `chunk` consecutive steps (two chunks,
    double-buffered); a full chunk ``[chunk, B, words]`` travels in ONE
    collective, issued asynchronously, that overlaps the kernels of the next
    chunk. xGMI moves 128 KB in about a microsecond but a collective costs tens
    of microseconds of launch latency, i.e. more than one env step: shipping
    `chunk` steps per message takes the gather off the critical path whatever
    the number of ranks ("gather per T-step rollout chunk
    when the consumer allows"). `chunk=1` gathers every step.
    Works on any device / backend (RCCL for GPU tensors, gloo for CPU tests).
    """

    def __init__(self, local_envs: int, rank: int, world_size: int, device, horizon: int = 128, words: int = RECORD_WORDS, chunk: int = 8,
                 collectives: Optional[bool] = None):
        self.rank, self.world_size = rank, world_size
        # staged + gathered (several ranks) or produced straight into the ring (one rank);
        # `collectives=True` forces the gather path on a one-rank group (RCCL smoke test on one GPU)
        self.collectives = world_size > 1 if collectives is None else bool(collectives)
        self.chunk = max(1, int(chunk))
        self.num_chunks = max(2, -(-int(horizon) // self.chunk))  # ring of chunks on rank 0
        self.horizon = self.num_chunks * self.chunk
        K = self.chunk
        f32 = dict(dtype=torch.float32, device=device)
        self.staging = torch.zeros((2, K, local_envs, words), **f32) if self.collectives else None
        self._work = [None, None]
        self._work_chunk = [None, None]  # which chunk of the rollout each gather in flight fills
        # rank 0: `consumer(chunk_index)` is called, in step order, once chunk `chunk_index % num_chunks` of `rollout` holds
        # every rank's records of those `chunk` steps (stream-ordered behind the gather that delivered them): the rollout
        # consumer of BASELINE configs[3] (bench.py --config c4 hangs generalized advantage estimation here)
        self.consumer = None
        self._step = 0  # index of the step being produced
        self._flushed_to = 0  # steps of the current chunk that a flush has already shipped
        self.rollout: Optional[torch.Tensor] = None
        if rank == 0:
            # [chunks, world, K, B, words]: step t of rank r at [t // K % chunks, r, t % K]
            self.rollout = torch.zeros((self.num_chunks, world_size, K, local_envs, words), **f32)

    def _slot(self, step: int) -> torch.Tensor:
        # one view per slot, made once: indexing a tensor costs microseconds of host time, a third of a 15 us env step
        views = self.__dict__.get("_views")
        if views is None:
            K = self.chunk
            if not self.collectives:  # single rank: produce straight into the ring
                views = [self.rollout[c, 0, k] for c in range(self.num_chunks) for k in range(K)]
            else:
                views = [self.staging[c, k] for c in range(2) for k in range(K)]
            self._views = views
            self._view_ptrs = [v.data_ptr() for v in views]
        return views[step % len(views)]

    def slot_pointers(self):
        """(device address of the previous step's records, of the current step's):
        what a step launch needs of `previous` / `begin_step()`, as plain integers."""
        if self.__dict__.get("_views") is None:
            self._slot(0)
        ptrs = self._view_ptrs
        n = len(ptrs)
        return ptrs[(self._step - 1) % n], ptrs[self._step % n]

    @property
    def current(self) -> torch.Tensor:
        return self._slot(self._step)

    @property
    def previous(self) -> torch.Tensor:
        return self._slot(self._step - 1)

    def begin_step(self) -> torch.Tensor:
        if self.collectives and self._step % self.chunk == 0:
            c = (self._step // self.chunk) % 2
            self._landed(c)  # the gather issued two chunks ago has read this buffer
        return self.current

    def begin_steps(self, n: int) -> torch.Tensor:
        """Buffer ``[n, B, words]`` the next `n` steps write into (they must
        fit into the current chunk: ``step % chunk + n <= chunk``)."""
        k = self._step % self.chunk
        if k + n > self.chunk:
            raise ValueError(f"{n} steps from position {k} cross the boundary of a {self.chunk}-step chunk")
        first = self.begin_step()
        if self.collectives:
            block = self.staging[(self._step // self.chunk) % 2]
        else:
            block = self.rollout[(self._step // self.chunk) % self.num_chunks, 0]
        assert block[k].data_ptr() == first.data_ptr()
        return block[k : k + n]

    def end_steps(self, n: int) -> None:
        self._step += n - 1
        self.end_step()

    def _gather_chunk(self, chunk_index: int, first: int = 0, stop: Optional[int] = None) -> None:
        """Steps ``first .. stop - 1`` of a chunk (default: all of it) in one collective: the step axis is the
        slowest one of a chunk on both sides, so a range of steps is one contiguous block per rank."""
        c = chunk_index % 2
        stop = self.chunk if stop is None else stop
        gather_list: Optional[List[torch.Tensor]] = None
        if self.rank == 0:
            gather_list = [block[first:stop] for block in self.rollout[chunk_index % self.num_chunks].unbind(0)]
        self._landed(c)
        self._work[c] = dist.gather(self.staging[c, first:stop], gather_list, dst=0, async_op=True)
        self._work_chunk[c] = chunk_index if stop == self.chunk else None  # (a flush's partial chunk is not handed to the consumer)

    def _landed(self, c: int) -> None:
        """Wait (on the stream) for the gather in flight on staging buffer `c`; rank 0 hands the chunk it completed to the consumer."""
        work = self._work[c]
        if work is None:
            return
        work.wait()
        self._work[c] = None
        chunk_index, self._work_chunk[c] = self._work_chunk[c], None
        if chunk_index is not None and self.consumer is not None and self.rank == 0:
            self.consumer(chunk_index)

    def end_step(self) -> None:
        step = self._step
        self._step += 1
        if step % self.chunk == self.chunk - 1:
            if self.collectives:
                self._gather_chunk(step // self.chunk, self._flushed_to)  # (what a flush has shipped is on rank 0 already)
                self._flushed_to = 0
            elif self.consumer is not None:
                self.consumer(step // self.chunk)  # one rank: the step kernels wrote the chunk straight into the ring

    def flush(self) -> None:
        """Ship a partially filled chunk and wait for every gather in flight
        (end of a rollout / of the bench). Every rank must call it at the same
        step."""
        filled = self._step % self.chunk
        if self.collectives and filled > self._flushed_to:
            # only the steps produced since the last flush (a flush in the middle of a 64-step chunk used to ship all 64
            # steps' worth, 8 MB per rank at 4096 envs, with nothing left to overlap it with -- and the chunk again, in
            # full, when it completed)
            self._gather_chunk(self._step // self.chunk, self._flushed_to, filled)
            self._flushed_to = filled
        first = 0 if (self._work_chunk[0] or 0) <= (self._work_chunk[1] or 0) else 1  # (in step order)
        for c in (first, 1 - first):
            self._landed(c)

    def records(self, step: int) -> Optional[torch.Tensor]:
        """Rank 0: records ``[world, B, words]`` of absolute step `step` (one of
        the last `horizon` steps, after `flush`)."""
        if self.rollout is None:
            return None
        return self.rollout[(step // self.chunk) % self.num_chunks, :, step % self.chunk]

    def last(self, steps_back: int = 0) -> Optional[torch.Tensor]:
        """Rank 0: records of the step issued `steps_back` steps before the
        latest one (after `flush`)."""
        return self.records(self._step - 1 - steps_back)


class ShardedPendulum:
    """This rank's shard of a batch of Upkie-Pendulum envs plus the pipelined
    gather of per-step records into rank 0's rollout buffer."""

    def __init__(self, config, device: str, rank: int = 0, world_size: int = 1, model=None, horizon: int = 128, chunk: int = 8, sim_factory=None,
                 collectives: Optional[bool] = None):
        from .sim import BatchedSim

        self.rank, self.world_size = rank, world_size
        # (sim_factory: test doubles only; the product always builds a BatchedSim)
        self.sim = sim_factory(config, model, device) if sim_factory is not None else BatchedSim(config, model, device=device)
        self.gather = RolloutGather(self.sim.num_envs, rank, world_size, self.sim.device, horizon=horizon, chunk=chunk, collectives=collectives)
        self._collectives = self.gather.collectives
        self._device = self.sim.device

    def reset(self) -> None:
        obs6 = self.sim.reset()
        self.gather.flush()
        if self.gather.staging is not None:
            self.gather.staging.zero_()
        self.gather.previous.zero_()
        # the agent's first input: the reset observation, upkie_pendulum.py:17
        self.gather.previous[:, :4] = obs6[:, [1, 0, 4, 3]]

    @property
    def lanes_per_env(self) -> int:
        return int(getattr(self.sim, "lanes_per_env", 1))

    @property
    def fused_rollouts(self) -> bool:
        """True when `rollout_agent(n)` is ONE launch (the multi-lane mappings of small and medium batches)."""
        return self.lanes_per_env > 1

    @property
    def kernel_name(self) -> str:
        lanes = self.lanes_per_env
        return {1: "step_kernel<MODE_PENDULUM_AGENT> (one env per lane)", 2: "step_kernel_pair<MODE_PENDULUM_AGENT / _ROLLOUT> (two lanes per env)",
                8: "step_kernel_octet<MODE_PENDULUM_AGENT / _ROLLOUT> (eight lanes per env: one quad per leg, one lane per body)"}.get(
            lanes, f"step kernel, {lanes} lanes per env")

    def step_agent(self) -> None:
        """One env.step() of every local env with the on-device linear agent;
        the records of this step travel to rank 0 while the next step runs."""
        g = self.gather
        raw = getattr(self.sim, "step_pendulum_records_raw", None)
        if raw is None:  # (test doubles)
            out = g.begin_step()
            self.sim.step_pendulum_records(g.previous, out)
        else:
            g.begin_step()
            raw(*g.slot_pointers())
        g.end_step()

    def rollout_agent(self, n: int) -> None:
        """`n` env.step() with the on-device agent in one launch (two lanes per
        env; `n` launches beyond 32768 envs): same records as `n` calls of
        `step_agent`. The steps must stay inside one gather chunk."""
        prev = self.gather.previous
        out = self.gather.begin_steps(n)
        self.sim.rollout_pendulum_records(prev, out)
        self.gather.end_steps(n)

    def step(self, act: torch.Tensor) -> None:
        out = self.gather.begin_step()
        self.sim.step_pendulum_packed(out, act)
        self.gather.end_step()

    def flush(self) -> None:
        self.gather.flush()

    def barrier(self) -> None:
        if self._collectives:
            dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self._collectives:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self._device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def total_resets(self) -> int:
        """Episodes started so far, summed over ranks."""
        from . import abi

        n = self.sim.state[abi.S_EPISODE].sum().to(torch.float64).reshape(1)
        if self._collectives:
            dist.all_reduce(n, op=dist.ReduceOp.SUM)
        return int(n.item())

    def shutdown(self, destroy_group: bool = True) -> None:
        """Close the handle; leave the process group (`destroy_group=False`: another sharded env of this process follows)."""
        self.gather.flush()
        self.sim.close()
        if destroy_group and self._collectives and dist.is_initialized():
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# Every env kind, sharded (BASELINE.json configs[4]: UpkieServos on 8 GPUs)

ENV_KINDS = {
    # kind: (observation shape, action shape) per env -- the layouts of upkie_amd.envs.vec_env
    "pendulum": ((4,), (1,)),
    "gyropod": ((6,), (2,)),
    "servos": ((6, 5), (6, 6)),
    "base_velocity": ((3,), (2,)),
}


def _prod(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


class StepBlob:
    """Layout of ONE step's outputs of a rank as one contiguous message: the
    four buffers a step call writes -- ``obs [B, *obs_shape]`` f32, ``reward
    [B]`` f32, ``terminated [B]`` u8, ``truncated [B]`` u8 -- back to back,
    each region 16-byte aligned, the whole rounded up to 16 bytes. The step
    kernel is handed the four region addresses of a slot of the staging buffer
    and stores straight into the message the collective ships: no packing
    kernel, no copy, whatever the env kind (the Pendulum's 8-word records of
    `ShardedPendulum` are the special case where a record fits two stores)."""

    def __init__(self, num_envs: int, obs_shape):
        B, d = int(num_envs), _prod(obs_shape)
        up = lambda n: (n + 15) // 16 * 16  # noqa: E731
        self.num_envs, self.obs_shape = B, tuple(int(x) for x in obs_shape)
        self.obs_offset = 0
        self.reward_offset = up(4 * B * d)
        self.terminated_offset = self.reward_offset + up(4 * B)
        self.truncated_offset = self.terminated_offset + up(B)
        self.nbytes = self.truncated_offset + up(B)
        self.words = self.nbytes // 4

    def views(self, blob: torch.Tensor):
        """``(obs, reward, terminated, truncated)`` views of a blob ``[..., words]`` f32
        (leading dimensions kept: a rank-0 ring slice ``[world, words]`` decodes to ``[world, B, ...]``)."""
        B, lead = self.num_envs, tuple(blob.shape[:-1])
        raw = blob.view(torch.uint8)
        obs = blob[..., : B * _prod(self.obs_shape)].view(*lead, B, *self.obs_shape)
        reward = blob[..., self.reward_offset // 4 : self.reward_offset // 4 + B]
        terminated = raw[..., self.terminated_offset : self.terminated_offset + B]
        truncated = raw[..., self.truncated_offset : self.truncated_offset + B]
        return obs, reward, terminated, truncated

    def addresses(self, blob: torch.Tensor):
        base = blob.data_ptr()
        return base + self.obs_offset, base + self.reward_offset, base + self.terminated_offset, base + self.truncated_offset


class ShardedVecEnv:
    """This rank's shard of a batch of envs of ANY kind ("pendulum" |
    "gyropod" | "servos" | "base_velocity") with the same pipelined gather of
    per-step outputs into rank 0's rollout ring as `ShardedPendulum`, and the
    reverse path for a policy that lives on rank 0.

    Envs are sharded by index (``config.env_id_offset`` = first global env id
    of the shard: random streams are keyed by the global id, so results do not
    depend on the number of ranks); a step has no data-path collective.

    * Per-rank policies (every rank holds a policy replica -- data-parallel
      learners -- or the servo-level law runs inside the launch,
      `servo_policy=`): ``obs, reward, terminated, truncated = env.step(actions)``
      with this rank's actions; outputs are staged in chunks of `chunk` steps
      and ONE asynchronous gather per chunk ships them to rank 0's ring
      ``[chunks, world, chunk, blob]`` (`records(step)` decodes a step).
    * A policy on rank 0: ``env.step_from_root(policy)`` = gather of the
      latest observations to rank 0 -> ``policy(obs [world * B, ...])`` there
      -> scatter of the actions -> step. Both collectives are issued
      asynchronously into double-buffered tensors and waited for on the
      stream (RCCL) right where the data is consumed, so the host runs one
      step ahead of the device; they ARE on the step's critical path (two
      latency-bound collectives per step, tens of microseconds each over
      xGMI): the price of a central policy, stated in DESIGN.md section 7.
    """

    def __init__(self, kind: str, config, device: str, rank: int = 0, world_size: int = 1, model=None, horizon: int = 128, chunk: int = 8,
                 sim_factory=None, collectives: Optional[bool] = None, servo_policy=None, mpc_config=None, mpc_factory=None, sim=None):
        from . import abi
        from .sim import BatchedSim

        if kind not in ENV_KINDS:
            raise ValueError(f"unknown env kind '{kind}' (one of {sorted(ENV_KINDS)})")
        self.kind, self.rank, self.world_size = kind, rank, world_size
        self.obs_shape, self.act_shape = ENV_KINDS[kind]
        # `sim`: an existing handle (e.g. the `.sim` of a vector env built with this rank's `env_id_offset`, its
        # inertia randomisation and joint properties already applied) instead of a new one from `config`
        # (sim_factory / mpc_factory: test doubles only; the product always builds a BatchedSim / BatchedMpc)
        self._owns_sim = sim is None  # (a caller's handle stays the caller's to close: `shutdown` leaves it open)
        self._stale_primed, self._stale_scatter = False, None  # `step_from_root(policy, stale=1)`
        if sim is not None:
            self.sim = sim
        else:
            self.sim = sim_factory(config, model, device) if sim_factory is not None else BatchedSim(config, model, device=device)
        B = self.num_envs = self.sim.num_envs
        self._device = self.sim.device
        self.blob = StepBlob(B, self.obs_shape)
        self.gather = RolloutGather(1, rank, world_size, self._device, horizon=horizon, words=self.blob.words, chunk=chunk, collectives=collectives)
        self._collectives = self.gather.collectives
        f32 = dict(dtype=torch.float32, device=self._device)
        self.actions = torch.zeros((2, B) + self.act_shape, **f32)  # double-buffered: scatter target of step t + 1 while step t reads the other
        self._action_slot = 0
        self._root_obs = torch.zeros((2, world_size, B) + self.obs_shape, **f32) if rank == 0 else None
        self._works = []
        extra = {}
        step_kind = kind
        if kind == "servos" and servo_policy is not None:
            step_kind, extra = "servos_policy", {"policy": servo_policy}
        self.mpc = None
        if kind == "base_velocity":
            if mpc_config is None:
                mpc_config = abi.default_mpc_config(B, 16)
            if mpc_factory is None:
                from .mpc import BatchedMpc

                mpc_factory = BatchedMpc
            self.mpc = mpc_factory(mpc_config, device=str(self._device))
            self._mpc_x0 = torch.zeros((B, 4), **f32)
            self._mpc_contact = torch.zeros(B, dtype=torch.uint8, device=self._device)
            extra = {"mpc": self.mpc, "mpc_x0": self._mpc_x0, "mpc_contact": self._mpc_contact}
        self._servo_policy = servo_policy
        self._step_fn = self.sim.step_into_fn(step_kind, **extra)
        self._slot_cache = {}
        self.last = None  # (obs, reward, terminated, truncated) views of the latest step, this rank

    # ------------------------------------------------------------------ steps
    def _slot(self):
        """Views and addresses of the slot the step being produced writes into."""
        g = self.gather
        blob = g.begin_step()[0]
        key = blob.data_ptr()
        hit = self._slot_cache.get(key)
        if hit is None:
            hit = self._slot_cache[key] = (self.blob.views(blob), self.blob.addresses(blob))
        return hit

    def reset(self) -> torch.Tensor:
        """Reset every local env; returns this rank's first observation ``[B, *obs_shape]``."""
        from . import abi

        if self._stale_scatter is not None and self._stale_scatter[0] is not None:
            self._stale_scatter[0].wait()  # (a lagged exchange still in flight: its actions die with the episode)
        self._stale_primed, self._stale_scatter = False, None
        obs6 = self.sim.reset()
        self.gather.flush()
        if self.kind == "pendulum":
            obs = obs6[:, [1, 0, 4, 3]].contiguous()  # upkie_pendulum.py:17
        elif self.kind == "gyropod":
            obs = obs6.clone()
        elif self.kind == "servos":
            st = self.sim.state
            obs = torch.empty((self.num_envs, 6, 5), dtype=torch.float32, device=self._device)
            obs[:, :, 0] = st[abi.S_Q : abi.S_Q + 6].t()
            obs[:, :, 1] = st[abi.S_QD : abi.S_QD + 6].t()
            obs[:, :, 2] = st[abi.S_TORQUE : abi.S_TORQUE + 6].t()
            obs[:, :, 3] = 42.0
            obs[:, :, 4] = 18.0
        else:  # base_velocity: the dead-reckoned pose restarts at the origin (upkie_base_velocity.py:160-162)
            self._mpc_x0.copy_(obs6[:, [0, 1, 3, 4]])
            self._mpc_contact.copy_((self.sim.state[abi.S_CONTACT] != 0).to(torch.uint8))
            self.mpc.reset(None)
            obs = torch.cat([torch.zeros((self.num_envs, 2), dtype=torch.float32, device=self._device), obs6[:, 2:3]], dim=1)
        self._reset_obs = obs
        self.last = (obs, None, None, None)
        return obs

    def step(self, actions: Optional[torch.Tensor] = None):
        """One env.step() of every local env. `actions`: this rank's actions
        ``[B, *act_shape]`` (fp32, contiguous, on the device); None = the buffer
        the last `scatter_actions` filled (or nothing at all when the
        servo-level policy runs inside the launch). Returns views of this step's
        slot of the staging buffer: valid until the ring wraps
        (``2 * chunk`` steps with collectives, `horizon` steps without)."""
        if actions is None:
            actions = self.actions[self._action_slot]
        elif not (actions.dtype is torch.float32 and actions.device == self._device and actions.is_contiguous()
                  and actions.numel() == self.num_envs * _prod(self.act_shape)):
            actions = actions.to(self._device, torch.float32).reshape((self.num_envs,) + self.act_shape).contiguous()
        views, addresses = self._slot()
        self._step_fn(actions.data_ptr(), *addresses)
        self.gather.end_step()
        self.last = views
        return views

    # -------------------------------------------------- a policy on rank 0
    def gather_observations(self) -> Optional[torch.Tensor]:
        """Collective: the latest observation of every rank to rank 0,
        ``[world * B, *obs_shape]`` there (a persistent double buffer), None elsewhere."""
        obs = self.last[0]
        if not self._collectives:
            return obs.reshape((self.num_envs,) + self.obs_shape)
        slot = self._action_slot
        out = None
        if self.rank == 0:
            out = list(self._root_obs[slot].unbind(0))
        work = dist.gather(obs.contiguous(), out, dst=0, async_op=True)
        work.wait()  # (RCCL: a stream dependency, the host does not block)
        return self._root_obs[slot].reshape((self.world_size * self.num_envs,) + self.obs_shape) if self.rank == 0 else None

    def scatter_actions(self, actions: Optional[torch.Tensor]) -> torch.Tensor:
        """Collective: rank 0 hands over the actions of ALL envs ``[world * B, *act_shape]``
        (None elsewhere); every rank receives its block into the action buffer the next `step()` reads."""
        self._action_slot ^= 1
        mine = self.actions[self._action_slot]
        if not self._collectives:
            mine.copy_(actions.reshape(mine.shape))
            return mine
        chunks = None
        if self.rank == 0:
            chunks = list(actions.to(torch.float32).reshape((self.world_size, self.num_envs) + self.act_shape).contiguous().unbind(0))
        work = dist.scatter(mine, chunks, src=0, async_op=True)
        work.wait()
        return mine

    def step_from_root(self, policy, stale: int = 0):
        """``obs -> policy (rank 0) -> actions -> step`` with the policy on rank 0
        only: `policy` maps ``[world * B, *obs_shape]`` to ``[world * B, *act_shape]``
        there and is not called on the other ranks.

        ``stale=0``: step k + 1 acts on ``policy(o_k)`` -- gather, policy and
        scatter all sit between two steps (two latency-bound collectives on every
        step's critical path).
        ``stale=1`` (round 5): step k + 1 acts on ``policy(o_{k-1})``, the
        observation of ONE STEP EARLIER (the first two steps both on
        ``policy(o_0)``): the gather of ``o_k`` is issued in front of step
        k + 1's launch and collected behind it, so it overlaps the step; rank 0
        then evaluates the policy and issues the scatter whose actions step
        k + 2 waits for. What is left between two steps is the policy and one
        scatter; an on-policy learner must know that its actions are one step
        late (asynchronous / "lagged" actors accept exactly this)."""
        if not stale:
            if self._stale_scatter is not None:  # (a lagged exchange left in flight by an earlier stale=1 call: finished and dropped)
                if self._stale_scatter[0] is not None:
                    self._stale_scatter[0].wait()
                self._stale_primed, self._stale_scatter = False, None
            obs_all = self.gather_observations()
            self.scatter_actions(policy(obs_all) if self.rank == 0 else None)
            return self.step(None)
        if stale != 1:
            raise ValueError("stale must be 0 or 1")
        if not self._stale_primed:
            # the first step: nothing older than the reset observation exists; its actions also serve the second step
            obs_all = self.gather_observations()
            self.scatter_actions(policy(obs_all) if self.rank == 0 else None)
            self._stale_primed = True
            self._stale_scatter = None
            return self.step(None)
        # the scatter issued behind the previous step delivers this step's actions, policy(o_{k-1})
        if self._stale_scatter is not None:
            work, ready_slot = self._stale_scatter
            if work is not None:
                work.wait()
            self._action_slot = ready_slot
            self._stale_scatter = None
        # gather of the latest observation o_k: issued now, collected behind the step
        obs = self.last[0]
        slot = self._action_slot ^ 1  # (the exchange in flight uses the OTHER half of both double buffers: this step reads `_action_slot`'s)
        gather_work = None
        if self._collectives:
            out = list(self._root_obs[slot].unbind(0)) if self.rank == 0 else None
            gather_work = dist.gather(obs.contiguous(), out, dst=0, async_op=True)
        views = self.step(None)
        # rank 0: policy on o_k (which the gather has delivered meanwhile), scatter for the step after the next call's
        if gather_work is not None:
            gather_work.wait()
        obs_all = None
        if self.rank == 0:
            obs_all = (self._root_obs[slot].reshape((self.world_size * self.num_envs,) + self.obs_shape) if self._collectives
                       else obs.reshape((self.num_envs,) + self.obs_shape))
        actions = policy(obs_all) if self.rank == 0 else None
        mine = self.actions[slot]
        if not self._collectives:
            mine.copy_(actions.reshape(mine.shape))
            self._stale_scatter = (None, slot)
        else:
            chunks = None
            if self.rank == 0:
                chunks = list(actions.to(torch.float32).reshape((self.world_size, self.num_envs) + self.act_shape).contiguous().unbind(0))
            self._stale_scatter = (dist.scatter(mine, chunks, src=0, async_op=True), slot)
        return views

    # ------------------------------------------------------------ rank 0 reads
    def records(self, step: int):
        """Rank 0: ``(obs [world, B, ...], reward [world, B], terminated, truncated)`` of absolute
        step `step` (one of the last `horizon` steps, after `flush`); None elsewhere."""
        blob = self.gather.records(step)
        if blob is None:
            return None
        return self.blob.views(blob[:, 0])

    def flush(self) -> None:
        self.gather.flush()

    def barrier(self) -> None:
        if self._collectives:
            dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self._collectives:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self._device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def total_resets(self) -> int:
        from . import abi

        n = self.sim.state[abi.S_EPISODE].sum().to(torch.float64).reshape(1)
        if self._collectives:
            dist.all_reduce(n, op=dist.ReduceOp.SUM)
        return int(n.item())

    @property
    def lanes_per_env(self) -> int:
        return int(getattr(self.sim, "lanes_per_env", 1))

    def shutdown(self, destroy_group: bool = True) -> None:
        self.gather.flush()
        if self.mpc is not None:
            self.mpc.close()
        if self._owns_sim:
            self.sim.close()
        if destroy_group and self._collectives and dist.is_initialized():
            dist.destroy_process_group()

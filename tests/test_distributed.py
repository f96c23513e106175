"""Multi-rank path on CPU: world_size 2 over gloo (the GPU path uses the same
code with backend "nccl" = RCCL)."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from upkie_amd.distributed import RECORD_WORDS, RecordGather, init_distributed, shard_range


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_range_partitions_the_batch():
    for total, world in ((65536, 8), (10, 3), (7, 8), (4096, 1)):
        covered = []
        for rank in range(world):
            lo, hi = shard_range(rank, world, total)
            assert 0 <= lo <= hi <= total
            covered.extend(range(lo, hi))
        assert covered == list(range(total))


def _worker(rank: int, world: int, port: int, envs: int, steps: int, out_path: str):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = init_distributed(world, backend="gloo")
    assert (r, w) == (rank, world)
    gather = RecordGather(envs, rank, world, device="cpu")
    lo, hi = shard_range(rank, world, envs * world)
    ok = True
    for step in range(steps):
        # record of global env g at step s: every word = 1000 s + g + word / 10
        g = torch.arange(lo, hi, dtype=torch.float32)[:, None]
        gather.local.copy_(1000.0 * step + g + torch.arange(RECORD_WORDS)[None, :] / 10.0)
        out = gather.gather()
        if rank == 0:
            flat = out.reshape(world * envs, RECORD_WORDS)
            expect = 1000.0 * step + torch.arange(world * envs, dtype=torch.float32)[:, None] + torch.arange(RECORD_WORDS)[None, :] / 10.0
            ok = ok and torch.equal(flat, expect)
        else:
            ok = ok and out is None
    # max-over-ranks reduction used for the timing contract
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    dist.barrier()
    if rank == 0:
        with open(out_path, "w") as f:
            f.write("ok" if ok else "bad")
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def test_record_gather_world_size_2_gloo(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, free_port(), 33, 3, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def test_record_gather_single_rank():
    gather = RecordGather(5, 0, 1, device="cpu")
    gather.local.copy_(torch.arange(5 * RECORD_WORDS, dtype=torch.float32).reshape(5, RECORD_WORDS))
    out = gather.gather()
    assert out.shape == (1, 5, RECORD_WORDS) and torch.equal(out[0], gather.local)


def test_world_size_mismatch_is_an_error(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(RuntimeError):
        init_distributed(expected_world=2)

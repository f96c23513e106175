"""Run under `python -m torch.distributed.run --nproc-per-node 1` with
UPKIE_FORCE_PROCESS_GROUP=1 on a GPU box (tests/test_sharded_gpu.py): the
rank-0-policy path of `ShardedVecEnv` -- observation gather, action scatter,
chunked record gather -- through RCCL on a one-rank group, for every env kind,
against the same env stepped without any collective. Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests.helpers import randomized_config  # noqa: E402
from tests.test_distributed import _kind_policy  # noqa: E402
from upkie_amd import abi  # noqa: E402
from upkie_amd.distributed import ShardedVecEnv, init_distributed  # noqa: E402


def make(kind, B, collectives):
    cfg = randomized_config(B, seed=9, autoreset=True)
    cfg.fall_pitch = 0.2
    return ShardedVecEnv(kind, cfg, "cuda:0", rank=0, world_size=1, horizon=32, chunk=4, collectives=collectives,
                         mpc_config=abi.default_mpc_config(B, 16) if kind == "base_velocity" else None)


def main():
    rank, world, local = init_distributed(1)
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    out = {}
    B, steps = 300, 14
    for kind in ("pendulum", "gyropod", "servos", "base_velocity"):
        policy = lambda o, k=kind: _kind_policy(k)(o.cpu()).to("cuda:0")  # noqa: E731
        a, b = make(kind, B, True), make(kind, B, False)
        a.reset()
        ob = b.reset()
        for _ in range(steps):
            a.step_from_root(policy)
            ob = b.step(policy(ob))[0]
        a.flush()
        b.flush()
        torch.cuda.synchronize()
        same = True
        for step in range(steps - 8, steps):
            for x, y in zip(a.records(step), b.records(step)):
                same = same and torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0))
        out[kind] = {"bit_equal": bool(same), "resets": a.total_resets(), "resets_plain": b.total_resets()}
        a.sim.close()
        b.sim.close()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

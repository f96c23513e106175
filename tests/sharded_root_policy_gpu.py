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
        if kind in ("pendulum", "servos"):
            # stale = 1: rank 0's policy one step behind, its observation gather overlapped with the step (round 5),
            # against one process that acts on the observation of one step earlier
            a, b = make(kind, B, True), make(kind, B, False)
            a.reset()
            ob = b.reset()
            older = None
            for _ in range(steps):
                a.step_from_root(policy, stale=1)
                before = ob.clone()
                ob = b.step(policy(older if older is not None else before))[0]
                older = before
            a.flush()
            b.flush()
            torch.cuda.synchronize()
            same = True
            for step in range(steps - 8, steps):
                for x, y in zip(a.records(step), b.records(step)):
                    same = same and torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0))
            out[kind]["stale_bit_equal"] = bool(same)
            a.sim.close()
            b.sim.close()
    # what the lag buys on a one-rank RCCL group (the only one a one-GPU box can host): us per step of the rank-0-policy loop
    import time

    device_policy = lambda o: (10.0 * o[:, 0] + o[:, 1] + 0.1 * o[:, 3]).clamp(-0.99, 0.99)[:, None]  # noqa: E731
    timing = {}
    for stale in (0, 1):
        env = make("pendulum", 4096, True)
        env.reset()
        for _ in range(50):
            env.step_from_root(device_policy, stale=stale)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            env.step_from_root(device_policy, stale=stale)
        env.flush()
        torch.cuda.synchronize()
        timing[f"stale_{stale}_us_per_step"] = (time.perf_counter() - t0) / 300 * 1e6
        env.sim.close()
    out["root_policy_loop_4096_envs_one_rank"] = timing
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""Multi-rank path on CPU: world_size 2 over gloo (the GPU path runs the same
code with backend "nccl" = RCCL)."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from upkie_amd.distributed import RECORD_WORDS, RolloutGather, init_distributed, shard_range


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


def record_pattern(step: int, lo: int, hi: int) -> torch.Tensor:
    """Record of global env g at step s: word w = 1000 s + g + w / 10."""
    g = torch.arange(lo, hi, dtype=torch.float32)[:, None]
    return 1000.0 * step + g + torch.arange(RECORD_WORDS)[None, :] / 10.0


def produce(gather: RolloutGather, steps: int, lo: int, hi: int) -> bool:
    ok = True
    for step in range(steps):
        out = gather.begin_step()
        if step > 0:  # the agent's input: last step's records are still intact
            ok = ok and torch.equal(gather.previous, record_pattern(step - 1, lo, hi))
        out.copy_(record_pattern(step, lo, hi))
        gather.end_step()
    gather.flush()
    return ok


def produce_more(gather: RolloutGather, first: int, last: int, lo: int, hi: int) -> bool:
    ok = True
    for step in range(first, last):
        out = gather.begin_step()
        ok = ok and torch.equal(gather.previous, record_pattern(step - 1, lo, hi))
        out.copy_(record_pattern(step, lo, hi))
        gather.end_step()
    gather.flush()
    return ok


def _worker(rank: int, world: int, port: int, envs: int, steps: int, horizon: int, chunk: int, out_path: str):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = init_distributed(world, backend="gloo")
    assert (r, w) == (rank, world)
    gather = RolloutGather(envs, rank, world, device="cpu", horizon=horizon, chunk=chunk)
    lo, hi = shard_range(rank, world, envs * world)
    ok = produce(gather, steps, lo, hi)
    if rank == 0:
        kept = gather.horizon - gather.chunk  # whole chunks older than the one being filled
        for step in range(max(0, steps - kept), steps):  # the ring keeps the last chunks
            flat = gather.records(step).reshape(world * envs, RECORD_WORDS)
            ok = ok and torch.equal(flat, record_pattern(step, 0, world * envs))
        ok = ok and torch.equal(gather.last(), gather.records(steps - 1))
        ok = ok and torch.equal(gather.last(2), gather.records(steps - 3))
    else:
        ok = ok and gather.rollout is None and gather.last() is None
    # a rollout continues after a flush in the middle of a chunk
    ok = ok and produce_more(gather, steps, steps + 5, lo, hi)
    if rank == 0:
        for step in range(steps, steps + 5):
            ok = ok and torch.equal(gather.records(step).reshape(world * envs, RECORD_WORDS), record_pattern(step, 0, world * envs))
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


@pytest.mark.parametrize("steps,horizon,chunk", [(7, 16, 1), (21, 8, 1), (21, 16, 4), (37, 24, 8), (5, 32, 8)])
def test_pipelined_gather_world_size_2_gloo(tmp_path, steps, horizon, chunk):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, free_port(), 33, steps, horizon, chunk, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


@pytest.mark.parametrize("chunk", [1, 2])
def test_single_rank_writes_straight_into_the_ring(chunk):
    gather = RolloutGather(5, 0, 1, device="cpu", horizon=4, chunk=chunk)
    assert produce(gather, 6, 0, 5)
    for step in (2, 3, 4, 5):
        assert torch.equal(gather.records(step)[0], record_pattern(step, 0, 5))
    assert gather.staging is None and gather.current.data_ptr() == gather.records(6)[0].data_ptr()  # no staging copy


def test_world_size_mismatch_is_an_error(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(RuntimeError):
        init_distributed(expected_world=2)


def _sharded_worker(rank: int, world: int, port: int, out_path: str):
    """bench.py's driver object on two ranks (gloo), each rank's shard backed
    by the oracle double: sharding by global env id, chunked gather into rank
    0's rollout, reset counting and max-over-ranks timing."""
    import numpy as np

    from tests.fake_sim import OracleSim
    from tests.helpers import randomized_config
    from upkie_amd import abi
    from upkie_amd.distributed import ShardedPendulum
    from upkie_amd.model.default_model import default_model

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = init_distributed(world, backend="gloo")
    B = 6
    factory = lambda cfg, model, device: OracleSim(cfg, model if model is not None else default_model(), device)  # noqa: E731

    def config(num_envs, offset):
        cfg = randomized_config(num_envs, seed=4, autoreset=True)
        cfg.fall_pitch = 0.12
        cfg.env_id_offset = offset
        return cfg

    env = ShardedPendulum(config(B, rank * B), device="cpu", rank=rank, world_size=world, horizon=32, chunk=4, sim_factory=factory)
    env.reset()
    steps = 22
    for _ in range(steps // 2):
        env.rollout_agent(2)  # two steps per call, inside the 4-step chunks; the single-rank run below steps one by one
    with pytest.raises(ValueError):
        env.gather.begin_steps(3)  # 22 % 4 = 2: three more steps would cross the chunk boundary
    env.flush()
    env.barrier()
    assert env.max_over_ranks(float(rank + 1)) == float(world)
    resets = env.total_resets()
    ok = True
    if rank == 0:
        # one process stepping all 2 B envs gives the same records: results do not depend on the sharding
        whole = ShardedPendulum(config(world * B, 0), device="cpu", rank=0, world_size=1, horizon=32, chunk=4, sim_factory=factory)
        whole.reset()
        for _ in range(steps):
            whole.step_agent()
        for back in range(8):
            sharded = env.gather.last(back).reshape(world * B, -1)
            single = whole.gather.last(back).reshape(world * B, -1)
            ok = ok and np.allclose(sharded.numpy(), single.numpy(), atol=1e-6)
        ok = ok and resets == whole.total_resets() and resets >= world * B
        with open(out_path, "w") as f:
            f.write("ok" if ok else "bad")
    dist.barrier()
    env.gather.flush()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def test_sharded_pendulum_two_ranks_equal_one_rank(tmp_path):
    out = tmp_path / "sharded.txt"
    mp.spawn(_sharded_worker, args=(2, free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _shard_worker_8(rank: int, world: int, port: int, out_path: str):
    """Eight ranks (gloo): the shard arithmetic of BASELINE configs[3] / [4]
    (65536 = 8 x 8192, 32768 = 8 x 4096) on the real gather path, with tiny
    per-rank batches standing for the full ones, and an uneven total."""
    from upkie_amd.distributed import RolloutGather, shard_range

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_distributed(world, backend="gloo")
    ok = True
    # the configs' own numbers: contiguous, disjoint, complete
    for total in (65536, 32768, 65536 + 5):
        ranges = [shard_range(r, world, total) for r in range(world)]
        ok = ok and ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        ok = ok and max(hi - lo for lo, hi in ranges) - min(hi - lo for lo, hi in ranges) <= 1
    ok = ok and shard_range(3, 8, 65536) == (3 * 8192, 4 * 8192) and shard_range(7, 8, 32768) == (7 * 4096, 8 * 4096)
    # the gather: 8 ranks x 3 envs, 11 steps in chunks of 4 (a partial chunk is flushed)
    B = 3
    gather = RolloutGather(B, rank, world, device="cpu", horizon=16, chunk=4)
    for step in range(11):
        out = gather.begin_step()
        out.copy_(record_pattern(step, rank * B, (rank + 1) * B))
        gather.end_step()
    gather.flush()
    if rank == 0:
        for step in range(11):
            got = gather.records(step)
            for r in range(world):
                ok = ok and torch.equal(got[r], record_pattern(step, r * B, (r + 1) * B))
        with open(out_path, "w") as f:
            f.write("ok" if ok else "bad")
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def test_eight_ranks_shard_arithmetic_and_gather(tmp_path):
    out = tmp_path / "eight.txt"
    mp.spawn(_shard_worker_8, args=(8, free_port(), str(out)), nprocs=8, join=True)
    assert out.read_text() == "ok"


@pytest.mark.parametrize("world,flags,total,per_rank,ghosts", [
    (2, ["--envs-per-gpu", "6"], 12, 6, 0),  # weak scaling
    (3, ["--total-envs", "16"], 16, 6, 2),  # strong scaling, 3 does not divide 16: blocks of 6, two ghost envs on the last rank
    (2, ["--config", "c4", "--envs-per-gpu", "6"], 12, 6, 0),  # BASELINE configs[3]: gather + the rollout consumer on rank 0
])
def test_bench_launch_line_on_cpu_doubles(world, flags, total, per_rank, ghosts):
    """The driver's launch line (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    --steps K --warmup W`) with bench.py's `main` running on the oracle-backed
    double over gloo (tests/bench_double.py): one JSON line from rank 0, whole-job
    figures, the gather text, weak and strong scaling with a remainder."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "tests", "bench_double.py"), "--gpus", str(world), "--steps", "12", "--warmup", "3",
           "--gather-chunk", "4"] + flags
    env = dict(os.environ, OMP_NUM_THREADS="1")
    result = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 12 and out["warmup"] == 3
    cfg = out["config"]
    assert cfg["total_envs"] == total and cfg["envs_per_gpu"] == per_rank and cfg["ghost_envs"] == ghosts
    assert out["scaling"] == ("strong" if "--total-envs" in flags else "weak")
    assert cfg["gather"].startswith("RCCL gather") and "4-step chunk" in cfg["gather"]
    assert out["value"] == pytest.approx(total * 12 / (out["ms_per_step"] * 1e-3 * 12), rel=1e-6)
    assert cfg["steps_per_launch"] == 1 and out["fused_rollout"]["value"] > 0  # (the double steps its rollouts one by one)
    assert out["cpu_baseline"] is None  # N = 1 on a GPU only
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["frac"] > 0
    steady = out["steady_state"]  # the window of SURVEY 8d, timed before the contract region (shrunk by the double)
    assert steady["steps"] == 6 and steady["warmup"] == 2 and steady["value"] == pytest.approx(total * 6 / (steady["ms_per_step"] * 1e-3 * 6), rel=1e-6)
    assert "secondary" not in out  # one GPU, rank 0 only
    if "c4" in flags:  # every completed chunk of the timed and untimed steps went through the consumer, in order
        assert cfg["baseline_config"].startswith("configs[3]") and "rollout consumer" in cfg["workload"]
        assert cfg["rollout_consumer"]["chunks_consumed"] >= (12 + 3 + 6 + 2) // 4 - 1 and cfg["rollout_consumer"]["steps_per_chunk"] == 4
    else:
        assert cfg["baseline_config"].startswith("weak-scaling line of configs[1]") and "--config c4" in cfg["baseline_config"]


def test_a_flush_ships_only_the_steps_it_has_not_shipped(monkeypatch):
    """A flush in the middle of a chunk (the end of bench.py's timed region: the driver times 20 steps against a
    64-step chunk) sends the steps produced since the last flush, not the whole chunk buffer, and the completed
    chunk only what no flush has sent."""
    from upkie_amd import distributed as D

    sent = []

    class Done:
        def wait(self):
            return None

    def fake_gather(tensor, gather_list=None, dst=0, async_op=False):
        sent.append(tuple(tensor.shape))
        assert gather_list is not None and all(tuple(g.shape) == tuple(tensor.shape) and g.is_contiguous() for g in gather_list)
        gather_list[0].copy_(tensor)
        return Done()

    monkeypatch.setattr(D.dist, "gather", fake_gather)
    g = RolloutGather(5, 0, 1, device="cpu", horizon=16, chunk=8, collectives=True)

    def advance(first, last):
        for step in range(first, last):
            g.begin_step().copy_(record_pattern(step, 0, 5))
            g.end_step()

    advance(0, 3)
    g.flush()
    g.flush()  # nothing new: nothing sent
    advance(3, 5)
    g.flush()
    advance(5, 8)  # the chunk completes: its remaining steps are sent
    advance(8, 9)
    g.flush()
    assert sent == [(3, 5, RECORD_WORDS), (2, 5, RECORD_WORDS), (3, 5, RECORD_WORDS), (1, 5, RECORD_WORDS)], sent
    for step in range(9):
        assert torch.equal(g.records(step)[0], record_pattern(step, 0, 5))


# ------------------------------------------------------------------ every env kind, sharded (ShardedVecEnv)
def _kind_policy(kind):
    """A deterministic policy of the observation per env kind: ``[n, *obs_shape] -> [n, *act_shape]``."""
    import math

    def pendulum(o):
        return (10.0 * o[:, 0] + o[:, 1] + 0.1 * o[:, 3]).clamp(-0.99, 0.99)[:, None]

    def gyropod(o):
        v = (10.0 * o[:, 1] + o[:, 0] + 0.1 * o[:, 3]).clamp(-0.99, 0.99)
        return torch.stack([v, torch.full_like(v, 0.1)], dim=1)

    def servos(o):
        a = torch.zeros((o.shape[0], 6, 6), dtype=torch.float32)
        a[:, :, 3] = 1.0
        a[:, :, 4] = 1.0
        a[:, :, 5] = 16.0
        a[:, [2, 5], 0] = math.nan
        a[:, 2, 1] = (20.0 * o[:, 0, 0]).clamp(-5.0, 5.0)  # wheel velocity targets from the hip angles: any function of the observation
        a[:, 5, 1] = -a[:, 2, 1]
        return a

    def base_velocity(o):
        return torch.stack([0.2 + 0.0 * o[:, 0], 0.1 + 0.0 * o[:, 0]], dim=1)

    return {"pendulum": pendulum, "gyropod": gyropod, "servos": servos, "base_velocity": base_velocity}[kind]


def _make_sharded(kind, num_envs, offset, rank, world, collectives=None, servo_policy=None, chunk=4):
    from tests.fake_sim import OracleMpc, OracleSim
    from tests.helpers import randomized_config
    from upkie_amd import abi
    from upkie_amd.distributed import ShardedVecEnv
    from upkie_amd.model.default_model import default_model

    cfg = randomized_config(num_envs, seed=7, autoreset=True)
    cfg.fall_pitch = 0.15  # episodes end (and restart) within the test's few steps
    cfg.env_id_offset = offset
    factory = lambda c, model, device: OracleSim(c, model if model is not None else default_model(), device)  # noqa: E731
    return ShardedVecEnv(kind, cfg, device="cpu", rank=rank, world_size=world, horizon=16, chunk=chunk, sim_factory=factory, collectives=collectives,
                         servo_policy=servo_policy, mpc_config=abi.default_mpc_config(num_envs, 16) if kind == "base_velocity" else None,
                         mpc_factory=OracleMpc)


def _run_kind(env, kind, mode, steps):
    """`steps` env.step() of a (sharded or whole) env under `mode`: "local" = every rank evaluates the policy on its
    own observations, "root" = rank 0 evaluates it for everybody (gather + scatter), "policy" = the servo-level law
    inside the step."""
    policy = _kind_policy(kind)
    obs = env.reset()
    for _ in range(steps):
        if mode == "local":
            obs = env.step(policy(obs))[0]
        elif mode == "root":
            obs = env.step_from_root(policy)[0]
        elif mode == "root_stale":  # rank 0's policy one step behind, its gather overlapped with the step (round 5)
            obs = env.step_from_root(policy, stale=1)[0]
        elif mode == "local_stale":  # what `root_stale` must equal: one process acting on the observation of one step earlier
            before, older = obs, getattr(env, "_test_older", None)
            obs = env.step(policy(older if older is not None else before))[0]
            env._test_older = before.clone()
        else:
            obs = env.step(None)[0]
    env.flush()


def _sharded_kinds_worker(rank: int, world: int, port: int, per_rank: int, total: int, cases, out_path: str):
    from upkie_amd import abi

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    init_distributed(world, backend="gloo")
    ok, steps, why = True, 11, ""
    for kind, mode in cases:
        law = abi.velocity_balancing_policy(0.06, 0.15, 1.0) if mode == "policy" else None
        env = _make_sharded(kind, per_rank, rank * per_rank, rank, world, servo_policy=law)
        _run_kind(env, kind, mode, steps)
        resets = env.total_resets()
        if rank == 0:
            # one process stepping all envs (ghost envs of an uneven split included: they are simulated, not counted)
            whole = _make_sharded(kind, world * per_rank, 0, 0, 1, collectives=False, servo_policy=law)
            _run_kind(whole, kind, {"root": "local", "root_stale": "local_stale"}.get(mode, mode), steps)
            for step in range(steps - 8, steps):  # what the 16-step ring still holds of both
                got = env.records(step)
                ref = whole.records(step)
                for name, g, r in zip(("obs", "reward", "terminated", "truncated"), got, ref):
                    g = g.reshape((world * per_rank,) + tuple(g.shape[2:]))[:total]
                    r = r.reshape((world * per_rank,) + tuple(r.shape[2:]))[:total]
                    same = torch.equal(torch.nan_to_num(g.float(), nan=-7.0), torch.nan_to_num(r.float(), nan=-7.0))
                    if not same:
                        ok, why = False, f"{kind}/{mode}: {name} of step {step} differs"
            if resets != whole.total_resets():
                ok, why = False, f"{kind}/{mode}: {resets} resets sharded, {whole.total_resets()} whole"
            if kind != "base_velocity" and mode != "policy" and resets < 1:
                ok, why = False, f"{kind}/{mode}: no episode ended (the test is meant to cross autoresets)"
            whole.sim.close()
        dist.barrier()
        env.gather.flush()
        env.sim.close()
    if rank == 0:
        with open(out_path, "w") as f:
            f.write("ok" if ok else "bad: " + why)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


ALL_KIND_CASES = [("pendulum", "local"), ("pendulum", "root"), ("gyropod", "local"), ("gyropod", "root"), ("servos", "local"), ("servos", "root"),
                  ("servos", "policy"), ("base_velocity", "local"), ("base_velocity", "root"),
                  ("pendulum", "root_stale"), ("gyropod", "root_stale"), ("servos", "root_stale")]


def test_sharded_vec_env_every_kind_two_ranks_equal_one_rank(tmp_path):
    """Pendulum / Gyropod / Servos / BaseVelocity on two ranks (gloo, oracle
    doubles): per-rank policies, a rank-0 policy (observation gather + action
    scatter) and the in-launch servo law all give, bit for bit, the outputs of
    one process stepping all the envs."""
    out = tmp_path / "kinds2.txt"
    mp.spawn(_sharded_kinds_worker, args=(2, free_port(), 5, 10, ALL_KIND_CASES, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def test_sharded_vec_env_eight_ranks_equal_one_rank(tmp_path):
    """BASELINE configs[4]'s shape (Servos over 8 ranks, the law inside the
    launch) and the rank-0 policy path on eight ranks, tiny shards."""
    out = tmp_path / "kinds8.txt"
    cases = [("servos", "policy"), ("servos", "root"), ("pendulum", "root"), ("pendulum", "root_stale")]
    mp.spawn(_sharded_kinds_worker, args=(8, free_port(), 3, 24, cases, str(out)), nprocs=8, join=True)
    assert out.read_text() == "ok"


def test_sharded_vec_env_uneven_split_has_ghost_envs(tmp_path):
    """16 envs over 3 ranks: blocks of 6, the last rank carries two ghost envs
    that are stepped (equal messages) but are nobody's results: the first 16
    rows equal the 16-env single-process run."""
    out = tmp_path / "ghosts.txt"
    mp.spawn(_sharded_kinds_worker, args=(3, free_port(), 6, 16, [("gyropod", "root"), ("servos", "local")], str(out)), nprocs=3, join=True)
    assert out.read_text() == "ok"


def test_step_blob_layout_is_aligned_and_decodes():
    from upkie_amd.distributed import StepBlob

    for B, shape in ((5, (4,)), (4096, (6, 5)), (7, (3,)), (1, (6,))):
        blob = StepBlob(B, shape)
        assert blob.nbytes % 16 == 0 and blob.reward_offset % 16 == 0 and blob.terminated_offset % 16 == 0 and blob.truncated_offset % 16 == 0
        buf = torch.zeros((3, blob.words))
        obs, rew, term, trunc = blob.views(buf)
        assert obs.shape == (3, B) + shape and rew.shape == (3, B) and term.shape == (3, B) and term.dtype == torch.uint8
        obs[1].fill_(2.0)
        rew[1].fill_(3.0)
        term[1].fill_(1)
        trunc[1].fill_(1)
        again = blob.views(buf[1])
        assert float(again[0].sum()) == 2.0 * obs[1].numel() and float(again[1].sum()) == 3.0 * B and int(again[2].sum()) == B and int(again[3].sum()) == B
        assert float(buf[0].abs().sum()) == 0.0 and float(buf[2].abs().sum()) == 0.0  # nothing spills into the neighbours
        a = blob.addresses(buf[1])
        assert a[0] == buf[1].data_ptr() and a[3] - a[0] == blob.truncated_offset


def test_bench_default_multi_gpu_line_also_measures_configs_3_and_4():
    """The driver's multi-GPU runs use the default flags:
    `bench.py --gpus N --steps K --warmup W` with N > 1 prints the weak-scaling
    line of configs[1]'s per-GPU workload AND, under `secondary`, BASELINE
    configs[3] (`c4`: the Pendulum env at its own batch size, gather + rollout
    consumer) and configs[4] (`c5`: UpkieServos with pushes and randomised
    inertias), each with its own steady_state, all three on one process group."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "tests", "bench_double.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--gather-chunk", "4"]
    env = dict(os.environ, OMP_NUM_THREADS="1", UPKIE_BENCH_DOUBLE_ENVS="6,10")
    result = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["total_envs"] == 12 and out["config"]["baseline_config"].startswith("weak-scaling line of configs[1]")
    sec = out["secondary"]
    c4, c5 = sec["c4"], sec["c5"]
    assert "error" not in c4 and "error" not in c5, (c4, c5)
    assert c4["config"]["baseline_config"].startswith("configs[3]") and c4["config"]["total_envs"] == 20 and c4["n_gpus"] == 2
    assert c4["config"]["rollout_consumer"]["chunks_consumed"] > 0 and c4["steady_state"]["steps"] == 6
    assert c4["value"] == pytest.approx(20 * 12 / (c4["ms_per_step"] * 1e-3 * 12), rel=1e-6)
    assert "C5" in c5["config"]["workload"] and c5["config"]["total_envs"] == 12 and c5["steady_state"]["steps"] == 6
    assert c5["value"] == pytest.approx(12 * 12 / (c5["ms_per_step"] * 1e-3 * 12), rel=1e-6)
    assert c5["config"]["gather"].startswith("RCCL gather") and c4["config"]["gather"].startswith("RCCL gather")


def test_bench_c5_launch_line_on_cpu_doubles():
    """`bench.py --config c5 --gpus 2` under the driver's launch line (gloo,
    oracle doubles): BASELINE configs[4]'s workload on the sharded runner of
    every env kind, one JSON line from rank 0 with the whole-job figure."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "tests", "bench_double.py"), "--gpus", "2", "--steps", "9", "--warmup", "2",
           "--gather-chunk", "4", "--config", "c5", "--envs-per-gpu", "5", "--law", "velocity"]
    result = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=root)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 9 and out["config"]["total_envs"] == 10 and out["scaling"] == "weak"
    assert "UpkieServos" in out["metric"] and "C5" in out["config"]["workload"] and out["config"]["gather"].startswith("RCCL gather")
    assert out["value"] == pytest.approx(10 * 9 / (out["ms_per_step"] * 1e-3 * 9), rel=1e-6)
    assert out["roofline"]["algorithmic_bytes_per_env_step"] == 630 and out["steady_state"]["steps"] == 6


def test_the_contract_line_survives_hanging_secondary_blocks():
    """`bench.py --gpus N` (N > 1, default flags) measures BASELINE's configs[3] / configs[4] behind the contract figure; those
    blocks run collectives of their own, and a rank stuck in one must not cost the line: the watchdog prints it and ends the
    process (rank 0), or just ends it (the others)."""
    import json
    import subprocess
    import sys

    script = (
        "import sys, time; sys.path.insert(0, %r); import bench\n"
        "line = {'metric': 'm', 'value': 1.0} if sys.argv[1] == '0' else None\n"
        "bench._arm_line_watchdog(line, 0.5)\n"
        "time.sleep(30)\n"
        "print('never')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    )
    for rank, want_line in (("0", True), ("1", False)):
        out = subprocess.run([sys.executable, "-c", script, rank], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0 and "never" not in out.stdout
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == (1 if want_line else 0)
        if want_line:
            parsed = json.loads(lines[0])
            assert parsed["value"] == 1.0 and "did not finish" in parsed["secondary"]["error"]

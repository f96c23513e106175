"""Headline benchmark: env-steps/s of the batched Upkie-Pendulum env.step().

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): Upkie-Pendulum,
4096 envs per GPU, 200 Hz control (5 x 1 ms physics substeps), fp32, the
README's PD-gain balancer evaluated on-device, init-state randomisation pitch
+-0.1 rad, x +-0.05 m, omega_y +-0.1 rad/s, v_x +-0.05 m/s, fall_pitch 1.0,
NEXT_STEP autoreset. One "step" = one env.step() of every env; the agent runs
on the device, so up to --steps-per-launch (32) consecutive steps share ONE
kernel launch per GPU in which the state stays in registers (every step's
records are still written; results are bit-identical to one launch per step,
whose rate is reported beside it as "single_step_launch"); for N > 1 ranks the packed (obs, reward, terminated,
truncated) records of every step are gathered to rank 0 over RCCL, one
asynchronous collective per 64-step chunk (two per 128-step rollout).

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the driver contract) with two extra objects:
"roofline" (algorithmic bytes / measured kernel time vs HBM peak) and
"cpu_baseline" (the fp64 oracle timed on the host cores, rank 0, N = 1 only).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
STEPS_PER_LAUNCH = 32  # the state stays in registers between the steps of a launch; 1 = one launch per env.step()
GATHER_CHUNK = 64  # steps per collective: a gather costs ~27 us of queue time whatever its size (profiles/r01_gather_chunk_sweep.txt); two per 128-step rollout
# 29 fp32 state words read + written (232 B), action 4,
# obs 16, reward 4, terminated 1, truncated 1.
ALGORITHMIC_BYTES_PER_ENV_STEP = 258
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3


def make_config(num_envs: int, env_id_offset: int = 0, seed: int = 0):
    from upkie_amd import abi

    cfg = abi.default_sim_config(num_envs, frequency=200.0, seed=seed)
    cfg.rand_pitch = 0.1
    cfg.rand_x = 0.05
    cfg.rand_omega_y = 0.1
    cfg.rand_linvel[0] = 0.05
    cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
    cfg.env_id_offset = env_id_offset
    return cfg


def cpu_baseline(envs: int, budget_s: float = 15.0) -> dict:
    """Time the fp64 oracle (a port: the reference's PyBullet path cannot run
    here) on the host cores with the same workload, bounded to ~budget_s."""
    from oracle import oracle as O
    from upkie_amd.model.default_model import default_model

    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    ref = O.Oracle(default_model(), make_config(envs))
    obs = ref.reset()[:, [1, 0, 4, 3]]
    obs, *_ = ref.step_pendulum_agent(obs)  # warm up
    steps = 0
    t0 = time.perf_counter()
    while True:
        obs, *_ = ref.step_pendulum_agent(obs)
        steps += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= budget_s or steps >= 5000:
            break
    # the reference's own execution model: one env, one thread
    try:
        import ctypes

        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)  # the oracle's OpenMP runtime: no team to wake for one env
    except OSError:
        pass
    single = O.Oracle(default_model(), make_config(1))
    o1 = single.reset()[:, [1, 0, 4, 3]]
    n1, t1 = 0, time.perf_counter()
    while True:
        o1, *_ = single.step_pendulum_agent(o1)
        n1 += 1
        e1 = time.perf_counter() - t1
        if e1 >= min(2.0, budget_s) or n1 >= 100000:
            break
    return {
        "value": envs * steps / elapsed,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{steps} env.step() of {envs} envs, fp64 C oracle, OpenMP over envs ({elapsed:.1f} s)",
        "single_env_single_thread": {"value": n1 / e1, "unit": "env-steps/s", "sample": f"{n1} env.step() of 1 env ({e1:.1f} s)"},
    }


def issue_floor(step_us: float, steps_per_launch: float):
    """What actually bounds the step at this batch size: 4096 envs are 128
    waves on 1024 SIMDs, one wave per SIMD, and a lone gfx950 wave issues one
    instruction per >= 4.5 cycles whatever its kind (tools/microbench/
    issue_rate.hip, profiles/r01_issue_rate_microbench.txt). Instructions per
    wave come from the committed PMC passes of this same kernel, batch and
    steps per launch; the step duration is the live one."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary_b4096_final.json")
    meta = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path) or not os.path.exists(meta):
        return None
    with open(meta) as f:
        profiled_steps = json.load(f).get("steps_per_launch", 1)
    if profiled_steps != steps_per_launch:
        return None  # the committed counters describe another launch shape
    with open(path) as f:
        pmc = {k: v["mean_per_launch"] for k, v in json.load(f).items()}
    try:
        instructions = sum(pmc[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"))
        per_wave = instructions / pmc["SQ_WAVES"] / profiled_steps
    except KeyError:
        return None
    cycles, ghz = 4.5, 2.4
    floor_us = per_wave * cycles / (ghz * 1e3)
    return {
        "instructions_per_wave_per_step": per_wave,
        "cycles_per_instruction_lone_wave": cycles,
        "clock_ghz": ghz,
        "floor_us_per_step": floor_us,
        "achieved_us_per_step": step_us,
        "frac": floor_us / step_us,
        "source": "profiles/r01_pmc_summary_b4096_final.json (rocprofv3 --pmc), profiles/r01_issue_rate_microbench.txt",
    }


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=2000)
    parser.add_argument("--warmup", type=int, default=200)
    parser.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    parser.add_argument("--total-envs", type=int, default=0,
                        help="strong scaling (SURVEY 8d): this many envs in total, split evenly over the ranks; default: --envs-per-gpu each (weak)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-single-step", action="store_true", help="skip the extra one-launch-per-step measurement (profiling runs)")
    parser.add_argument("--gather-chunk", type=int, default=GATHER_CHUNK, help="steps per RCCL gather (N > 1)")
    parser.add_argument("--steps-per-launch", type=int, default=STEPS_PER_LAUNCH,
                        help="env.step() per kernel launch (the agent runs on the device: nothing returns to the host between steps)")
    args = parser.parse_args()

    import torch

    from upkie_amd.distributed import ShardedPendulum, init_distributed

    rank, world, local_rank = init_distributed(args.gpus)
    B = args.total_envs // world if args.total_envs > 0 else args.envs_per_gpu
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    # UPKIE_FORCE_PROCESS_GROUP=1: run the RCCL gather path on a one-rank group (test of the N > 1 code on one GPU)
    forced = True if os.environ.get("UPKIE_FORCE_PROCESS_GROUP") == "1" else None
    env = ShardedPendulum(make_config(B, env_id_offset=rank * B), device=device, rank=rank, world_size=world, collectives=forced, chunk=args.gather_chunk)
    env.reset()

    def advance(total: int, per_launch: int) -> int:
        """`total` env.step() of every local env, up to `per_launch` of them per
        kernel launch (never across a gather chunk); returns the launches made."""
        done = launches = 0
        while done < total:
            room = env.gather.chunk - env.gather._step % env.gather.chunk
            n = min(per_launch, total - done, room)
            if n == 1:
                env.step_agent()
            else:
                env.rollout_agent(n)
            launches += 1 if B <= 32768 else n  # beyond 32768 envs the library launches step by step
            done += n
        return launches

    def timed(total: int, per_launch: int):
        env.barrier()
        torch.cuda.synchronize()
        start_evt = torch.cuda.Event(enable_timing=True)
        stop_evt = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        start_evt.record()  # same stream the kernels are launched on
        launches = advance(total, per_launch)
        env.flush()  # records of the last steps must have reached rank 0
        stop_evt.record()
        env.barrier()
        torch.cuda.synchronize()
        elapsed = env.max_over_ranks(time.perf_counter() - t0)
        return elapsed, start_evt.elapsed_time(stop_evt), launches

    advance(args.warmup, args.steps_per_launch)
    elapsed, device_ms, launches = timed(args.steps, args.steps_per_launch)
    resets = env.total_resets()
    # for the record: the same steps launched one by one (what a host-side policy would see)
    single_elapsed = None
    if not args.no_single_step:
        single_elapsed = timed(args.steps, 1)[0] if args.steps_per_launch > 1 else elapsed

    if rank != 0:
        env.shutdown()
        return

    total_envs = B * world
    value = total_envs * args.steps / elapsed
    step_us = device_ms * 1e3 / args.steps  # device time per env.step() of the batch, this rank
    launch_us = device_ms * 1e3 / launches  # avg per-launch device time
    steps_per_launch = args.steps / launches
    achieved = ALGORITHMIC_BYTES_PER_ENV_STEP * B * steps_per_launch / (launch_us * 1e-6) / 1e9
    traffic = None
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        with open(traffic_file) as f:
            measured = json.load(f)
        if measured.get("steps_per_launch", 1) == args.steps_per_launch and measured.get("launch_envs") == B:
            traffic = measured.get("hbm_bytes_per_launch")  # PMC bytes of a launch of this shape
    line = {
        "metric": "env-steps/sec (batched Upkie-Pendulum, 200 Hz)",
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if args.total_envs > 0 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "Upkie-Pendulum batched env.step(), PD-gain balancer on device, 200 Hz (5 x 1 ms substeps), NEXT_STEP autoreset",
            "envs_per_gpu": B,
            "total_envs": total_envs,
            "gather": f"RCCL gather of the packed obs/reward/done records of every step into rank 0's rollout ring buffer, one asynchronous collective per {env.gather.chunk}-step chunk, overlapped with the next chunk's kernels" if env.gather.collectives else "none (single GPU): records written straight into the rollout ring buffer",
            "episode_resets_in_timed_region": resets,
            "steps_per_launch": steps_per_launch,
        },
        # the same steps launched one by one (a policy on the host side of the boundary sees this rate)
        "single_step_launch": None if single_elapsed is None else {"value": total_envs * args.steps / single_elapsed, "ms_per_step": single_elapsed / args.steps * 1e3},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "kernel": "step_kernel_pair<MODE_PENDULUM_AGENT> (two lanes per env; step_kernel<...> above 32768 envs per GPU)" if B <= 32768 else "step_kernel<MODE_PENDULUM_AGENT>",
            "avg_launch_us": launch_us,
            "env_steps_per_launch": B * steps_per_launch,
            "avg_step_us": step_us,
            "algorithmic_bytes_per_env_step": ALGORITHMIC_BYTES_PER_ENV_STEP,
            "note": "the step is fp32-VALU/latency bound (~2e4 VALU instructions vs 258 B per env-step), not HBM bound: see DESIGN.md section 6",
        },
    }
    issue = issue_floor(step_us, args.steps_per_launch) if B == ENVS_PER_GPU else None
    if issue is not None:
        line["roofline"]["issue_floor"] = issue
    if world == 1 and not args.no_cpu_baseline:
        # (the budget can be shortened for tests; the default sample is ~15 s of CPU work)
        line["cpu_baseline"] = cpu_baseline(B, float(os.environ.get("UPKIE_CPU_BASELINE_BUDGET_S", "15")))
    else:
        line["cpu_baseline"] = None
    env.shutdown()
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

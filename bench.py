"""Headline benchmark: env-steps/s of the batched Upkie-Pendulum env.step().

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): Upkie-Pendulum,
4096 envs per GPU, 200 Hz control (5 x 1 ms physics substeps), fp32, the
README's PD-gain balancer evaluated on-device, init-state randomisation pitch
+-0.1 rad, x +-0.05 m, omega_y +-0.1 rad/s, v_x +-0.05 m/s, fall_pitch 1.0,
NEXT_STEP autoreset. One "step" = one env.step() of every env = ONE kernel
launch per GPU (what `VecEnv.step` gives a policy on the host side of the
boundary: SURVEY 8d defines the metric there); that is `value`. Because the
agent runs on the device, up to 32 consecutive steps can also share one launch
in which the state stays in registers (every step's records still written,
results bit-identical): that rate is reported beside it as "fused_rollout".
For N > 1 ranks the packed (obs, reward, terminated, truncated) records of
every step are gathered to rank 0 over RCCL, one asynchronous collective per
64-step chunk (two per 128-step rollout).

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the driver contract) with two extra objects:
"roofline" (algorithmic bytes / measured kernel time vs HBM peak, and under
"valu" the roofline that actually binds: VALU issue utilisation from PMC) and
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
# the host driver of the GPU boxes only supports dmabuf IPC: without this RCCL's intra-node transport fails with
# `hipIpcGetMemHandle: invalid argument` (already exported on the boxes; kept here for any env that drops it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def usable_cores() -> int:
    """Host cores this process may actually run on: the scheduler affinity
    mask capped by the cgroup CPU quota (os.cpu_count() reports the machine's
    hardware threads, which a container is rarely given in full: timing 256
    OpenMP threads on a 16-core quota measures oversubscription, not the CPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (
        ("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else int(t.split()[0]) / int(t.split()[1])),
        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())),
    ):
        try:
            with open(path) as f:
                quota = parse(f.read().strip())
            if quota:
                n = max(1, min(n, int(quota)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(envs: int, budget_s: float = 15.0) -> dict:
    """The fp64 C oracle (a port: the reference's PyBullet path cannot run
    here) on the host cores, same workload, bounded to ~budget_s of CPU work:
    built -O3 -march=native on this host, the whole rollout inside ONE OpenMP
    parallel region (envs are independent: each thread carries its chunk of
    envs through every step, no barrier, no Python in the timed loop),
    threads = the cores this process may use. Beside it: the same code on ONE
    thread (the scaling factor follows) and one env on one thread (the
    reference's own execution model)."""
    import ctypes

    from oracle import oracle as O

    native = O.use_native_build()  # before anything touches the oracle
    from upkie_amd.model.default_model import default_model

    cores = usable_cores()
    try:
        omp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        omp = None

    def rate(num_envs: int, threads: int, seconds: float):
        if omp is not None:
            omp.omp_set_num_threads(threads)
        ref = O.Oracle(default_model(), make_config(num_envs))
        obs = ref.reset()[:, [1, 0, 4, 3]]
        obs, _ = ref.rollout_pendulum_agent(obs, 2)  # warm up (thread team, caches)
        steps, chunk, elapsed = 0, 8, 0.0
        while elapsed < seconds and steps < 200000:
            t0 = time.perf_counter()
            obs, _ = ref.rollout_pendulum_agent(obs, chunk)
            dt = time.perf_counter() - t0
            elapsed += dt
            steps += chunk
            if dt < 0.25 * seconds:
                chunk = min(4 * chunk, 4096)  # few, long calls: the Python call is outside what matters
        return num_envs * steps / elapsed, steps, elapsed

    all_cores, steps, elapsed = rate(envs, cores, 0.6 * budget_s)
    per_thread_envs = max(1, envs // cores)  # the share of one thread in the run above
    one_thread, steps1, elapsed1 = rate(per_thread_envs, 1, 0.25 * budget_s)
    single, n1, e1 = rate(1, 1, min(2.0, 0.15 * budget_s))
    return {
        "value": all_cores,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{steps} env.step() of {envs} envs, fp64 C oracle ({'-O3 -march=native' if native else '-O2'}), one OpenMP region over the rollout, {cores} threads ({elapsed:.1f} s)",
        "hardware_threads": os.cpu_count(),
        "one_thread": {"value": one_thread, "unit": "env-steps/s", "sample": f"{steps1} env.step() of {per_thread_envs} envs on 1 thread ({elapsed1:.1f} s)"},
        "scaling_vs_one_thread": all_cores / one_thread,
        "parallel_efficiency": all_cores / one_thread / cores,
        "single_env_single_thread": {"value": single, "unit": "env-steps/s", "sample": f"{n1} env.step() of 1 env ({e1:.1f} s)"},
    }


PMC_FILE = os.path.join(ROOT, "profiles", "pmc_step_b4096.json")  # written by tools/pmc_summary.py from rocprofv3 --pmc passes


def pmc_of_launch_shape(launch_envs: int, steps_per_launch: int):
    """Committed PMC counters (mean per launch) of the step kernel, or None when
    they were collected on another launch shape than the one just timed."""
    if not os.path.exists(PMC_FILE):
        return None
    with open(PMC_FILE) as f:
        pmc = json.load(f)
    if pmc.get("launch_envs") != launch_envs or pmc.get("steps_per_launch") != steps_per_launch:
        return None
    return pmc


SIMDS, CLOCK_GHZ = 1024, 2.4  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz max


def valu_roofline(pmc, launch_us: float):
    """The roofline that binds this kernel (~480 flop/B, fp32 VALU):
    VALU issue utilisation = wave-level VALU instructions per launch / what the
    chip's 1024 SIMD-32s can issue in the measured launch time (a wave64 VALU
    instruction takes 2 cycles of a SIMD-32), and the fp32 rate it stands for
    when every instruction is counted as an FMA on 64 lanes (an upper bound)."""
    if pmc is None or "SQ_INSTS_VALU" not in pmc.get("counters", {}):
        return None
    c = pmc["counters"]
    insts = c["SQ_INSTS_VALU"]
    capacity = launch_us * 1e-6 * CLOCK_GHZ * 1e9 * SIMDS / 2.0
    out = {
        "bound": "valu",
        "valu_instructions_per_launch": insts,
        "issue_utilisation": insts / capacity,
        "tflops_upper_bound": insts * 64 * 2 / (launch_us * 1e-6) / 1e12,
        "peak_tflops": FP32_VALU_PEAK_TFLOPS,
        "waves_per_launch": c.get("SQ_WAVES"),
        "source": "profiles/pmc_step_b4096.json (rocprofv3 --pmc passes of this launch shape) and the live launch duration",
    }
    if c.get("SQ_WAVES"):
        per_wave = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS")) / c["SQ_WAVES"]
        out["instructions_per_wave"] = per_wave
        # a wave that has its SIMD to itself issues one instruction per >= 4.5 cycles (profiles/r01_issue_rate_microbench.txt)
        out["lone_wave_floor_us"] = per_wave * 4.5 / (CLOCK_GHZ * 1e3)
    return out


def main(argv=None, sim_factory=None, backend=None) -> None:
    """`sim_factory` / `backend` exist for tests/ only (a CPU double of the
    simulation handle over gloo, so that the N > 1 launch line, the shard
    arithmetic and the JSON contract are covered where there is no GPU); the
    product run never passes them."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=2000)
    parser.add_argument("--warmup", type=int, default=200)
    parser.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    parser.add_argument("--total-envs", type=int, default=0,
                        help="strong scaling (SURVEY 8d): this many envs in total, split evenly over the ranks; default: --envs-per-gpu each (weak)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-fused", action="store_true", help="skip the extra fused-rollout measurement (profiling runs)")
    parser.add_argument("--gather-chunk", type=int, default=GATHER_CHUNK, help="steps per RCCL gather (N > 1)")
    parser.add_argument("--steps-per-launch", type=int, default=1,
                        help="env.step() per kernel launch in the TIMED region: 1 (the contract figure: one launch per env.step(), what "
                             "VecEnv.step gives a policy on the host side of the boundary); > 1 times the fused rollout instead (profiling runs)")
    args = parser.parse_args(argv)

    import torch

    from upkie_amd.distributed import ShardedPendulum, init_distributed

    rank, world, local_rank = init_distributed(args.gpus, backend=backend)
    on_gpu = sim_factory is None
    # weak scaling: --envs-per-gpu each. Strong scaling (--total-envs T): blocks of ceil(T / N) consecutive env ids per
    # rank; when N does not divide T the last block runs past T with "ghost" envs that are simulated (the gather needs
    # equal messages) but not counted. Env i gives the same results whatever N (streams keyed by the global id).
    B = -(-args.total_envs // world) if args.total_envs > 0 else args.envs_per_gpu
    counted_envs = args.total_envs if args.total_envs > 0 else B * world
    device = f"cuda:{local_rank}" if on_gpu else "cpu"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    # UPKIE_FORCE_PROCESS_GROUP=1: run the RCCL gather path on a one-rank group (test of the N > 1 code on one GPU)
    forced = True if os.environ.get("UPKIE_FORCE_PROCESS_GROUP") == "1" else None
    env = ShardedPendulum(make_config(B, env_id_offset=rank * B), device=device, rank=rank, world_size=world, collectives=forced, chunk=args.gather_chunk,
                          sim_factory=sim_factory)
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
            launches += 1 if env.fused_rollouts else n  # (large batches: the library launches step by step)
            done += n
        return launches

    def timed(total: int, per_launch: int):
        env.barrier()
        sync()
        episodes_before = env.total_resets()
        env.barrier()
        sync()
        if on_gpu:
            start_evt = torch.cuda.Event(enable_timing=True)
            stop_evt = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if on_gpu:
            start_evt.record()  # same stream the kernels are launched on
        launches = advance(total, per_launch)
        env.flush()  # records of the last steps must have reached rank 0
        if on_gpu:
            stop_evt.record()
        env.barrier()
        sync()
        wall = time.perf_counter() - t0
        elapsed = env.max_over_ranks(wall)
        device_ms = start_evt.elapsed_time(stop_evt) if on_gpu else wall * 1e3
        return elapsed, device_ms, launches, env.total_resets() - episodes_before

    advance(args.warmup, args.steps_per_launch)
    elapsed, device_ms, launches, autoresets = timed(args.steps, args.steps_per_launch)
    # beside it: the same number of steps fused STEPS_PER_LAUNCH per launch (the on-device agent needs nothing from the host between steps)
    fused = None
    if not args.no_fused and args.steps_per_launch == 1:
        advance(min(args.warmup, STEPS_PER_LAUNCH), STEPS_PER_LAUNCH)
        fused = timed(args.steps, STEPS_PER_LAUNCH)

    if rank != 0:
        env.shutdown()
        return

    total_envs = counted_envs
    value = total_envs * args.steps / elapsed
    step_us = device_ms * 1e3 / args.steps  # device time per env.step() of the batch, this rank
    launch_us = device_ms * 1e3 / launches  # avg per-launch device time (HIP events on the launching stream)
    steps_per_launch = args.steps / launches
    achieved = ALGORITHMIC_BYTES_PER_ENV_STEP * B * steps_per_launch / (launch_us * 1e-6) / 1e9
    pmc = pmc_of_launch_shape(B, args.steps_per_launch)
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
            "ghost_envs": B * world - total_envs,  # strong scaling with N not dividing the total: simulated, not counted
            "gather": f"RCCL gather of the packed obs/reward/done records of every step into rank 0's rollout ring buffer, one asynchronous collective per {env.gather.chunk}-step chunk, overlapped with the next chunk's kernels" if env.gather.collectives else "none (single GPU): records written straight into the rollout ring buffer",
            "launches": f"{steps_per_launch:g} env.step() per kernel launch",
            "steps_per_launch": steps_per_launch,
            "autoresets_in_timed_region": autoresets,  # episodes that ended and restarted inside the timed steps (all ranks)
            "lanes_per_env": env.lanes_per_env,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None if pmc is None else pmc.get("hbm_bytes_per_launch"),
            "kernel": env.kernel_name,
            "avg_launch_us": launch_us,
            "env_steps_per_launch": B * steps_per_launch,
            "avg_step_us": step_us,
            "algorithmic_bytes_per_env_step": ALGORITHMIC_BYTES_PER_ENV_STEP,
            "note": "the step is fp32-VALU issue bound (~2e4 VALU lane-instructions vs 258 B per env-step), not HBM bound: the binding roofline is under \"valu\" (DESIGN.md section 6)",
            "valu": valu_roofline(pmc, launch_us),
        },
    }
    if fused is not None:
        f_elapsed, f_ms, f_launches, f_resets = fused
        line["fused_rollout"] = {
            "value": total_envs * args.steps / f_elapsed,
            "ms_per_step": f_elapsed / args.steps * 1e3,
            "steps_per_launch": args.steps / f_launches,
            "avg_launch_us": f_ms * 1e3 / f_launches,
            "autoresets_in_timed_region": f_resets,
            "note": "same steps, same results bit for bit, up to 32 env.step() per launch with the state in registers (upkie_sim_step_pendulum_agent_rollout): what an on-device policy gets",
        }
    if world == 1 and not args.no_cpu_baseline and on_gpu:
        # (the budget can be shortened for tests; the default sample is ~15 s of CPU work)
        line["cpu_baseline"] = cpu_baseline(B, float(os.environ.get("UPKIE_CPU_BASELINE_BUDGET_S", "15")))
    else:
        line["cpu_baseline"] = None
    env.shutdown()
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

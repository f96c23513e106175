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
C4_ENVS_PER_GPU = 8192  # BASELINE configs[3]: 65536 envs over 8 GPUs
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
    """The fp64 C oracle (a port: the reference's PyBullet path cannot run
    here) on the host cores, same workload, bounded to ~budget_s of CPU work:
    built -O3 -march=native on this host, the whole rollout inside ONE OpenMP
    parallel region (envs are independent: each thread carries its chunk of
    envs through every step, no barrier, no Python in the timed loop),
    threads = the cores this process may use. Beside it: the same code on ONE
    thread (the scaling factor follows) and one env on one thread (the
    reference's own execution model)."""
    from oracle import oracle as O

    native = O.use_native_build()  # before anything touches the oracle
    from upkie_amd.model.default_model import default_model

    cores = O.usable_cores()

    def rate(num_envs: int, threads: int, seconds: float):
        O.set_threads(threads)
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
        "sample": f"{steps} env.step() of {envs} envs, fp64 C oracle ({'-O3 -march=native' if native else '-O3'}), one OpenMP region over the rollout, {cores} threads ({elapsed:.1f} s)",
        "hardware_threads": os.cpu_count(),
        "one_thread": {"value": one_thread, "unit": "env-steps/s", "sample": f"{steps1} env.step() of {per_thread_envs} envs on 1 thread ({elapsed1:.1f} s)"},
        "scaling_vs_one_thread": all_cores / one_thread,
        "parallel_efficiency": all_cores / one_thread / cores,
        "single_env_single_thread": {"value": single, "unit": "env-steps/s", "sample": f"{n1} env.step() of 1 env ({e1:.1f} s)"},
    }


STEADY_WARMUP, STEADY_STEPS = 200, 2000  # "measured over >= 2000 steps after 200 warm-up steps ... autoresets included"
PUSH_PERIOD, PUSH_HOLD, PUSH_MAX_NORM = 400, 20, 20.0  # every 400 steps a force of norm ~ U(0, 20) N, held 20 steps
TARGET_PERIOD = 400  # v* ~ U(-0.5, 0.5) per env resampled every 400 steps
C4_VALUE_WEIGHTS, C4_GAMMA, C4_LAMBDA = (0.5, 0.1, 0.05, 0.02), 0.99, 0.95  # the rollout consumer of --config c4
C3_BYTES_PER_ENV_STEP, C5_BYTES_PER_ENV_STEP = 554, 630  # SURVEY 8d, algorithmic bytes


def _timed_loop(step, steps: int, warmup: int):
    """`warmup` + `steps` calls of `step(k)` (k counts from 0 across both);
    wall seconds and device milliseconds (HIP events on the launching stream)
    of the timed ones."""
    import torch

    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start.record()
    for k in range(warmup, warmup + steps):
        step(k)
    stop.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, start.elapsed_time(stop)


def secondary_c3(envs: int = 16384, steps: int = STEADY_STEPS, warmup: int = STEADY_WARMUP, seed: int = 0, horizon: int = 16) -> dict:
    """BASELINE.json configs[2] as SURVEY.md section 8d writes it (C3):
    UpkieBaseVelocity (the MPC balancer in front of UpkieGyropod,
    upkie_base_velocity.py:164-202), horizon N = 16 (T = 0.02 s), leg length
    0.58 m, a_max 10, v_max 3 (mpc_balancer.py:170-178), target
    v* ~ U(-0.5, 0.5) per env RESAMPLED every 400 steps, yaw rate 0, warm
    started over-relaxed ADMM iterations on the matrix cores (15 at N <= 16, 30
    beyond; v_mfma_f32_16x16x32_f16 on two fp16 terms per operand with fp32
    accumulation: csrc/mpc.hpp), NEXT_STEP autoreset. `horizon` = 50: the reference's own default
    (mpc_balancer.py:174), reported beside BASELINE's N = 16 as `n50`."""
    import numpy as np
    import torch

    import upkie_amd.envs as envs_mod
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
    env = envs_mod.make("Upkie-HIP-BaseVelocity-Vec", num_envs=envs, frequency=200.0, nb_timesteps=horizon, init_state=init, seed=seed)
    env.reset(seed=seed)
    gen = torch.Generator(device=env.device)
    gen.manual_seed(seed)
    act = torch.zeros(envs, 2, device=env.device)

    def step(k):
        if k % TARGET_PERIOD == 0:
            act[:, 0].uniform_(-0.5, 0.5, generator=gen)
        env.step(act)

    wall, device_ms = _timed_loop(step, steps, warmup)
    us = wall / steps * 1e6
    out = {
        "config": f"C3: UpkieBaseVelocity + MPC balancer N = {horizon} ({int(env.mpc_balancer.config.admm_iterations)} over-relaxed ADMM iterations, v_mfma_f32_16x16x32_f16 on two fp16 terms per operand), v* ~ U(-0.5, 0.5) resampled every 400 steps, NEXT_STEP autoreset; "
                  + ("ONE launch per env.step() (upkie_sim_step_base_velocity_mpc: the balancer's QPs solved by the step's own wavefronts)" if env.fuse_mpc and horizon <= 16
                     else "two launches per env.step() (the balancer's kernel, then the step)") + ", Python loop",
        "horizon": horizon,
        "launches_per_step": 1 if env.fuse_mpc and horizon <= 16 else 2,
        "envs": envs, "steps": steps, "warmup": warmup, "us_per_step": us, "device_us_per_step": device_ms * 1e3 / steps, "env_steps_per_s": envs / (us * 1e-6),
        "lanes_per_env": env.sim.lanes_per_env, "episodes": int(env.sim.state[40].sum().item()),
        "algorithmic_bytes_per_env_step": C3_BYTES_PER_ENV_STEP,
        "hbm_frac": C3_BYTES_PER_ENV_STEP * envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
    }
    env.close()
    return out


def vec_env_api(envs: int = ENVS_PER_GPU, steps: int = STEADY_STEPS, warmup: int = STEADY_WARMUP, seed: int = 0) -> dict:
    """SURVEY.md 8d words the metric as the wall time of `VecEnv.step`: the
    PUBLIC path, `obs, reward, terminated, truncated, info = env.step(policy(obs))`
    on `Upkie-HIP-Pendulum-Vec`, the README balancer written as a three-op
    PyTorch policy on the device (matmul, clamp, unsqueeze) -- from a plain
    Python loop, and as one hipGraph launch per step
    (`upkie_amd.graphs.GraphedEnvStep`: policy kernels + step kernel recorded
    once). Same workload as the headline (C2), NEXT_STEP and SAME_STEP
    autoreset; the step kernel is `step_kernel_octet<MODE_PENDULUM>` (its own
    launch per step, action read from HBM), not the agent-in-the-launch one."""
    import numpy as np
    import torch

    import upkie_amd.envs as envs_mod
    from upkie_amd.graphs import GraphedEnvStep
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    out = {"policy": "act = (obs @ gains).clamp(-0.99, 0.99).unsqueeze(1), gains = [10, 1, 0, 0.1] (README.md:60-67), torch ops on the device",
           "envs": envs, "steps": steps, "warmup": warmup,
           "note": "the loop is GPU bound: step kernel (step_kernel_octet<MODE_PENDULUM>, ~14.5 us) + the policy's two kernels (rocblas gemv ~4.5 us, clamp ~2 us) "
                   "+ three dependent-launch gaps; rocprofv3 per-kernel times under profiles/ (r04_vec_env_kernel_stats.csv). A hipGraph of the same "
                   "three kernels replays SLOWER than the eager loop on ROCm 7.2 (per-node latency), see graphed_us_per_env_step; "
                   "python_loop_one_launch_policy_us_per_env_step: the same policy as ONE kernel (upkie_amd.policies.LinearPolicy) in the same Python loop; "
                   "python_loop_policy_in_the_step_launch_us_per_env_step: env.step_linear_policy(gains, clip), the policy evaluated by the step kernel itself"}
    for mode in ("next_step", "same_step"):
        init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
        env = envs_mod.make("Upkie-HIP-Pendulum-Vec", num_envs=envs, frequency=200.0, autoreset_mode=mode, init_state=init, seed=seed)
        obs, _ = env.reset(seed=seed)
        gains = torch.tensor([10.0, 1.0, 0.0, 0.1], device=env.device)
        policy = lambda o: (o @ gains).clamp(-0.99, 0.99).unsqueeze(1)
        state = {"obs": obs}

        def eager(k):
            state["obs"] = env.step(policy(state["obs"]))[0]

        wall, device_ms = _timed_loop(eager, steps, warmup)
        # where a loop iteration goes: the policy's own kernels (rocBLAS gemv + clamp: the same three ops, result dropped),
        # and what the interpreter needs to ISSUE one iteration (no synchronisation inside the loop)
        wall_policy, _ = _timed_loop(lambda k: policy(state["obs"]), steps, 10)
        t0 = time.perf_counter()
        for k in range(steps):
            eager(k)
        host_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        graphed = GraphedEnvStep(env, policy)
        wall_graph, device_ms_graph = _timed_loop(lambda k: graphed(), steps, warmup)
        # the same loop with the same policy as ONE launch (`upkie_amd.policies.LinearPolicy`, upkie_linear_policy) instead
        # of torch's two kernels: what the step's side of the boundary leaves of an iteration
        from upkie_amd.policies import LinearPolicy

        one_launch = LinearPolicy(gains, clip=0.99, device=env.device)

        def eager_one_launch(k):
            state["obs"] = env.step(one_launch(state["obs"]))[0]

        wall_one, _ = _timed_loop(eager_one_launch, steps, warmup)
        # ... and with the policy inside the step's launch (`UpkiePendulumVecEnv.step_linear_policy`: NEXT_STEP / disabled autoreset)
        wall_in = None
        if mode != "same_step":
            host_gains = [10.0, 1.0, 0.0, 0.1]
            wall_in, _ = _timed_loop(lambda k: env.step_linear_policy(host_gains, clip=0.99), steps, warmup)
        episodes = int(env.sim.state[40].sum().item())
        out[mode] = {
            "python_loop_us_per_env_step": wall / steps * 1e6, "python_loop_device_us": device_ms * 1e3 / steps,
            "policy_ops_alone_us": wall_policy / steps * 1e6, "host_time_to_issue_one_iteration_us": host_issue / steps * 1e6,
            "graphed_us_per_env_step": wall_graph / steps * 1e6,
            "python_loop_one_launch_policy_us_per_env_step": wall_one / steps * 1e6,
            "python_loop_policy_in_the_step_launch_us_per_env_step": None if wall_in is None else wall_in / steps * 1e6,
            "env_steps_per_s_python_loop": envs * steps / wall, "env_steps_per_s_graphed": envs * steps / wall_graph,
            "episodes": episodes, "lanes_per_env": env.sim.lanes_per_env,
        }
        env.close()
    return out


def secondary_c5_share(law: str, envs: int = 4096, steps: int = STEADY_STEPS, warmup: int = STEADY_WARMUP, seed: int = 0, census_steps: int = 400,
                       contact_model: str = "default", lanes: int = 0) -> dict:
    """One GPU's share of BASELINE.json configs[4] as SURVEY.md section 8d
    writes it (C5: 32768 envs over 8 GPUs): UpkieServos, inertia_variation 0.2
    per env and link (pybullet_backend.py:571-601), wheel friction 0.1
    (examples/pybullet/joint_friction.py:22), and the push schedule -- every
    400 steps a world-frame force on the torso, norm ~ U(0, 20) N, uniformly
    random horizontal direction, held 20 steps (set_external_forces,
    pybullet_backend.py:603-658) -- drawn ON THE DEVICE (upkie_sim_sample_pushes,
    Philox keyed by env and push number). Servo-level law evaluated by the
    step's own lanes, fallen robots restart (NEXT_STEP): one launch per step.
    `law`: "torque" = examples/pybullet/torque_balancing.py:15-37 (8d's law:
    wheel torques +-10 x pitch, kd_scale 0), "velocity" = the README balancer
    through the wheels' velocity loop. `contact_model`: "default" or
    "bullet_like" (persistent manifolds, 50 fixed sweeps, cone friction: on
    eight lanes per env for Servos steps too, joint stops inside the same
    sweeps since round 6; `lanes` = 1 asks for the one-lane kernels, which also
    keep several cached points per tire)."""
    import numpy as np
    import torch

    import upkie_amd.envs as envs_mod
    from upkie_amd import abi
    from upkie_amd.model.joint_properties import JointProperties
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
    env = envs_mod.make("Upkie-HIP-Servos-Vec", num_envs=envs, frequency=200.0, inertia_variation=0.2, init_state=init, autoreset_mode="next_step", seed=seed,
                        contact_model=contact_model, joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")})
    env.reset(seed=seed)
    sim = env.sim
    if lanes:
        sim.set_lanes_per_env(lanes)
    push = torch.zeros((3, envs), dtype=torch.float32, device=env.device)
    sim.set_external_force(push)  # the kernels read this buffer at every substep from now on
    m = env.model.struct
    policy = (abi.torque_balancing_policy(10.0, 1.0, float(m.left_sign)) if law == "torque"
              else abi.velocity_balancing_policy(float(m.wheel_radius), 1.0, float(m.left_sign)))

    def step(k):
        phase = k % PUSH_PERIOD
        if phase == 0:
            sim.sample_pushes(k // PUSH_PERIOD, PUSH_MAX_NORM, out=push)
        elif phase == PUSH_HOLD:
            push.zero_()
        sim.step_servos_policy(policy)

    wall, device_ms = _timed_loop(step, steps, warmup)
    us = wall / steps * 1e6
    out = {
        "config": f"C5 share: UpkieServos, inertia_variation 0.2, wheel friction 0.1, torso push every {PUSH_PERIOD} steps (norm ~ U(0, {PUSH_MAX_NORM:g}) N, random heading, "
                  f"held {PUSH_HOLD} steps, drawn on device), servo-level law = " + ("examples/pybullet/torque_balancing.py (wheel torque +-10 x pitch, kd_scale 0)" if law == "torque"
                  else "README balancer through the wheel velocity loop") + ", evaluated inside the step's launch, NEXT_STEP autoreset of fallen robots; one launch per step, Python loop"
                  + ("" if contact_model == "default" else "; contact model: Bullet-like (upkie_sim_set_contact_manifold)"),
        "contact_model": contact_model,
        "envs": envs, "steps": steps, "warmup": warmup, "us_per_step": us, "device_us_per_step": device_ms * 1e3 / steps, "env_steps_per_s": envs / (us * 1e-6),
        "lanes_per_env": sim.lanes_per_env_of(abi.OBSERVATION_SERVOS), "episodes": int(sim.state[40].sum().item()),
        "algorithmic_bytes_per_env_step": C5_BYTES_PER_ENV_STEP,
        "hbm_frac": C5_BYTES_PER_ENV_STEP * envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
    }
    if census_steps and sim.lanes_per_env_of(abi.OBSERVATION_SERVOS) == 8:  # (the census is counted by the eight-lane kernels)
        # rare-path census on its own steps afterwards (its atomics are not free), continuing the same schedule
        sim.enable_census()
        for k in range(warmup + steps, warmup + steps + census_steps):
            step(k)
        c = sim.census_counts()
        sim.enable_census(False)
        substeps = envs * 5 * census_steps
        out["census"] = {
            "env_substeps_in_gauss_seidel_sweeps": c["friction_cone"] / substeps,  # (those whose direct solution was not admissible)
            "env_substeps_with_a_joint_at_its_stop": c["joint_limit"] / substeps,
            "sweeps_per_infeasible_env_substep": c["sweeps_total"] / max(c["friction_cone"], 1),
            "infeasible_env_substeps_answered_by_an_active_set": c["active_set_solves"] / max(c["friction_cone"], 1),
            "sweep_cap_hits": c["sweep_cap_hits"],
            "sweeps_max": c["sweeps_max"],
            # per wavefront-substep that swept: the most sweeps among its eight envs (what the launch waits for); quantiles
            "wavefront_sweeps_quantiles": _histogram_quantiles(c["wavefront_max_sweeps_histogram"], (0.5, 0.9, 0.99, 0.999)),
        }
    env.close()
    return out


def secondary_bullet_like(envs: int = ENVS_PER_GPU, steps: int = 400, warmup: int = 100, seed: int = 0) -> dict:
    """The headline workload (C2: Upkie-Pendulum, PD agent on device, NEXT_STEP
    autoreset) under the Bullet-like contact model
    (`upkie_sim_set_contact_manifold`: persistent manifolds, 50 fixed
    sequential-impulse sweeps, cone friction) on the lane mapping the handle
    picks (eight lanes per env at this size: octet.hpp) and on the one-lane
    kernels that cover every case (bullet_like.hpp), with the default model on
    the same two mappings beside it: what the fidelity option costs."""
    from upkie_amd.sim import BatchedSim

    out = {"config": "C2 workload, one launch per env.step(); *_one_lane rows: UPKIE_LANES_PER_ENV=1", "envs": envs, "steps": steps, "warmup": warmup}
    for name, bullet, forced in (("bullet_like", True, None), ("bullet_like_one_lane", True, "1"), ("default", False, None), ("default_one_lane", False, "1")):
        saved = os.environ.get("UPKIE_LANES_PER_ENV")
        if forced is not None:
            os.environ["UPKIE_LANES_PER_ENV"] = forced
        try:
            sim = BatchedSim(make_config(envs, seed=seed))
        finally:
            if saved is None:
                os.environ.pop("UPKIE_LANES_PER_ENV", None)
            else:
                os.environ["UPKIE_LANES_PER_ENV"] = saved
        if bullet:
            sim.use_bullet_like_contacts()
        o6 = sim.reset()
        sim.obs4.copy_(o6[:, [1, 0, 4, 3]])
        wall, device_ms = _timed_loop(lambda k: sim.step_pendulum_agent(), steps, warmup)
        out[name] = {"us_per_step": wall / steps * 1e6, "device_us_per_step": device_ms * 1e3 / steps, "env_steps_per_s": envs * steps / wall,
                     "lanes_per_env": sim.lanes_per_env, "episodes": int(sim.state[40].sum().item())}
        if name == "bullet_like":
            # the mode closest to pybullet_backend.py:306, reported like the headline: algorithmic bytes over the
            # launch time against HBM, and the roofline that binds it -- VALU issue -- from the committed counters of THIS kernel
            # (tools/pmc_bullet_like.sh; refused when they belong to another version of the sources: kernel_fingerprint)
            launch_us = device_ms * 1e3 / steps
            bytes_per_env_step = ALGORITHMIC_BYTES_PER_ENV_STEP + 2 * (16 + 11) * 4  # + the manifold: 16 words read, 11 written per tire
            pmc = pmc_of_launch_shape(envs, 1, PMC_BULLET_LIKE_FILE)
            out[name]["roofline"] = {"bound": "hbm", "achieved": bytes_per_env_step * envs / (launch_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": bytes_per_env_step * envs / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_env_step": bytes_per_env_step,
                                     "traffic": None if pmc is None else pmc.get("hbm_bytes_per_launch"), "kernel": "step_kernel_octet<MODE_PENDULUM_AGENT, false, false, false, BULLET_LIKE = true>",
                                     "avg_launch_us": launch_us, "valu": valu_roofline(pmc, launch_us, "profiles/pmc_bullet_like_b4096.json")}
        sim.close()
    return out


def _histogram_quantiles(hist, qs):
    total = sum(hist)
    out = {}
    for q in qs:
        acc = 0
        for value, count in enumerate(hist):
            acc += count
            if total and acc >= q * total:
                out[f"p{q * 100:g}"] = value
                break
    return out


PMC_FILE = os.path.join(ROOT, "profiles", "pmc_step_b4096.json")  # written by tools/pmc_summary.py from rocprofv3 --pmc passes


# what the headline kernel (step_kernel_octet<MODE_PENDULUM_AGENT>) is compiled from, and with which flags
KERNEL_SOURCES = ("octet.hpp", "dynamics.hpp", "step_kernels.hpp", "state_words.hpp", "bullet_like.hpp")


def kernel_fingerprint() -> str:
    """sha256[:16] of the headline kernel's sources with comments and whitespace removed, plus the compiler flags: what the
    committed PMC counters must have been collected on. (Comment edits keep it; any change of code or flags does not.)"""
    import hashlib
    import re

    from upkie_amd import lib

    h = hashlib.sha256(" ".join(lib.HIPCC_FLAGS).encode())
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "upkie_amd", "csrc", name)) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


PMC_BULLET_LIKE_FILE = os.path.join(ROOT, "profiles", "pmc_bullet_like_b4096.json")  # tools/pmc_bullet_like.sh


def pmc_of_launch_shape(launch_envs: int, steps_per_launch: int, path: str = None):
    """Committed PMC counters (mean per launch) of the step kernel, or None when
    they were collected on another launch shape than the one just timed -- or on
    another version of the kernel (`kernel_fingerprint`, round 5: a stale file
    yields `traffic: null`, not a number that belongs to other code)."""
    path = path or PMC_FILE
    if not os.path.exists(path):
        return None
    with open(path) as f:
        pmc = json.load(f)
    if pmc.get("launch_envs") != launch_envs or pmc.get("steps_per_launch") != steps_per_launch:
        return None
    if pmc.get("kernel_fingerprint") != kernel_fingerprint():
        return None
    return pmc


SIMDS, CLOCK_GHZ = 1024, 2.4  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz max


def valu_roofline(pmc, launch_us: float, source: str = "profiles/pmc_step_b4096.json"):
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
        "source": source + " (rocprofv3 --pmc passes of this launch shape) and the live launch duration",
    }
    if c.get("SQ_WAVES"):
        per_wave = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS")) / c["SQ_WAVES"]
        out["instructions_per_wave"] = per_wave
        # a wave that has its SIMD to itself issues one instruction per >= 4.5 cycles (profiles/r01_issue_rate_microbench.txt)
        out["lone_wave_floor_us"] = per_wave * 4.5 / (CLOCK_GHZ * 1e3)
    return out


def main_c5(args, sim_factory=None, backend=None, keep_group: bool = False):
    """`python bench.py --config c5 --gpus N`: BASELINE.json configs[4] (SURVEY
    8d "C5"): UpkieServos, 4096 envs per GPU (32768 on 8), per-link inertia
    randomisation 0.2, wheel friction 0.1, the device-drawn push schedule,
    `examples/pybullet/torque_balancing.py`'s law (`--law velocity`: the
    README balancer through the wheel loop) evaluated inside the step's launch,
    NEXT_STEP autoreset of fallen robots; every step's outputs (servo
    observations ``[B, 6, 5]``, reward, flags: 126 B per env) staged and gathered
    to rank 0's ring, one asynchronous RCCL gather per `--gather-chunk` steps
    (`upkie_amd.distributed.ShardedVecEnv`). Same timing contract as the
    headline: W warm-up steps, K timed steps between barriers, max over ranks."""
    import numpy as np
    import torch

    import upkie_amd.envs as envs_mod
    from upkie_amd import abi
    from upkie_amd.distributed import ShardedVecEnv, init_distributed
    from upkie_amd.model.joint_properties import JointProperties
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    rank, world, local_rank = init_distributed(args.gpus, backend=backend)
    on_gpu = sim_factory is None
    B = -(-args.total_envs // world) if args.total_envs > 0 else args.envs_per_gpu
    counted_envs = args.total_envs if args.total_envs > 0 else B * world
    device = f"cuda:{local_rank}" if on_gpu else "cpu"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    forced = True if os.environ.get("UPKIE_FORCE_PROCESS_GROUP") == "1" else None
    init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
    base = envs_mod.make("Upkie-HIP-Servos-Vec", num_envs=B, device=device, frequency=200.0, inertia_variation=0.2, init_state=init, autoreset_mode="next_step", seed=0,
                         env_id_offset=rank * B, joint_properties={n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")},
                         **({"sim_factory": sim_factory} if sim_factory is not None else {}))
    m = base.model.struct
    policy = (abi.torque_balancing_policy(10.0, 1.0, float(m.left_sign)) if args.law == "torque"
              else abi.velocity_balancing_policy(float(m.wheel_radius), 1.0, float(m.left_sign)))
    env = ShardedVecEnv("servos", None, device, rank=rank, world_size=world, chunk=args.gather_chunk, collectives=forced, servo_policy=policy, sim=base.sim)
    sim = base.sim
    env.reset()
    push = torch.zeros((3, B), dtype=torch.float32, device=sim.device)
    sim.set_external_force(push)
    counter = {"k": 0}

    def advance(n):
        for _ in range(n):
            k = counter["k"]
            phase = k % PUSH_PERIOD
            if phase == 0:
                sim.sample_pushes(k // PUSH_PERIOD, PUSH_MAX_NORM, out=push)
            elif phase == PUSH_HOLD:
                push.zero_()
            env.step(None)
            counter["k"] = k + 1

    def timed(n):
        env.flush()
        env.barrier()
        sync()
        before = env.total_resets()
        env.barrier()
        sync()
        if on_gpu:
            start_evt, stop_evt = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start_evt.record()
        t0 = time.perf_counter()
        advance(n)
        env.flush()
        if on_gpu:
            stop_evt.record()
        env.barrier()
        sync()
        wall = time.perf_counter() - t0
        return env.max_over_ranks(wall), (start_evt.elapsed_time(stop_evt) if on_gpu else wall * 1e3), env.total_resets() - before

    steady = None
    if not args.no_steady_state:
        advance(STEADY_WARMUP)
        steady = timed(STEADY_STEPS)
    advance(args.warmup)
    elapsed, device_ms, resets = timed(args.steps)
    if rank != 0:
        env.shutdown(destroy_group=not keep_group)
        base.close()  # (the handle is `base`'s: ShardedVecEnv leaves a caller's handle open)
        return
    step_us = device_ms * 1e3 / args.steps
    achieved = C5_BYTES_PER_ENV_STEP * B / (step_us * 1e-6) / 1e9
    blob = env.blob.nbytes
    line = {
        "metric": "env-steps/sec (batched UpkieServos, BASELINE configs[4] \"C5\", 200 Hz)",
        "value": counted_envs * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.total_envs > 0 else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"UpkieServos batched env.step() (C5): inertia_variation 0.2, wheel friction 0.1, torso push every {PUSH_PERIOD} steps (norm ~ U(0, {PUSH_MAX_NORM:g}) N, held {PUSH_HOLD}, "
                        f"drawn on device), servo-level law '{args.law}' inside the launch, NEXT_STEP autoreset",
            "baseline_config": f"configs[4] (full 6-DoF UpkieServos, floor contact, push + inertia randomisation, 32768 envs at 8 GPUs): {B} envs per GPU x {world} GPU(s) = {counted_envs} envs",
            "envs_per_gpu": B, "total_envs": counted_envs, "ghost_envs": B * world - counted_envs,
            "gather": (f"RCCL gather of every step's outputs (servo observations [B, 6, 5], reward, flags: {blob} B per rank and step) into rank 0's ring, one asynchronous "
                       f"collective per {env.gather.chunk}-step chunk ({blob * env.gather.chunk / 1e6:.1f} MB per rank), overlapped with the next chunk's kernels") if env.gather.collectives
                      else "none (single GPU): outputs written straight into the rollout ring",
            "autoresets_in_timed_region": resets, "lanes_per_env": env.lanes_per_env,
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "step_kernel_octet<MODE_SERVOS> (policy in the launch)", "avg_launch_us": step_us, "algorithmic_bytes_per_env_step": C5_BYTES_PER_ENV_STEP},
        "cpu_baseline": None,
    }
    if env.gather.collectives:
        # a gather costs ~27 us of queue time plus its wire time; rank 0 receives (N - 1) x chunk x blob
        # bytes per chunk over its xGMI links (one link per peer, ~50 GB/s achievable each): overlapped unless it outlasts the chunk
        t_chunk = env.gather.chunk * step_us * 1e-6
        wire = blob * env.gather.chunk / 50e9
        line["config"]["predicted_weak_scaling_efficiency_steady_state"] = t_chunk / (max(t_chunk, wire) + 27e-6)
    if steady is not None:
        s_elapsed, s_ms, s_resets = steady
        line["steady_state"] = {"value": counted_envs * STEADY_STEPS / s_elapsed, "unit": "env-steps/s", "steps": STEADY_STEPS, "warmup": STEADY_WARMUP,
                                "ms_per_step": s_elapsed / STEADY_STEPS * 1e3, "avg_launch_us": s_ms * 1e3 / STEADY_STEPS, "autoresets_in_timed_region": s_resets}
    env.shutdown(destroy_group=not keep_group)
    base.close()
    return line


def c4_rollout_consumer(env, consumed: dict, on_gpu: bool = True):
    """BASELINE configs[3]'s "PPO rollout consumer" on rank 0: every chunk of K steps x (all ranks' envs) that lands in the
    rollout ring is turned into advantages and returns where it lies (a linear value function of the observation, the
    rewards and episode ends the step kernels wrote, upkie_rollout_gae / nothing on the CPU double): stream-ordered
    behind the gather that delivered the chunk, in front of rank 0's next step -- part of the timed work. Returns
    `consume(chunk_index)` for `env.gather.consumer`; it leaves its last results in `consumed`."""
    import torch

    from upkie_amd.rollout import compute_gae

    value_weights = torch.tensor(C4_VALUE_WEIGHTS, device=env.gather.rollout.device)

    def consume(chunk_index: int) -> None:
        ring = env.gather.rollout[chunk_index % env.gather.num_chunks]  # [world, K, B, 8]: records = obs (4), reward, terminated, truncated, 0
        K, N = ring.shape[1], ring.shape[0] * ring.shape[2]
        time_major = lambda t: t.permute(1, 0, 2).reshape(K, N)  # noqa: E731  ([world, K, B] -> [K, world * B])
        values = time_major((ring[..., :4] * value_weights).sum(-1))
        ended = time_major((ring[..., 5] + ring[..., 6]) != 0)
        starts = torch.zeros_like(ended)
        starts[1:] = ended[:-1]
        if on_gpu:
            consumed["advantages"], consumed["returns"] = compute_gae(time_major(ring[..., 4]), values, starts, values[-1], ended[-1], C4_GAMMA, C4_LAMBDA)
        consumed["chunks"] += 1

    return consume


def main(argv=None, sim_factory=None, backend=None, json_out=None) -> None:
    """`sim_factory` / `backend` exist for tests/ only (a CPU double of the
    simulation handle over gloo, so that the N > 1 launch line, the shard
    arithmetic and the JSON contract are covered where there is no GPU); the
    product run never passes them."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=2000)
    parser.add_argument("--warmup", type=int, default=200)
    parser.add_argument("--envs-per-gpu", type=int, default=None, help=f"default: {ENVS_PER_GPU} (BASELINE configs[1] at one GPU); --config c4: 8192 (configs[3]: 65536 envs at 8 GPUs)")
    parser.add_argument("--total-envs", type=int, default=0,
                        help="strong scaling (SURVEY 8d): this many envs in total, split evenly over the ranks; default: --envs-per-gpu each (weak)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-fused", action="store_true", help="skip the extra fused-rollout measurement (profiling runs)")
    parser.add_argument("--no-steady-state", action="store_true",
                        help="skip the steady_state block (SURVEY 8d's own window: 2000 steps after 200 warm-up, whatever --steps / --warmup say)")
    parser.add_argument("--no-secondary", action="store_true", help="skip the C3 / C5-share blocks (rank 0, one GPU)")
    parser.add_argument("--gather-chunk", type=int, default=GATHER_CHUNK, help="steps per RCCL gather (N > 1)")
    parser.add_argument("--steps-per-launch", type=int, default=1,
                        help="env.step() per kernel launch in the TIMED region: 1 (the contract figure: one launch per env.step(), what "
                             "VecEnv.step gives a policy on the host side of the boundary); > 1 times the fused rollout instead (profiling runs)")
    parser.add_argument("--config", choices=("c2", "c4", "c5"), default="c2",
                        help="c2 (default): the headline, Upkie-Pendulum with the PD agent on device, 4096 envs per GPU (BASELINE configs[1]; with --gpus N the "
                             "weak-scaling line of that per-GPU workload); c4: BASELINE configs[3] -- the same env at 8192 envs per GPU (65536 at --gpus 8), every "
                             "step's records gathered to rank 0 AND consumed there by a PPO rollout consumer (generalized advantage estimation over each "
                             "gathered chunk, upkie_rollout_gae); c5: UpkieServos with push / inertia randomisation (BASELINE configs[4]: 4096 envs per GPU, "
                             "32768 at --gpus 8), every env kind's sharded runner")
    parser.add_argument("--law", choices=("torque", "velocity"), default="torque", help="--config c5: the servo-level law evaluated inside the launch")
    args = parser.parse_args(argv)
    explicit_envs = args.envs_per_gpu is not None
    if args.envs_per_gpu is None:
        args.envs_per_gpu = C4_ENVS_PER_GPU if args.config == "c4" else ENVS_PER_GPU
    # The driver's multi-GPU runs use the default flags. Beside the weak-scaling line of configs[1]'s per-GPU workload such a
    # run (N > 1, no --config, default sizes) ALSO measures BASELINE's own multi-GPU configs in the same launch, each with its
    # own steady_state: configs[3] (`--config c4`: 8192 Pendulum envs per GPU = 65536 at 8 GPUs, chunked RCCL gather, rollout
    # consumer on rank 0) and configs[4] (`--config c5`: 4096 UpkieServos envs per GPU = 32768 at 8) -> "secondary": {"c4", "c5"}.
    both_multi_gpu_configs = args.gpus > 1 and args.config == "c2" and not explicit_envs and args.total_envs == 0 and not args.no_secondary
    if args.config == "c5":
        line = main_c5(args, sim_factory=sim_factory, backend=backend)
    else:
        line = run_pendulum(args, sim_factory=sim_factory, backend=backend, keep_group=both_multi_gpu_configs)
    if both_multi_gpu_configs:
        import copy

        def block(config, envs_per_gpu, keep_group):
            a = copy.copy(args)
            a.config, a.envs_per_gpu, a.no_fused, a.no_cpu_baseline = config, envs_per_gpu, True, True
            try:
                if config == "c5":
                    out = main_c5(a, sim_factory=sim_factory, backend=backend, keep_group=keep_group)
                else:
                    out = run_pendulum(a, sim_factory=sim_factory, backend=backend, keep_group=keep_group)
            except Exception as exc:  # noqa: BLE001 (a block beside the contract figure never costs the line itself)
                import traceback

                return {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}
            if out is None:
                return None
            keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "config", "roofline", "steady_state")
            return {k: out[k] for k in keep if k in out}

        # The blocks below run collectives of their own: a rank that fails inside one leaves the others waiting for it. The
        # contract figure is measured by now -- a watchdog prints it if the blocks do not come back (never seen; no multi-GPU
        # box to try it on: the 2 / 4 / 8-GPU runs are the driver's).
        watchdog = _arm_line_watchdog(line, SECONDARY_BLOCKS_TIMEOUT_S, json_out)
        c4 = block("c4", C4_ENVS_PER_GPU, True)
        c5 = block("c5", ENVS_PER_GPU, False)
        watchdog.cancel()
        if line is not None:
            line["secondary"] = {"c4": c4, "c5": c5,
                                 "note": "BASELINE configs[3] and configs[4] at this run's N, measured in the same launch behind the weak-scaling line above "
                                         "(same W / K, each with SURVEY 8d's steady_state window of its own); `--config c4|c5` prints either as a line of its own"}
    if line is not None:
        print(json.dumps(line), file=json_out or sys.stdout, flush=True)


SECONDARY_BLOCKS_TIMEOUT_S = float(os.environ.get("UPKIE_BENCH_SECONDARY_TIMEOUT_S", "300"))


def _arm_line_watchdog(line, seconds: float, json_out=None):
    """A timer that ends the process if the multi-GPU secondary blocks hang: rank 0 (the one that holds `line`) prints
    the contract line first, with the reason in place of the blocks; the other ranks leave a little later."""
    import threading

    def fire():
        if line is not None:
            out = dict(line)
            out["secondary"] = {"error": f"the c4 / c5 blocks behind the contract figure did not finish within {seconds:g} s (a rank stuck in a collective?); the line above them is complete"}
            print(json.dumps(out), file=json_out or sys.stdout, flush=True)
        os._exit(0)

    timer = threading.Timer(seconds if line is not None else seconds + 5.0, fire)
    timer.daemon = True
    timer.start()
    return timer


def run_pendulum(args, sim_factory=None, backend=None, keep_group: bool = False):
    """The Upkie-Pendulum lines (`--config c2`: the headline; `--config c4`:
    BASELINE configs[3]); returns the line on rank 0, None on the others."""
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
    consumed = {"chunks": 0}
    if args.config == "c4" and rank == 0:
        consume = c4_rollout_consumer(env, consumed, on_gpu)
        consume(0)  # (once, untimed: the first call loads the kernels' code objects)
        sync()
        consumed["chunks"] = 0
        env.gather.consumer = consume

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
        env.flush()  # (what earlier steps left in the current chunk travels outside the timed region)
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

    # SURVEY 8d's own definition of the metric, in the same run and on the same envs: >= 2000 steps after 200 warm-up
    # steps, autoresets counted, one launch per env.step() -- regardless of --steps / --warmup. It runs FIRST: the
    # contract's W + K steps below then continue this rollout (episodes spread over their whole life, the way a training
    # run sees them, and a GPU that is already clocked up) instead of timing the landing transient right behind reset().
    steady = None
    if not args.no_steady_state:
        advance(STEADY_WARMUP, 1)
        steady = timed(STEADY_STEPS, 1)
    advance(args.warmup, args.steps_per_launch)
    elapsed, device_ms, launches, autoresets = timed(args.steps, args.steps_per_launch)
    # beside it: the same number of steps fused STEPS_PER_LAUNCH per launch (the on-device agent needs nothing from the host between steps)
    fused = None
    if not args.no_fused and args.steps_per_launch == 1:
        advance(min(args.warmup, STEPS_PER_LAUNCH), STEPS_PER_LAUNCH)
        fused = timed(args.steps, STEPS_PER_LAUNCH)

    if rank != 0:
        env.shutdown(destroy_group=not keep_group)
        return None

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
            "workload": "Upkie-Pendulum batched env.step(), PD-gain balancer on device, 200 Hz (5 x 1 ms substeps), NEXT_STEP autoreset"
                        + ("; every gathered chunk consumed on rank 0 by generalized advantage estimation (the PPO rollout consumer)" if args.config == "c4" else ""),
            # which BASELINE.json config this line is, in words (the driver's 1 / 2 / 4 / 8-GPU runs use the default flags: weak scaling at 4096 envs per GPU)
            "baseline_config": (f"configs[3] (Upkie-Pendulum 65536 envs sharded over 8 GPUs, RCCL gather, PPO rollout consumer): {B} envs per GPU x {world} GPU(s) = {total_envs} envs"
                                if args.config == "c4" else
                                ("configs[1] (Upkie-Pendulum batched 4096 envs on one GPU, PD-gain balancer)" if (world, B) == (1, ENVS_PER_GPU) else
                                 f"weak-scaling line of configs[1]'s per-GPU workload: {B} envs per GPU x {world} GPU(s) = {total_envs} envs"
                                 + (" (configs[3] itself -- 65536 envs at 8 GPUs with the rollout consumer -- is `--config c4`; configs[4] is `--config c5`)" if world > 1 else ""))),
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
    if args.config == "c4":
        line["config"]["rollout_consumer"] = {
            "what": "generalized advantage estimation (gamma 0.99, lambda 0.95, linear value function of the observation) over every K-step chunk of all "
                    "ranks' records as it lands in rank 0's ring (upkie_rollout_gae), stream-ordered between the gather and rank 0's next step",
            "chunks_consumed": consumed["chunks"], "steps_per_chunk": env.gather.chunk, "envs_per_chunk": total_envs + B * world - total_envs,
        }
    if env.gather.collectives:
        # DESIGN.md section 7's model, printed beside the measurement so that a multi-GPU run can be judged against it:
        # ranks share nothing but one asynchronous gather per chunk of K steps, which costs ~27 us of queue time whatever
        # its size (profiles/r01_gather_chunk_sweep.txt, one-rank RCCL group) => N-GPU weak-scaling efficiency
        # ~ K t_step / (K t_step + 27 us), independent of N (0.983 measured at K = 64 on one rank)
        # -- for THIS run's window: a window shorter than a chunk (the driver's --steps 20) ends on one gather of its own
        # steps with nothing left to overlap it with, so its efficiency is lower than the steady one beside it
        k_steps, t_step = env.gather.chunk, launch_us * 1e-6 * (launches / args.steps)
        gathers = -(-args.steps // k_steps)
        line["config"]["predicted_weak_scaling_efficiency"] = args.steps * t_step / (args.steps * t_step + gathers * 27e-6)
        line["config"]["predicted_weak_scaling_efficiency_steady_state"] = k_steps * t_step / (k_steps * t_step + 27e-6)
    if steady is not None:
        s_elapsed, s_ms, s_launches, s_resets = steady
        s_launch_us = s_ms * 1e3 / s_launches
        line["steady_state"] = {
            "value": total_envs * STEADY_STEPS / s_elapsed,
            "unit": "env-steps/s",
            "steps": STEADY_STEPS,
            "warmup": STEADY_WARMUP,
            "ms_per_step": s_elapsed / STEADY_STEPS * 1e3,
            "avg_launch_us": s_launch_us,
            "autoresets_in_timed_region": s_resets,
            "hbm_frac": ALGORITHMIC_BYTES_PER_ENV_STEP * B / (s_launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "note": "SURVEY.md 8d's window (>= 2000 steps after 200 warm-up, autoresets included), one launch per env.step(); timed in this run BEFORE "
                    "the --warmup / --steps region, which continues the same rollout",
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
    env.shutdown(destroy_group=not keep_group)
    if world == 1 and on_gpu and not args.no_secondary:
        del env
        torch.cuda.synchronize()
        def guarded(block, *args):
            # (the blocks beside the contract figure must never cost the line itself: a failure is reported in place)
            try:
                return block(*args)
            except Exception as exc:  # noqa: BLE001
                import traceback

                return {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}

        line["vec_env_api"] = guarded(vec_env_api, B)
        c3 = guarded(secondary_c3)
        if isinstance(c3, dict):
            c3["n50"] = guarded(secondary_c3, 16384, 600, 100, 0, 50)  # the reference's default horizon (missing #6)
        line["secondary"] = {
            "c3": c3,
            "c5_share_torque_law": guarded(secondary_c5_share, "torque"),
            "c5_share_velocity_law": guarded(secondary_c5_share, "velocity"),
            "c5_share_bullet_like": {law: guarded(secondary_c5_share, law, 4096, 600, 100, 0, 200, "bullet_like") for law in ("torque", "velocity")},
            "c5_share_bullet_like_one_lane": {law: guarded(secondary_c5_share, law, 4096, 600, 100, 0, 0, "bullet_like", 1) for law in ("torque", "velocity")},
            "c2_bullet_like_contact_model": guarded(secondary_bullet_like, B),
        }
    return line


if __name__ == "__main__":
    # Native libraries write to file descriptor 1 too (RCCL prints a five-line version banner there when its first
    # communicator comes up, on every rank): the contract's ONE JSON line goes to the real stdout, everything else
    # -- of this process, whatever its rank -- to stderr.
    sys.stdout.flush()
    json_stream = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    main(json_out=json_stream)

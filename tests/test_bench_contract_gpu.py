"""bench.py's one-line JSON contract, on the GPU box."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    env = dict(os.environ, UPKIE_CPU_BASELINE_BUDGET_S="2")
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "10"],
                            capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert result.returncode == 0, result.stderr[-2000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["unit"] == "env-steps/s" and out["n_gpus"] == 1 and out["steps"] == 60 and out["warmup"] == 10
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f32" and out["data"] == "synthetic" and "workload" in out["config"] and "model" not in out["config"]
    assert out["value"] == pytest.approx(out["config"]["total_envs"] * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"]), rel=1e-6)
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"]) and roof["achieved"] > 1.0
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu and cpu["unit"] == "env-steps/s"
    assert out["value"] > 100 * cpu["value"] / cpu["cores"]  # sanity: a GPU is not slower than a CPU core
    # SURVEY 8d's own window beside the contract figure, whatever --steps / --warmup were
    steady = out["steady_state"]
    assert steady["steps"] == 2000 and steady["warmup"] == 200 and steady["unit"] == "env-steps/s"
    assert steady["value"] == pytest.approx(out["config"]["total_envs"] * 2000 / (steady["ms_per_step"] * 1e-3 * 2000), rel=1e-6)
    assert steady["autoresets_in_timed_region"] > 0  # 2000 steps = 10 s per env: the README agent's robots fall about once per 11 s
    assert steady["avg_launch_us"] <= steady["ms_per_step"] * 1e3 * 1.001
    # the secondary BASELINE configs as SURVEY 8d writes them
    sec = out["secondary"]
    assert set(sec) == {"c3", "c5_share_torque_law", "c5_share_velocity_law", "c5_share_bullet_like", "c5_share_bullet_like_one_lane", "c2_bullet_like_contact_model"}
    # round 6: on eight lanes too a joint within reach of its stop is a row of the Bullet-like sweeps; the one-lane kernels (several
    # cached points per tire) beside it
    for law in ("torque", "velocity"):
        assert sec["c5_share_bullet_like"][law]["lanes_per_env"] == 8 and sec["c5_share_bullet_like_one_lane"][law]["lanes_per_env"] == 1
        assert sec["c5_share_bullet_like"][law]["census"]["env_substeps_with_a_joint_at_its_stop"] < 1e-3
    assert sec["c3"]["n50"]["horizon"] == 50 and sec["c3"]["n50"]["us_per_step"] > 0  # the reference's default horizon beside BASELINE's N = 16
    for law in ("torque", "velocity"):
        block = sec["c5_share_bullet_like"][law]
        assert block["contact_model"] == "bullet_like" and block["us_per_step"] > 0, block
    bl = sec["c2_bullet_like_contact_model"]  # the fidelity option beside the default model on the same mapping
    assert bl["bullet_like"]["lanes_per_env"] == 8 and bl["bullet_like_one_lane"]["lanes_per_env"] == 1 and bl["default_one_lane"]["lanes_per_env"] == 1
    assert bl["bullet_like"]["us_per_step"] > bl["default"]["us_per_step"] > 0
    assert bl["bullet_like_one_lane"]["us_per_step"] > bl["default_one_lane"]["us_per_step"] > 0
    # the public path: env.step(policy(obs)) of the vector env, NEXT_STEP and SAME_STEP
    api = out["vec_env_api"]
    for mode in ("next_step", "same_step"):
        assert api[mode]["python_loop_us_per_env_step"] > 0 and api[mode]["lanes_per_env"] == 8 and api[mode]["episodes"] > 0
        assert 0 < api[mode]["python_loop_one_launch_policy_us_per_env_step"] < api[mode]["python_loop_us_per_env_step"]  # the same policy as ONE kernel
    assert 0 < api["next_step"]["python_loop_policy_in_the_step_launch_us_per_env_step"] < api["next_step"]["python_loop_us_per_env_step"]  # env.step_linear_policy
    assert api["same_step"]["python_loop_policy_in_the_step_launch_us_per_env_step"] is None
    assert sec["c3"]["envs"] == 16384 and "resampled every 400 steps" in sec["c3"]["config"] and sec["c3"]["algorithmic_bytes_per_env_step"] == 554
    for key in ("c5_share_torque_law", "c5_share_velocity_law"):
        c5 = sec[key]
        assert c5["envs"] == 4096 and c5["algorithmic_bytes_per_env_step"] == 630 and "drawn on device" in c5["config"]
        assert c5["episodes"] > 4096  # pushed robots do fall and restart
        assert 0 < c5["hbm_frac"] < 1 and c5["us_per_step"] >= c5["device_us_per_step"] * 0.999
        assert c5["census"]["sweep_cap_hits"] >= 0


def test_bench_under_torchrun_through_rccl_on_one_rank():
    """The driver's N > 1 launch line (`python -m torch.distributed.run
    --nproc-per-node N ... bench.py --gpus N`) with N = 1 and the collective
    path forced on: RCCL init with `device_id`, the asynchronous chunked
    `dist.gather` into the rollout ring, barrier, MAX / SUM all-reduces and the
    shutdown all run on the GPU (a one-GPU box cannot host two ranks)."""
    env = dict(os.environ, UPKIE_CPU_BASELINE_BUDGET_S="1", UPKIE_FORCE_PROCESS_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "100", "--warmup", "20"]
    result = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 1e6
    assert out["config"]["gather"].startswith("RCCL gather")


def test_rccl_gather_delivers_every_step_to_the_ring():
    """RolloutGather over a one-rank RCCL group: what reaches the rollout ring
    through the collectives equals what a rank without collectives writes."""
    code = r'''
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from upkie_amd.distributed import ShardedPendulum
from upkie_amd import abi
def run(collectives):
    cfg = abi.default_sim_config(512, frequency=200.0, seed=3)
    cfg.rand_pitch = 0.1; cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
    env = ShardedPendulum(cfg, device="cuda:0", horizon=32, chunk=8, collectives=collectives)
    env.reset()
    for _ in range(29):  # not a multiple of the chunk: flush ships the partial one
        env.step_agent()
    env.flush(); env.barrier()
    return torch.stack([env.gather.last(k).clone() for k in range(24)]), env.total_resets(), env.max_over_ranks(1.5)
a, ra, ma = run(True)
b, rb, mb = run(False)
assert torch.equal(a, b) and ra == rb and ma == mb == 1.5, (float((a - b).abs().max()), ra, rb)
assert float(a.abs().sum()) > 0
dist.destroy_process_group()
print("RCCL_GATHER_OK")
'''
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    result = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert result.returncode == 0 and "RCCL_GATHER_OK" in result.stdout, (result.stdout[-1000:], result.stderr[-3000:])


def test_bench_config_c4_runs_the_rollout_consumer_on_the_device():
    """`bench.py --config c4` (BASELINE configs[3]: 8192 envs per GPU, every chunk of gathered records consumed on rank 0
    by generalized advantage estimation) on one GPU: the consumer's kernel runs inside the timed work, the line says
    which BASELINE config it is."""
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4", "--steps", "200", "--warmup", "20", "--no-secondary",
                             "--no-cpu-baseline", "--no-steady-state", "--no-fused"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert result.returncode == 0, result.stderr[-2000:]
    out = json.loads([line for line in result.stdout.splitlines() if line.startswith("{")][0])
    cfg = out["config"]
    assert cfg["envs_per_gpu"] == 8192 and cfg["baseline_config"].startswith("configs[3]") and cfg["lanes_per_env"] == 8
    assert cfg["rollout_consumer"]["chunks_consumed"] >= 3 and cfg["rollout_consumer"]["steps_per_chunk"] == 64
    assert out["value"] > 1e8  # (16.5 us per step of 8192 envs is 5e8; the consumer costs a few per cent of it)

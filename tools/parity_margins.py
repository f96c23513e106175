"""Measured HIP-vs-oracle errors of the regimes the GPU parity tests hold loosely (tire slip, torque saturation,
fall-threshold crossings): the numbers the tolerances in tests/test_parity_gpu.py and tests/test_baseline_configs_gpu.py
are set from. Usage (GPU box): python tools/parity_margins.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tests.helpers import make_pair, randomized_config, state_errors
from upkie_amd import abi
from upkie_amd.model.model import Model

def q(name, err):
    err = np.asarray(err, dtype=np.float64).ravel()
    print(f"{name:58s} median {np.median(err):.2e}  p97 {np.quantile(err, 0.97):.2e}  p99 {np.quantile(err, 0.99):.2e}  p99.9 {np.quantile(err, 0.999):.2e}  max {err.max():.2e}")

# gyropod, 5 steps of random +-1.5 m/s commands (test_gyropod_step_matches_oracle)
oracle, sim = make_pair(256, seed=5); oracle.reset(); sim.reset(); rng = np.random.default_rng(1)
for _ in range(5):
    act = rng.uniform(-1.5, 1.5, (256, 2)).astype(np.float32)
    obs_o, *_ = oracle.step_gyropod(act.astype(np.float64)); obs_h, *_ = sim.step_gyropod(torch.from_numpy(act))
q("gyropod 5 steps: per-env max |obs error|", np.abs(obs_h.cpu().numpy() - obs_o).max(axis=1))
# servos, 3 steps (test_servos_step_matches_oracle)
oracle, sim = make_pair(256, seed=9); oracle.reset(); sim.reset(); rng = np.random.default_rng(2)
act = np.zeros((256, 6, 6), dtype=np.float32)
act[:, :, 0] = rng.uniform(-0.05, 0.05, (256, 6)); act[:, [2, 5], 0] = np.nan
act[:, [2, 5], 1] = rng.uniform(-0.5, 0.5, (256, 2)); act[:, [2, 5], 2] = rng.uniform(-0.2, 0.2, (256, 2))
act[:, :, 3] = rng.uniform(0.5, 1.5, (256, 6)); act[:, :, 4] = rng.uniform(0.5, 1.5, (256, 6)); act[:, :, 5] = 16.0
for _ in range(3):
    obs_o, *_ = oracle.step_servos(act.astype(np.float64)); obs_h, *_ = sim.step_servos(torch.from_numpy(act))
obs_h = obs_h.cpu().numpy()
q("servos 3 steps: per-env max |velocity error| [rad/s]", np.abs(obs_h[:, :, 1] - obs_o[:, :, 1]).max(axis=1))
q("servos 3 steps: per-env max |torque error| [N m]", np.abs(obs_h[:, :, 2] - obs_o[:, :, 2]).max(axis=1))
# closed loop 200 steps (test_closed_loop_200_steps_matches_oracle)
oracle, sim = make_pair(128, seed=7); obs_o = oracle.reset()[:, [1, 0, 4, 3]]; sim.reset(); sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
for _ in range(200):
    obs_o, *_ = oracle.step_pendulum_agent(obs_o); obs_h, *_ = sim.step_pendulum_agent()
d = np.abs(obs_h.cpu().numpy() - obs_o)
for i, n in enumerate(("pitch", "position", "pitch rate", "velocity")): q(f"closed loop 200 steps, 128 envs: |{n} error|", d[:, i])
# autoreset synchrony (test_autoreset...): fall_pitch just above the reset range
cfg = randomized_config(64, seed=2, autoreset=True); cfg.fall_pitch = 0.12
oracle, sim = make_pair(64, cfg=cfg); oracle.reset(); sim.reset(); act = np.zeros(64, dtype=np.float32); sync = np.ones(64, dtype=bool)
for _ in range(300):
    _, _, t_o, _ = oracle.step_pendulum(act.astype(np.float64)); _, _, t_h, _ = sim.step_pendulum(torch.from_numpy(act)); sync &= t_h.cpu().numpy() == t_o
print(f"autoreset: envs whose episodes ended on the same steps over 300 steps: {sync.mean():.3f}")
# C5 share against the oracle, 4096 envs x 10 steps (test_c5_servos_share_4096_envs_matches_oracle)
B = 4096; cfg = randomized_config(B, seed=2); cfg.joint_friction[2] = cfg.joint_friction[5] = 0.1; model = Model().struct
oracle, sim = make_pair(B, cfg=cfg, model=model); sim.randomize_inertias(0.2); oracle.body_inertials = oracle.sample_body_inertials(0.2)
rng = np.random.default_rng(5); angle, norm = rng.uniform(0, 2 * np.pi, B), rng.uniform(0.0, 20.0, B)
force = np.stack([norm * np.cos(angle), norm * np.sin(angle), np.zeros(B)])
oracle.ext_force = force; oracle.ext_point = np.array([0.0, 0.0, -0.1]); sim.set_external_force(torch.from_numpy(force).float(), point=(0.0, 0.0, -0.1))
obs_o = oracle.reset(); sim.reset()
act = np.zeros((B, 6, 6)); act[:, :, 3] = 1.0; act[:, :, 4] = 1.0; act[:, :, 5] = 16.0; act[:, [2, 5], 0] = np.nan; act[:, [2, 5], 4] = 0.0
pitch = obs_o[:, 1]
for _ in range(10):
    act[:, 2, 2] = 10.0 * pitch; act[:, 5, 2] = -10.0 * pitch
    so, *_ = oracle.step_servos(act); sh, *_ = sim.step_servos(torch.from_numpy(act).float())
    st = oracle.state; pitch = np.arcsin(np.clip(2.0 * (st[abi.S_QUAT] * st[abi.S_QUAT + 2] - st[abi.S_QUAT + 3] * st[abi.S_QUAT + 1]), -1, 1))
sh = sh.cpu().numpy(); print("C5 x 10 steps state errors", {k: f"{v:.2e}" for k, v in state_errors(oracle.state, sim.state_numpy()).items()})
q("C5 x 10 steps: |hip / knee angle error|", np.abs(sh[:, [0, 1, 3, 4], 0] - so[:, [0, 1, 3, 4], 0]))
q("C5 x 10 steps: |wheel angle error|", np.abs(sh[:, [2, 5], 0] - so[:, [2, 5], 0]))
q("C5 x 10 steps: |hip / knee velocity error|", np.abs(sh[:, [0, 1, 3, 4], 1] - so[:, [0, 1, 3, 4], 1]))
q("C5 x 10 steps: |wheel velocity error| [rad/s]", np.abs(sh[:, [2, 5], 1] - so[:, [2, 5], 1]))
q("C5 x 10 steps: |wheel torque error| [N m]", np.abs(sh[:, [2, 5], 2] - so[:, [2, 5], 2]))

"""Step time of the Pendulum agent per lane mapping with one feature switched on at a time
(noise, friction, inertia randomisation, pushes, wide initial states): where a mapping pays."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from upkie_amd import abi
from upkie_amd.model.model import Model
from upkie_amd.sim import BatchedSim

B = 4096
def config(noise=False, friction=False, wide=False, limit=False):
    cfg = abi.default_sim_config(B, seed=1)
    cfg.rand_pitch, cfg.rand_x, cfg.rand_omega_y = (0.2, 0.05, 0.3) if wide else (0.1, 0.05, 0.1)
    if wide: cfg.rand_roll = 0.05
    cfg.rand_linvel[0] = 0.1 if wide else 0.05
    cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
    if limit: cfg.max_episode_steps = 1500
    for j in range(6):
        if noise: cfg.torque_control_noise[j] = 0.05
        if friction: cfg.joint_friction[j] = 0.05
    return cfg

variants = {"plain": {}, "noise": dict(noise=True), "friction": dict(friction=True), "wide init": dict(wide=True), "time limit": dict(limit=True),
            "inertia 0.3": dict(inertia=True), "pushes": dict(push=True), "all": dict(noise=True, friction=True, wide=True, limit=True, inertia=True, push=True)}
for name, kw in variants.items():
    row = []
    for lanes in ("2", "8"):
        os.environ["UPKIE_LANES_PER_ENV"] = lanes
        inertia, push = kw.get("inertia", False), kw.get("push", False)
        sim = BatchedSim(config(**{k: v for k, v in kw.items() if k not in ("inertia", "push")}), Model().struct)
        if inertia: sim.randomize_inertias(0.3)
        sim.reset()
        sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
        if push:
            f = (torch.rand((3, B), device=sim.device) * 2 - 1) * torch.tensor([[15.0], [8.0], [5.0]], device=sim.device)
            sim.set_external_force(f, point=(0.0, 0.0, 0.1))
        census = sim.enable_census() if lanes == "8" else None
        for _ in range(100): sim.step_pendulum_agent()
        if census is not None: census.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(600): sim.step_pendulum_agent()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 600
        extra = ""
        if census is not None:
            c = sim.census_counts(); es = B * 5 * 600
            extra = f" [sweeps {c['friction_cone'] / es:.2%} limits {c['joint_limit'] / es:.2%}]"
        row.append(f"lanes {lanes}: {dt * 1e6:6.1f} us{extra}")
    print(f"{name:12s} " + "   ".join(row) + f"   episodes {int(sim.state[40].sum())}", flush=True)

"""Soak run: long randomised rollouts in every mode and lane mapping, checking
that no state ever becomes non-finite and that invariants hold (unit
quaternions, joints within their stops, bounded speeds).
Usage: python tools/soak.py [steps] [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upkie_amd import abi  # noqa: E402
from upkie_amd.model.model import Model  # noqa: E402
from upkie_amd.sim import BatchedSim  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096


def check(sim, tag, k):
    st = sim.state
    bad = ~torch.isfinite(st[:41]).all(dim=0)
    assert not bad.any(), f"{tag}: {int(bad.sum())} non-finite envs at step {k}"
    qn = (st[abi.S_QUAT : abi.S_QUAT + 4] ** 2).sum(0).sqrt()
    assert (qn - 1).abs().max() < 1e-3, f"{tag}: quaternion norm {float(qn.min())}..{float(qn.max())} at step {k}"
    lo = torch.tensor(list(sim.model.joint_lower), device=st.device)
    hi = torch.tensor(list(sim.model.joint_upper), device=st.device)
    q = st[abi.S_Q : abi.S_Q + 6]
    for j in (0, 1, 3, 4):
        assert q[j].min() > lo[j] - 0.1 and q[j].max() < hi[j] + 0.1, f"{tag}: joint {j} outside its stops at step {k}"
    assert st[abi.S_QD : abi.S_QD + 6].abs().max() <= 100.0 + 1e-3
    assert st[abi.S_LINVEL : abi.S_LINVEL + 3].abs().max() < 100.0, f"{tag}: base speed {float(st[abi.S_LINVEL:abi.S_LINVEL+3].abs().max())}"


def config(seed):
    cfg = abi.default_sim_config(B, seed=seed)
    cfg.rand_pitch, cfg.rand_roll, cfg.rand_x, cfg.rand_omega_y = 0.2, 0.05, 0.05, 0.3
    cfg.rand_linvel[0] = 0.1
    cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
    cfg.max_episode_steps = 1500  # time limit kept by the kernel
    for j in range(6):
        cfg.torque_control_noise[j] = 0.05
        cfg.joint_friction[j] = 0.05
    return cfg


for lanes in ("8", "2", "1"):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    t0 = time.time()
    # Pendulum with the README agent, randomised inertias, random pushes renewed every 500 steps
    sim = BatchedSim(config(1), Model().struct)  # URDF model: 13 links behind the 7 bodies, randomised one by one
    sim.randomize_inertias(0.3)
    sim.reset()
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    for k in range(steps):
        if k % 500 == 0:
            f = (torch.rand((3, B), device=sim.device) * 2 - 1) * torch.tensor([[15.0], [8.0], [5.0]], device=sim.device)
            sim.set_external_force(f, point=(0.0, 0.0, 0.1))
        sim.step_pendulum_agent()
        if k % 1000 == 999:
            check(sim, f"pendulum lanes={lanes}", k)
    resets = int(sim.state[abi.S_EPISODE].sum()) - B
    print(f"lanes={lanes} pendulum agent + inertia 0.3 + pushes + noise: {steps} steps ok, {resets} episode resets, {time.time() - t0:.1f} s")
    # the same agent in fused 32-step launches (state carried in registers from step to step)
    sim = BatchedSim(config(3), Model().struct)
    sim.randomize_inertias(0.3)
    o6 = sim.reset()
    prev = torch.zeros((B, 8), device=sim.device)
    prev[:, :4] = o6[:, [1, 0, 4, 3]]
    window = torch.zeros((32, B, 8), device=sim.device)
    for k in range(0, steps, 32):
        sim.rollout_pendulum_records(prev, window)
        prev.copy_(window[31])
        assert torch.isfinite(window).all(), f"rollout lanes={lanes}: non-finite records at step {k}"
        if k % 1024 == 992:
            check(sim, f"rollout lanes={lanes}", k)
    print(f"lanes={lanes} fused rollouts (32 steps per launch): {steps} steps ok, {int(sim.state[abi.S_EPISODE].sum()) - B} episode resets")
    # Gyropod with random commands
    sim = BatchedSim(config(2))
    sim.reset()
    for k in range(steps // 2):
        act = (torch.rand((B, 2), device=sim.device) * 2 - 1) * torch.tensor([2.0, 1.5], device=sim.device)
        sim.step_gyropod(act)
        if k % 1000 == 999:
            check(sim, f"gyropod lanes={lanes}", k)
    print(f"lanes={lanes} gyropod random commands: {steps // 2} steps ok, {int(sim.state[abi.S_EPISODE].sum()) - B} resets")
    # Servos: random torques and position targets, no termination: robots flail, fall, fold
    cfg = config(3)
    cfg.autoreset_mode = abi.AUTORESET_DISABLED
    sim = BatchedSim(cfg)
    sim.reset()
    scale = torch.tensor([16.0, 16.0, 1.7, 16.0, 16.0, 1.7], device=sim.device)
    act = torch.zeros((B, 6, 6), device=sim.device)
    for k in range(steps // 2):
        if k % 20 == 0:
            act[:, :, 0] = (torch.rand((B, 6), device=sim.device) * 2 - 1) * 3.0
            act[:, :, 0] = torch.where(torch.rand((B, 6), device=sim.device) < 0.3, torch.full_like(act[:, :, 0], float("nan")), act[:, :, 0])
            act[:, :, 1] = (torch.rand((B, 6), device=sim.device) * 2 - 1) * 10.0
            act[:, :, 2] = (torch.rand((B, 6), device=sim.device) * 2 - 1) * scale
            act[:, :, 3] = torch.rand((B, 6), device=sim.device) * 2.0
            act[:, :, 4] = torch.rand((B, 6), device=sim.device) * 2.0
            act[:, :, 5] = torch.rand((B, 6), device=sim.device) * scale
        sim.step_servos(act)
        if k % 1000 == 999:
            check(sim, f"servos lanes={lanes}", k)
    print(f"lanes={lanes} servos random commands, no resets: {steps // 2} steps ok")
# The Bullet-like contact model (upkie_sim_set_contact_manifold): the Pendulum agent with pushes and noise on eight lanes
# and on one, then Servos with random commands on one lane -- flailing robots go through the general row list: joints
# at their stops and tire contacts in ONE fixed-sweep solve -- a tenth of the steps (the model is 4-7 x slower).
for lanes in ("8", "1"):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    sim = BatchedSim(config(5), Model().struct)
    sim.randomize_inertias(0.3)
    sim.use_bullet_like_contacts()
    assert sim.lanes_per_env == int(lanes)
    sim.reset()
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    n = max(steps // 10, 200)
    for k in range(n):
        if k % 500 == 0:
            f = (torch.rand((3, B), device=sim.device) * 2 - 1) * torch.tensor([[15.0], [8.0], [5.0]], device=sim.device)
            sim.set_external_force(f, point=(0.0, 0.0, 0.1))
        sim.step_pendulum_agent()
        if k % 200 == 199:
            check(sim, f"bullet-like pendulum lanes={lanes}", k)
            m = sim.contact_manifold.reshape(2, 4, 8, B)
            assert torch.isfinite(m).all()
            several = m[:, :, 7].sum(dim=1) > 1  # [tire, env]: only a robot flat on its side caches several points on a tire
            if bool(several.any()):
                q = sim.state[abi.S_QUAT : abi.S_QUAT + 4]
                un = torch.hypot(2 * (q[1] * q[3] - q[2] * q[0]), 1 - 2 * (q[1] ** 2 + q[2] ** 2))  # |world z projected on the wheel plane|
                assert lanes == "1" and float(un[several.any(dim=0)].max()) < 0.5, "several cached points on a tire of a robot that is not lying on its side"
    print(f"lanes={lanes} Bullet-like contacts, pendulum agent + inertia 0.3 + pushes + noise: {n} steps ok, {int(sim.state[abi.S_EPISODE].sum()) - B} episode resets")
# (round 5: Servos steps run the eight-lane Bullet-like kernel too; round 6: limit rows and tire contacts in ONE fixed-sweep solve on
# both mappings)
for lanes in ("8", "1"):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    cfg = config(6)
    cfg.autoreset_mode = abi.AUTORESET_DISABLED
    sim = BatchedSim(cfg)
    sim.use_bullet_like_contacts()
    assert sim.lanes_per_env_of(abi.OBSERVATION_SERVOS) == int(lanes)
    sim.reset()
    scale = torch.tensor([16.0, 16.0, 1.7, 16.0, 16.0, 1.7], device=sim.device)
    act = torch.zeros((B, 6, 6), device=sim.device)
    n = max(steps // 20, 100)
    for k in range(n):
        if k % 20 == 0:
            act[:, :, 0] = (torch.rand((B, 6), device=sim.device) * 2 - 1) * 3.0
            act[:, :, 1] = (torch.rand((B, 6), device=sim.device) * 2 - 1) * 10.0
            act[:, :, 2] = (torch.rand((B, 6), device=sim.device) * 2 - 1) * scale
            act[:, :, 3] = torch.rand((B, 6), device=sim.device) * 2.0
            act[:, :, 4] = torch.rand((B, 6), device=sim.device) * 2.0
            act[:, :, 5] = torch.rand((B, 6), device=sim.device) * scale
        sim.step_servos(act)
        if k % 50 == 49:
            check(sim, f"bullet-like servos lanes={lanes}", k)
            assert torch.isfinite(sim.contact_manifold).all()
    print(f"lanes={lanes} Bullet-like contacts, servos random commands, no resets: {n} steps ok")
# Round 6, the non-finite guard (include/upkie_hip.h, "Non-finite commands and states"): NaN / +-Inf / 1e30 written into a random word of
# the action of 1 % of the envs at EVERY step (every word of a Servos action: velocity, feedforward torque, gain scales, maximum torque,
# position; the ground / yaw velocity of the Gyropod family), NaN forces on a few envs for a while: no value that is not finite may leave
# a step, poisoned commands are replaced and counted, poisoned states end their episode and are re-initialised by the autoreset.
POISON = torch.tensor([float("nan"), float("inf"), float("-inf"), 1e30, -1e30])
for lanes in ("8", "2", "1"):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    for kind, width in (("pendulum", 1), ("gyropod", 2), ("servos", 36)):
        sim = BatchedSim(config(7))
        sim.reset()
        n = max(steps // 5, 400)
        force = torch.zeros((3, B), device=sim.device)
        sim.set_external_force(force)
        poison = POISON.to(sim.device)
        for k in range(n):
            if kind == "servos":
                act = torch.zeros((B, 6, 6), device=sim.device)
                act[:, :, 0] = float("nan")
                act[:, :, 1] = (torch.rand((B, 6), device=sim.device) * 2 - 1) * 3.0
                act[:, :, 3:5] = 1.0
                act[:, :, 5] = 16.0
            else:
                act = (torch.rand((B, width), device=sim.device) * 2 - 1) * 0.5
            flat = act.view(B, -1)
            hit = torch.nonzero(torch.rand(B, device=sim.device) < 0.01).flatten()
            flat[hit, torch.randint(0, width, (len(hit),), device=sim.device)] = poison[torch.randint(0, 5, (len(hit),), device=sim.device)]
            if k == n // 2:
                force[1, :8] = float("nan")  # eight envs pushed by a NaN force for 50 steps: caught and re-initialised every step
            if k == n // 2 + 50:
                force.zero_()
            obs, _, term, _ = {"pendulum": sim.step_pendulum, "gyropod": sim.step_gyropod, "servos": sim.step_servos}[kind](act.view(B) if width == 1 else act)
            assert torch.isfinite(obs).all(), f"{kind} lanes={lanes}: a non-finite observation left step {k}"
            if k % 100 == 99:
                assert torch.isfinite(sim.state).all(), f"{kind} lanes={lanes}: non-finite state after step {k}"
        counts = sim.guard_counts()
        assert counts["commands_replaced"] > 0 and counts["states_replaced"] >= 8 * 20, counts  # (a guarded env spends its next step in the autoreset, which applies no force)
        print(f"lanes={lanes} {kind}: {n} steps with 1 % of the actions poisoned and a NaN force on 8 envs for 50 steps: finite throughout, {counts}")
os.environ.pop("UPKIE_LANES_PER_ENV", None)
print("soak passed")

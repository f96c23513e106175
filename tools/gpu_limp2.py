"""Bisect the pair-mapping blow-up found by tools/gpu_limp.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upkie_amd import abi  # noqa: E402
from upkie_amd.model.default_model import default_model  # noqa: E402
from upkie_amd.sim import BatchedSim  # noqa: E402

B = 64
act = None


def make(lanes, limits=True, substeps=5):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    cfg = abi.default_sim_config(B, frequency=200.0 if substeps == 5 else 1000.0, nb_substeps=substeps)
    cfg.autoreset_mode = abi.AUTORESET_DISABLED
    model = default_model()
    model.enforce_joint_limits = 1 if limits else 0
    return BatchedSim(cfg, model)


act = torch.zeros((B, 6, 6), device="cuda:0")
act[:, :, 0] = float("nan")
act[:, :, 3] = 1.0
act[:, :, 5] = 16.0
ref = make("2")
ref.reset()
states = []
for k in range(400):
    states.append(ref.state.clone())
    ref.step_servos(act)
    if not torch.isfinite(ref.state[:25]).all():
        print("pair trajectory: non-finite after step", k)
        break
# replay every recorded state through one step of each mapping / setting
for label, lanes, limits in (("pair+limits", "2", True), ("pair, no limits", "2", False), ("single+limits", "1", True)):
    sim = make(lanes, limits)
    first = None
    for k, st in enumerate(states):
        sim.state.copy_(st)
        sim.step_servos(act)
        if not torch.isfinite(sim.state[:25]).all():
            first = k
            break
    print(label, "first non-finite when stepping from recorded state", first)
    if first is not None and label == "pair+limits":
        bad = first
bad = len(states) - 1 if "bad" not in dir() else bad
os.makedirs('gpurun_out', exist_ok=True)
torch.save(states[bad].cpu(), 'gpurun_out/bad_state.pt')
# substep resolution from the bad state
sim1 = make("2", True, substeps=1)
sim0 = make("1", True, substeps=1)
sim1.state.copy_(states[bad])
sim0.state.copy_(states[bad])
act1 = act.clone()
for sub in range(5):
    before = sim1.state[:25, 0].clone()
    sim1.step_servos(act1)
    sim0.step_servos(act1)
    a, b = sim1.state[:25, 0], sim0.state[:25, 0]
    print("substep", sub, "pair finite", bool(torch.isfinite(a).all()), "max |pair - single|", float((a - b).abs().max()))
    if not torch.isfinite(a).all():
        torch.save(dict(before=before.cpu(), sub=sub), 'gpurun_out/bad_substate.pt')
        print(" state before:", [round(float(v), 5) for v in before])
        print(" single after:", [round(float(v), 5) for v in b])
        break

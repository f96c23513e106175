"""Collapse test: every joint limp, robots fall and fold into their joint
stops. Reports non-finite states per lane mapping. python tools/gpu_limp.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upkie_amd import abi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for lanes in ("1", "2", "8"):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    import upkie_amd.envs as envs

    env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, autoreset_mode="disabled")
    env.reset(seed=0)
    act = env.get_neutral_action()
    act[:, :, 0] = float("nan")
    act[:, :, 4] = 0.0
    first = None
    prev = None
    for k in range(400):
        prev = env.sim.state.clone()
        env.sim.step_servos(act)
        bad = ~torch.isfinite(env.sim.state[:25]).all(dim=0)
        if bad.any() and first is None:
            first = k
            e = int(torch.nonzero(bad)[0])
            print(f"lanes={lanes}: first non-finite state at step {k}, {int(bad.sum())} envs; env {e} before the step:")
            print("  ", [round(float(v), 4) for v in prev[:25, e]])
            print("  after:", [round(float(v), 4) for v in env.sim.state[:25, e]])
            break
    if first is None:
        st = env.sim.state
        print(f"lanes={lanes}: finite through 400 steps; z mean {st[abi.S_POS + 2].mean().item():.3f}")
    env.close()

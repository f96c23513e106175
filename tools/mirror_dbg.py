import sys; sys.path.insert(0, '/root/repo')
import torch
from tests import test_full_size_gpu as T
import types
src = open('/root/repo/tests/test_full_size_gpu.py').read()
# re-run the body with prints
from upkie_amd import abi
from upkie_amd.sim import BatchedSim
from tests.helpers import randomized_config
B = 4096
cfg = randomized_config(B, seed=5); cfg.rand_roll = 0.05
sim = BatchedSim(cfg); sim.reset(); mirrored = BatchedSim(cfg)
m = sim.state.clone()
m[abi.S_POS + 1] *= -1; m[abi.S_LINVEL + 1] *= -1; m[abi.S_QUAT + 1] *= -1; m[abi.S_QUAT + 3] *= -1; m[abi.S_ANGVEL + 0] *= -1; m[abi.S_ANGVEL + 2] *= -1
for word in (abi.S_Q, abi.S_QD):
    left = m[word:word+3].clone(); m[word:word+3] = -m[word+3:word+6]; m[word+3:word+6] = -left
legl = m[abi.S_LEGREF:abi.S_LEGREF+2].clone(); m[abi.S_LEGREF:abi.S_LEGREF+2] = -m[abi.S_LEGREF+2:abi.S_LEGREF+4]; m[abi.S_LEGREF+2:abi.S_LEGREF+4] = -legl
mirrored.state.copy_(m)
act = torch.linspace(-0.2, 0.2, B, device="cuda:0")
for k in range(20):
    o1, *_ = sim.step_pendulum(act); o2, *_ = mirrored.step_pendulum(act)
    err = (o1 - o2).abs().max(dim=1).values
    print(k, "frac<2e-4", float((err < 2e-4).float().mean()), "max", float(err.max()), "n>1e-2", int((err > 1e-2).sum()))

"""The C2 workload under the Bullet-like contact model, 1000 steps at 4096 envs on the eight-lane kernel (for
`rocprofv3 --kernel-trace --stats -- python tools/profile_bullet_like.py`: profiles/r05_kernel_stats_bullet_like.csv),
then the C5 share under the model (Servos steps, README law, pushes), 600 steps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from upkie_amd.sim import BatchedSim  # noqa: E402

sim = BatchedSim(bench.make_config(4096))
sim.use_bullet_like_contacts()
o6 = sim.reset()
sim.obs4.copy_(o6[:, [1, 0, 4, 3]])
for _ in range(1100):
    sim.step_pendulum_agent()
torch.cuda.synchronize()
sim.close()
if "c2" not in sys.argv[1:]:  # (`c2`: the Pendulum loop alone, for counter passes)
    print(bench.secondary_c5_share("velocity", 4096, 600, 100, 0, 0, "bullet_like")["us_per_step"])

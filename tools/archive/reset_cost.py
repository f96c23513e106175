"""What a restart inside a launch costs: the headline loop (4096 envs, C2, `ShardedPendulum.step_agent()`, NEXT_STEP autoreset)
timed window by window -- before any robot falls (steps 200-1400), while the first episodes end (1500-2200), and with
episodes spread over their whole life (2200-6200: about two restarts per launch, the regime bench.py's `value` is timed in).
Usage (GPU box): python tools/reset_cost.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from upkie_amd.distributed import ShardedPendulum  # noqa: E402

env = ShardedPendulum(bench.make_config(4096), device="cuda:0", chunk=bench.GATHER_CHUNK)
env.reset()
k = 0
for name, stop in (("warm-up", 200), ("no restarts yet", 1400), ("-", 1500), ("first episodes end", 2200), ("episodes spread out", 4200), ("episodes spread out", 6200)):
    before = env.total_resets()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = stop - k
    for _ in range(n):
        env.step_agent()
    z.record()
    torch.cuda.synchronize()
    k = stop
    resets = env.total_resets() - before
    print(f"steps {stop - n:5d}-{stop:5d}  {name:22s} {a.elapsed_time(z) * 1e3 / n:6.2f} us per env.step()   {resets / n:5.2f} restarts per launch", flush=True)

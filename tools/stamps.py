"""Where the time of a headline launch goes that is not substeps: the s_memtime stamps of a PROBE build of the library
(`python tools/build_variant.py ab/libupkie_hip_stamps.so -DUPKIE_STAMPS`, then `UPKIE_HIP_LIBRARY=ab/libupkie_hip_stamps.so
python tools/stamps.py`): the bench's kernel (step_kernel_octet<MODE_PENDULUM_AGENT>, 4096 envs, 512 wavefronts, records entry
point), stamps of every wavefront of 200 launches in the calm phase (no robot has fallen yet) behind the census words.
Shader clock 2.4 GHz (profiles/r02_kernarg_latency.txt); the 100 MHz wall clock at exit orders the wavefronts of a launch.
Output: profiles/r06_fixed_cost.txt."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
from upkie_amd import abi
from upkie_amd.sim import BatchedSim

B = int(sys.argv[1]) if len(sys.argv) > 1 else bench.ENVS_PER_GPU
GHZ = 2.4
sim = BatchedSim(bench.make_config(B))
assert sim.lanes_per_env == 8
waves = -(-B * 8 // 64)
buf = torch.zeros(abi.CENSUS_WORDS + 16 * waves, dtype=torch.int32, device=sim.device)
sim.census = buf
sim._check(sim._lib.upkie_sim_set_census(sim._handle, buf.data_ptr()))
o6 = sim.reset()
prev = torch.zeros((B, 8), device=sim.device)
prev[:, :4] = o6[:, [1, 0, 4, 3]]
rec = torch.zeros((2, B, 8), device=sim.device)
rec[1].copy_(prev)
for k in range(100):
    sim.step_pendulum_records(rec[(k + 1) % 2], rec[k % 2])
torch.cuda.synchronize()
names = ["entry -> prologue done (settings, lane constants, state, action map; loads landed)", "first substep", "substeps 2-5",
         "guard, observation, flags", "stores issued", "stores acknowledged"]
rows, spans, exits = [], [], []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
launch_us = []
for k in range(200):
    e0.record()
    sim.step_pendulum_records(rec[(k + 1) % 2], rec[k % 2])
    e1.record()
    torch.cuda.synchronize()
    launch_us.append(e0.elapsed_time(e1) * 1e3)
    st = buf[abi.CENSUS_WORDS:].cpu().numpy().view(np.uint64).reshape(waves, 8)
    if (st[:, 0] == 0).any():
        raise SystemExit("no stamps: is UPKIE_HIP_LIBRARY a -DUPKIE_STAMPS build?")
    d = np.diff(st[:, :7].astype(np.int64), axis=1)  # [waves, 6] cycles
    rows.append(d)
    spans.append((st[:, 6].max() - st[:, 0].min(), st[:, 0].max() - st[:, 0].min(), (st[:, 7].max() - st[:, 7].min()) * 10.0))
rows = np.stack(rows).astype(np.float64)  # [launches, waves, 6]
mean = rows.mean(axis=(0, 1))
slow = rows.sum(axis=2).max(axis=1).mean()
print(f"# tools/stamps.py, {B} envs, {waves} wavefronts, 200 launches of the records entry point with one synchronisation each (calm phase); probe build -DUPKIE_STAMPS")
print(f"# (the stamps cost the probe kernel two s_waitcnt 0 and 16 scalar registers: its in-wave total is an upper bound of the shipped kernel's)")
for n, c in zip(names, mean):
    print(f"  {n:88s} {c:8.0f} cycles  {c / GHZ / 1e3:6.2f} us")
total = mean[:5].sum()
print(f"  {'in the wave, entry to stores issued (mean over wavefronts)':88s} {total:8.0f} cycles  {total / GHZ / 1e3:6.2f} us")
print(f"  {'slowest wavefront of a launch, entry to stores acknowledged (mean over launches)':88s} {slow:8.0f} cycles  {slow / GHZ / 1e3:6.2f} us")
sp = np.array(spans, dtype=np.float64)
# (the shader clocks of different XCDs / CUs are not aligned: stamps of different wavefronts cannot be subtracted; the 100 MHz wall clock can)
print(f"  spread of the wavefronts' exits within a launch, 100 MHz wall clock (a launch from an idle queue: one synchronisation per launch here) {sp[:, 2].mean() / 1e3:6.2f} us")
print(f"  launch to launch with a synchronisation each (events on the stream): {np.mean(launch_us):6.2f} us")

"""Where do 0.7 us go between `upkie_sim_step_pendulum_agent` (separate obs / reward / flag arrays: 14.4 us per launch at
4096 envs) and the records entry point the headline line times (`upkie_sim_step_pendulum_agent_records`: 15.2 us)?
The same kernel instantiation, the same handle settings, one process per row: (a) the plain entry, (b) records into two
alternating buffers, (c) records into a ring of 128 slots (what `RolloutGather` hands out on one rank), (d) ring of 4 slots,
(e) the bench's own `ShardedPendulum.step_agent()` loop.
Usage (GPU box): python tools/ab_records_entry.py > gpurun_out/r05_ab_records_entry.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import torch, bench
from upkie_amd.sim import BatchedSim
from upkie_amd.distributed import ShardedPendulum
mode, B = sys.argv[2], 4096
if mode == "sharded":
    env = ShardedPendulum(bench.make_config(B), device="cuda:0", chunk=bench.GATHER_CHUNK)
    env.reset()
    step = env.step_agent
else:
    sim = BatchedSim(bench.make_config(B)); o6 = sim.reset(); sim.obs4.copy_(o6[:, [1, 0, 4, 3]])
    if mode == "plain":
        step = sim.step_pendulum_agent
    else:
        n = int(mode)
        ring = torch.zeros((n, B, 8), device="cuda:0")
        ring[n - 1, :, :4] = sim.obs4
        ptrs = [ring[i].data_ptr() for i in range(n)]
        state = {"k": 0}
        raw = sim.step_pendulum_records_raw
        def step():
            k = state["k"]
            raw(ptrs[(k - 1) % n], ptrs[k % n])
            state["k"] = k + 1
for _ in range(200): step()
out, host = [], []
for rep in range(3):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    for _ in range(400): step()
    z.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    out.append(a.elapsed_time(z) * 1e3 / 400)
    host.append((t1 - t0) * 1e6 / 400)
print(" ".join(f"{t:.2f}" for t in out), "|", " ".join(f"{t:.2f}" for t in host))
'''


def main():
    print("4096 envs, C2 workload, steps 200..1400 of the episodes; us per env.step(): device (HIP events around 400 launches; median, min of 9) | host time to ISSUE one launch (median)")
    rows = {m: ([], []) for m in ("plain", "2", "4", "128", "sharded")}
    for _ in range(3):
        for mode in rows:
            out = subprocess.run([sys.executable, "-c", CHILD, ROOT, mode], capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(mode, out.stderr[-400:])
                continue
            dev, host = out.stdout.split("|")
            rows[mode][0].extend(float(x) for x in dev.split())
            rows[mode][1].extend(float(x) for x in host.split())
    names = {"plain": "upkie_sim_step_pendulum_agent (obs4 / reward / flags arrays)", "2": "records, two alternating buffers", "4": "records, ring of 4 slots",
             "128": "records, ring of 128 slots (16.8 MB)", "sharded": "ShardedPendulum.step_agent() (bench.py's loop)"}
    for mode, (dev, host) in rows.items():
        if dev:
            print(f"{names[mode]:62s} {sorted(dev)[len(dev) // 2]:6.2f} ({min(dev):.2f})  | {sorted(host)[len(host) // 2]:6.2f}", flush=True)


if __name__ == "__main__":
    main()

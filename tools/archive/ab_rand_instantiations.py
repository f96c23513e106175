"""Do the RAND = false instantiations of the one- and two-lane step kernels earn their place in the library?
(prune instantiations to those a measured >= 1 % gain justifies.) The same library, the same
workload (C2: Upkie-Pendulum, PD agent on device, no inertial records, no forces) at the batch sizes those mappings
serve, with the plain instantiation and -- UPKIE_ALWAYS_RAND_KERNELS=1 -- with the randomisation-capable one handed
null pointers; each in its own process, interleaved, three rounds.
Usage (GPU box): python tools/ab_rand_instantiations.py > gpurun_out/r05_ab_rand_instantiations.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, bench
from upkie_amd.sim import BatchedSim
B = int(sys.argv[2])
sim = BatchedSim(bench.make_config(B)); sim.reset(); sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
for _ in range(100): sim.step_pendulum_agent()
out = []
for rep in range(3):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300): sim.step_pendulum_agent()
    z.record(); torch.cuda.synchronize()
    out.append(a.elapsed_time(z) * 1e3 / 300)
print(sim.lanes_per_env, " ".join(f"{t:.2f}" for t in out))
'''


def main():
    rounds = 3
    print("C2 workload, us per env.step() launch (three windows of 300 steps per process); plain = RAND false instantiation, capable = RAND true with null pointers")
    for B, forced in ((32768, None), (65536, None), (262144, None), (4096, "2"), (4096, "1")):
        rows = {"plain": [], "capable": []}
        lanes = "?"
        for _ in range(rounds):
            for name, flag in (("plain", "0"), ("capable", "1")):
                env = dict(os.environ, UPKIE_ALWAYS_RAND_KERNELS=flag)
                if forced:
                    env["UPKIE_LANES_PER_ENV"] = forced
                out = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(B)], capture_output=True, text=True, env=env, timeout=300)
                if out.returncode != 0:
                    print(out.stderr[-500:])
                    continue
                parts = out.stdout.split()
                lanes = parts[0]
                rows[name] += [float(x) for x in parts[1:]]
        med = {k: sorted(v)[len(v) // 2] if v else float("nan") for k, v in rows.items()}
        print(f"{B:7d} envs, {lanes} lane(s) per env: plain {med['plain']:.2f} (min {min(rows['plain']):.2f})   capable {med['capable']:.2f} (min {min(rows['capable']):.2f})   "
              f"capable / plain = {med['capable'] / med['plain']:.4f}")


if __name__ == "__main__":
    main()

"""A/B of builds of the library on ONE box: the bench workload's step (4096 envs, fused agent, one launch per
step), each build in its own process, interleaved, several rounds (box-to-box differences are larger than the
few-percent effects this is for). Usage: python tools/ab_step.py libA.so libB.so ... [--rounds N]"""
import os, subprocess, sys

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), ".."))
import torch, bench
from upkie_amd.sim import BatchedSim
sim = BatchedSim(bench.make_config(4096)); sim.reset(); sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
for _ in range(100): sim.step_pendulum_agent()
out = []
for rep in range(3):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(400): sim.step_pendulum_agent()
    z.record(); torch.cuda.synchronize()
    out.append(a.elapsed_time(z) * 1e3 / 400)
print(" ".join(f"{t:.2f}" for t in out))
'''

args = sys.argv[1:]
rounds = 3
if "--rounds" in args:
    i = args.index("--rounds")
    rounds = int(args[i + 1])
    del args[i:i + 2]
libs = args
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, UPKIE_HIP_LIBRARY=os.path.abspath(lib))
        res = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print(f"round {r} {os.path.basename(lib):28s} us/step (steps 100-500, 500-900, 900-1300): {res.stdout.strip() or res.stderr[-300:]}", flush=True)

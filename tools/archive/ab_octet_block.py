"""Lanes per workgroup of the eight-lane step kernel: one wavefront (64, shipped) against two and four wavefronts of one
workgroup on one CU (-DUPKIE_OCTET_BLOCK=128 / 256). Builds the variant libraries beside the shipped one
(`--build`, here or on the GPU box: ~35 s each on eight cores), then times the headline launch (C2, 4096 envs, one launch
per env.step()) and the 8192- and 16384-env launches with each, one process per measurement, interleaved, three rounds.
Usage: python tools/ab_octet_block.py --build; (GPU box) python tools/ab_octet_block.py > gpurun_out/r05_ab_octet_block.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANTS = {64: None, 128: "libupkie_hip_block128.so", 256: "libupkie_hip_block256.so"}
CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, bench
from upkie_amd.sim import BatchedSim
B = int(sys.argv[2])
sim = BatchedSim(bench.make_config(B)); sim.reset(); sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
for _ in range(200): sim.step_pendulum_agent()
out = []
for rep in range(3):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(400): sim.step_pendulum_agent()
    z.record(); torch.cuda.synchronize()
    out.append(a.elapsed_time(z) * 1e3 / 400)
print(sim.lanes_per_env, " ".join(f"{t:.2f}" for t in out))
'''


def build():
    from upkie_amd import lib

    shipped = lib.LIB_PATH
    for block, name in VARIANTS.items():
        if name is None:
            continue
        lib.LIB_PATH = os.path.join(os.path.dirname(shipped), name)
        flags = list(lib.HIPCC_FLAGS)
        lib.HIPCC_FLAGS.append(f"-DUPKIE_OCTET_BLOCK={block}")
        try:
            print(lib.build(force=True))
        finally:
            lib.HIPCC_FLAGS[:] = flags
            lib.LIB_PATH = shipped


def main():
    if "--build" in sys.argv:
        return build()
    lib_dir = os.path.join(ROOT, "upkie_amd", "_lib")
    print("C2 workload, us per env.step() launch (three windows of 400 steps per process, three processes per cell, interleaved): median (min)")
    for B in (4096, 8192, 16384):
        cells = {b: [] for b in VARIANTS}
        for _ in range(3):
            for block, name in VARIANTS.items():
                env = dict(os.environ)
                if name:
                    env["UPKIE_HIP_LIBRARY"] = os.path.join(lib_dir, name)
                out = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(B)], capture_output=True, text=True, env=env, timeout=300)
                if out.returncode != 0:
                    print(block, out.stderr[-400:])
                    continue
                parts = out.stdout.split()
                assert parts[0] == "8"
                cells[block] += [float(x) for x in parts[1:]]
        print(f"{B:6d} envs: " + "   ".join(f"{b:3d} lanes per workgroup {sorted(v)[len(v) // 2]:.2f} ({min(v):.2f})" for b, v in cells.items() if v), flush=True)


if __name__ == "__main__":
    main()

import os, sys, json
sys.path.insert(0, '/root/repo')
import bench
r = bench.secondary_c3(steps=1000, warmup=200)
print(f"c3 N=16 16384 envs: {r['us_per_step']:.2f} us", flush=True)

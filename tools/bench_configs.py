"""Secondary workloads of BASELINE.json as SURVEY.md section 8d writes them
(configs[2] = C3, one GPU's share of configs[4] = C5): the same functions
bench.py puts under "secondary" in its JSON line, one JSON line per config.
Usage: python tools/bench_configs.py [steps [warmup]]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else bench.STEADY_STEPS
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else bench.STEADY_WARMUP
print(json.dumps(bench.secondary_c3(steps=steps, warmup=warmup)), flush=True)
for law in ("torque", "velocity"):
    print(json.dumps(bench.secondary_c5_share(law, steps=steps, warmup=warmup)), flush=True)

"""Long runs of the three BASELINE workloads on one GPU (bench.py's own functions): every state word stays finite, episodes
keep ending and restarting, no contact system ends at the sweep cap. (tools/soak.py is the invariant checker over every mode and lane mapping that
tests/test_soak_gpu.py runs; this one is the long run of the three BASELINE workloads.) Usage: python tools/soak_workloads.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from upkie_amd.sim import BatchedSim

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sim = BatchedSim(bench.make_config(4096)); sim.reset(); sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
for _ in range(steps):
    sim.step_pendulum_agent()
torch.cuda.synchronize()
print(json.dumps({"workload": "C2 Pendulum 4096 envs", "steps": steps, "env_steps": steps * 4096, "finite": bool(torch.isfinite(sim.state).all()),
                  "episodes": int(sim.state[40].sum().item())}), flush=True)
out = bench.secondary_c3(steps=steps // 4, warmup=200)
print(json.dumps({"workload": "C3 16384 envs", "steps": steps // 4, "us_per_step": out["us_per_step"], "episodes": out["episodes"]}), flush=True)
for law in ("torque", "velocity"):
    out = bench.secondary_c5_share(law, steps=steps // 2, warmup=200, census_steps=2000)
    print(json.dumps({"workload": f"C5 share, {law} law", "steps": steps // 2, "us_per_step": out["us_per_step"], "episodes": out["episodes"], "census": out["census"]}), flush=True)

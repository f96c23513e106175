"""Rare-path census of the eight-lane kernel on the bench workload: how many
env-substeps leave the eight-lane substep for the general one, by reason, and
how many wavefront-substeps pay for it (see include/upkie_hip.h,
upkie_sim_set_census). Usage: python tools/census.py [envs] [steps] [window]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from upkie_amd.sim import BatchedSim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2200
window = int(sys.argv[3]) if len(sys.argv) > 3 else 200
sim = BatchedSim(bench.make_config(B))
sim.reset()
sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
census = sim.enable_census()
print(f"lanes per env: {sim.lanes_per_env}, {B} envs, 5 substeps per step; per window of {window} steps:")
done = 0
while done < steps:
    census.zero_()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(window):
        sim.step_pendulum_agent()
    stop.record()
    torch.cuda.synchronize()
    c = sim.census_counts()
    env_sub = B * 5 * window
    wave_sub = (B * 8 // 64) * 5 * window
    done += window
    print(f"steps {done - window:5d}-{done:5d}: {start.elapsed_time(stop) * 1e3 / window:6.2f} us/step  env-substeps: limit {c['joint_limit'] / env_sub:.4%} "
          f"cone {c['friction_cone'] / env_sub:.4%}  wavefront-substeps: joint-stop path {c['wavefront_substeps_limit'] / wave_sub:.3%} sweeps {c['wavefront_substeps_sweeps'] / wave_sub:.3%}  episodes {int(sim.state[40].sum())}", flush=True)

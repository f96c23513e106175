"""Histogram of the sweeps a wavefront waits for (largest count among its eight envs, per substep that swept) on one
GPU's share of BASELINE's C5 under both servo laws: the rare-path census of the eight-lane kernel
(upkie_sim_set_census, words 8..71). Usage: python tools/sweeps_hist.py [census steps]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from upkie_amd.sim import BatchedSim

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
_orig = BatchedSim.census_counts
last = {}
def counts(self):
    c = _orig(self)
    last["hist"] = c["wavefront_max_sweeps_histogram"]
    return c
BatchedSim.census_counts = counts
for law in ("torque", "velocity"):
    out = bench.secondary_c5_share(law, steps=600, warmup=200, census_steps=steps)
    print(law, f"{out['us_per_step']:.1f} us/step", json.dumps(out["census"]))
    print("   wavefront-substeps by largest sweep count 0..63:", last["hist"])

"""One-off (round 6): the C5 share (UpkieServos, pushes, randomised inertias, both servo-level laws) on the TWO-lane kernels -- forced at 4096
envs, and at 16384 envs where they are the default mapping of Servos batches -- for A/B runs of the library with and without the
active-set solve in those kernels (UPKIE_HIP_LIBRARY = a -DUPKIE_AB_PAIR_SWEEPS_ONLY build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import bench  # noqa: E402

if __name__ == "__main__":
    out = []
    for envs in (4096, 16384):
        for law in ("torque", "velocity"):
            r = bench.secondary_c5_share(law, envs, 800, 200, 0, 0, "default", 2)
            out.append(f"{envs} envs {law} law {r['us_per_step']:.2f} us ({r.get('lanes_per_env')} lanes, {r.get('episodes')} episodes)")
    print(os.path.basename(os.environ.get("UPKIE_HIP_LIBRARY", "shipped library")) + ": " + "; ".join(out), flush=True)

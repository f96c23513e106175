"""One-off (round 6): bench.py's C3 block alone (UpkieBaseVelocity + balancer N = 16, 16384 envs), for A/B runs of library builds
(UPKIE_HIP_LIBRARY): the fp16 matrix path of the balancer inside the step's launch against a -DUPKIE_FUSED_MPC_FP32 build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import bench  # noqa: E402

if __name__ == "__main__":
    r = bench.secondary_c3(steps=1000, warmup=200)
    print(f"c3 N=16 16384 envs: {r['us_per_step']:.2f} us", flush=True)

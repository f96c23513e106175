"""Time of one launch of the MPC balancer's ADMM kernel by horizon and batch size (round 6): N = 16 / 48 / 50 (the reference's
default) / 64 on the fp16 matrix path with two terms per operand (mpc_tile_h). `UPKIE_MPC_FP32=1`
in the environment selects the fp32 MFMA kernels (N = 50: three row tiles + rows 48 / 49 on the vector unit; with
`UPKIE_MPC_FOUR_TILES=1` as well, round 5's padding to four tiles): the A/Bs of profiles/r06_mpc_tail.txt and
profiles/r06_mpc_f16_split.txt.

Usage (GPU box): python tools/mpc_time.py; UPKIE_MPC_FP32=1 python tools/mpc_time.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from upkie_amd import abi  # noqa: E402
from upkie_amd.mpc import BatchedMpc  # noqa: E402

if __name__ == "__main__":
    fp32 = os.environ.get("UPKIE_MPC_FP32") == "1"
    four = os.environ.get("UPKIE_MPC_FOUR_TILES") == "1"
    print("product: " + (("fp32 MFMA, " + ("four tiles at N = 50" if four else "three tiles + two rows on the vector unit at N = 50")) if fp32 else "fp16 MFMA, two terms per operand"))
    for N in (16, 48, 50, 64):
        for B in (2048, 16384):
            cfg = abi.default_mpc_config(B, N)
            mpc = BatchedMpc(cfg)
            x0 = torch.zeros((B, 4), device="cuda")
            x0[:, 1] = 0.05
            vt = torch.full((B,), 0.3, device="cuda")
            ct = torch.ones(B, dtype=torch.uint8, device="cuda")
            for _ in range(20):
                mpc.step(x0, vt, ct, 0.005)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(200):
                mpc.step(x0, vt, ct, 0.005)
            e.record()
            torch.cuda.synchronize()
            print(f"N={N} B={B} iterations={cfg.admm_iterations}: {s.elapsed_time(e) * 1e3 / 200:.2f} us per mpc.step", flush=True)

"""Run by tests/test_parity_gpu.py in a process of its own with UPKIE_MPC_FP32=1 (and UPKIE_MPC_FOUR_TILES=1): the fp32 MFMA
kernels of the balancer -- the A/B partners of the fp16 matrix path -- against the fp64 checker, at the
tolerance they were held to until round 6 (2e-3 a_max on the first input; 4e-3 at N = 49)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from upkie_amd import abi  # noqa: E402
from upkie_amd.mpc import BatchedMpc  # noqa: E402

if __name__ == "__main__":
    assert os.environ.get("UPKIE_MPC_FP32") == "1"
    for N in (16, 32, 48, 49, 50):
        B = 500
        cfg = abi.default_mpc_config(B, N)
        mpc = BatchedMpc(cfg)
        rng = np.random.default_rng(0)
        ws = np.zeros((2 * N, B))
        v_o, first_o = np.zeros(B), np.zeros(B)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        for step in range(4):
            scale = 1.0 if step < 2 else 5.0
            x0 = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.15, 0.15, B) * scale, rng.uniform(-0.5, 0.5, B) * scale, rng.uniform(-0.5, 0.5, B) * scale], axis=1)
            vt = rng.uniform(-0.5, 0.5, B)
            contact = (rng.uniform(size=B) > 0.1).astype(np.uint8)
            O.lib().oracle_mpc_step(C.byref(cfg), p(ws), p(np.ascontiguousarray(x0)), p(vt), p(contact), C.c_double(0.005), p(v_o), p(first_o))
            v_h, first_h = mpc.step(torch.from_numpy(x0).float(), torch.from_numpy(vt).float(), torch.from_numpy(contact), dt=0.005)
            err = float(np.max(np.abs(first_h.cpu().numpy() - first_o)))
            print(f"N={N} step {step}: |first input - checker| {err:.2e}")
            assert err <= (4e-3 if N == 49 else 2e-3) * cfg.max_ground_accel, (N, step, err)
            assert np.max(np.abs(v_h.cpu().numpy() - v_o)) <= 1e-4
    print("ok")

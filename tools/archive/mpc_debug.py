"""One-off (round 6): the N = 49 / 50 balancer kernel against the fp64 checker, error by step and by horizon index."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from upkie_amd import abi  # noqa: E402
from upkie_amd.mpc import BatchedMpc  # noqa: E402

if __name__ == "__main__":
    for N in (50, 49, 32, 64):
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
            wh = mpc.workspace.cpu().numpy()
            dz = np.abs(wh[:N] - ws[:N]).max(axis=1)
            dy = np.abs(wh[N:2 * N] - ws[N:]).max(axis=1)
            print(f"N={N} step {step}: first_input error {np.max(np.abs(first_h.cpu().numpy() - first_o)):.3e}; z error by index (max over envs):")
            print("   z", " ".join(f"{d:.0e}" for d in dz))
            print("   y", " ".join(f"{d:.0e}" for d in dy))

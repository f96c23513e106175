"""Which contact systems run the projected Gauss-Seidel sweeps into their iteration cap? Runs one GPU's share of
BASELINE's C5 (Servos, randomised inertias, +-5 N pushes, README balancer through the wheel velocity loop, fallen
robots reset) on the fp64 oracle (CPU), prints the sweep histogram and saves the captured systems.
Usage: python tools/pgs_cap_cases.py [envs] [steps] [out.npz] [sweeps threshold] [torque]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

from oracle import oracle as O
from tests.helpers import randomized_config
from upkie_amd import abi
from upkie_amd.model.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/pgs_cap_cases.npz"
threshold = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # capture the systems that needed at least this many sweeps (0: the cap)
torque_law = len(sys.argv) > 5 and sys.argv[5] == "torque"  # examples/pybullet/torque_balancing.py instead of the README balancer
cfg = randomized_config(B, seed=0)
cfg.rand_pitch = 0.1
cfg.joint_friction[2] = cfg.joint_friction[5] = 0.1
model = Model().struct
oracle = O.Oracle(model, cfg)
oracle.body_inertials = oracle.sample_body_inertials(0.2)
rng = np.random.default_rng(0)
force = np.zeros((3, B)); force[0] = rng.uniform(-5, 5, B)
oracle.ext_force = force
oracle.ext_point = np.array([0.0, 0.0, -0.1])
oracle.reset()
lib = O.load() if hasattr(O, "load") else O._lib
hist = (C.c_long * 64).in_dll(lib, "oracle_debug_sweep_hist")
for i in range(64): hist[i] = 0
C.c_long.in_dll(lib, "oracle_debug_captured").value = -int(os.environ.get("PGS_SKIP_CASES", "0"))  # skip the first cases (landing after the initial reset)
C.c_long.in_dll(lib, "oracle_debug_capture_threshold").value = threshold
r = float(model.wheel_radius)
act = np.zeros((B, 6, 6)); act[:, :, 3] = 1.0; act[:, :, 4] = 1.0; act[:, :, 5] = 16.0
act[:, [2, 5], 0] = np.nan
resets = 0
for k in range(steps):
    st = oracle.state
    pitch = 2.0 * st[abi.S_QUAT + 2]
    pos = 0.5 * (st[abi.S_Q + 2] - st[abi.S_Q + 5]) * r * float(model.left_sign)
    v = np.clip(10.0 * pitch + pos, -0.99, 0.99) / r
    if torque_law:  # wheel torques +-10 x pitch, no velocity feedback (kd_scale 0 on the wheels)
        act[:, [2, 5], 4] = 0.0
        act[:, 2, 2] = float(model.left_sign) * 10.0 * pitch
        act[:, 5, 2] = -float(model.left_sign) * 10.0 * pitch
    else:
        act[:, 2, 1] = float(model.left_sign) * v
        act[:, 5, 1] = -float(model.left_sign) * v
    oracle.step_servos(act)
    fallen = np.abs(2.0 * oracle.state[abi.S_QUAT + 2]) > 1.0
    if fallen.any():
        resets += int(fallen.sum())
        oracle.reset(mask=fallen.astype(np.uint8))
h = np.array(list(hist))
n = h.sum()
print(f"{B} envs x {steps} steps, {resets} resets; infeasible substeps {n} ({n / (B * steps * 5):.3%}); mean sweeps {np.dot(h, np.arange(64)) / max(n, 1):.1f}; at the cap {h[50:].sum()} ({h[50:].sum() / max(n, 1):.2%})")
print("sweeps histogram (1..50):", h[1:51].tolist())
captured = max(0, min(C.c_long.in_dll(lib, "oracle_debug_captured").value, 4096))
cases = np.ctypeslib.as_array((C.c_double * (4096 * 55)).in_dll(lib, "oracle_debug_capture")).reshape(4096, 55)[:captured].copy()
np.savez(out, cases=cases, mu=float(model.friction_mu))
print("captured", captured, "cases ->", out)

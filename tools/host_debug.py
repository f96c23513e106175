import ctypes as C, numpy as np, sys
sys.path.insert(0, ".")
from upkie_amd import abi
from upkie_amd.model.default_model import default_model
from oracle import oracle as O
from tests.helpers import randomized_config
H = C.CDLL("tests/_host_harness.so")
m = default_model()
cfg = randomized_config(16, seed=1)
o = O.Oracle(m, cfg)
o.reset()
rng = np.random.default_rng(0)
# run a few oracle steps to get into a generic contact state
obs = o.step_pendulum(rng.uniform(-1,1,16))
for trial, tau in enumerate([np.zeros(6), np.array([0,0,1.0,0,0,-1.0]), np.array([2.0,-1.0,0.5,1.0,0.3,0.2])]):
    for e in range(3):
        s64 = o.state[:, e].copy()
        s32 = s64.astype(np.float32)
        so = s64.copy()
        c_o = O.lib().oracle_substep(C.byref(m), so.ctypes.data_as(C.c_void_p), tau.ctypes.data_as(C.c_void_p), C.c_double(1e-3), None, None, None)
        t32 = tau.astype(np.float32)
        c_h = H.harness_substep(C.byref(m), s32.ctypes.data_as(C.c_void_p), t32.ctypes.data_as(C.c_void_p), C.c_float(1e-3), None, None, None)
        d = so[:25] - s32[:25]
        print(trial, e, c_o, c_h, "max err", np.abs(d).max(), "linvel", d[7:10], "angvel", d[10:13], "qd", d[19:25])

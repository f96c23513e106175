"""How many ADMM iterations does the MPC balancer need? |first input - exact|
(the only entry MPCBalancer.step uses, mpc_balancer.py:307; contract: 2e-3 x
a_max = 0.02 m/s2) of the warm-started ADMM of upkie_amd/csrc/mpc.hpp, in fp64
numpy, as a function of the iteration count K and the over-relaxation alpha:
(a) from an UNRELATED warm start (random states four steps in a row, the later
ones saturating the bounds: tests/test_parity_gpu.py::test_mpc_step_matches_oracle's
inputs) and (b) in closed loop (UpkieBaseVelocity on the oracle doubles, 64
envs x 500 steps, targets redrawn every 100 steps). CPU only.
Usage: python tools/mpc_iterations.py > profiles/rNN_mpc_iterations.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import upkie_amd.envs as envs
from oracle import oracle as O
from tests.fake_sim import OracleMpc, oracle_sim_factory
from upkie_amd import abi
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
KS = (8, 10, 12, 15, 20, 30)


def problem(N):
    cfg = abi.default_mpc_config(1, N)
    P, Kx, kv = np.zeros((N, N)), np.zeros((N, 4)), np.zeros(N)
    O.lib().oracle_mpc_build(C.byref(cfg), p(P), p(Kx), p(kv))
    return cfg, P, Kx, kv


def exact(P, q, bound):
    u = np.zeros(len(q))
    assert O.lib().oracle_mpc_solve_exact(len(q), p(P), p(np.ascontiguousarray(q)), C.c_double(bound), p(u)) >= 0
    return u[0]


def admm(Mi, rho, alpha, bound, K, q, z, y):
    for _ in range(K):
        U = (rho * (z - y) - q) @ Mi.T
        xh = alpha * U + (1 - alpha) * z
        zn = np.clip(xh + y, -bound, bound)
        y = y + xh - zn
        z = zn
    return z, y


for N in (16, 50):
    cfg, P, Kx, kv = problem(N)
    rho, bound = cfg.admm_rho, cfg.max_ground_accel
    Mi = np.linalg.inv(P + rho * np.eye(N))
    print(f"N = {N}, rho = {rho}, a_max = {bound}, default: {cfg.admm_iterations} iterations, alpha = {cfg.admm_relaxation}")
    print(" (a) unrelated warm starts, 400 envs x 4 steps: worst |U0 - exact| [m/s2]")
    B = 400
    for alpha in (1.0, 1.5, 1.6):
        line = []
        for K in KS:
            rng = np.random.default_rng(0)
            z, y, worst = np.zeros((B, N)), np.zeros((B, N)), 0.0
            for step in range(4):
                scale = 1.0 if step < 2 else 5.0
                x0 = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.15, 0.15, B) * scale, rng.uniform(-0.5, 0.5, B) * scale, rng.uniform(-0.5, 0.5, B) * scale], axis=1)
                q = x0 @ Kx.T + np.outer(rng.uniform(-0.5, 0.5, B), kv)
                z, y = admm(Mi, rho, alpha, bound, K, q, z, y)
                worst = max(worst, max(abs(z[i, 0] - exact(P, q[i], bound)) for i in range(B)))
            line.append(f"K = {K}: {worst:.1e}")
        print(f"   alpha {alpha}:  " + "   ".join(line))
    # closed loop: the states an env actually hands the balancer
    Bc, steps = 64, 500
    init = RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0])))
    env = envs.make("Upkie-HIP-BaseVelocity-Vec", num_envs=Bc, frequency=200.0, nb_timesteps=N, init_state=init, seed=0, sim_factory=oracle_sim_factory, mpc_factory=OracleMpc)
    env.reset(seed=0)
    rng = np.random.default_rng(0)
    act = torch.zeros(Bc, 2)
    Q = []
    for k in range(steps):
        if k % 100 == 0:
            act[:, 0] = torch.from_numpy(rng.uniform(-0.5, 0.5, Bc)).float()
        Q.append(env._x0.double().numpy() @ Kx.T + np.outer(act[:, 0].double().numpy(), kv))
        env.step(act)
    E = np.array([[exact(P, Q[k][e], bound) for e in range(Bc)] for k in range(steps)])
    print(f" (b) closed loop, {Bc} envs x {steps} steps (cold start at step 0, targets redrawn every 100 steps): |U0 - exact| median / p99 / worst over all steps")
    for alpha in (1.0, 1.5):
        line = []
        for K in KS:
            z, y, err = np.zeros((Bc, N)), np.zeros((Bc, N)), []
            for k in range(steps):
                z, y = admm(Mi, rho, alpha, bound, K, Q[k], z, y)
                err.append(np.abs(z[:, 0] - E[k]))
            err = np.array(err)
            line.append(f"K = {K}: {np.median(err):.0e} / {np.quantile(err, 0.99):.0e} / {err.max():.0e}")
        print(f"   alpha {alpha}:  " + "   ".join(line))

"""The balancer's host setup (upkie_amd/csrc/mpc.hpp::mpc_host_setup) on the CPU, through tests/host_harness.hip: the condensed
QP's Kx / kv and Minv = (P + rho I)^-1 against the fp64 oracle's own build (oracle_mpc_build) and numpy's inverse; the two
operand layouts the kernels read -- fp32 (rounds 2-6) and, since round 6, two fp16 terms in the A-operand layout of
v_mfma_f32_16x16x32_f16 (Minv / scale = hi + lo, scale a power of two) -- element by element; and the constant part of the re-associated iteration,
Minv Kx and Minv kv."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests.test_device_arithmetic_on_host import harness  # noqa: E402,F401  (the fixture that builds the harness)
from upkie_amd import abi  # noqa: E402

p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def logical(t, g, r):  # horizon index of element (row tile t, lane group g, register r): csrc/mpc.hpp::mpc_index
    return 16 * t + 4 * r + g


@pytest.mark.parametrize("N, rho", [(16, 1e-3), (17, 1e-3), (32, 1e-3), (50, 1e-3), (64, 1e-3), (50, 1e-6), (16, 1e-1)])
def test_host_setup_layouts_hold_the_checkers_matrix(harness, N, rho):  # noqa: F811
    cfg = abi.default_mpc_config(1, N)
    cfg.admm_rho = rho
    tiles = (N + 15) // 16
    kj = (tiles + 1) // 2
    npad = 16 * tiles
    minv_perm = np.zeros(npad * npad, dtype=np.float32)
    kx, kv = np.zeros(npad * 4, dtype=np.float32), np.zeros(npad, dtype=np.float32)
    minv_h = np.zeros(64 * tiles * kj * 16, dtype=np.uint16)
    gx, gv = np.zeros(npad * 4, dtype=np.float32), np.zeros(npad, dtype=np.float32)
    harness.harness_mpc_host_setup.restype = C.c_int
    scale_f = C.c_float(0.0)
    assert harness.harness_mpc_host_setup(C.byref(cfg), npad, p(minv_perm), p(kx), p(kv), p(minv_h), p(gx), p(gv), C.byref(scale_f)) == minv_h.size
    scale_h = float(scale_f.value)
    # the checker's problem
    P, Kx, Kv = np.zeros((N, N)), np.zeros((N, 4)), np.zeros(N)
    O.lib().oracle_mpc_build(C.byref(cfg), p(P), p(Kx), p(Kv))
    Minv = np.linalg.inv(P + cfg.admm_rho * np.eye(N))
    scale = np.abs(Minv).max()
    assert scale_h == 2.0 ** round(np.log2(scale_h)) and 8.0 <= max(scale, 1.0 / (1.0 + rho)) / scale_h < 16.0, (scale, scale_h)  # a power of two; largest entry in [8, 16)
    np.testing.assert_allclose(kx.reshape(npad, 4)[:N], Kx, rtol=2e-7, atol=1e-7 * np.abs(Kx).max())
    np.testing.assert_allclose(kv[:N], Kv, rtol=2e-7, atol=1e-7 * np.abs(Kv).max())
    np.testing.assert_allclose(gx.reshape(npad, 4)[:N], Minv @ Kx, rtol=1e-6, atol=1e-6 * np.abs(Minv @ Kx).max())
    np.testing.assert_allclose(gv[:N], Minv @ Kv, rtol=1e-6, atol=1e-6 * np.abs(Minv @ Kv).max())

    def entry(row, col):  # the padded matrix: identity / (1 + rho) outside the horizon
        if row < N and col < N:
            return Minv[row, col]
        return 1.0 / (1.0 + cfg.admm_rho) if row == col else 0.0

    a32 = minv_perm.reshape(64, tiles, 4 * tiles)  # [lane][t][k-step]
    halves = minv_h.view(np.float16).astype(np.float64).reshape(64, tiles, kj, 2, 8)  # [lane][t][K-step][hi | lo][slot]
    worst32 = worst16 = 0.0
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for t in range(tiles):
            row = logical(t, i // 4, i % 4)  # A row i of tile t is output row 4 g' + r' = i: element (t, i / 4, i % 4)
            for s in range(4 * tiles):  # fp32 form: k-step s reads element (s / 4, g, s % 4)
                worst32 = max(worst32, abs(a32[lane, t, s] - entry(row, logical(s // 4, g, s % 4))))
            for j in range(kj):
                for slot in range(8):  # fp16 form: K-step j, slot c reads element (2 j + c / 4, g, c % 4)
                    tk = 2 * j + slot // 4
                    want = entry(row, logical(tk, g, slot % 4)) / scale_h if tk < tiles else 0.0
                    got = halves[lane, t, j, 0, slot] + halves[lane, t, j, 1, slot]
                    worst16 = max(worst16, abs(got - want) * scale_h)
    assert worst32 <= 1e-7 * scale, worst32  # one fp32 rounding
    assert worst16 <= 6e-7 * scale, worst16  # two fp16 terms: 22 bits (2.4e-7) of the largest entry, and fp16's subnormal floor (6e-8 x scale) on the small ones

"""oracle/_ref: the part of the reference's C++ that compiles from its own
sources with g++ alone (upkie/cpp/utils/low_pass_filter.h) against the oracle's
restatements of it. The .so is built in the build container (where
/root/reference exists) and travels to the GPU box."""

import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from upkie_amd import abi


@pytest.fixture(scope="module")
def ref():
    library = O.ref_lib()
    if library is None:
        pytest.skip("oracle/_ref was not built (no /root/reference here)")
    return library


def test_low_pass_filter_restatement_is_the_reference(ref):
    rng = np.random.default_rng(0)
    threw = C.c_int(0)
    O.lib().oracle_low_pass_filter.restype = C.c_double
    O.lib().oracle_low_pass_filter.argtypes = [C.c_double] * 4
    for _ in range(500):
        prev, new = rng.uniform(-10, 10, 2)
        dt = float(rng.uniform(1e-4, 0.02))
        cutoff = float(rng.uniform(2.0 * dt * 1.0001, 2.0))
        want = ref.ref_low_pass_filter(prev, cutoff, new, dt, C.byref(threw))
        assert threw.value == 0
        assert O.lib().oracle_low_pass_filter(prev, cutoff, new, dt) == want  # bit-identical doubles
    # Nyquist guard: cutoff <= 2 dt throws FilterError (low_pass_filter.h:22-30)
    for cutoff, dt, expect in ((0.01, 0.005, 1), (0.0100001, 0.005, 0), (0.2, 0.1, 1), (0.008, 0.005, 1)):
        ref.ref_low_pass_filter(0.0, cutoff, 1.0, dt, C.byref(threw))
        assert threw.value == expect
        cfg = abi.default_observer_config(1, dt)
        cfg.wheel_cutoff_period = cutoff
        leg_filter_ok = 0.01 > 2.0 * dt
        assert (O.lib().oracle_observers_check(C.byref(cfg)) == 0) == (expect == 0 and leg_filter_ok)


def test_observer_filters_follow_the_reference_filter(ref):
    """The three low-pass filters inside WheelContact::observe
    (WheelContact.cpp:26-31) and the upper-leg torque filter
    (FloorContact.cpp:87-90), replayed with the reference's compiled filter."""
    dt = 1e-3
    cfg = abi.default_observer_config(1, dt)
    oracle = O.ObserverOracle(cfg)
    rng = np.random.default_rng(1)
    threw = C.c_int(0)
    f = lambda prev, cutoff, new: ref.ref_low_pass_filter(prev, cutoff, new, dt, C.byref(threw))
    v = a = t = leg = 0.0
    for _ in range(400):
        servo = np.zeros((1, 6, 5))
        servo[0, :, 1] = rng.uniform(-5, 5, 6)
        servo[0, :, 2] = rng.uniform(-2, 2, 6)
        out = oracle.step(servo)
        prev = v
        v = f(v, 0.2, servo[0, 2, 1])
        a = f(a, 0.2, abs((v - prev) / dt))
        t = f(t, 0.2, abs(servo[0, 2, 2]))
        leg = f(leg, 0.01, float(np.sqrt(np.sum(servo[0, [0, 1, 3, 4], 2] ** 2))))
        assert oracle.state[abi.O_WHEEL + 0, 0] == v
        assert out["wheel_contact"][0, 0, 0] == a and out["wheel_contact"][0, 0, 1] == t
        assert out["upper_leg_torque"][0] == leg

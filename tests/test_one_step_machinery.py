"""The one-step (teacher-forced) comparison of tests/one_step.py, run WITHOUT a
GPU on a second oracle as the "implementation": zero defect by construction,
so what is checked here is the machinery itself -- that both sides really start
each step from the same bits, that the regime classifier finds every regime the
GPU test states a tolerance for, and that a perturbed implementation shows up
in the bins with the size it was given."""

import numpy as np
import pytest

from upkie_amd import abi
from upkie_amd.model.default_model import default_model

from .fake_sim import servo_policy_action
from .one_step import REGIMES, OracleTwin, window


def c5_oracle(B, seed=0, bullet_like=False):
    """The oracle's side of `bench.secondary_c5_share` (per-link inertia
    randomisation 0.2, wheel friction 0.1, init-state randomisation)."""
    import bench
    from oracle import oracle as O

    cfg = bench.make_config(B, seed=seed)
    cfg.joint_friction[2] = cfg.joint_friction[5] = 0.1
    model = default_model()
    # the device holds the model in fp32: a joint resting ON its stop is "at the stop" or not by the last bit of the limit, so
    # the checker is given the limits the device has (everything else of the model enters through products, not comparisons)
    for j in range(abi.NJ):
        model.joint_lower[j] = float(np.float32(model.joint_lower[j]))
        model.joint_upper[j] = float(np.float32(model.joint_upper[j]))
    ref = O.Oracle(model, cfg)
    if bullet_like:
        ref.use_bullet_like_contacts()
    ref.body_inertials = ref.sample_body_inertials(0.2)
    ref.ext_force = np.zeros((3, B))
    ref.ext_point = np.zeros(3)
    ref.reset()
    return ref, model, cfg


def c5_push_schedule(ref):
    import bench

    def schedule(k):
        phase = k % bench.PUSH_PERIOD
        if phase == 0:
            return ref.sample_pushes(k // bench.PUSH_PERIOD, bench.PUSH_MAX_NORM)
        if phase == bench.PUSH_HOLD:
            return np.zeros((3, ref.B))
        return None

    return schedule


@pytest.mark.parametrize("bullet_like", [False, True])
def test_machinery_on_a_second_oracle(bullet_like):
    B, steps = 256, 520
    ref, model, cfg = c5_oracle(B, bullet_like=bullet_like)
    policy = abi.torque_balancing_policy(10.0, 1.0, float(model.left_sign))
    rs = float(model.left_sign) * float(model.wheel_radius)
    twin = OracleTwin(model, cfg, ref.body_inertials, bullet_like)
    noisy = OracleTwin(model, cfg, ref.body_inertials, bullet_like, perturb=1e-6)
    bins, census, points, flags = window(ref, model, {"twin": twin, "noisy": noisy}, lambda s, _: servo_policy_action(policy, s, rs), steps,
                                         "servos", c5_push_schedule(ref), bullet_like)
    table = bins.table()
    assert sum(census.values()) == B * steps
    # the window visits what the GPU test states tolerances for (256 envs, 520 steps: through the first falls and the second push)
    for regime in ("reset", "airborne", "sliding", "saturated", "push", "rolling"):
        assert census[regime] > 0, census
    for regime in REGIMES:
        row = table["twin"][regime]
        assert row["env_steps"] == census[regime]
        if row["env_steps"]:
            # same bits in, same code: same bits out (fp32 rounding of the outputs aside)
            assert row["position"]["q1"] <= 1e-6 and row["torque"]["q1"] <= 1e-6 and row["velocity"]["q1"] <= 2e-5, (regime, row)
    # and a defect that IS there is seen at its size: relative 1e-6 on every state word
    row = table["noisy"]["rolling"]
    assert 1e-8 <= row["position"]["q0.5"] <= 1e-5 and row["velocity"]["q0.99"] >= 1e-7, row
    if bullet_like:
        assert points["twin"][1] == points["twin"][0]

"""How far does the MEASURED one-step defect of the fp32 kernels carry in the chaotic C5 window?  (CPU only: the fp64 oracle.)

tests/test_timed_windows_gpu.py finds, over 4096 envs x 1200 steps of BASELINE configs[4]'s share under
examples/pybullet/torque_balancing.py's law, about 280 envs that fall on the device only and about 285 on the oracle only
(of 2300 that fall on both). tests/test_one_step_parity_gpu.py measures what ONE device step differs from the oracle's on
the very states of that window (gpurun_out/parity_windows/one_step_c5_torque_law.json): velocities 2e-6 (median, rolling),
1e-5 (sliding), 3e-4 at the 99th percentile of the sliding steps.

This tool closes the argument: TWO fp64 oracle runs of the window from the same initial states, the second one perturbed
after every step by a random state change of exactly that measured size (relative to max(1, |value|), per regime: the
median as the scale of a Gaussian, i.e. smaller than the device's own tail). If two runs that differ by nothing but the
measured one-step defect part as far as device and oracle do, the one-sided falls are the window's own sensitivity to
that defect -- not a defect of another kind.

Usage: python tools/one_step_amplification.py [envs] [steps] > profiles/r06_one_step_amplification.txt
"""

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.fake_sim import servo_policy_action  # noqa: E402
from tests.one_step import contact_summary  # noqa: E402
from tests.test_one_step_machinery import c5_oracle, c5_push_schedule  # noqa: E402
from upkie_amd import abi  # noqa: E402

# measured medians of the one-step defect (gpurun_out/parity_windows/one_step_c5_torque_law.json, eight lanes; round 6)
POSITION = {"rolling": 8.3e-8, "sliding": 1.3e-7}
VELOCITY = {"rolling": 1.7e-6, "sliding": 9.1e-6}


def run(B, steps, scale, seed=1):
    """The window on the oracle; `scale` x the measured one-step defect is added after every step (0: the clean run).
    Returns the step on which each env's episodes ended [steps, B]."""
    ref, model, _ = c5_oracle(B)
    policy = abi.torque_balancing_policy(10.0, 1.0, float(model.left_sign))
    rs = float(model.left_sign) * float(model.wheel_radius)
    schedule = c5_push_schedule(ref)
    rng = np.random.default_rng(seed)
    ends = np.zeros((steps, B), dtype=np.uint8)
    mu = float(model.friction_mu)
    pos_words = list(range(abi.S_POS, abi.S_POS + 7)) + list(range(abi.S_Q, abi.S_Q + 6))
    vel_words = list(range(abi.S_LINVEL, abi.S_LINVEL + 6)) + list(range(abi.S_QD, abi.S_QD + 6))
    for k in range(steps):
        force = schedule(k)
        if force is not None:
            ref.ext_force = np.ascontiguousarray(force)
        act, fallen = servo_policy_action(policy, ref.state, rs)
        ref.state[abi.S_DONE] = np.where(fallen, 1.0, ref.state[abi.S_DONE])
        ends[k] = fallen
        ref.step_servos(act)
        if scale > 0.0:
            _, sliding = contact_summary(ref.contact_points(), mu)
            p = np.where(sliding, POSITION["sliding"], POSITION["rolling"]) * scale
            v = np.where(sliding, VELOCITY["sliding"], VELOCITY["rolling"]) * scale
            # median |N(0, s)| = 0.6745 s: the Gaussian whose median magnitude is the measured median
            for words, size in ((pos_words, p), (vel_words, v)):
                s = ref.state[words]
                ref.state[words] = s + rng.standard_normal(s.shape) * (size / 0.6745)[None, :] * np.maximum(1.0, np.abs(s))
            q = ref.state[abi.S_QUAT:abi.S_QUAT + 4]
            ref.state[abi.S_QUAT:abi.S_QUAT + 4] = q / np.linalg.norm(q, axis=0)
    return ends


def compare(a, b):
    fell_a, fell_b = a.sum(axis=0) > 0, b.sum(axis=0) > 0
    first_a = np.where(fell_a, a.argmax(axis=0), -1)
    first_b = np.where(fell_b, b.argmax(axis=0), -1)
    both = fell_a & fell_b
    counts_equal = a.sum(axis=0) == b.sum(axis=0)
    return {
        "episodes_ended": [int(a.sum()), int(b.sum())],
        "envs_fell_in_the_first_run_only": int((fell_a & ~fell_b).sum()),
        "envs_fell_in_the_second_run_only": int((~fell_a & fell_b).sum()),
        "envs_fell_in_both": int(both.sum()),
        "envs_with_the_same_number_of_episode_ends": float(counts_equal.mean()),
        "envs_whose_first_episode_ends_within_2_steps": float((np.abs(first_a - first_b) <= 2)[both].mean()) if both.any() else 1.0,
        "median_first_fall_step": [int(np.median(first_a[fell_a])), int(np.median(first_b[fell_b]))],
    }


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
    clean = run(B, steps, 0.0)
    print(f"C5 share under torque_balancing.py's law on the fp64 oracle, {B} envs x {steps} steps; a second oracle run perturbed after every step by")
    print("scale x the MEASURED one-step defect of the fp32 kernels (medians per regime: position 8.3e-8 / 1.3e-7, velocity 1.7e-6 / 9.1e-6 rolling / sliding)")
    print("device against oracle over the same window (tests/test_timed_windows_gpu.py, round 5): 278 envs fell on the device only, 285 on the oracle only, 2296 on both;")
    print("per-env agreement within 2 steps 60.6 %")
    for scale in (1.0, 0.1, 0.01):
        out = compare(clean, run(B, steps, scale))
        print(f"scale {scale:g}: " + json.dumps(out))

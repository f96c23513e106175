"""Table of the timed-window parity reports (tests/test_timed_windows_gpu.py
writes them to gpurun_out/parity_windows/ on the GPU box): DESIGN.md section 4.
Usage: python tools/parity_windows_table.py [directory] > profiles/rNN_parity_windows.txt"""
import json
import os
import sys

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "parity_windows")


def load(name):
    with open(os.path.join(D, name + ".json")) as f:
        return json.load(f)


def q(d, col=0):
    return "  ".join(f"{k[1:]}: {v[col]:.2e}" for k, v in d.items() if k.startswith("q"))


def falls(r, slack):
    print(f"  episode ends: device {r['episodes_ended_device']}, oracle {r['episodes_ended_oracle']} ({r['envs']} envs x {r['steps']} steps)")
    print(f"  envs with the same number of ends {r['envs_same_number_of_episode_ends']:.4f}, every end within {slack} step(s) {r[f'envs_every_end_within_{slack}_step']:.4f}, "
          f"every end on the same step {r['envs_every_end_on_the_same_step']:.4f}")
    print(f"  first end: median step {r['first_end_median_step_device']} (device) / {r['first_end_median_step_oracle']} (oracle), largest difference {r['first_end_largest_difference_steps']} steps")


for spl in (1, 32):
    r = load(f"c2_window_{spl}_steps_per_launch")
    print(f"C2 window, {spl} env.step() per launch (bench.py: {'value / steady_state' if spl == 1 else 'fused_rollout'})")
    falls(r, 1)
    for m in (500, 1000, 1500):
        e = r[f"obs_error_step_{m}"]
        print(f"  |obs - oracle| at step {m} over {e['envs_in_phase']} envs in phase (quantiles over envs):")
        for i, c in enumerate(e["columns"]):
            print(f"    {c:12s} {q(e, i)}")
r = load("c3_window")
print("C3 window (UpkieBaseVelocity + MPC in the launch, v* redrawn at steps", r["target_redraws"], ")")
falls(r, 1)
print("  commanded ground velocity |device - oracle|, worst over the window per env [m/s]:", q(r["commanded_velocity_worst_over_window"]))
for k, v in r["commanded_velocity_error_at_step"].items():
    print(f"    at step {k:>4s}: {q(v)}")
for i, c in enumerate(r["pose_worst_over_window"]["columns"]):
    print(f"  pose {c}: worst over the window {q(r['pose_worst_over_window'], i)}")
print("  final pitch error [rad]:", q(r["final_pitch_error"]), "  final base x error [m]:", q(r["final_base_x_error"]))
for law in ("velocity", "torque"):
    r = load(f"c5_share_window_{law}_law")
    print(f"C5 share window, {law} law, pushes drawn on device vs oracle twin: max |dF| = {max(r['push_draw_max_abs_error_newton']):.1e} N over {r['pushes']} pushes")
    falls(r, 2)
    print(f"  envs that fell on both {r['envs_fell_on_both']}, on the device only {r['envs_fell_on_device_only']}, on the oracle only {r['envs_fell_on_oracle_only']}")
    for m, v in r["marks"].items():
        print(f"  step {m:>4s}, {v['envs_that_never_fell']} envs that never fell on either side: pitch [{q(v['pitch'])}]  wheel velocity [{q(v['wheel_velocity'])}]  wheel torque [{q(v['wheel_torque'])}]")

"""Tables of the one-step (teacher-forced) parity reports tests/test_one_step_parity_gpu.py writes on the GPU box
(gpurun_out/parity_windows/one_step_*.json): per window, lane mapping and regime the one-step defect of the fp32 kernels
against the fp64 oracle from the SAME fp32 state (median / 99 % / worst; relative to max(1, |value|)). DESIGN.md section 4.
Usage: python tools/one_step_table.py [directory] > profiles/r06_one_step_parity.txt"""
import glob
import json
import os
import sys

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "parity_windows")
TITLES = {
    "one_step_c2": "C2 (BASELINE configs[1]): 4096 Upkie-Pendulum envs, README agent, 2200 steps through the falls and NEXT_STEP autoresets",
    "one_step_c5_torque_law": "C5 share (BASELINE configs[4]), examples/pybullet/torque_balancing.py's law (the chaotic window), 4096 envs x 1200 steps, pushes, randomised inertias",
    "one_step_c5_torque_law_bullet_like": "the same under the Bullet-like contact model (eight lanes: one cached point per tire; one lane: up to four)",
    "one_step_c5_velocity_law": "C5 share, README law through the wheels' velocity loop",
    "one_step_joint_stops": "UpkieServos agents that HOLD hips and knees against their stops, 1024 envs x 300 steps, default contact model",
    "one_step_joint_stops_bullet_like": "the same under the Bullet-like model (limit rows inside the specification's 50 sweeps on both mappings since round 6)",
}
for path in sorted(glob.glob(os.path.join(D, "one_step_*.json"))):
    name = os.path.basename(path)[:-5]
    with open(path) as f:
        r = json.load(f)
    print(f"== {TITLES.get(name, name)}")
    total = sum(r["env_steps_per_regime"].values())
    print("   env-steps per regime: " + ", ".join(f"{k} {v} ({100.0 * v / total:.1f} %)" for k, v in r["env_steps_per_regime"].items() if v))
    if r.get("terminated_flag_mismatches"):
        print("   terminated-flag mismatches: " + ", ".join(f"{k}: {v}" for k, v in r["terminated_flag_mismatches"].items()))
    if r.get("env_steps_with_the_oracles_live_points"):
        print("   env-steps whose contact manifold holds the oracle's live points: " + ", ".join(f"{k}: {v:.6f}" for k, v in r["env_steps_with_the_oracles_live_points"].items()))
    print(f"   {'mapping':8s} {'regime':14s} {'env-steps':>9s}   position (median / 99 % / worst)   velocity                      wheel rate                    joint torque")
    for mapping, table in r["one_step_defect"].items():
        for regime, row in table.items():
            if not row["env_steps"]:
                continue
            cells = "   ".join(f"{row[m]['q0.5']:.1e} / {row[m]['q0.99']:.1e} / {row[m]['q1']:.1e}" for m in ("position", "velocity", "wheel_rate", "torque"))
            print(f"   {mapping:8s} {regime:14s} {row['env_steps']:9d}   {cells}")
    print()

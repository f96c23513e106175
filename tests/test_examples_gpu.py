"""The scripts under examples/ run on the GPU box (shortened)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = ["pd_balancing", "batched_balancing", "domain_randomization", "count_wheel_contacts", "mpc_balancing", "ppo_rollout", "torque_balancing", "contact_models",
            "sharded_servos"]


@pytest.mark.parametrize("name", EXAMPLES)
def test_example_runs(name):
    env = dict(os.environ, EXAMPLE_STEPS="120" if name != "ppo_rollout" else "16")
    result = subprocess.run([sys.executable, os.path.join(ROOT, "examples", name + ".py")], capture_output=True, text=True, timeout=600,
                            env=env, cwd=os.path.join(ROOT, "examples"))
    assert result.returncode == 0, result.stderr[-3000:]
    out = result.stdout
    assert out.strip(), "the example prints what it did"
    if name == "count_wheel_contacts":
        assert "left tire 1 contact(s), right tire 1 contact(s)" in out and "PointContact(link_name='left_wheel_tire'" in out
    if name == "torque_balancing":
        assert "torque_balancing:" in out and "velocity_balancing:" in out
    if name == "domain_randomization":
        assert out.count("inertia_variation") == 3


@pytest.mark.parametrize("workload, steps", [("pendulum", 150), ("servos", 450), ("mpc", 150)])
def test_compare_with_pybullet_tool_runs_end_to_end_against_the_doubles(workload, steps, tmp_path):
    """tools/compare_with_pybullet.py is the one command that produces the a8 / a18 pins on a machine with pybullet /
    proxsuite. Here its plumbing runs end to end with the reference replaced by the fp64 oracle
    doubles (`--against doubles`): both loops, the push schedule, the JSON report -- and the numbers it reports are
    those of the device against the oracle."""
    import json

    report_path = tmp_path / "report.json"
    result = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_with_pybullet.py"), "--workload", workload, "--steps", str(steps),
                             "--against", "doubles", "--horizon", "16", "--json", str(report_path)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert result.returncode == 0, result.stderr[-3000:]
    report = json.loads(report_path.read_text())
    # (servos: the loop stops when a robot falls -- under torque_balancing.py's law the second push, at step 400, may tip it over)
    assert report["workload"] == workload and report["against"] == "doubles" and report["steps_compared"] >= (400 if workload == "servos" else steps - 1)
    if workload == "pendulum":
        assert report["observation_error"]["pitch"]["q1"] < 1e-3 and report["observation_error"]["position"]["q1"] < 1e-3
    elif workload == "servos":
        assert report["pushes_applied_on_both_sides"] is True
        assert report["joint_torque_error_first_100_steps_worst_joint"]["q0.5"] < 1e-3 and report["pitch_error"]["q0.5"] < 1e-4
    else:
        assert report["commanded_velocity_error_m_per_s"]["q1"] < 5e-4

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

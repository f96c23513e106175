"""Short soak on the GPU: randomised rollouts in every mode and lane mapping
stay finite and keep their invariants (tools/soak.py, 20000 steps x 4096 envs
passes offline; here a few thousand steps)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_runs_clean():
    result = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "3000", "1024"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert result.returncode == 0, result.stdout[-1500:] + result.stderr[-1500:]
    assert "soak passed" in result.stdout

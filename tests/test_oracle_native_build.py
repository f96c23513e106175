"""bench.py's cpu_baseline times the checker built `-O3 -march=native` ON THE
HOST IT RUNS ON (oracle.use_native_build); no other test loads that build, and
in round 6 it crashed on the GPU box (an aligned AVX load from a 16-byte
aligned frame slot once a row of the contact system was 96 bytes) while every
test of the portable build was green. Run in a subprocess: the tuned library
replaces the one this process has loaded."""

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import numpy as np
import bench
from oracle import oracle as O
from upkie_amd.model.default_model import default_model
native = O.use_native_build()
assert native, "the native build did not load"
B = 4096  # (the batch cpu_baseline times: enough initial states for every branch of the contact solve)
ref = O.Oracle(default_model(), bench.make_config(B))
obs = ref.reset()[:, [1, 0, 4, 3]]
obs, falls = ref.rollout_pendulum_agent(obs, 40)
assert np.isfinite(obs).all() and np.isfinite(ref.state).all()
np.save(r"%s", obs)
print("NATIVE_OK")
"""


def test_the_native_build_of_the_checker_runs_the_bench_workload(tmp_path):
    import numpy as np

    from oracle import oracle as O
    import bench
    from upkie_amd.model.default_model import default_model

    out = tmp_path / "native_obs.npy"
    result = subprocess.run([sys.executable, "-c", CODE % str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert result.returncode == 0 and "NATIVE_OK" in result.stdout, (result.returncode, result.stderr[-2000:])
    # the same rollout on the portable build: the tuned one contracts multiply-adds, so to rounding through 40 steps
    ref = O.Oracle(default_model(), bench.make_config(4096))
    obs = ref.reset()[:, [1, 0, 4, 3]]
    obs, _ = ref.rollout_pendulum_agent(obs, 40)
    assert np.abs(np.load(out) - obs).max() < 1e-8

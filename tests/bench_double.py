"""bench.py's `main` on CPU doubles (TEST INFRASTRUCTURE): the simulation
handle is the oracle-backed `OracleSim`, the process group is gloo. Launched by
tests/test_distributed.py under `python -m torch.distributed.run` exactly as
the driver launches bench.py, so the N > 1 launch line, the shard arithmetic
and the JSON contract are covered where there is no GPU."""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tests.fake_sim import OracleSim  # noqa: E402
from upkie_amd.model.default_model import default_model  # noqa: E402

if __name__ == "__main__":
    # the steady-state window of SURVEY 8d (2000 steps after 200) shrunk to what the fp64 oracle steps in seconds
    bench.STEADY_WARMUP, bench.STEADY_STEPS = 2, 6
    if os.environ.get("UPKIE_BENCH_DOUBLE_ENVS"):  # "c2,c4": the default per-GPU batch sizes shrunk the same way (the no-flags multi-GPU line)
        bench.ENVS_PER_GPU, bench.C4_ENVS_PER_GPU = (int(v) for v in os.environ["UPKIE_BENCH_DOUBLE_ENVS"].split(","))
    bench.main(sys.argv[1:], sim_factory=lambda cfg, model, device: OracleSim(cfg, model if model is not None else default_model(), device), backend="gloo")

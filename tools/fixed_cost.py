"""Fixed and per-substep cost of a step launch: the same Pendulum step with 1 .. 5 physics substeps
(frequency adjusted so that h stays 1 ms), per lane mapping; with --rollout the same for 32 env.step() per launch
(upkie_sim_step_pendulum_agent_rollout): what a step costs there beyond its substeps. Usage: python tools/fixed_cost.py [B] [--rollout]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from upkie_amd import abi
from upkie_amd.sim import BatchedSim

rollout = "--rollout" in sys.argv
if rollout:
    sys.argv.remove("--rollout")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for lanes in ("2", "8"):
    os.environ["UPKIE_LANES_PER_ENV"] = lanes
    times = {}
    for n in (1, 2, 3, 5):
        cfg = abi.default_sim_config(B, frequency=1000.0 / n, nb_substeps=n, seed=0)
        cfg.rand_pitch = 0.05
        sim = BatchedSim(cfg)
        sim.reset()
        sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if rollout:
            prev = torch.zeros(B, 8, device=sim.device); prev[:, :4] = sim.obs6[:, [1, 0, 4, 3]]
            ring = torch.zeros(32, B, 8, device=sim.device)
            def launch():
                sim.rollout_pendulum_records(prev, ring)
                prev.copy_(ring[-1])
            for _ in range(4): launch()
            torch.cuda.synchronize(); s.record()
            for _ in range(20): launch()
            e.record(); torch.cuda.synchronize()
            times[n] = s.elapsed_time(e) * 1e3 / 640
            continue
        for _ in range(100): sim.step_pendulum_agent()
        torch.cuda.synchronize(); s.record()
        for _ in range(400): sim.step_pendulum_agent()
        e.record(); torch.cuda.synchronize()
        times[n] = s.elapsed_time(e) * 1e3 / 400
    per = (times[5] - times[1]) / 4
    print(f"lanes {lanes}{' (32 steps per launch)' if rollout else ''}: " + "  ".join(f"{n} substeps {t:.2f} us" for n, t in times.items()) + f"   per substep {per:.2f} us, fixed {times[1] - per:.2f} us", flush=True)

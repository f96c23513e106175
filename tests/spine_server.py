"""Test helper: a `HipSpine` in its own process (as `python -m upkie_amd.spine`
runs it), over the CPU double of the simulation handle, serving env #1 of a
three-env batch. Usage: python -m tests.spine_server <shm_name>"""

import signal
import sys

import upkie_amd.envs as envs
from upkie_amd.spine import HipSpine

from .fake_sim import oracle_sim_factory


def main() -> None:
    name = sys.argv[1]
    env = envs.make("Upkie-HIP-Servos-Vec", num_envs=3, frequency=200.0, autoreset_mode="disabled", sim_factory=oracle_sim_factory)
    env.reset(seed=0)
    spine = HipSpine(env, shm_name=name, shm_size=1 << 16, env_index=1)
    signal.signal(signal.SIGTERM, lambda *a: spine.interrupt())
    print("ready", flush=True)
    spine.run(idle_sleep=1e-4)
    # what the other envs of the batch did meanwhile: env #0's base position (untouched by the agent's reset)
    print(f"over {float(env.sim.state[0, 0]):.6f} {float(env.sim.state[0, 1]):.6f}", flush=True)
    spine.close()


if __name__ == "__main__":
    main()

"""Start a HIP spine: the counterpart of ``start_bullet_spine.sh`` /
``spines/bullet_spine.cpp`` with a GPU batch behind the shared memory.

    python -m upkie_amd.spine --num-envs 4096 --shm-name /upkie

An agent of the reference (``gym.make("Upkie-Spine-Pendulum", ...)``) then
drives env `--env-index` of the batch; the other envs hold the neutral action.
"""

import argparse
import signal

from ..envs.entry_points import make
from .hip_spine import HipSpine


def main() -> None:
    parser = argparse.ArgumentParser(description=__doc__)
    parser.add_argument("--num-envs", type=int, default=1)
    parser.add_argument("--env-index", type=int, default=0)
    parser.add_argument("--shm-name", default="/upkie")  # Spine.h:66
    parser.add_argument("--shm-size", type=int, default=1 << 20)  # Spine.h:69
    parser.add_argument("--frequency", type=float, default=200.0, help="agent frequency: one env step per action")
    parser.add_argument("--nb-substeps", type=int, default=None)
    parser.add_argument("--device", default="cuda:0")
    parser.add_argument("--spine-observers", action="store_true", help="append the C++ spine's observer outputs")
    args = parser.parse_args()
    env = make(
        "Upkie-HIP-Servos-Vec",
        num_envs=args.num_envs,
        device=args.device,
        frequency=args.frequency,
        nb_substeps=args.nb_substeps,
        autoreset_mode="disabled",
        spine_observers=args.spine_observers,
    )
    env.reset(seed=0)
    spine = HipSpine(env, shm_name=args.shm_name, shm_size=args.shm_size, env_index=args.env_index)
    signal.signal(signal.SIGINT, lambda *_: spine.interrupt())
    signal.signal(signal.SIGTERM, lambda *_: spine.interrupt())
    try:
        spine.run(idle_sleep=1e-5)
    finally:
        spine.close()
        env.close()
    print("SEE YOU SPACE COWBOY...")  # Spine.cpp:116


if __name__ == "__main__":
    main()

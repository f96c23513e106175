"""Side-by-side run of the TRUE reference (Upkie-PyBullet-Pendulum on PyBullet
+ upkie_description) and this repository's HIP path, for machines where the
reference's dependencies are installed (they are not in the build container:
SURVEY.md section 8c). Prints per-step differences of the Pendulum observation
for the README agent and, with --time, env-steps/s of the reference on one
core (the B2 baseline of BASELINE.md).

    python tools/compare_with_pybullet.py --steps 200 [--time] [--contact-model bullet_like]

--contact-model bullet_like (the default here) runs this repository's side
under the Bullet-like contact specification on the device
(`upkie_sim_set_contact_manifold`: persistent 4-point manifolds, 50 fixed
sweeps, cone friction) -- the restatement of what pybullet.stepSimulation() is
published to do, and the first thing to hold against the real thing;
--contact-model default runs the product's fast specification.

Expect agreement of the wrapper arithmetic and qualitative agreement of the
dynamics only: Bullet's contact solver and the real URDF differ from the
synthetic model and contact spec documented in DESIGN.md section 4.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--time", action="store_true")
    parser.add_argument("--contact-model", choices=("bullet_like", "default"), default="bullet_like")
    args = parser.parse_args()
    try:
        import gymnasium as gym
        import upkie.envs  # the reference package
    except ImportError as exc:
        print(f"reference not importable here ({exc}); nothing to compare")
        return 0
    upkie.envs.register()
    import upkie_amd.envs as envs
    from upkie_amd.model.model import Model

    ref = gym.make("Upkie-PyBullet-Pendulum", frequency=200.0, gui=False, regulate_frequency=False, frequency_checks=False)
    ours = envs.make("Upkie-HIP-Pendulum", frequency=200.0, model=Model(), contact_model=args.contact_model)  # Model() picks up upkie_description's URDF
    gain = np.array([10.0, 1.0, 0.0, 0.1])
    obs_r, _ = ref.reset(seed=0)
    obs_o, _ = ours.reset(seed=0)
    worst = np.zeros(4)
    t0 = time.perf_counter()
    for step in range(args.steps):
        action_r = np.clip(gain.dot(obs_r), -0.99, 0.99).reshape((1,)).astype(np.float32)
        action_o = np.clip(gain.dot(obs_o), -0.99, 0.99).reshape((1,)).astype(np.float32)
        obs_r, _, term_r, trunc_r, _ = ref.step(action_r)
        obs_o, _, term_o, trunc_o, _ = ours.step(action_o)
        worst = np.maximum(worst, np.abs(obs_r - obs_o))
        if step % 20 == 0:
            print(f"step {step:4d}  reference {obs_r}  hip {obs_o}")
        if term_r or term_o:
            print(f"terminated at step {step}: reference={term_r} hip={term_o}")
            break
    print("max |reference - hip| per observation component [pitch, position, pitch rate, velocity]:", worst)
    if args.time:
        obs_r, _ = ref.reset(seed=0)
        t0 = time.perf_counter()
        n = 2000
        for _ in range(n):
            obs_r, _, term, trunc, _ = ref.step(np.clip(gain.dot(obs_r), -0.99, 0.99).reshape((1,)).astype(np.float32))
            if term or trunc:
                obs_r, _ = ref.reset()
        dt = time.perf_counter() - t0
        print(f"reference (PyBullet, one process, one core): {n / dt:.1f} env-steps/s")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""One robot balancing with the README's linear feedback, as the reference's
examples/pybullet/pd_balancing.py and README.md:53-68 do on PyBullet: the same
id scheme, reset/step signature and observation layout, on the HIP backend."""
import numpy as np

from _common import steps

import upkie_amd.envs as envs

if __name__ == "__main__":
    envs.register()
    with envs.make("Upkie-HIP-Pendulum", frequency=200.0) as env:
        observation, _ = env.reset()
        gain = np.array([10.0, 1.0, 0.0, 0.1])
        for step in range(steps(1000)):
            action = np.clip(gain.dot(observation), -0.9, 0.9).reshape((1,)).astype(np.float32)
            observation, reward, terminated, truncated, info = env.step(action)
            if step % 200 == 0:
                torque = info["spine_observation"]["servo"]["left_wheel"]["torque"]
                print(f"step {step:4d}: pitch {observation[0]:+.4f} rad, ground position {observation[1]:+.4f} m, left wheel {torque:+.3f} N.m")
            if terminated or truncated:
                observation, _ = env.reset()
    print("done")

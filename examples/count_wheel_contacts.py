"""Contact points between the tires and the floor while balancing, as the
reference's examples/pybullet/count_wheel_contacts.py reads them from
`env.unwrapped.backend.get_contact_points(link_name)`."""
import numpy as np

from _common import steps

import upkie_amd.envs as envs

if __name__ == "__main__":
    with envs.make("Upkie-HIP-Pendulum", frequency=200.0) as env:
        observation, _ = env.reset()
        simulator = env.unwrapped.backend
        for step in range(steps(600)):
            v = 10.0 * observation[0] + 1.0 * observation[1] + 0.1 * observation[3]
            observation, _, terminated, truncated, _ = env.step(np.clip([v], -0.9, 0.9).astype(np.float32))
            if step % 100 == 0:
                left = simulator.get_contact_points("left_wheel_tire")
                right = simulator.get_contact_points("right_wheel_tire")
                print(f"step {step:3d}: left tire {len(left)} contact(s), right tire {len(right)} contact(s)")
                for contact in simulator.get_contact_points():
                    print(f"    {contact}")
            if terminated or truncated:
                observation, _ = env.reset()

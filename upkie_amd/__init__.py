"""upkie_amd -- MI355X-native batched implementation of Upkie's simulated
``env.step()`` hot path (Upkie-PyBullet-Pendulum / Gyropod / Servos /
BaseVelocity), behind the reference's environment and backend interfaces.

    import upkie_amd.envs
    env = upkie_amd.envs.make("Upkie-HIP-Pendulum-Vec", num_envs=4096, frequency=200.0)
"""

__version__ = "0.1.0"

"""Per-joint simulation properties (upkie/model/joint_properties.py:4-40)."""


class JointProperties:
    def __init__(
        self,
        friction: float = 0.0,
        torque_control_noise: float = 0.0,
        torque_measurement_noise: float = 0.0,
    ):
        self.friction = friction
        self.torque_control_noise = torque_control_noise
        self.torque_measurement_noise = torque_measurement_noise

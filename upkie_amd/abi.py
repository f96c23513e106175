"""ctypes mirror of ``include/upkie_hip.h`` (structs, enums, state words).

Keep this file in lock-step with the header: ``tests/test_abi.py`` compiles a
small C probe against the header and compares ``sizeof``/``offsetof`` of every
struct with the classes below.
"""

import ctypes as C

NB = 7  # merged bodies
NJ = 6  # actuated joints
MAX_LINKS = 24  # UPKIE_MAX_LINKS
INERTIAL_WORDS = 10  # UPKIE_INERTIAL_WORDS: mass, com (3), inertia about the com (6)

## Joint order of the reference (static_config.h:64-69).
JOINT_NAMES = (
    "left_hip",
    "left_knee",
    "left_wheel",
    "right_hip",
    "right_knee",
    "right_wheel",
)

## ACTION_KEYS of UpkieServos (upkie_servos.py:98-105).
ACTION_KEYS = (
    "position",
    "velocity",
    "feedforward_torque",
    "kp_scale",
    "kd_scale",
    "maximum_torque",
)

## Observation keys per servo (upkie_servos.py:218-254).
SERVO_OBS_KEYS = ("position", "velocity", "torque", "temperature", "voltage")

# State word indices (enum UpkieStateWord)
S_POS = 0
S_QUAT = 3
S_LINVEL = 7
S_ANGVEL = 10
S_Q = 13
S_QD = 19
S_LEGREF = 25
S_YAW = 29
S_YAWVEL = 30
S_TORQUE = 31
S_IMUVEL = 37
S_EPISODE = 40
S_DONE = 41
S_MPC_V = 42
S_SE2_X = 43
S_SE2_Y = 44
S_CONTACT = 45
S_STEP = 46
S_ELAPSED = 47  # steps of the current episode (time limit)
STATE_WORDS = 48
CONTACT_MANIFOLD_WORDS = 64  # UPKIE_CONTACT_MANIFOLD_WORDS (upkie_sim_set_contact_manifold)
CENSUS_WORDS = 72  # UPKIE_CENSUS_WORDS: eight counters + a 64-bin histogram (upkie_sim_set_census)
PENDULUM_STATE_WORDS = 29

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_UNSUPPORTED_MODEL = -2
ERR_HIP = -3
ERR_NO_DEVICE = -4

AUTORESET_DISABLED = 0
AUTORESET_NEXT_STEP = 1


class UpkieModel(C.Structure):
    _fields_ = [
        ("mass", C.c_double * NB),
        ("com", (C.c_double * 3) * NB),
        ("inertia", (C.c_double * 6) * NB),
        ("joint_pos", (C.c_double * 3) * NJ),
        ("joint_axis", (C.c_double * 3) * NJ),
        ("joint_lower", C.c_double * NJ),
        ("joint_upper", C.c_double * NJ),
        ("joint_effort", C.c_double * NJ),
        ("joint_velocity", C.c_double * NJ),
        ("joint_damping", C.c_double * NJ),
        ("wheel_radius", C.c_double),
        ("wheel_center", (C.c_double * 3) * 2),
        ("wheel_base", C.c_double),
        ("left_sign", C.c_double),
        ("imu_pos", C.c_double * 3),
        ("rot_base_to_imu", C.c_double * 9),
        ("gravity", C.c_double),
        ("contact_stiffness", C.c_double),
        ("contact_damping", C.c_double),
        ("friction_mu", C.c_double),
        ("friction_cfm", C.c_double),
        ("contact_breaking_threshold", C.c_double),
        ("base_linear_damping", C.c_double),
        ("base_angular_damping", C.c_double),
        ("max_joint_velocity", C.c_double),
        ("pgs_tolerance", C.c_double),
        ("pgs_iterations", C.c_int32),
        ("enforce_joint_limits", C.c_int32),
        ("num_links", C.c_int32),
        ("link_body", C.c_int32 * MAX_LINKS),
        ("link_randomized", C.c_int32 * MAX_LINKS),
        ("link_mass", C.c_double * MAX_LINKS),
        ("link_com", (C.c_double * 3) * MAX_LINKS),
        ("link_inertia", (C.c_double * 6) * MAX_LINKS),
    ]


class UpkieSimConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32),
        ("nb_substeps", C.c_int32),
        ("dt", C.c_double),
        ("torque_control_kp", C.c_double),
        ("torque_control_kd", C.c_double),
        ("joint_friction", C.c_double * NJ),
        ("torque_control_noise", C.c_double * NJ),
        ("torque_measurement_noise", C.c_double * NJ),
        ("fall_pitch", C.c_double),
        ("max_ground_velocity", C.c_double),
        ("max_yaw_velocity", C.c_double),
        ("leg_gain_scale", C.c_double),
        ("max_gain_scale", C.c_double),
        ("init_pos", C.c_double * 3),
        ("init_quat", C.c_double * 4),
        ("init_linvel", C.c_double * 3),
        ("init_angvel", C.c_double * 3),
        ("init_joint", C.c_double * NJ),
        ("rand_roll", C.c_double),
        ("rand_pitch", C.c_double),
        ("rand_x", C.c_double),
        ("rand_z", C.c_double),
        ("rand_omega_x", C.c_double),
        ("rand_omega_y", C.c_double),
        ("rand_linvel", C.c_double * 3),
        ("seed", C.c_uint64),
        ("env_id_offset", C.c_int64),
        ("autoreset_mode", C.c_int32),
        ("max_episode_steps", C.c_int32),
        ("agent_gains", C.c_double * 4),
        ("agent_clip", C.c_double),
    ]


class UpkieSpineObservation(C.Structure):
    _fields_ = [
        ("pitch", C.c_void_p),
        ("angular_velocity", C.c_void_p),
        ("linear_velocity", C.c_void_p),
        ("rotation_base_to_world", C.c_void_p),
        ("floor_contact", C.c_void_p),
        ("imu_orientation", C.c_void_p),
        ("imu_angular_velocity", C.c_void_p),
        ("imu_linear_acceleration", C.c_void_p),
        ("imu_raw_linear_acceleration", C.c_void_p),
        ("servo", C.c_void_p),
        ("wheel_odometry", C.c_void_p),
    ]


class UpkieMpcConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32),
        ("nb_timesteps", C.c_int32),
        ("admm_iterations", C.c_int32),
        ("reserved0", C.c_int32),
        ("sampling_period", C.c_double),
        ("leg_length", C.c_double),
        ("max_ground_accel", C.c_double),
        ("max_ground_velocity", C.c_double),
        ("fall_pitch", C.c_double),
        ("stage_input_cost_weight", C.c_double),
        ("stage_state_cost_weight", C.c_double),
        ("terminal_cost_weight", C.c_double),
        ("admm_rho", C.c_double),
        ("admm_relaxation", C.c_double),
    ]


MAX_EXTERNAL_FORCES = 16


class UpkieExternalForces(C.Structure):
    """Bodies, frames and application points of the external forces."""

    _fields_ = [
        ("count", C.c_int32),
        ("body", C.c_int32 * MAX_EXTERNAL_FORCES),
        ("local", C.c_int32 * MAX_EXTERNAL_FORCES),
        ("reserved0", C.c_int32),
        ("point", (C.c_double * 3) * MAX_EXTERNAL_FORCES),
    ]


class UpkieServoPolicy(C.Structure):
    """On-device servo-level policy (include/upkie_hip.h, upkie_sim_servo_policy)."""

    _fields_ = [
        ("action", (C.c_float * 6) * NJ),
        ("pitch_to_torque", C.c_float * NJ),
        ("pitch_to_velocity", C.c_float * NJ),
        ("position_to_velocity", C.c_float * NJ),
        ("velocity_to_velocity", C.c_float * NJ),
        ("velocity_feedback_clip", C.c_float * NJ),
        ("fall_pitch", C.c_float),
    ]


def torque_balancing_policy(gain: float = 10.0, fall_pitch: float = 1.0, left_sign: float = 1.0) -> "UpkieServoPolicy":
    """examples/pybullet/torque_balancing.py:15-37: legs held at zero, no
    velocity feedback in the wheels, wheel torques +-gain x pitch."""
    policy = UpkieServoPolicy()
    for j in range(NJ):
        wheel = j in (2, 5)
        policy.action[j][0] = float("nan") if wheel else 0.0  # position
        policy.action[j][1] = 0.0  # velocity
        policy.action[j][2] = 0.0  # feedforward torque
        policy.action[j][3] = 1.0  # kp_scale
        policy.action[j][4] = 0.0 if wheel else 1.0  # kd_scale
        policy.action[j][5] = 1.7 if wheel else 16.0  # maximum torque (clamped to the joint's effort limit by the step)
    policy.pitch_to_torque[2] = left_sign * gain
    policy.pitch_to_torque[5] = -left_sign * gain
    policy.fall_pitch = fall_pitch
    return policy


def velocity_balancing_policy(wheel_radius: float, fall_pitch: float = 1.0, left_sign: float = 1.0, gains=(10.0, 1.0, 0.1),
                              clip: float = 0.99) -> "UpkieServoPolicy":
    """The README's balancer at the servo level: legs held at zero, wheel
    velocity targets +-clip(g0 pitch + g1 p + g2 pdot, +-clip) / wheel_radius
    through the servos' velocity loop (kd_scale 1, no position target)."""
    policy = torque_balancing_policy(gain=0.0, fall_pitch=fall_pitch, left_sign=left_sign)
    for j, sign in ((2, left_sign), (5, -left_sign)):
        policy.action[j][4] = 1.0
        policy.pitch_to_velocity[j] = sign * gains[0] / wheel_radius
        policy.position_to_velocity[j] = sign * gains[1] / wheel_radius
        policy.velocity_to_velocity[j] = sign * gains[2] / wheel_radius
        policy.velocity_feedback_clip[j] = clip / wheel_radius
    return policy


class UpkieObserverConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32),
        ("reserved0", C.c_int32),
        ("dt", C.c_double),
        ("upper_leg_torque_threshold", C.c_double),
        ("wheel_cutoff_period", C.c_double),
        ("liftoff_inertia", C.c_double),
        ("min_touchdown_acceleration", C.c_double),
        ("min_touchdown_torque", C.c_double),
        ("touchdown_inertia", C.c_double),
        ("signed_radius", C.c_double * 2),
        ("rotation_base_to_imu", C.c_double * 9),
        ("rotation_ars_to_world", C.c_double * 9),
    ]


class UpkieObserverInput(C.Structure):
    _fields_ = [
        ("servo", C.c_void_p),
        ("imu_orientation", C.c_void_p),
        ("imu_angular_velocity", C.c_void_p),
        ("cross_button", C.c_void_p),
    ]


class UpkieObserverOutput(C.Structure):
    _fields_ = [
        ("base_pitch", C.c_void_p),
        ("base_angular_velocity", C.c_void_p),
        ("rotation_base_to_world", C.c_void_p),
        ("floor_contact", C.c_void_p),
        ("upper_leg_torque", C.c_void_p),
        ("wheel_contact", C.c_void_p),
        ("wheel_odometry", C.c_void_p),
    ]


# enum UpkieStructId -> the mirror of that struct here (`upkie_hip_struct_bytes`: lib.load() compares sizes)
STRUCT_IDS = {
    0: UpkieModel,
    1: UpkieSimConfig,
    2: UpkieExternalForces,
    3: UpkieServoPolicy,
    4: UpkieSpineObservation,
    5: UpkieMpcConfig,
    6: UpkieObserverConfig,
    7: UpkieObserverInput,
    8: UpkieObserverOutput,
}
MAX_GRAPH_CAPTURES = 8  # UPKIE_MAX_GRAPH_CAPTURES

# observer memory words (enum UpkieObserverStateWord)
O_WHEEL = 0
O_UPPER_LEG_TORQUE = 10
O_CONTACT = 11
O_ODOMETRY_POSITION = 12
O_ODOMETRY_VELOCITY = 13
OBSERVER_STATE_WORDS = 16
CONTACT_POINT_WORDS = 8  # UPKIE_CONTACT_POINT_WORDS
# enum UpkieObservationLayout
OBSERVATION_PENDULUM, OBSERVATION_PENDULUM_RECORDS, OBSERVATION_GYROPOD, OBSERVATION_SERVOS = 1, 2, 3, 4


def default_sim_config(
    num_envs: int = 1,
    frequency: float = 200.0,
    nb_substeps=None,
    seed: int = 0,
) -> UpkieSimConfig:
    """Config with the reference's defaults.

    ``dt = 1 / frequency`` (entry_points.py:57-58), ``nb_substeps =
    int(1000 * dt)`` (pybullet_backend.py:85-87), kp = 20, kd = 1
    (pybullet_backend.py:64-65), fall_pitch = 1, max_ground_velocity = 3
    (upkie_pendulum.py:70-71), max_yaw_velocity = 1, leg_gain_scale = 1
    (upkie_gyropod.py:107-110), max_gain_scale = 5 (upkie_servos.py:122),
    initial base position (0, 0, 0.6) (upkie_env.py:87-90).
    """
    cfg = UpkieSimConfig()
    dt = 1.0 / frequency
    cfg.num_envs = num_envs
    cfg.dt = dt
    cfg.nb_substeps = (
        int(nb_substeps) if nb_substeps is not None else int(1000.0 * dt)
    )
    cfg.torque_control_kp = 20.0
    cfg.torque_control_kd = 1.0
    cfg.fall_pitch = 1.0
    cfg.max_ground_velocity = 3.0
    cfg.max_yaw_velocity = 1.0
    cfg.leg_gain_scale = 1.0
    cfg.max_gain_scale = 5.0
    cfg.init_pos[:] = [0.0, 0.0, 0.6]
    cfg.init_quat[:] = [1.0, 0.0, 0.0, 0.0]
    cfg.seed = seed
    cfg.env_id_offset = 0
    cfg.autoreset_mode = AUTORESET_DISABLED
    cfg.agent_gains[:] = [10.0, 1.0, 0.0, 0.1]  # README.md:62
    cfg.agent_clip = 0.99  # examples/pybullet/pd_balancing.py:29
    return cfg


def default_mpc_config(num_envs: int = 1, nb_timesteps: int = 50):
    """MPCBalancer defaults (mpc_balancer.py:168-180)."""
    cfg = UpkieMpcConfig()
    cfg.num_envs = num_envs
    cfg.nb_timesteps = nb_timesteps
    # ADMM iterations per solve (warm-started from the previous step's solution) and the over-relaxation factor: 15
    # relaxed iterations at N <= 16 leave the first input within 4e-4 m/s2 of the exact one even from an unrelated warm
    # start (a_max 10; the contract is 2e-3 a_max), 7e-6 in closed loop; the reference's N = 50 keeps 30
    # (profiles/r04_mpc_iterations.txt)
    cfg.admm_iterations = 15 if nb_timesteps <= 16 else 30
    cfg.admm_relaxation = 1.5
    cfg.sampling_period = 0.02
    cfg.leg_length = 0.58
    cfg.max_ground_accel = 10.0
    cfg.max_ground_velocity = 3.0
    cfg.fall_pitch = 1.0
    cfg.stage_input_cost_weight = 1e-3
    cfg.stage_state_cost_weight = 1e-3
    cfg.terminal_cost_weight = 1.0
    cfg.admm_rho = 1e-3
    return cfg


def default_observer_config(num_envs: int = 1, dt: float = 1e-3) -> UpkieObserverConfig:
    """The spine's observer defaults (spine_backend.py:89-105,
    FloorContact.h:82, BaseOrientation.h:161-164); `dt` is the spine period
    (spines/common/observers.h:31-38)."""
    cfg = UpkieObserverConfig()
    cfg.num_envs = num_envs
    cfg.dt = dt
    cfg.upper_leg_torque_threshold = 10.0
    cfg.wheel_cutoff_period = 0.2
    cfg.liftoff_inertia = 0.001
    cfg.min_touchdown_acceleration = 2.0
    cfg.min_touchdown_torque = 0.015
    cfg.touchdown_inertia = 0.004
    cfg.signed_radius[0] = +0.05
    cfg.signed_radius[1] = -0.05
    for i in range(9):
        cfg.rotation_base_to_imu[i] = 0.0
        cfg.rotation_ars_to_world[i] = 0.0
    for i, v in enumerate((-1.0, 1.0, -1.0)):
        cfg.rotation_base_to_imu[4 * i] = v
    for i, v in enumerate((1.0, -1.0, -1.0)):
        cfg.rotation_ars_to_world[4 * i] = v
    return cfg

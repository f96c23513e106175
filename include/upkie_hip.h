/*
 * upkie_hip.h -- C-ABI of the MI355X-native batched Upkie simulation step.
 *
 * This is the drop-in boundary for the reference's backend plugin interface
 * (upkie/envs/backends/backend.py:11-50: reset / step / get_spine_observation
 * / close) batched over B independent environments. Every entry point takes
 * plain pointers and sizes; device pointers are owned by the caller (PyTorch
 * tensors on the Python side), launches go on the caller's hipStream_t and
 * nothing here synchronises the device. No exception crosses this boundary:
 * functions return 0 on success and a negative UpkieStatus on error, with a
 * message available from upkie_sim_last_error().
 *
 * Data layout (all device buffers, fp32):
 *   state    [UPKIE_STATE_WORDS][B]   struct-of-arrays, word index below
 *   act/obs  row-major [B][d] as gymnasium.vector would hand them over
 */
#ifndef UPKIE_HIP_H_
#define UPKIE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UPKIE_NB 7 /* merged rigid bodies: trunk, L{thigh,calf,wheel}, R{..} */
#define UPKIE_NJ 6 /* actuated joints, reference order (static_config.h:64-69,
                      kinematic_tree.py:105-127): left_hip, left_knee,
                      left_wheel, right_hip, right_knee, right_wheel */

/* ---- per-env state words (SoA rows of the state buffer) ---------------- */
enum UpkieStateWord {
  UPKIE_S_POS = 0,      /* 3: base position in world                        */
  UPKIE_S_QUAT = 3,     /* 4: base->world quaternion (w, x, y, z)           */
  UPKIE_S_LINVEL = 7,   /* 3: base linear velocity, world frame             */
  UPKIE_S_ANGVEL = 10,  /* 3: base angular velocity, world frame            */
  UPKIE_S_Q = 13,       /* 6: joint angles                                  */
  UPKIE_S_QD = 19,      /* 6: joint velocities                              */
  UPKIE_S_LEGREF = 25,  /* 4: hip/knee low-pass targets (lh, lk, rh, rk),
                              upkie_gyropod.py:246-267                      */
  UPKIE_S_YAW = 29,     /* 1: integrated commanded yaw, upkie_gyropod.py:383 */
  UPKIE_S_YAWVEL = 30,  /* 1: last commanded yaw velocity                   */
  UPKIE_S_TORQUE = 31,  /* 6: last commanded joint torques (servo "torque"
                              observation, pybullet_backend.py:456-458)     */
  UPKIE_S_IMUVEL = 37,  /* 3: previous IMU linear velocity (world), for the
                              finite-difference accelerometer :405-408      */
  UPKIE_S_EPISODE = 40, /* 1: number of resets done so far (RNG stream id),
                              modulo 2^24 (UPKIE_COUNTER_MASK)              */
  UPKIE_S_DONE = 41,    /* 1: 1.0 when the env terminated and awaits reset  */
  UPKIE_S_MPC_V = 42,   /* 1: MPCBalancer.commanded_velocity                */
  UPKIE_S_SE2_X = 43,   /* 1: dead-reckoned x, upkie_base_velocity.py:197   */
  UPKIE_S_SE2_Y = 44,   /* 1: dead-reckoned y                               */
  UPKIE_S_CONTACT = 45, /* 1: floor contact flag after the last substep     */
  UPKIE_S_STEP = 46,    /* 1: env.step() calls so far (torque-noise stream),
                              modulo 2^24                                    */
  UPKIE_S_ELAPSED = 47, /* 1: steps of the current episode (time limit)      */
  UPKIE_STATE_WORDS = 48
};

/* The counters are held as fp32 VALUES (so that they read naturally from a
 * float tensor): exact up to 2^24, then they wrap to 0 instead of sticking. */
#define UPKIE_COUNTER_MASK 0xFFFFFFu

/* Words the fused Pendulum step reads and writes (29 physics/filter words +
 * the done flag); everything else is untouched by that kernel. */
#define UPKIE_PENDULUM_STATE_WORDS 29

/* URDF links the model remembers (upkie_description's Upkie has 15 with mass). */
#define UPKIE_MAX_LINKS 24
/* Per-env inertial record of one composite body: mass, com (3, body frame),
 * inertia about the com (xx yy zz xy xz yz, body axes). */
#define UPKIE_INERTIAL_WORDS 10

enum UpkieStatus {
  UPKIE_OK = 0,
  UPKIE_ERR_INVALID_ARGUMENT = -1,
  UPKIE_ERR_UNSUPPORTED_MODEL = -2,
  UPKIE_ERR_HIP = -3,
  UPKIE_ERR_NO_DEVICE = -4
};

enum UpkieAutoreset {
  UPKIE_AUTORESET_DISABLED = 0, /* caller resets through a mask             */
  UPKIE_AUTORESET_NEXT_STEP = 1 /* gymnasium.vector default: an env that
                                   terminated at step t is reset by step t+1,
                                   which ignores its action                 */
};

/* Merged-fixed-link rigid-body model. Body frames are located at their joint
 * origin and aligned with the base frame at the zero configuration, so the
 * tree transforms are pure translations (the host-side URDF loader does this
 * canonicalisation). Replaces what the reference gets from upkie_description's
 * URDF through pybullet.loadURDF (pybullet_backend.py:121-125) and
 * upkie/model/model.py:63-110. */
typedef struct UpkieModel {
  double mass[UPKIE_NB];
  double com[UPKIE_NB][3];        /* centre of mass in body frame           */
  double inertia[UPKIE_NB][6];    /* about the com: xx yy zz xy xz yz       */
  double joint_pos[UPKIE_NJ][3];  /* joint origin in the parent body frame  */
  double joint_axis[UPKIE_NJ][3]; /* unit rotation axis in body frame       */
  double joint_lower[UPKIE_NJ];
  double joint_upper[UPKIE_NJ];
  double joint_effort[UPKIE_NJ];   /* N.m   */
  double joint_velocity[UPKIE_NJ]; /* rad/s */
  double joint_damping[UPKIE_NJ];  /* URDF <dynamics damping>, N.m.s/rad    */
  double wheel_radius;             /* tire collision cylinder radius        */
  double wheel_center[2][3];       /* tire centre in the wheel body frame   */
  double wheel_base;               /* model.py:88 */
  double left_sign;                /* +1 if left-wheeled, model.py:104      */
  double imu_pos[3];               /* IMU origin in base frame              */
  double rot_base_to_imu[9];       /* row-major, model.py:106               */
  double gravity;                  /* 9.81, pybullet_backend.py:110         */
  double contact_stiffness;        /* tire <contact> stiffness              */
  double contact_damping;          /* tire <contact> damping                */
  double friction_mu;              /* tire x plane lateral friction         */
  double friction_cfm;             /* compliance added to friction rows, 1/kg:
                                      the two tires' lateral rows coincide in a
                                      symmetric stance (singular otherwise)  */
  double contact_breaking_threshold; /* floor_contact flag distance         */
  double base_linear_damping;      /* Bullet default 0.04                   */
  double base_angular_damping;     /* Bullet default 0.04                   */
  double max_joint_velocity;       /* Bullet maxCoordinateVelocity, 100     */
  double pgs_tolerance;            /* sweeps stop once no impulse moved by more
                                      than this fraction of the largest one;
                                      the fp32 kernels do not go below 1e-5  */
  int32_t pgs_iterations;          /* Bullet numSolverIterations, 50        */
  int32_t enforce_joint_limits;    /* hip/knee limit rows in the solver     */
  /* The URDF links each composite body was fused from: what Bullet keeps as
   * separate links (no URDF_MERGE_FIXED_LINKS, pybullet_backend.py:121-125) and
   * randomize_inertias() scales one by one (pybullet_backend.py:555-601).
   * num_links = 0: one link per body. */
  int32_t num_links;
  int32_t link_body[UPKIE_MAX_LINKS];       /* composite body of the link   */
  int32_t link_randomized[UPKIE_MAX_LINKS]; /* 0 for the URDF root link: Bullet's
                                               base (index -1) is not in
                                               range(getNumJoints), :563     */
  double link_mass[UPKIE_MAX_LINKS];
  double link_com[UPKIE_MAX_LINKS][3];      /* link com in its body's frame  */
  double link_inertia[UPKIE_MAX_LINKS][6];  /* about the link com, body axes:
                                               xx yy zz xy xz yz             */
} UpkieModel;

/* Everything gym.make(...) kwargs decide for one batch of environments
 * (entry_points.py:41-54,99-105; upkie_servos.py:115-124;
 * upkie_gyropod.py:104-111; joint_properties.py:4-40;
 * robot_state.py:52-120; robot_state_randomization.py:53-90). */
typedef struct UpkieSimConfig {
  int32_t num_envs;     /* B, envs hosted by this handle                    */
  int32_t nb_substeps;  /* int(1000 * dt) unless given, backend :85-87      */
  double dt;            /* 1 / frequency                                    */
  double torque_control_kp; /* 20.0 */
  double torque_control_kd; /* 1.0  */
  double joint_friction[UPKIE_NJ];           /* JointProperties.friction   */
  double torque_control_noise[UPKIE_NJ];     /* std dev, N.m               */
  double torque_measurement_noise[UPKIE_NJ]; /* std dev, N.m               */
  double fall_pitch;          /* 1.0 */
  double max_ground_velocity; /* 3.0 */
  double max_yaw_velocity;    /* 1.0 */
  double leg_gain_scale;      /* 1.0 */
  double max_gain_scale;      /* 5.0 */
  /* nominal initial state (RobotState) */
  double init_pos[3];
  double init_quat[4]; /* w x y z */
  double init_linvel[3];
  double init_angvel[3];
  double init_joint[UPKIE_NJ];
  /* RobotStateRandomization magnitudes */
  double rand_roll, rand_pitch, rand_x, rand_z, rand_omega_x, rand_omega_y;
  double rand_linvel[3];
  uint64_t seed;          /* Philox key                                     */
  int64_t env_id_offset;  /* global index of local env 0 (multi-GPU shards) */
  int32_t autoreset_mode; /* UpkieAutoreset                                 */
  int32_t max_episode_steps; /* gymnasium TimeLimit for the batch: the step that
                                brings an episode to this many steps reports
                                `truncated` (unless the robot fell in it) and
                                flags the env done; 0 = no limit            */
  /* Fused linear-feedback agent (README.md:62-64): a = clip(g . obs, +-c)  */
  double agent_gains[4];
  double agent_clip;
} UpkieSimConfig;

typedef struct UpkieSim UpkieSim;

/* Library / device probe. Returns the number of visible HIP devices (>= 0)
 * or a negative status. */
int upkie_hip_device_count(void);

/* sizeof() of a public struct AS THIS BUILD OF THE LIBRARY SEES IT (-1 for an
 * unknown id): a binding written against one version of this header checks it
 * against its own mirror of the struct before the first call -- the config
 * structs grow at the tail from release to release (UpkieMpcConfig gained
 * admm_relaxation), and a library loaded from elsewhere (UPKIE_HIP_LIBRARY)
 * that reads a longer struct than the caller wrote reads garbage silently.
 * upkie_amd/lib.py refuses to load a library whose sizes differ from
 * upkie_amd/abi.py's. No counterpart in the reference (its backends are
 * Python objects). */
enum UpkieStructId {
  UPKIE_STRUCT_MODEL = 0,
  UPKIE_STRUCT_SIM_CONFIG = 1,
  UPKIE_STRUCT_EXTERNAL_FORCES = 2,
  UPKIE_STRUCT_SERVO_POLICY = 3,
  UPKIE_STRUCT_SPINE_OBSERVATION = 4,
  UPKIE_STRUCT_MPC_CONFIG = 5,
  UPKIE_STRUCT_OBSERVER_CONFIG = 6,
  UPKIE_STRUCT_OBSERVER_INPUT = 7,
  UPKIE_STRUCT_OBSERVER_OUTPUT = 8,
  UPKIE_STRUCT_COUNT = 9
};
int64_t upkie_hip_struct_bytes(int which);

/* Streams and hipGraphs. Every launching entry point takes the caller's
 * hipStream_t (`stream`, NULL = the default stream) and only enqueues work on
 * it. The eight-lane step kernels read the handle's limits and configuration
 * from a block in device memory, refreshed by a small store kernel enqueued in
 * front of the step whenever a setting changed (upkie_sim_set_config, ...) or
 * the launching stream did. What that allows:
 *   - eager launches: the handle keeps TWO such blocks and alternates, so the
 *     steps of one handle may be in flight on up to two streams at a time
 *     (a third stream must wait for the first to drain: the handle does not
 *     track completion);
 *   - launches recorded into a hipGraph (stream capture, e.g.
 *     torch.cuda.graph / upkie_amd.graphs): each CAPTURE gets a block of its
 *     own and records its own store of the settings as they are at capture
 *     time, so several graphs of one handle -- recorded at different settings,
 *     replayed in any order, mixed with eager launches -- each replay with the
 *     settings they were recorded with. The blocks are allocated with the
 *     handle (a capture cannot allocate): at most UPKIE_MAX_GRAPH_CAPTURES
 *     captures per handle, the next one is refused with UPKIE_ERR_HIP and a
 *     message. The library cannot see a graph being destroyed: a long-lived
 *     env that re-captures (after a setting change, after an aborted capture)
 *     hands the blocks back with upkie_sim_release_graph_captures() once none
 *     of the graphs recorded so far will be replayed again. A setting changed
 *     AFTER a capture does not reach that graph's replays: re-capture.
 * The one- and two-lane kernels take their settings by value: no limit. */
#define UPKIE_MAX_GRAPH_CAPTURES 8

/* Create a simulation handle on the current HIP device: validates the model
 * (all joint axes must be lateral, i.e. +-y of the base frame, as on every
 * Upkie; UPKIE_ERR_UNSUPPORTED_MODEL otherwise) and uploads constants.
 * Replaces PyBulletBackend.__init__ (pybullet_backend.py:55-197). */
int upkie_sim_create(const UpkieSimConfig* config, const UpkieModel* model,
                     UpkieSim** out);
int upkie_sim_destroy(UpkieSim* sim);
const char* upkie_sim_last_error(const UpkieSim* sim);
/* "Streams and hipGraphs" above: every hipGraph recorded from this handle so
 * far is dead to the caller; its settings blocks may be handed out again. */
int upkie_sim_release_graph_captures(UpkieSim* sim);

/* Replace the configuration of an existing handle (same num_envs): lets a
 * caller change init state, randomisation magnitudes, seed, gains, fall pitch
 * ... between resets, as the reference's mutable env attributes allow
 * (upkie_env.py:244-251 update_init_rand, upkie_gyropod.py:176-184). */
int upkie_sim_set_config(UpkieSim* sim, const UpkieSimConfig* config);

/* Size in bytes the caller must allocate for the state buffer. */
int64_t upkie_sim_state_bytes(const UpkieSim* sim);

/* The relative tolerance the Gauss-Seidel sweeps of this handle's kernels stop
 * at: UpkieModel.pgs_tolerance, but never below 1e-5 -- the kernels sweep in
 * fp32, where a six-term residual cannot be told from zero below that, and a
 * tighter setting only ran systems into the iteration cap (the fp64 oracle
 * honours the model's value). */
double upkie_sim_pgs_tolerance(const UpkieSim* sim);

/* Lanes of a wavefront that share one env in the step kernels of this handle:
 * chosen from the batch size at creation (small batches spread every env over
 * several lanes so that the chip's SIMDs all hold a wave; large ones keep one
 * env per lane), or forced by the environment variable UPKIE_LANES_PER_ENV
 * (tests, sweeps). Results agree across mappings to fp32 rounding -- where a
 * contact solution leaves its friction box, to the solver's tolerance: the
 * eight-lane kernel answers such substeps by an active-set solve, the others
 * by sweeps -- and bit for bit within one. */
int upkie_sim_lanes_per_env(const UpkieSim* sim);
/* ... of the step kernel a given entry point launches, named by the layout of
 * its observation output (UpkieObservationLayout): the Servos kernels, which
 * take the whole register file, leave the eight-lane mapping at 8192 envs, the
 * others at 16384 -- a caller that keys on the mapping (the census below is
 * counted by the eight-lane kernels only; the SAME_STEP autoreset runs inside
 * the launch there) asks for the entry point it uses. */
int upkie_sim_lanes_per_env_of(const UpkieSim* sim, int observation_layout);
/* Force the mapping of this handle's later launches: 1, 2 or 8 lanes per env,
 * 0 = back to the choice by batch size (what UPKIE_LANES_PER_ENV sets at
 * creation). A forced mapping still yields where it does not exist (the
 * eight-lane kernels beyond 2^32 bytes of state or with forces on leg links,
 * the two-lane kernels under the Bullet-like contact model). */
int upkie_sim_set_lanes_per_env(UpkieSim* sim, int lanes);

/* Non-finite commands and states. The reference has one robot and asserts
 * (`assert not np.isnan(target_velocity)`, pybullet_backend.py:519); a batch of
 * thousands must not stop -- nor keep a poisoned env for ever -- because one
 * policy output diverged. Every step entry point therefore
 *  1. replaces what is still NOT FINITE behind the reference's clamp
 *     (upkie_servos.py:331-342; only NaN survives a clamp, and an infinite
 *     position target of a joint without position limits) by the NEUTRAL
 *     action's value (upkie_servos.py:255-262): velocity 0, feedforward torque
 *     0, kp_scale 1, kd_scale 1, maximum_torque = the joint's effort limit; an
 *     infinite position becomes NaN, which is the neutral position ("no
 *     position term"). A NaN ground / yaw velocity action of the Pendulum /
 *     Gyropod / BaseVelocity steps is 0, an infinite yaw velocity -- the yaw
 *     word integrates the unclamped action, upkie_gyropod.py:383-385 -- is
 *     max_yaw_velocity with its sign; a non-finite target velocity of the MPC
 *     balancer is 0;
 *  2. looks at the env's state behind the substeps: if a word of it is not
 *     finite (a force, an inertial record or an uploaded state word was not),
 *     the env is put into UpkieSimConfig's initial state WITHOUT randomisation,
 *     at rest, its contact cache dropped; the step reports `terminated = 1`
 *     for it -- every env kind, UpkieServos included -- and flags it done, so
 *     the autoreset treats it like a fall (NEXT_STEP: re-initialised by its
 *     next step; SAME_STEP: inside this call; disabled: until the caller
 *     resets it). No NaN leaves a step in an observation.
 * Finite values, however large, go through the reference's arithmetic
 * untouched (its clamps bound them). Both events are counted per handle:
 * counts[0] = command words replaced, counts[1] = env states replaced since
 * creation (or the last call with reset != 0); the call waits for `stream`.
 * The sound envs pay about twenty compares per step for this. */
int upkie_sim_guard_counts(UpkieSim* sim, uint32_t counts[2], int reset, void* stream);

/* gymnasium's SAME_STEP autoreset completed by the step calls themselves: with
 * `final_obs` set (a device buffer shaped like the step's observation output:
 * [B][4] Pendulum, [B][6] Gyropod, [B][6][5] Servos) and
 * UpkieSimConfig::autoreset_mode == UPKIE_AUTORESET_DISABLED (no NEXT_STEP reset),
 * upkie_sim_step_pendulum / _gyropod / _servos return with every env's last
 * observation in `final_obs`, the finished envs re-initialised and their rows
 * of the observation output replaced by the reset observation (reward and flags
 * are those of the step) -- what upkie_sim_autoreset_done does as a second call,
 * inside the same launch where the lane mapping allows it (batches up to 8192
 * envs), as a second launch otherwise. NULL (the default): the step calls only
 * flag the finished envs. Mirrors gymnasium.vector's AutoresetMode.SAME_STEP
 * (the reference's envs are single robots: no counterpart there). */
int upkie_sim_set_final_observation(UpkieSim* sim, float* final_obs);

/* Contact model. Default (`manifold` NULL): the product's specification -- one
 * contact point per tire, exact solve; when the solution leaves the friction
 * box, an active-set solve (eight-lane kernel) and Gauss-Seidel sweeps to
 * convergence, friction CFM 0.01. With `manifold` set -- the caller's device buffer
 * [UPKIE_CONTACT_MANIFOLD_WORDS][B] fp32, zeroed by the caller, kept between
 * steps -- every step of this handle solves contacts and joint limits the way
 * Bullet's multibody solver is published to inside pybullet.stepSimulation()
 * (call sites pybullet_backend.py:228,306; third
 * party, absent here: restated, unverified): a persistent manifold of up to
 * four points per tire (refresh against the breaking threshold, nearest cached
 * point replaced), one normal + two friction rows per point along / across
 * its sliding velocity, no friction CFM, a FIXED number of sequential-impulse
 * sweeps (UpkieModel.pgs_iterations = 50; joint limits, normals, then each
 * point's friction pair projected onto the cone), normal impulses warm-started
 * with 0.85 x the last applied ones. Per env and tire four records of 8 words:
 * point in the wheel frame (3), on the plane in world coordinates (3), applied
 * normal impulse, live flag. A reset clears an env's manifold. Two kernels run
 * this model: the one-env-per-lane step kernels cover every case (several
 * points on a tire, joints at their stops in the same solve, any batch size,
 * every entry point); up to 16384 envs (upkie_sim_step_servos: 8192) the step
 * entry points run it on eight lanes per env, in the case a rolling wheel
 * produces (one cached point per tire, which the tire's deepest point replaces
 * every substep: the default model's contact point with the friction rows
 * rotated into the sliding direction; a robot lying flat on its side, whose
 * tires may cache several points under Bullet's rule, keeps the deepest one
 * on this variant -- upkie_sim_set_lanes_per_env(sim, 1) selects the one-lane
 * kernels; a joint within reach of its stop is a row of the same 50 sweeps on
 * both, since round 6 -- until then the eight-lane variant answered such a
 * substep with the default model's joint-stop solve --, and is counted by the
 * census, word [0]).
 * Both keep complete manifold records, so either continues from a manifold the
 * other wrote. About 2.3 x the default model's Pendulum step (the sweeps are ~36 packed-
 * fp32 instructions each); the default stays the product's fast
 * specification. This is the model to answer "what would PyBullet's contact
 * pipeline do", e.g. the first thing tools/compare_with_pybullet.py holds
 * against the real thing. */
#define UPKIE_CONTACT_MANIFOLD_WORDS 64
int upkie_sim_set_contact_manifold(UpkieSim* sim, float* manifold);

/* Rare-path census of the eight-lanes-per-env step kernel (octet.hpp): `counters`
 * is the caller's device buffer of UPKIE_CENSUS_WORDS uint32 (zeroed by the
 * caller) or NULL to switch the census off (the default). Env-substeps:
 *   [0] a hip / knee at its stop (contacts and limit rows through the general
 *       solver over scratch memory)
 *   [2] contact impulses outside the friction cone / pulling (an active-set
 *       solve, then projected Gauss-Seidel sweeps)
 *   [3] those of [2] an active-set solve answered (no sweep; since round 5)
 * wavefront-substeps (what the paths cost) that took, for at least one of their
 * eight envs, [4] the joint-stop path, [5] the sweeps. Sweeps run by the
 * env-substeps of [2]: [6] their sum, [7] the largest count, [1] how many
 * stopped at the iteration cap. Words [8 .. 71]: histogram
 * over the wavefront-substeps of [5] of the LARGEST sweep count among the
 * wavefront's envs (bin 63: 63 or more) -- what a launch waits for, since a
 * wavefront leaves the sweeps with its slowest env. Diagnostics only: no entry
 * point of the reference corresponds to it, and its atomics (up to five per
 * wavefront and substep on a rare path) are not free: time without it. */
#define UPKIE_CENSUS_WORDS 72
int upkie_sim_set_census(UpkieSim* sim, uint32_t* counters);

/* Optional per-env domain randomisation buffers (device pointers, may be
 * NULL): body_inertials[UPKIE_NB * UPKIE_INERTIAL_WORDS][B] replaces mass,
 * centre of mass and inertia of every composite body of every env (filled by
 * upkie_sim_sample_body_inertials; pybullet_backend.py:571-601); ext_force[3][B]
 * is a world-frame force applied at point `ext_point` of the trunk, re-applied
 * every substep until overwritten (pybullet_backend.py:603-658). */
int upkie_sim_set_randomization(UpkieSim* sim, const float* body_inertials,
                                const float* ext_force,
                                const double ext_point[3]);

/* External forces on any link, PyBulletBackend.set_external_forces /
 * __apply_external_forces (pybullet_backend.py:603-658): up to
 * UPKIE_MAX_EXTERNAL_FORCES forces act at the same time, each on one composite
 * body (0 trunk, 1-3 left thigh / calf / wheel, 4-6 right) at `point` given in
 * that body's frame (Bullet applies the force at the link's centre of mass),
 * expressed in the world frame (pybullet.WORLD_FRAME) or, with `local`, in the
 * body frame (pybullet.LINK_FRAME). forces[count][3][B] is a device buffer
 * read at every substep until replaced; NULL or count = 0 removes all forces.
 * Replaces whatever upkie_sim_set_randomization installed as ext_force. */
#define UPKIE_MAX_EXTERNAL_FORCES 16 /* one slot per link: upkie_description's Upkie has 15 */
typedef struct UpkieExternalForces {
  int32_t count;
  int32_t body[UPKIE_MAX_EXTERNAL_FORCES];
  int32_t local[UPKIE_MAX_EXTERNAL_FORCES];
  int32_t reserved0;
  double point[UPKIE_MAX_EXTERNAL_FORCES][3];
} UpkieExternalForces;

int upkie_sim_set_external_forces(UpkieSim* sim, const float* forces,
                                  const UpkieExternalForces* slots);

/* PyBulletBackend.randomize_inertias (pybullet_backend.py:571-601) for every
 * env: one epsilon ~ U(-v, v) per URDF link and env scales that link's mass
 * and inertia by (1 + epsilon); the links of each composite body are then
 * fused again (mass, centre of mass, inertia about it) into
 * body_inertials[UPKIE_NB * UPKIE_INERTIAL_WORDS][B], row 10 * body + word.
 * link_scale (may be NULL) receives the factors, [UPKIE_MAX_LINKS][B]. */
int upkie_sim_sample_body_inertials(UpkieSim* sim, float* body_inertials,
                                    float* link_scale,
                                    double inertia_variation, void* stream);

/* Push domain randomisation (BASELINE.json configs[4] "push-domain-randomisation";
 * the reference applies pushes through PyBulletBackend.set_external_forces,
 * pybullet_backend.py:603-658, with forces the user script draws:
 * examples/pybullet/apply_external_forces.py:37-43): push number `push_index`
 * of every env, a world-frame force with norm ~ U(0, max_norm) and a uniformly
 * random horizontal direction, drawn on the device from the Philox stream
 * keyed by (seed, global env id, push_index) into force[3][B] -- the buffer
 * upkie_sim_set_randomization / upkie_sim_set_external_forces read at every
 * substep. Zero the buffer to end the push. */
int upkie_sim_sample_pushes(UpkieSim* sim, float* force, uint32_t push_index,
                            double max_norm, void* stream);

/* Diagnostic: the projected Gauss-Seidel sweeps the step kernels run when the
 * direct contact solution leaves its friction cone (what Bullet's
 * btMultiBodyConstraintSolver iterates inside pybullet.stepSimulation(),
 * pybullet_backend.py:306), on caller-provided 6 x 6 systems of the two tires
 * (rows: normal, rolling, lateral of the left tire, then of the right one), one
 * system per lane, with this handle's friction coefficient, tolerance and
 * iteration cap: A[n][21] packed lower by rows with the CFM on the diagonal,
 * rhs[n][6], lam[n][6] (in: warm start, out: impulses), both_tires[n] (0: one
 * tire is off the floor, its rows are identity rows with zero right-hand
 * sides and no coupling; informative since round 4: every system runs the
 * same loop, which such rows pass through unchanged), sweeps[n] (may be NULL)
 * receives the sweeps each system ran. Device pointers. */
int upkie_sim_contact_sweeps(UpkieSim* sim, int32_t num_systems, const float* A,
                             const float* rhs, float* lam,
                             const uint8_t* both_tires, int32_t* sweeps,
                             void* stream);

/* Reset the envs whose mask byte is non-zero (all when mask is NULL):
 * sample the initial state in the reference's draw order (robot_state.py:
 * 182-187), zero joint velocities, run the one extra torque-free physics
 * substep (pybullet_backend.py:220-232) and seed the Gyropod filter state
 * (upkie_gyropod.py:236-240). obs6 (may be NULL) receives the Gyropod
 * observation [B][6]. */
int upkie_sim_reset(UpkieSim* sim, float* state, const uint8_t* mask,
                    float* obs6, void* stream);

/* One env.step() of UpkiePendulum (upkie_pendulum.py:124-142) for every env:
 * act[B] ground velocity -> obs[B][4] = [pitch, position, pitch rate, velocity],
 * reward[B] = 0, terminated[B], truncated[B] = 0. */
int upkie_sim_step_pendulum(UpkieSim* sim, float* state, const float* act,
                            float* obs, float* reward, uint8_t* terminated,
                            uint8_t* truncated, void* stream);

/* Same step with the README's linear-feedback agent evaluated on-device from
 * the previous observation held in `obs` (in/out). */
int upkie_sim_step_pendulum_agent(UpkieSim* sim, float* state, float* obs,
                                  float* reward, uint8_t* terminated,
                                  uint8_t* truncated, void* stream);

/* Packed-record variants for rollout collection across GPUs: instead of four
 * output arrays each env gets one 32-byte record records[B][8] =
 * [obs(4) | reward, terminated, truncated, 0] (two 16-byte stores per lane),
 * which is what one RCCL gather per step ships to rank 0. The agent variant
 * reads the previous observation from the record. */
int upkie_sim_step_pendulum_packed(UpkieSim* sim, float* state,
                                   const float* act, float* records,
                                   void* stream);
int upkie_sim_step_pendulum_agent_packed(UpkieSim* sim, float* state,
                                         float* records, void* stream);
/* Double-buffered form: the agent reads the previous observation from
 * prev_records (a different buffer) so that a gather of prev_records can still
 * be in flight while this step runs. */
int upkie_sim_step_pendulum_agent_records(UpkieSim* sim, float* state,
                                          const float* prev_records,
                                          float* records, void* stream);

/* num_steps consecutive env.step() of the same fused-agent Pendulum env, i.e.
 * num_steps calls of upkie_sim_step_pendulum_agent_records with
 * records[k - 1] as the previous records of step k (prev_records for k = 0):
 * records [num_steps][B][8]. Up to 32768 envs this is ONE launch in which the
 * state stays in registers from step to step (the agent needs nothing from the
 * host in between); results are bit-identical to the step-by-step calls. */
int upkie_sim_step_pendulum_agent_rollout(UpkieSim* sim, float* state,
                                          const float* prev_records,
                                          float* records, int32_t num_steps,
                                          void* stream);

/* One env.step() of UpkieGyropod (upkie_gyropod.py:354-392):
 * act[B][2] -> obs[B][6]. Buffers of row-major observations, actions and
 * records are read and written with 8- and 16-byte accesses: keep them
 * 16-byte aligned (any allocator's default). */
int upkie_sim_step_gyropod(UpkieSim* sim, float* state, const float* act,
                           float* obs, float* reward, uint8_t* terminated,
                           uint8_t* truncated, void* stream);

/* One env.step() of UpkieBaseVelocity (upkie_base_velocity.py:164-202) after
 * upkie_mpc_step_env(): act[B][2] = [linear velocity, yaw velocity]; the
 * ground velocity commanded to the Gyropod layer is commanded_velocity[B]
 * (the MPC balancer's output). obs[B][3] = dead-reckoned [x, y, yaw]. Also
 * writes what the balancer reads at the next step: mpc_x0[B][4] = [ground
 * position, pitch, ground velocity, pitch rate], mpc_contact[B]. */
int upkie_sim_step_base_velocity(UpkieSim* sim, float* state, const float* act,
                                 const float* commanded_velocity, float* obs,
                                 float* mpc_x0, uint8_t* mpc_contact,
                                 float* reward, uint8_t* terminated,
                                 uint8_t* truncated, void* stream);

/* The same step with MPCBalancer.step() in front of it in ONE launch
 * (upkie_base_velocity.py:164-202 as a whole): equivalent to
 * upkie_mpc_step_env(mpc, workspace, mpc_x0, act, mpc_contact, <done words of
 * the state when autoreset is on>, dt, commanded_velocity) followed by
 * upkie_sim_step_base_velocity(...), bit for bit. One kernel when envs are
 * mapped two lanes each and the horizon fits one 16-row tile (the wavefront
 * that steps 32 envs first solves their condensed QPs on the matrix cores and
 * hands the velocities over through LDS); the two launches otherwise. */
struct UpkieMpc;
int upkie_sim_step_base_velocity_mpc(UpkieSim* sim, struct UpkieMpc* mpc, float* state, float* workspace,
                                     const float* act, float* commanded_velocity, float* obs,
                                     float* mpc_x0, uint8_t* mpc_contact, float* reward,
                                     uint8_t* terminated, uint8_t* truncated, void* stream);

/* One env.step() of UpkieServos (upkie_servos.py:316-344 + upkie_env.py:
 * 196-242): act[B][6][6] in ACTION_KEYS order (position, velocity,
 * feedforward_torque, kp_scale, kd_scale, maximum_torque) -> obs[B][6][5]
 * (position, velocity, torque, temperature, voltage). */
int upkie_sim_step_servos(UpkieSim* sim, float* state, const float* act,
                          float* obs, float* reward, uint8_t* terminated,
                          uint8_t* truncated, void* stream);

/* On-device servo-level policy for UpkieServos: ONE small launch that writes the
 * action buffer act[B][6][6] of the next upkie_sim_step_servos from the state,
 * so that a servo-level agent needs no host round trip (BASELINE.json
 * configs[4]). The law is examples/pybullet/torque_balancing.py:15-37 with
 * room for the README's velocity feedback: every joint j starts from
 * `action[j]` (ACTION_KEYS order) and gets
 *   feedforward_torque += pitch_to_torque[j] * pitch
 *   velocity           += clip(pitch_to_velocity[j] * pitch + position_to_velocity[j] * p + velocity_to_velocity[j] * pdot,
 *                               +-velocity_feedback_clip[j])        (no clipping where velocity_feedback_clip[j] <= 0)
 * with pitch = base_orientation.pitch of the spine observation and p, pdot the
 * ground position / velocity of upkie_gyropod.py:186-214 (wheel odometry).
 * Envs with |pitch| > fall_pitch get their UPKIE_S_DONE word set: with
 * UPKIE_AUTORESET_NEXT_STEP the step that follows re-initialises them (what the
 * example's `if terminated or truncated: env.reset()` does); fall_pitch <= 0
 * switches that off. */
typedef struct UpkieServoPolicy {
  float action[UPKIE_NJ][6];
  float pitch_to_torque[UPKIE_NJ];
  float pitch_to_velocity[UPKIE_NJ];
  float position_to_velocity[UPKIE_NJ];
  float velocity_to_velocity[UPKIE_NJ];
  float velocity_feedback_clip[UPKIE_NJ];
  float fall_pitch;
} UpkieServoPolicy;
int upkie_sim_servo_policy(UpkieSim* sim, float* state, const UpkieServoPolicy* policy, float* act, void* stream);
/* upkie_sim_servo_policy + upkie_sim_step_servos as one call: on the
 * eight-lanes-per-env mapping (batches up to 8192 envs) the policy is evaluated
 * inside the step's launch, each joint lane computing its own command from the
 * state the step starts from, and `act` is left untouched; on the other
 * mappings the two launches run one behind the other through `act` ([B][6][6],
 * caller's scratch). Same results up to the rounding of the feedback sum. */
int upkie_sim_step_servos_policy(UpkieSim* sim, float* state, const UpkieServoPolicy* policy, float* act, float* obs, float* reward,
                                 uint8_t* terminated, uint8_t* truncated, void* stream);

/* Full spine observation (pybullet_backend.py:313-490), materialised lazily.
 * Any pointer may be NULL. */
typedef struct UpkieSpineObservation {
  float* pitch;                  /* [B]                                     */
  float* angular_velocity;       /* [B][3] base in base                     */
  float* linear_velocity;        /* [B][3] base in world                    */
  float* rotation_base_to_world; /* [B][9] row-major                        */
  uint8_t* floor_contact;        /* [B]                                     */
  float* imu_orientation;        /* [B][4] w x y z, IMU in ARS              */
  float* imu_angular_velocity;   /* [B][3]                                  */
  float* imu_linear_acceleration;     /* [B][3]                             */
  float* imu_raw_linear_acceleration; /* [B][3]                             */
  float* servo;                  /* [B][6][5]                               */
  float* wheel_odometry;         /* [B][2] position, velocity               */
} UpkieSpineObservation;

/* gymnasium.vector AutoresetMode.SAME_STEP: the step that ends an episode
 * also returns the first observation of the next one, the terminal observation
 * going to info["final_obs"]. Call after a step made with
 * UPKIE_AUTORESET_DISABLED: every env whose DONE word is set (by the step when
 * the robot fell, or by the caller for a time limit: row UPKIE_S_DONE of the
 * state) is re-initialised exactly as upkie_sim_reset does it and its row of
 * `obs` (the buffer the step wrote, in that step's layout) replaced by the
 * reset observation; `final_obs` (same layout, may be NULL; records: [B][4])
 * receives the step's observation of EVERY env first. Reward and flags of the
 * terminal step stay; envs that are not done are otherwise untouched. One
 * launch. */
enum UpkieObservationLayout {
  UPKIE_OBSERVATION_PENDULUM = 1,         /* [B][4], upkie_sim_step_pendulum            */
  UPKIE_OBSERVATION_PENDULUM_RECORDS = 2, /* [B][8] records, ..._step_pendulum_packed   */
  UPKIE_OBSERVATION_GYROPOD = 3,          /* [B][6], upkie_sim_step_gyropod             */
  UPKIE_OBSERVATION_SERVOS = 4            /* [B][6][5], upkie_sim_step_servos           */
};
int upkie_sim_autoreset_done(UpkieSim* sim, int observation, float* state,
                             float* obs, float* final_obs, void* stream);

/* Output buffers must be 16-byte aligned (the rows of 64 consecutive envs are
 * streamed out with 16-byte stores).
 * update_imu != 0 advances the finite-difference accelerometer memory the
 * way one get_spine_observation() call does (pybullet_backend.py:405-408). */
int upkie_sim_observe(UpkieSim* sim, float* state,
                      const UpkieSpineObservation* out, int update_imu,
                      void* stream);

/* Replaces PyBulletBackend.get_contact_points (upkie/envs/backends/
 * pybullet_backend.py:660-716) for the whole batch. The only links that can
 * touch the floor are the two tires, with at most one contact point each:
 *   out [B][2][UPKIE_CONTACT_POINT_WORDS] = per tire (0 "left_wheel_tire",
 *   1 "right_wheel_tire") {exists (0/1), position_contact_in_world (3),
 *   force_in_world (3) = normal + both friction forces in N, 0}.
 * The forces are those of the contact solve of one simulator substep from the
 * current state under the last commanded torques (Bullet reports the last
 * solved substep; the two differ by one 1 ms substep of motion). `state` is
 * only read. On a handle under the Bullet-like contact model
 * (upkie_sim_set_contact_manifold) the query solves THAT model, on a copy of
 * the env's manifold: per tire its (first) cached point and the force the
 * impulses of its points sum to. */
#define UPKIE_CONTACT_POINT_WORDS 8
int upkie_sim_contact_points(UpkieSim* sim, const float* state, float* out, void* stream);

/* ---- MPC balancer (upkie/controllers/mpc_balancer.py:168-312) ---------- */
typedef struct UpkieMpcConfig {
  int32_t num_envs;
  int32_t nb_timesteps;  /* N, 50 by default in the reference               */
  int32_t admm_iterations;
  int32_t reserved0;
  double sampling_period;         /* 0.02 */
  double leg_length;              /* 0.58 */
  double max_ground_accel;        /* 10.0 */
  double max_ground_velocity;     /* 3.0  */
  double fall_pitch;              /* 1.0  */
  double stage_input_cost_weight; /* 1e-3 */
  double stage_state_cost_weight; /* 1e-3 */
  double terminal_cost_weight;    /* 1.0  */
  double admm_rho;
  double admm_relaxation; /* over-relaxation alpha of the ADMM iteration (Boyd et al. 2011, section 3.4.3; OSQP's
                             default is 1.6): x^ = alpha x + (1 - alpha) z replaces x in the z- and y-updates.
                             1.0 (or 0: unset) = the plain iteration; 1.5 by default: same fixed point, and at the
                             reference's N = 50 an order of magnitude closer to it after the same 30 iterations */
} UpkieMpcConfig;

typedef struct UpkieMpc UpkieMpc;

int upkie_mpc_create(const UpkieMpcConfig* config, UpkieMpc** out);
int upkie_mpc_destroy(UpkieMpc* mpc);
const char* upkie_mpc_last_error(const UpkieMpc* mpc);
/* Bytes of the warm-start workspace [2 N][B] the caller allocates. */
int64_t upkie_mpc_workspace_bytes(const UpkieMpc* mpc);
/* MPCBalancer.reset (mpc_balancer.py:228-235) for masked envs. */
int upkie_mpc_reset(UpkieMpc* mpc, float* workspace, float* commanded_velocity,
                    const uint8_t* mask, void* stream);
/* MPCBalancer.step (mpc_balancer.py:237-312) for every env. x0[B][4] =
 * [ground position, pitch, ground velocity, pitch rate]; contact[B] floor
 * contact flags; commanded_velocity[B] in/out; first_input[B] (may be NULL)
 * receives plan.first_input.
 * Arithmetic: fixed-iteration ADMM on the condensed QP, its dense product on the
 * matrix cores: v_mfma_f32_16x16x32_f16 on two fp16 terms per operand with fp32
 * accumulation, the constant part of the product (Minv q) computed once per step
 * from fp64 host products (csrc/mpc.hpp: first input within 5e-4 m/s2 of the fp64
 * oracle's at N = 16 .. 64). A solve whose iterates leave fp16's range (a finite but
 * absurd target or state, e.g. a target velocity of 1e30: |64 rho (2 z - w)| > 65504)
 * is discarded as a whole: zero warm start, first_input 0, the commanded velocity
 * decays as for a fallen robot; nothing non-finite is ever stored (non-finite TARGETS
 * are replaced by 0 up front, see "Non-finite commands and states").
 * The environment variable UPKIE_MPC_FP32=1, read at the
 * first step of the process, selects the fp32 MFMA kernels of earlier rounds for this
 * entry point (A/B). */
int upkie_mpc_step(UpkieMpc* mpc, float* workspace, const float* x0,
                   const float* target_velocity, const uint8_t* contact,
                   double dt, float* commanded_velocity, float* first_input,
                   void* stream);

/* MPCBalancer.step as the first half of a fused UpkieBaseVelocity env.step():
 * the target velocity of env e is act[2 e] (act is the env's [B][2] action),
 * and envs whose `done` word (row UPKIE_S_DONE of the state, may be NULL) is
 * set get MPCBalancer.reset() instead of a solve, because the env step that
 * follows resets them (NEXT_STEP autoreset). */
int upkie_mpc_step_env(UpkieMpc* mpc, float* workspace, const float* x0,
                       const float* act, const uint8_t* contact,
                       const float* done, double dt, float* commanded_velocity,
                       void* stream);

/* ---- Spine observer pipeline ---------------------
 * Device restatement of the observers the real spine runs after every cycle
 * (spines/common/observers.h:22-42): BaseOrientation
 * (upkie/cpp/observers/BaseOrientation.{h,cpp}), FloorContact
 * (FloorContact.cpp:37-104) with its two WheelContact estimators
 * (WheelContact.cpp:19-48) and the velocity-integrating WheelOdometry
 * (WheelOdometry.cpp:16-54), in that order, for every env of a batch. Sim
 * observations then carry what an agent sees on the real robot. */
typedef struct UpkieObserverConfig {
  int32_t num_envs;
  int32_t reserved0;
  double dt; /* spine period, FloorContact::Parameters::dt / WheelOdometry::Parameters::dt */
  double upper_leg_torque_threshold; /* 10.0  (FloorContact.h:82)            */
  double wheel_cutoff_period;        /* 0.2   (spine_backend.py:93); < 1e-6 =
                                        "not configured": wheel observers idle
                                        (WheelContact.cpp:21-24)             */
  double liftoff_inertia;            /* 0.001 */
  double min_touchdown_acceleration; /* 2.0   */
  double min_touchdown_torque;       /* 0.015 */
  double touchdown_inertia;          /* 0.004 */
  double signed_radius[2];           /* left, right: +0.05, -0.05 (spine_backend.py:99-104) */
  double rotation_base_to_imu[9];    /* row-major, diag(-1, 1, -1) (BaseOrientation.h:161-162) */
  double rotation_ars_to_world[9];   /* row-major, diag(1, -1, -1) (BaseOrientation.h:163-164) */
} UpkieObserverConfig;

/* Observer memory, struct-of-arrays [UPKIE_OBSERVER_STATE_WORDS][B] fp32,
 * allocated by the caller. */
enum UpkieObserverStateWord {
  UPKIE_O_WHEEL = 0, /* 2 wheels x (velocity, abs_acceleration, abs_torque, inertia, contact) */
  UPKIE_O_UPPER_LEG_TORQUE = 10,
  UPKIE_O_CONTACT = 11,
  UPKIE_O_ODOMETRY_POSITION = 12,
  UPKIE_O_ODOMETRY_VELOCITY = 13,
  UPKIE_OBSERVER_STATE_WORDS = 16
};

/* What the observers read from the spine observation (device pointers, the
 * layouts of UpkieSpineObservation). imu_orientation == NULL skips
 * BaseOrientation (no "imu" key, BaseOrientation.cpp:17-19); cross_button
 * may be NULL. */
typedef struct UpkieObserverInput {
  const float* servo;                /* [B][6][5]                             */
  const float* imu_orientation;      /* [B][4] w x y z, IMU in ARS            */
  const float* imu_angular_velocity; /* [B][3]                                */
  const uint8_t* cross_button;       /* [B] joystick.cross_button             */
} UpkieObserverInput;

/* What they write (any pointer may be NULL). */
typedef struct UpkieObserverOutput {
  float* base_pitch;             /* [B]    base_orientation.pitch             */
  float* base_angular_velocity;  /* [B][3] base in base                       */
  float* rotation_base_to_world; /* [B][9] row-major                          */
  uint8_t* floor_contact;        /* [B]    floor_contact.contact              */
  float* upper_leg_torque;       /* [B]    floor_contact.upper_leg_torque     */
  float* wheel_contact;          /* [B][2][4] per wheel: abs_acceleration,
                                    abs_torque, contact (0/1), inertia
                                    (FloorContact.cpp:96-103)                 */
  float* wheel_odometry;         /* [B][2] position, velocity                 */
} UpkieObserverOutput;

typedef struct UpkieObservers UpkieObservers;

/* Fails with UPKIE_ERR_INVALID_ARGUMENT where the reference throws:
 * cutoff period <= 2 dt (FilterError, low_pass_filter.h:22-30) for the wheel
 * filters or for the 0.01 s upper-leg torque filter (FloorContact.cpp:87). */
int upkie_observers_create(const UpkieObserverConfig* config, UpkieObservers** out);
int upkie_observers_destroy(UpkieObservers* observers);
const char* upkie_observers_last_error(const UpkieObservers* observers);
int64_t upkie_observers_state_bytes(const UpkieObservers* observers);
/* Observer::reset for masked envs (all when mask is NULL): filters, contacts
 * and odometry back to zero (FloorContact.cpp:26-35, WheelOdometry.cpp:10-14). */
int upkie_observers_reset(UpkieObservers* observers, float* state, const uint8_t* mask, void* stream);
/* Run FloorContact / WheelContact / WheelOdometry INSIDE every env step, one
 * observer cycle per 1 ms physics substep: what the spine does under a slower
 * agent (Spine::simulate, Spine.cpp:119-141: nb_substeps cycles per action).
 * The spine period is the substep dt / nb_substeps (config->dt is ignored);
 * observer_state is the caller's [UPKIE_OBSERVER_STATE_WORDS][B] buffer, read
 * and written by the step kernels and cleared for envs that (auto)reset; its
 * words are the observers' outputs. NULL config or state detaches. */
int upkie_sim_attach_observers(UpkieSim* sim, const UpkieObserverConfig* config, float* observer_state);

/* One ObserverPipeline::run (read + write of the three observers). in->servo
 * == NULL: no "servo" block, only BaseOrientation runs (FloorContact.cpp:42-44). */
int upkie_observers_step(UpkieObservers* observers, float* state, const UpkieObserverInput* in,
                         const UpkieObserverOutput* out, void* stream);

/* A linear policy in one launch, for the loop `obs, ... = env.step(policy(obs))`
 * that the reference leaves to the agent (README.md:60-67: action = clamp(gains
 * . obs)): act[n][a] = clamp(sum_d obs[n][d] weights[d][a] + bias[a], -clip,
 * clip) for n < num_envs; bias may be NULL, clip <= 0 disables the clamp. obs
 * [num_envs][obs_dim], weights [obs_dim][act_dim], act [num_envs][act_dim],
 * contiguous fp32 device buffers. (As torch ops the same policy is two or
 * three launches of 2-5 us each behind a 14.5 us step.) */
int upkie_linear_policy(int32_t num_envs, int32_t obs_dim, int32_t act_dim, const float* obs,
                        const float* weights, const float* bias, double clip, float* act,
                        void* stream);

/* ---- Rollout consumer (BASELINE.json configs[3]) ---
 * Generalized advantage estimation over a rollout resident in HBM: rewards,
 * values, episode_starts, advantages, returns are [num_steps][num_envs]
 * (episode_starts[t][n] != 0 when step t is the first of an episode),
 * last_values / last_dones [num_envs] describe the state after the last step.
 * The reference has no learner (README.md:107-109 points to external RL
 * playgrounds); the recurrence is the published one (Schulman et al. 2016).
 * Errors are reported through upkie_sim_last_error(NULL). */
int upkie_rollout_gae(int32_t num_steps, int32_t num_envs, const float* rewards, const float* values,
                      const uint8_t* episode_starts, const float* last_values, const uint8_t* last_dones,
                      double gamma, double gae_lambda, float* advantages, float* returns, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UPKIE_HIP_H_ */

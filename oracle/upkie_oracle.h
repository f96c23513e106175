/*
 * upkie_oracle.h -- CPU fp64 restatement of the reference's env.step() path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under upkie_amd/ may import, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * do, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: the wrapper arithmetic (PD servo law, Gyropod/Pendulum/
 * Servos maps, observation maps, init-state sampling, MPC cost maps) follows
 * the reference line by line and is pinned by the reference's own known-answer
 * tests (restated in tests/). The rigid-body dynamics + contact solve live in
 * third-party Bullet 3.25 and the robot model in upkie_description 2.2.0,
 * neither present in /root/reference nor installable here: that part restates
 * the published algorithm (Featherstone dynamics, velocity-level PGS contact
 * with ERP/CFM, semi-implicit Euler) and is pinned only by the reference's
 * physics invariants. Step-for-step parity with Bullet is UNPINNED.
 *
 * State layout: the same [UPKIE_STATE_WORDS][B] struct-of-arrays as the HIP
 * library (include/upkie_hip.h) but in double precision.
 */
#ifndef UPKIE_ORACLE_H_
#define UPKIE_ORACLE_H_

#include <stdint.h>

#include "../include/upkie_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Servo command of one joint, upkie_servos.py:98-105 (ACTION_KEYS order). */
typedef struct OracleServoCommand {
  double position; /* may be NaN: no position feedback */
  double velocity;
  double feedforward_torque;
  double kp_scale;
  double kd_scale;
  double maximum_torque;
} OracleServoCommand;

/* Optional per-env randomisation inputs (any pointer may be NULL). */
typedef struct OracleRandomization {
  const double* body_inertials; /* [UPKIE_NB * UPKIE_INERTIAL_WORDS][B], oracle_sample_body_inertials */
  const double* ext_force;     /* [count][3][B] (count = 1 without ext_slots) */
  double ext_point[3];         /* without ext_slots: application point on the trunk, base frame */
  const UpkieExternalForces* ext_slots; /* bodies / points / frames of the forces, or NULL */
  /* spine observers run inside the step, one cycle per substep (or NULL) */
  const UpkieObserverConfig* observer_config;
  double* observer_state; /* [UPKIE_OBSERVER_STATE_WORDS][B] */
  /* non-NULL: contacts follow the Bullet-like specification of upkie_oracle.c
   * (persistent 4-point manifolds, 50 fixed warm-started sweeps, cone friction,
   * no friction CFM) instead of the product's; this is its memory, zeroed by
   * the caller before the first reset: [ORACLE_BULLET_MANIFOLD_WORDS][B] */
  double* bullet_manifold;
} OracleRandomization;
#define ORACLE_BULLET_MANIFOLD_WORDS 64 /* 2 tires x 4 points x 8 words */

/* Philox4x32-10 counter-based generator (Salmon et al., SC'11). */
void oracle_philox4x32_10(const uint32_t counter[4], const uint32_t key[2],
                          uint32_t out[4]);

/* moteus-like servo torque law, pybullet_backend.py:492-553. */
double oracle_joint_torque(double q, double qd, const OracleServoCommand* cmd,
                           double kp, double kd, double friction);

/* Small helpers exposed so that tests can pin them on golden vectors generated
 * from the reference's own modules (tools/make_golden.py). */
void oracle_quat_to_matrix(const double quat_wxyz[4], double R[9]);
void oracle_matrix_to_quat(const double R[9], double quat_wxyz[4]);
void oracle_euler_zyx_compose(const double base_wxyz[4], const double ypr[3],
                              double out_wxyz[4]);
double oracle_clamp(double value, double lower, double upper);
double oracle_low_pass_filter(double prev_output, double cutoff_period,
                              double new_input, double dt);
double oracle_pitch_from_quat(const double quat_wxyz[4]);

/* Mass matrix (12x12, row-major; generalised velocity = [v_base(world),
 * omega_base(world), qd(6)]) and bias vector at one configuration. */
void oracle_mass_matrix_and_bias(const UpkieModel* model, const double pos[3],
                                 const double quat[4], const double linvel[3],
                                 const double angvel[3], const double q[6],
                                 const double qd[6], double M[144],
                                 double h[12]);

/* Total mass and centre of mass in the base frame at configuration q. */
double oracle_total_mass(const UpkieModel* model);
void oracle_center_of_mass(const UpkieModel* model, const double q[6],
                           double com_in_base[3]);
/* Kinetic + potential energy at one state (for conservation tests). */
double oracle_energy(const UpkieModel* model, const double pos[3],
                     const double quat[4], const double linvel[3],
                     const double angvel[3], const double q[6],
                     const double qd[6]);

/* One physics substep of duration h for one env given as an array-of-words
 * state[UPKIE_STATE_WORDS]; tau = commanded joint torques. Returns the floor
 * contact flag. */
int oracle_substep(const UpkieModel* model, double* state, const double tau[6],
                   double h, const double* body_inertials /* [70] of this env or NULL */,
                   const double* ext_force, const double* ext_point);
/* Same with forces on any link: ext_forces[count][3], pybullet_backend.py:603-658. */
int oracle_substep_bullet_like(const UpkieModel* model, double* state, const double tau[6], double h, double* manifold);
int oracle_substep_ext(const UpkieModel* model, double* state, const double tau[6],
                       double h, const double* body_inertials /* [70] of this env or NULL */,
                       const double* ext_forces, const UpkieExternalForces* ext_slots);

/* Batched entry points mirroring the HIP C-ABI (state is [WORDS][B]). */
void oracle_sample_body_inertials(const UpkieModel* model, const UpkieSimConfig* cfg,
                                  double inertia_variation, double* records /* [70][B] */,
                                  double* link_scale /* [UPKIE_MAX_LINKS][B] or NULL */);
void oracle_sample_pushes(const UpkieSimConfig* cfg, uint32_t push_index, double max_norm,
                          double* force /* [3][B] */);
void oracle_reset(const UpkieModel* model, const UpkieSimConfig* cfg,
                  double* state, const uint8_t* mask,
                  const OracleRandomization* rnd, double* obs6);
void oracle_step_servos(const UpkieModel* model, const UpkieSimConfig* cfg,
                        double* state, const double* act, double* obs,
                        double* reward, uint8_t* terminated,
                        uint8_t* truncated, const OracleRandomization* rnd);
void oracle_step_gyropod(const UpkieModel* model, const UpkieSimConfig* cfg,
                         double* state, const double* act, double* obs,
                         double* reward, uint8_t* terminated,
                         uint8_t* truncated, const OracleRandomization* rnd);
void oracle_step_pendulum(const UpkieModel* model, const UpkieSimConfig* cfg,
                          double* state, const double* act, double* obs,
                          double* reward, uint8_t* terminated,
                          uint8_t* truncated, const OracleRandomization* rnd);
void oracle_step_pendulum_agent(const UpkieModel* model,
                                const UpkieSimConfig* cfg, double* state,
                                double* obs, double* reward,
                                uint8_t* terminated, uint8_t* truncated,
                                const OracleRandomization* rnd);
/* `steps` env.step() of every env in one parallel region (bench.py's CPU baseline) */
int64_t oracle_rollout_pendulum_agent(const UpkieModel* model, const UpkieSimConfig* cfg, double* state,
                                      double* obs, int32_t steps, const OracleRandomization* rnd);

/* PyBulletBackend.get_contact_points (pybullet_backend.py:660-716) of every
 * env: out [B][2][8] = per tire {exists, position in world (3), force in
 * world (3), 0}. Same contract as upkie_sim_contact_points. */
void oracle_contact_points(const UpkieModel* model, const UpkieSimConfig* cfg,
                           const double* state, const OracleRandomization* rnd, double* out);

typedef struct OracleSpineObservation {
  double* pitch;
  double* angular_velocity;
  double* linear_velocity;
  double* rotation_base_to_world;
  uint8_t* floor_contact;
  double* imu_orientation;
  double* imu_angular_velocity;
  double* imu_linear_acceleration;
  double* imu_raw_linear_acceleration;
  double* servo;
  double* wheel_odometry;
} OracleSpineObservation;

void oracle_observe(const UpkieModel* model, const UpkieSimConfig* cfg,
                    double* state, const OracleSpineObservation* out,
                    int update_imu);

/* ---- MPC (mpc_balancer.py + qpmpc WheeledInvertedPendulum, restated) ---- */
/* Condensed QP data: P[N][N], Kx[N][4], kv[N] with q = Kx x0 + kv v_target. */
void oracle_mpc_build(const UpkieMpcConfig* cfg, double* P, double* Kx,
                      double* kv);
/* Cost vector built the long way (Phi/Psi stacks and target states exactly as
 * get_target_states + MPCQP.update_cost_vector do), to check Kx/kv. */
void oracle_mpc_cost_vector(const UpkieMpcConfig* cfg, const double x0[4],
                            double v_target, double* q);
/* Exact box-QP solution by projected Newton / active set to 1e-12. */
int oracle_mpc_solve_exact(int n, const double* P, const double* q,
                           double bound, double* u);
/* Fixed-iteration ADMM with the same recurrences as the HIP kernel; z, y are
 * the warm-start (in/out). Minv = (P + rho I)^-1. */
void oracle_mpc_admm(int n, const double* Minv, const double* q, double rho,
                     double bound, int iterations, double* z, double* y,
                     double* u);
void oracle_mpc_minv(int n, const double* P, double rho, double* Minv);
/* Batched MPCBalancer.step, workspace [2N][B]. */
void oracle_mpc_step(const UpkieMpcConfig* cfg, double* workspace,
                     const double* x0, const double* v_target,
                     const uint8_t* contact, double dt, double* commanded,
                     double* first_input);


/* ---- command / observation maps for the golden tests ---- */
void oracle_servo_commands(const UpkieModel* model, const UpkieSimConfig* cfg,
                           const double act[36], OracleServoCommand cmd[6]);
void oracle_gyropod_commands(const UpkieModel* model, const UpkieSimConfig* cfg,
                             double* state, double ground_velocity, double yaw_velocity,
                             OracleServoCommand cmd[6]);
void oracle_gyropod_observation(const UpkieModel* model, const double* state, double obs6[6]);

/* ---- observer pipeline (upkie_oracle_observers.c) ---- */
void oracle_observers_cycle_env(const UpkieObserverConfig* p, double dt, double* st,
                                const double velocity[6], const double torque[6]);
int oracle_observers_check(const UpkieObserverConfig* c);
double oracle_pitch_frame_in_parent(const double R[9]);
void oracle_base_orientation_from_imu(const double q[4], const double base_to_imu[9],
                                      const double ars_to_world[9], double R[9]);
void oracle_observers_step(const UpkieObserverConfig* p, double* state, const double* servo,
                           const double* imu_orientation, const double* imu_angular_velocity,
                           const uint8_t* cross_button, double* base_pitch,
                           double* base_angular_velocity, double* rotation_base_to_world,
                           uint8_t* floor_contact, double* upper_leg_torque,
                           double* wheel_contact_out, double* wheel_odometry);

#ifdef __cplusplus
}
#endif
#endif

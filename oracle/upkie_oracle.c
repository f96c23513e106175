/*
 * upkie_oracle.c -- CPU fp64 restatement of the reference's env.step() path.
 * TEST INFRASTRUCTURE ONLY (see upkie_oracle.h for the parity status).
 *
 * Formulation (deliberately NOT the one the HIP kernels use): every body's
 * pose, velocity and Jacobians are written in the WORLD frame, the joint-space
 * mass matrix is assembled as sum_i m_i Jv_i^T Jv_i + Jw_i^T I_i Jw_i, the bias
 * vector by projecting Newton-Euler velocity-product forces, and the 12x12
 * system is solved by dense Cholesky. Reference citations are file:line under
 * /root/reference.
 */
#include "upkie_oracle.h"

#include <math.h>
#include <string.h>

#define NB UPKIE_NB
#define NJ UPKIE_NJ
#define NV 12
#define NW UPKIE_STATE_WORDS
#define MAXROWS (6 + NJ) /* two tires x three rows + a limit row per joint (a model may bound its wheel joints too) */

long oracle_debug_sweeps = 0, oracle_debug_fallbacks = 0, oracle_debug_substeps = 0;
long oracle_debug_sweep_hist[64] = {0}; /* fallbacks by number of sweeps they needed */
/* the first systems whose sweeps ran into the iteration cap (diagnostics of the solver, tools/pgs_cap_cases.py):
 * per case nrows, then W + CFM row-major [6][6], rhs [6], the warm start [6], the result [6] */
#define ORACLE_CAPTURE_CASES 4096
double oracle_debug_capture[ORACLE_CAPTURE_CASES][1 + 36 + 18] = {{0}};
/* Non-finite guard: [0] command words replaced by the neutral action's, [1] env states replaced by the initial state
 * (the twin of upkie_sim_guard_counts) */
int64_t oracle_guard_counts[2] = {0, 0};
long oracle_debug_captured = 0;
long oracle_debug_capture_threshold = 0; /* capture systems that needed at least this many sweeps (0: the iteration cap) */
long oracle_debug_capture_rows = 0; /* capture systems of this many rows only (0: any) */

/* ------------------------------------------------------------------ vec3 */
static void v3_cross(const double a[3], const double b[3], double c[3]) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static double v3_dot(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static void m3_mulv(const double R[9], const double v[3], double out[3]) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  out[0] = x; out[1] = y; out[2] = z;
}
static void m3_tmulv(const double R[9], const double v[3], double out[3]) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  out[0] = x; out[1] = y; out[2] = z;
}
static void m3_mul(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] +
                     A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
static void m3_transpose(const double A[9], double T[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
}

/* upkie/utils/rotations.py:52-71, quat = [w, x, y, z] */
static void quat_to_matrix(const double q[4], double R[9]) {
  double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  R[0] = 1 - 2 * (qy * qy + qz * qz);
  R[1] = 2 * (qx * qy - qz * qw);
  R[2] = 2 * (qw * qy + qx * qz);
  R[3] = 2 * (qx * qy + qz * qw);
  R[4] = 1 - 2 * (qx * qx + qz * qz);
  R[5] = 2 * (qy * qz - qx * qw);
  R[6] = 2 * (qx * qz - qy * qw);
  R[7] = 2 * (qy * qz + qx * qw);
  R[8] = 1 - 2 * (qx * qx + qy * qy);
}
static void quat_mul(const double a[4], const double b[4], double c[4]) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  c[0] = w; c[1] = x; c[2] = y; c[3] = z;
}
/* Rodrigues rotation about a unit axis. */
static void axis_angle_matrix(const double a[3], double angle, double R[9]) {
  double c = cos(angle), s = sin(angle), t = 1.0 - c;
  R[0] = c + t * a[0] * a[0];
  R[1] = t * a[0] * a[1] - s * a[2];
  R[2] = t * a[0] * a[2] + s * a[1];
  R[3] = t * a[0] * a[1] + s * a[2];
  R[4] = c + t * a[1] * a[1];
  R[5] = t * a[1] * a[2] - s * a[0];
  R[6] = t * a[0] * a[2] - s * a[1];
  R[7] = t * a[1] * a[2] + s * a[0];
  R[8] = c + t * a[2] * a[2];
}

/* scipy.spatial.transform.Rotation.from_matrix -> as_quat, as called by
 * upkie/utils/rotations.py:16-33 (output reordered to w, x, y, z). */
static void matrix_to_quat_scipy(const double m[9], double q_wxyz[4]) {
  double decision[4] = {m[0], m[4], m[8], m[0] + m[4] + m[8]};
  int choice = 0;
  for (int i = 1; i < 4; ++i)
    if (decision[i] > decision[choice]) choice = i;
  double q[4]; /* x y z w */
  if (choice != 3) {
    int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1 - decision[3] + 2 * m[3 * i + i];
    q[j] = m[3 * j + i] + m[3 * i + j];
    q[k] = m[3 * k + i] + m[3 * i + k];
    q[3] = m[3 * k + j] - m[3 * j + k];
  } else {
    q[0] = m[7] - m[5];
    q[1] = m[2] - m[6];
    q[2] = m[3] - m[1];
    q[3] = 1 + decision[3];
  }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q_wxyz[0] = q[3] / n;
  q_wxyz[1] = q[0] / n;
  q_wxyz[2] = q[1] / n;
  q_wxyz[3] = q[2] / n;
}

/* --------------------------------------------------------------- Philox */
void oracle_philox4x32_10(const uint32_t counter[4], const uint32_t key[2],
                          uint32_t out[4]) {
  uint32_t c0 = counter[0], c1 = counter[1], c2 = counter[2], c3 = counter[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { STREAM_RESET = 0, STREAM_NOISE = 1, STREAM_INERTIA = 2, STREAM_PUSH = 3 };

/* 24-bit uniforms in [0, 1): identical values in fp32 and fp64. */
static void philox_uniform4(uint64_t seed, int64_t env, uint32_t episode,
                            uint32_t stream, uint32_t block, double u[4]) {
  uint32_t ctr[4] = {(uint32_t)((uint64_t)env & 0xffffffffu),
                     (uint32_t)((uint64_t)env >> 32), episode,
                     (stream << 24) | block};
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  uint32_t r[4];
  oracle_philox4x32_10(ctr, key, r);
  for (int i = 0; i < 4; ++i) u[i] = (double)(r[i] >> 8) * (1.0 / 16777216.0);
}

/* ------------------------------------------------------------ servo law */
/* pybullet_backend.py:492-553 (noise handled by the caller). */
static double joint_torque_with_noise(double q, double qd, const OracleServoCommand* cmd,
                                      double kp_gain, double kd_gain, double friction, double noise) {
  double kp = cmd->kp_scale * kp_gain;                 /* :526 */
  double kd = cmd->kd_scale * kd_gain;                 /* :527 */
  double torque = cmd->feedforward_torque;             /* :530 */
  torque += kd * (cmd->velocity - qd);                 /* :531 */
  if (!isnan(cmd->position)) torque += kp * (cmd->position - q); /* :532 */
  if (fabs(qd) > 1e-3) {                               /* :536-541 */
    double sign = qd > 0.0 ? 1.0 : -1.0;
    torque += -friction * sign;
  }
  torque += noise; /* :545-550, drawn by the caller when sigma > 1e-10 */
  /* np.clip, :552 */
  if (torque < -cmd->maximum_torque) torque = -cmd->maximum_torque;
  if (torque > cmd->maximum_torque) torque = cmd->maximum_torque;
  return torque;
}

double oracle_joint_torque(double q, double qd, const OracleServoCommand* cmd,
                           double kp_gain, double kd_gain, double friction) {
  return joint_torque_with_noise(q, qd, cmd, kp_gain, kd_gain, friction, 0.0);
}

/* Six standard normals for (env, step, slot): Box-Muller on two Philox blocks.
 * slot = substep index for control noise, NOISE_SLOT_MEASUREMENT for the
 * measurement noise of the observation that follows step `step`. */
#define NOISE_SLOT_MEASUREMENT 0x7fffu
static void philox_normal6(uint64_t seed, int64_t env, uint32_t step, uint32_t slot, double z[6]) {
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  uint32_t r[8];
  for (uint32_t k = 0; k < 2; ++k) {
    uint32_t ctr[4] = {(uint32_t)((uint64_t)env & 0xffffffffu), (uint32_t)((uint64_t)env >> 32), step,
                       ((uint32_t)STREAM_NOISE << 24) | (slot * 2u + k)};
    oracle_philox4x32_10(ctr, key, r + 4 * k);
  }
  for (int p = 0; p < 3; ++p) {
    double u1 = ((double)(r[2 * p] >> 8) + 1.0) * (1.0 / 16777216.0); /* (0, 1] */
    double u2 = (double)(r[2 * p + 1] >> 8) * (1.0 / 16777216.0);
    double radius = sqrt(-2.0 * log(u1));
    z[2 * p] = radius * cos(6.283185307179586 * u2);
    z[2 * p + 1] = radius * sin(6.283185307179586 * u2);
  }
}

/* ------------------------------------------------------------ kinematics */
typedef struct Kin {
  double R[NB][9];  /* body -> world */
  double o[NB][3];  /* body frame origin (joint origin) in world */
  double c[NB][3];  /* centre of mass in world */
  double a[NJ][3];  /* joint axes in world */
  double Iw[NB][9]; /* inertia about com, world axes */
  double m[NB];
} Kin;

static int parent_of(int body) { return (body == 1 || body == 4) ? 0 : body - 1; }
/* is joint j (moving body j+1) on the path from the base to body i? */
static int on_path(int j, int i) {
  if (i == 0) return 0;
  int leg_i = (i - 1) / 3, leg_j = j / 3;
  return leg_i == leg_j && (j % 3) <= ((i - 1) % 3);
}

/* inertials: per-env records of the bodies (mass, com, inertia about it:
 * [10 * body + word], what randomize_inertias leaves behind) or NULL */
static void kinematics(const UpkieModel* model, const double* inertials,
                       const double pos[3], const double quat[4],
                       const double q[NJ], Kin* k) {
  quat_to_matrix(quat, k->R[0]);
  memcpy(k->o[0], pos, 3 * sizeof(double));
  for (int i = 1; i < NB; ++i) {
    int p = parent_of(i), j = i - 1;
    double Rj[9], r[3];
    axis_angle_matrix(model->joint_axis[j], q[j], Rj);
    m3_mul(k->R[p], Rj, k->R[i]);
    m3_mulv(k->R[p], model->joint_pos[j], r);
    for (int d = 0; d < 3; ++d) k->o[i][d] = k->o[p][d] + r[d];
    m3_mulv(k->R[i], model->joint_axis[j], k->a[j]);
  }
  for (int i = 0; i < NB; ++i) {
    const double* rec = inertials ? inertials + UPKIE_INERTIAL_WORDS * i : NULL;
    double r[3];
    m3_mulv(k->R[i], rec ? rec + 1 : model->com[i], r);
    for (int d = 0; d < 3; ++d) k->c[i][d] = k->o[i][d] + r[d];
    const double* I6 = rec ? rec + 4 : model->inertia[i];
    double Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5],
                    I6[4], I6[5], I6[2]};
    double Rt[9], T[9];
    m3_transpose(k->R[i], Rt);
    m3_mul(k->R[i], Ib, T);
    m3_mul(T, Rt, k->Iw[i]);
    k->m[i] = rec ? rec[0] : model->mass[i];
  }
}

/* Jacobian columns of a point P rigidly attached to body i:
 * Jv[:,k] linear, Jw[:,k] angular, for generalised velocity
 * [v_base(3), omega_base(3), qd(6)]. */
static void point_jacobian(const Kin* k, int i, const double P[3],
                           double Jv[3][NV], double Jw[3][NV]) {
  memset(Jv, 0, sizeof(double) * 3 * NV);
  memset(Jw, 0, sizeof(double) * 3 * NV);
  double r[3] = {P[0] - k->o[0][0], P[1] - k->o[0][1], P[2] - k->o[0][2]};
  for (int d = 0; d < 3; ++d) {
    Jv[d][d] = 1.0;
    Jw[d][3 + d] = 1.0;
    double e[3] = {0, 0, 0}, x[3];
    e[d] = 1.0;
    v3_cross(e, r, x);
    for (int r_ = 0; r_ < 3; ++r_) Jv[r_][3 + d] = x[r_];
  }
  for (int j = 0; j < NJ; ++j) {
    if (!on_path(j, i)) continue;
    double rj[3] = {P[0] - k->o[j + 1][0], P[1] - k->o[j + 1][1],
                    P[2] - k->o[j + 1][2]};
    double x[3];
    v3_cross(k->a[j], rj, x);
    for (int d = 0; d < 3; ++d) {
      Jv[d][6 + j] = x[d];
      Jw[d][6 + j] = k->a[j][d];
    }
  }
}

static void mass_matrix_and_bias(const Kin* k, double gravity,
                                 const double linvel[3],
                                 const double angvel[3], const double qd[NJ],
                                 double M[NV * NV], double h[NV]) {
  (void)linvel;
  memset(M, 0, sizeof(double) * NV * NV);
  memset(h, 0, sizeof(double) * NV);
  /* velocity-product accelerations, Newton-Euler outward pass */
  double w[NB][3], al[NB][3], ao[NB][3];
  for (int d = 0; d < 3; ++d) {
    w[0][d] = angvel[d];
    al[0][d] = 0.0;
    ao[0][d] = 0.0;
  }
  for (int i = 1; i < NB; ++i) {
    int p = parent_of(i), j = i - 1;
    double t[3], r[3], t2[3];
    for (int d = 0; d < 3; ++d) w[i][d] = w[p][d] + k->a[j][d] * qd[j];
    v3_cross(w[p], k->a[j], t);
    for (int d = 0; d < 3; ++d) al[i][d] = al[p][d] + t[d] * qd[j];
    for (int d = 0; d < 3; ++d) r[d] = k->o[i][d] - k->o[p][d];
    v3_cross(al[p], r, t);
    v3_cross(w[p], r, t2);
    v3_cross(w[p], t2, t2);
    for (int d = 0; d < 3; ++d) ao[i][d] = ao[p][d] + t[d] + t2[d];
  }
  for (int i = 0; i < NB; ++i) {
    double Jv[3][NV], Jw[3][NV];
    point_jacobian(k, i, k->c[i], Jv, Jw);
    /* M += m Jv^T Jv + Jw^T Iw Jw */
    for (int a = 0; a < NV; ++a) {
      double IJw[3];
      double col[3] = {Jw[0][a], Jw[1][a], Jw[2][a]};
      m3_mulv(k->Iw[i], col, IJw);
      for (int b = 0; b < NV; ++b) {
        double s = 0.0;
        for (int d = 0; d < 3; ++d)
          s += k->m[i] * Jv[d][a] * Jv[d][b] + IJw[d] * Jw[d][b];
        M[a * NV + b] += s;
      }
    }
    /* bias force of body i */
    double r[3], t[3], t2[3], ac[3], f[3], n[3], Iw_w[3], Iw_al[3];
    for (int d = 0; d < 3; ++d) r[d] = k->c[i][d] - k->o[i][d];
    v3_cross(al[i], r, t);
    v3_cross(w[i], r, t2);
    v3_cross(w[i], t2, t2);
    for (int d = 0; d < 3; ++d) ac[d] = ao[i][d] + t[d] + t2[d];
    ac[2] += gravity; /* a_c - g with g = (0, 0, -gravity) */
    for (int d = 0; d < 3; ++d) f[d] = k->m[i] * ac[d];
    m3_mulv(k->Iw[i], w[i], Iw_w);
    m3_mulv(k->Iw[i], al[i], Iw_al);
    v3_cross(w[i], Iw_w, t);
    for (int d = 0; d < 3; ++d) n[d] = Iw_al[d] + t[d];
    for (int a = 0; a < NV; ++a)
      for (int d = 0; d < 3; ++d) h[a] += Jv[d][a] * f[d] + Jw[d][a] * n[d];
  }
}

/* dense Cholesky M = L L^T (lower, in place in L), returns 0 on success */
static int cholesky(int n, const double* A, double* L) {
  memset(L, 0, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      if (i == j) {
        if (s <= 0.0) return -1;
        L[i * n + i] = sqrt(s);
      } else {
        L[i * n + j] = s / L[j * n + j];
      }
    }
  return 0;
}
static void cholesky_solve(int n, const double* L, const double* b, double* x) {
  double y[64];
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
    y[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

void oracle_mass_matrix_and_bias(const UpkieModel* model, const double pos[3],
                                 const double quat[4], const double linvel[3],
                                 const double angvel[3], const double q[6],
                                 const double qd[6], double M[144],
                                 double h[12]) {
  Kin k;
  kinematics(model, NULL, pos, quat, q, &k);
  mass_matrix_and_bias(&k, model->gravity, linvel, angvel, qd, M, h);
}

double oracle_total_mass(const UpkieModel* model) {
  double m = 0.0;
  for (int i = 0; i < NB; ++i) m += model->mass[i];
  return m;
}

void oracle_center_of_mass(const UpkieModel* model, const double q[6],
                           double com_in_base[3]) {
  const double pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
  Kin k;
  kinematics(model, NULL, pos, quat, q, &k);
  double m = 0.0, s[3] = {0, 0, 0};
  for (int i = 0; i < NB; ++i) {
    m += k.m[i];
    for (int d = 0; d < 3; ++d) s[d] += k.m[i] * k.c[i][d];
  }
  for (int d = 0; d < 3; ++d) com_in_base[d] = s[d] / m;
}

double oracle_energy(const UpkieModel* model, const double pos[3],
                     const double quat[4], const double linvel[3],
                     const double angvel[3], const double q[6],
                     const double qd[6]) {
  Kin k;
  kinematics(model, NULL, pos, quat, q, &k);
  double M[NV * NV], h[NV], nu[NV];
  mass_matrix_and_bias(&k, model->gravity, linvel, angvel, qd, M, h);
  for (int d = 0; d < 3; ++d) {
    nu[d] = linvel[d];
    nu[3 + d] = angvel[d];
  }
  for (int j = 0; j < NJ; ++j) nu[6 + j] = qd[j];
  double T = 0.0, V = 0.0;
  for (int a = 0; a < NV; ++a)
    for (int b = 0; b < NV; ++b) T += 0.5 * nu[a] * M[a * NV + b] * nu[b];
  for (int i = 0; i < NB; ++i) V += k.m[i] * model->gravity * k.c[i][2];
  return T + V;
}


/* The two LATERAL friction rows (third row of each tire): the tires share one
 * axle direction, so their 2 x 2 block [[a22, a25], [a25, a55]] is nearly
 * singular (only friction_cfm and the yaw lever arm of the wheel base separate
 * the rows); swept one at a time they converge like (a / (a + cfm))^2 per
 * sweep, hundreds of sweeps. They are therefore solved TOGETHER and exactly,
 * given every other row: the minimiser of 1/2 x'Ax - r'x over the box
 * |x_i| <= mu lam_n,i -- the free 2 x 2 solution when it lies inside the box,
 * otherwise the best point of the four edges (one row on a bound, the other
 * solved and clamped). Both rows count in the convergence test.
 * (Until round 3 the pair was solved as a FREE pair in sum / difference
 * coordinates and clamped afterwards, the weak coordinate left out of the
 * convergence test: with a row on its bound -- a tire sliding sideways, or
 * lifted -- the result violated the complementarity conditions of the
 * problem, by up to O(1) of the velocity scale on robots tumbling under
 * examples/pybullet/torque_balancing.py's law:
 * tests/test_oracle_contact_kkt.py::test_torque_law_systems.) */
static int lateral_pair_exists(int nrows, const int* kind, const int* normal_row) {
  int n = 0;
  for (int r = 0; r < nrows; ++r)
    if (kind[r] == 1 && r == normal_row[r] + 2) ++n;
  return n == 2;
}
static void lateral_pair_sweep(int nrows, const int* kind, const int* normal_row, const double* W, int ldw,
                               const double* cfm, const double* rhs, double mu, double* lam, double* change, double* scale) {
  int lat[2], n = 0;
  for (int r = 0; r < nrows; ++r)
    if (kind[r] == 1 && r == normal_row[r] + 2 && n < 2) lat[n++] = r;
  if (n != 2) return; /* a single lateral row is swept like any friction row */
  const int i2 = lat[0], i5 = lat[1];
  double r2 = rhs[i2], r5 = rhs[i5];
  for (int b = 0; b < nrows; ++b) {
    if (b == i2 || b == i5) continue;
    r2 -= W[i2 * ldw + b] * lam[b];
    r5 -= W[i5 * ldw + b] * lam[b];
  }
  const double a22 = W[i2 * ldw + i2] + cfm[i2], a55 = W[i5 * ldw + i5] + cfm[i5], a25 = W[i5 * ldw + i2];
  const double lim2 = mu * lam[normal_row[i2]], lim5 = mu * lam[normal_row[i5]];
  const double det = a22 * a55 - a25 * a25;
  double x2 = 0.0, x5 = 0.0;
  int inside = 0;
  if (det > 1e-12 * a22 * a55) {
    x2 = (a55 * r2 - a25 * r5) / det;
    x5 = (a22 * r5 - a25 * r2) / det;
    inside = fabs(x2) <= lim2 && fabs(x5) <= lim5;
  }
  if (!inside) {
    double best = 1e300;
    for (int e = 0; e < 4; ++e) {
      double c2, c5;
      if (e < 2) {
        c2 = e == 0 ? -lim2 : lim2;
        c5 = (r5 - a25 * c2) / a55;
        if (c5 < -lim5) c5 = -lim5;
        if (c5 > lim5) c5 = lim5;
      } else {
        c5 = e == 2 ? -lim5 : lim5;
        c2 = (r2 - a25 * c5) / a22;
        if (c2 < -lim2) c2 = -lim2;
        if (c2 > lim2) c2 = lim2;
      }
      const double value = 0.5 * (a22 * c2 * c2 + a55 * c5 * c5) + a25 * c2 * c5 - r2 * c2 - r5 * c5;
      if (value < best) {
        best = value;
        x2 = c2;
        x5 = c5;
      }
    }
  }
  if (fabs(x2 - lam[i2]) > *change) *change = fabs(x2 - lam[i2]);
  if (fabs(x5 - lam[i5]) > *change) *change = fabs(x5 - lam[i5]);
  if (fabs(x2) > *scale) *scale = fabs(x2);
  if (fabs(x5) > *scale) *scale = fabs(x5);
  lam[i2] = x2;
  lam[i5] = x5;
}

/* ------------------------------------------------- Bullet-like contact spec */
/* A SECOND contact specification, for measuring how far the product's (one
 * point per tire, exact solve + sweeps to convergence, box friction in the
 * wheel's rolling / lateral directions, friction CFM 0.01) is from what Bullet
 * 3.25's btMultiBodyConstraintSolver does inside pybullet.stepSimulation()
 * (pybullet_backend.py:306) -- restated from the published sources as
 * SURVEY.md Appendix B.1 / B.2 summarise them, [3P, unverified: Bullet is not
 * available here]:
 *   - a PERSISTENT manifold of up to 4 points per tire (btPersistentManifold):
 *     every step the cached points are refreshed (removed beyond the contact
 *     breaking threshold, along the normal or in the plane) and the deepest
 *     point of the tire (btConvexPlaneCollisionAlgorithm's single support
 *     point) replaces the cached point within the threshold of it in the
 *     wheel's frame, or is added;
 *   - one normal + two friction rows per cached point; friction directions
 *     along / across the sliding velocity of the point (btPlaneSpace1 when it
 *     does not slide); zero friction CFM; normal rows with the URDF contact
 *     stiffness / damping as CFM / ERP, a separated point may close its gap;
 *   - sequential impulses: a FIXED number of sweeps (numSolverIterations = 50),
 *     joint limits first, then all normal rows, then the friction rows of each
 *     point projected onto the cone |f| <= mu f_n (implicit cone friction),
 *     normal impulses warm-started with 0.85 x last step's;
 * everything else (articulated-body dynamics, torque law, integration, joint
 * speed clamp, base damping) is shared with the product's specification.
 * Enabled per call through OracleRandomization.bullet_manifold; used by
 * tools/bullet_like_deviation.py and tests/test_oracle_bullet_like.py only. */
/* Joint position limits (URDF revolute limits; Bullet: btMultiBodyJointLimitConstraint, SURVEY App. B.1 [third party,
 * restated]): one unilateral row per bound on the joint's velocity towards the free side. A joint still `gap` short of
 * its stop may close that gap within the substep (sign v >= -gap / h), a joint `pen` beyond it is pushed back with
 * Bullet's default ERP (sign v >= 0.2 pen / h); the row is listed while the joint could reach the stop within the
 * substep (gap <= (|qd| + 0.2 max_joint_velocity) h). Until round 6 the row existed only at or beyond the stop, with the ERP bias
 * alone: for a joint RESTING on its stop -- within rounding of gap = 0 -- its existence in a substep hung on the last
 * bit of q (in fp64: alternately a substep with the row and one of free acceleration into the stop; the fp32 kernels,
 * whose rounding put the joint back ON the stop, kept the row: the one-step parity test of round 6 found the two 1e-3
 * rad apart on 1 % of such steps). Twin: joint_limit_row, upkie_amd/csrc/dynamics.hpp. */
static int joint_limit_row(const UpkieModel* model, int j, double qj, double qdj, double h, double* sign, double* bias) {
  if (!(model->joint_lower[j] > -1e30 && model->joint_upper[j] < 1e30)) return 0;
  /* within reach of the stop in this substep: the joint's own speed plus what a substep can add to it (dynamics.hpp) */
  const double zone = (fabs(qdj) + 0.2 * model->max_joint_velocity) * h;
  double pen;
  if (qj - model->joint_lower[j] <= zone) {
    *sign = 1.0;
    pen = model->joint_lower[j] - qj;
  } else if (model->joint_upper[j] - qj <= zone) {
    *sign = -1.0;
    pen = qj - model->joint_upper[j];
  } else {
    return 0;
  }
  *bias = (pen > 0.0 ? 0.2 : 1.0) * pen / h;
  return 1;
}

static _Thread_local double* g_contact_sink; /* (defined below: where a contact-point query wants the points of the substep) */
#define BL_POINTS 4
#define BL_POINT_WORDS 8 /* point in the wheel frame (3), on the plane (3), applied normal impulse, live */
#define BL_ROWS (2 * BL_POINTS * 3 + NJ) /* every tire point's three rows + a limit row per joint (a model may bound its wheels too) */
static _Thread_local double g_bullet_manifold[2 * BL_POINTS * BL_POINT_WORDS];
static _Thread_local int g_bullet_active = 0;

static int bullet_like_contacts(const UpkieModel* model, const Kin* k, const double* L, const double* q, const double* qd, double h, double* nu) {
  const double breaking = model->contact_breaking_threshold;
  const double n[3] = {0, 0, 1};
  double J[BL_ROWS][NV], MinvJt[BL_ROWS][NV], rhs[BL_ROWS], cfm[BL_ROWS], lam[BL_ROWS];
  int kind[BL_ROWS], normal_row[BL_ROWS];
  double* applied_slot[BL_ROWS];
  int nrows = 0, any_contact = 0;
  int row_wheel[BL_ROWS]; /* contact rows: their tire and direction, for the contact-point query */
  double row_dir[BL_ROWS][3], first_point[2][3];
  int has_point[2] = {0, 0};
  /* joint limits (btMultiBodyJointLimitConstraint, ERP 0.2): solved first in every sweep */
  if (model->enforce_joint_limits) {
    for (int j = 0; j < NJ; ++j) {
      double sign, bias;
      if (!joint_limit_row(model, j, q[j], qd[j], h, &sign, &bias)) continue;
      memset(J[nrows], 0, sizeof(double) * NV);
      J[nrows][6 + j] = sign;
      kind[nrows] = 2; normal_row[nrows] = nrows; cfm[nrows] = 0.0; lam[nrows] = 0.0; applied_slot[nrows] = NULL;
      row_wheel[nrows] = -1;
      rhs[nrows] = -sign * nu[6 + j] + bias;
      ++nrows;
    }
  }
  const double kpc = model->contact_stiffness, kdc = model->contact_damping, denom = h * kpc + kdc;
  const double erp = denom > 0 ? h * kpc / denom : 0.2, cfm_n = denom > 0 ? 1.0 / (denom * h) : 0.0;
  for (int wheel = 0; wheel < 2; ++wheel) {
    const int body = 3 * wheel + 3, joint = 3 * wheel + 2;
    double* pts = g_bullet_manifold + wheel * BL_POINTS * BL_POINT_WORDS;
    /* btPersistentManifold::refreshContactPoints */
    for (int p = 0; p < BL_POINTS; ++p) {
      double* pt = pts + p * BL_POINT_WORDS;
      if (pt[7] == 0.0) continue;
      double r[3], A[3];
      m3_mulv(k->R[body], pt, r);
      for (int d = 0; d < 3; ++d) A[d] = k->o[body][d] + r[d];
      const double dist = A[2];
      const double dx = A[0] - pt[3], dy = A[1] - pt[4];
      if (dist > breaking || dx * dx + dy * dy > breaking * breaking) pt[7] = 0.0;
    }
    /* the deepest point of the tire circle (the support point of the tire against the plane) */
    {
      double center[3], r[3];
      m3_mulv(k->R[body], model->wheel_center[wheel], r);
      for (int d = 0; d < 3; ++d) center[d] = k->o[body][d] + r[d];
      const double* a = k->a[joint];
      const double u[3] = {-a[2] * a[0], -a[2] * a[1], 1.0 - a[2] * a[2]};
      const double un = sqrt(v3_dot(u, u));
      if (un >= 1e-6) {
        double P[3], rel[3], local[3];
        for (int d = 0; d < 3; ++d) P[d] = center[d] - model->wheel_radius * u[d] / un;
        if (P[2] <= breaking) {
          for (int d = 0; d < 3; ++d) rel[d] = P[d] - k->o[body][d];
          m3_tmulv(k->R[body], rel, local);
          /* btPersistentManifold::getCacheEntry: the nearest cached point within the threshold is replaced */
          int slot = -1;
          double nearest = breaking * breaking;
          for (int p = 0; p < BL_POINTS; ++p) {
            const double* pt = pts + p * BL_POINT_WORDS;
            if (pt[7] == 0.0) continue;
            const double d0 = pt[0] - local[0], d1 = pt[1] - local[1], d2 = pt[2] - local[2];
            const double dd = d0 * d0 + d1 * d1 + d2 * d2;
            if (dd < nearest) { nearest = dd; slot = p; }
          }
          double keep = 0.0;
          if (slot >= 0) {
            keep = pts[slot * BL_POINT_WORDS + 6]; /* replaceContactPoint keeps the applied impulse */
          } else {
            for (int p = 0; p < BL_POINTS && slot < 0; ++p)
              if (pts[p * BL_POINT_WORDS + 7] == 0.0) slot = p;
            if (slot < 0) { /* full: the shallowest cached point makes room (sortCachedPoints keeps the deepest) */
              double worst = -1e300;
              for (int p = 0; p < BL_POINTS; ++p) {
                double rr[3];
                m3_mulv(k->R[body], pts + p * BL_POINT_WORDS, rr);
                const double z = k->o[body][2] + rr[2];
                if (z > worst) { worst = z; slot = p; }
              }
            }
          }
          double* pt = pts + slot * BL_POINT_WORDS;
          for (int d = 0; d < 3; ++d) pt[d] = local[d];
          pt[3] = P[0]; pt[4] = P[1]; pt[5] = 0.0;
          pt[6] = keep;
          pt[7] = 1.0;
        }
      }
    }
    /* rows of every cached point */
    for (int p = 0; p < BL_POINTS; ++p) {
      double* pt = pts + p * BL_POINT_WORDS;
      if (pt[7] == 0.0) continue;
      any_contact = 1;
      double r[3], A[3], Jv[3][NV], Jw[3][NV], v[3];
      m3_mulv(k->R[body], pt, r);
      for (int d = 0; d < 3; ++d) A[d] = k->o[body][d] + r[d];
      point_jacobian(k, body, A, Jv, Jw);
      if (!has_point[wheel]) {
        has_point[wheel] = 1;
        for (int d = 0; d < 3; ++d) first_point[wheel][d] = A[d];
      }
      for (int d = 0; d < 3; ++d) {
        v[d] = 0.0;
        for (int c = 0; c < NV; ++c) v[d] += Jv[d][c] * nu[c];
      }
      const double vn = v3_dot(v, n), dist = A[2];
      double vt[3] = {v[0] - vn * n[0], v[1] - vn * n[1], v[2] - vn * n[2]};
      double t1[3], t2[3];
      const double lat2 = v3_dot(vt, vt);
      if (lat2 > 1.1920929e-07) { /* SIMD_EPSILON: friction along the sliding direction */
        const double inv = 1.0 / sqrt(lat2);
        for (int d = 0; d < 3; ++d) t1[d] = vt[d] * inv;
        v3_cross(t1, n, t2);
      } else { /* btPlaneSpace1(n) for n = z */
        t1[0] = 0; t1[1] = -1; t1[2] = 0;
        t2[0] = 1; t2[1] = 0; t2[2] = 0;
      }
      const double* dirs[3] = {n, t1, t2};
      for (int r_ = 0; r_ < 3; ++r_) {
        for (int c = 0; c < NV; ++c) J[nrows][c] = dirs[r_][0] * Jv[0][c] + dirs[r_][1] * Jv[1][c] + dirs[r_][2] * Jv[2][c];
        double rel = 0.0;
        for (int c = 0; c < NV; ++c) rel += J[nrows][c] * nu[c];
        row_wheel[nrows] = wheel;
        for (int d = 0; d < 3; ++d) row_dir[nrows][d] = dirs[r_][d];
        if (r_ == 0) {
          kind[nrows] = 0; normal_row[nrows] = nrows; cfm[nrows] = cfm_n;
          rhs[nrows] = dist <= 0.0 ? -rel + erp * (-dist) / h : -rel - dist / h;
          lam[nrows] = 0.85 * pt[6]; /* m_warmstartingFactor */
          applied_slot[nrows] = pt + 6;
        } else {
          kind[nrows] = 1; normal_row[nrows] = nrows - r_; cfm[nrows] = 0.0;
          rhs[nrows] = -rel;
          lam[nrows] = 0.0;
          applied_slot[nrows] = NULL;
        }
        ++nrows;
      }
    }
  }
  if (nrows == 0) return 0;
  static _Thread_local double W[BL_ROWS][BL_ROWS];
  for (int r_ = 0; r_ < nrows; ++r_) cholesky_solve(NV, L, J[r_], MinvJt[r_]);
  for (int a = 0; a < nrows; ++a)
    for (int b = 0; b < nrows; ++b) {
      double s_ = 0.0;
      for (int c = 0; c < NV; ++c) s_ += J[a][c] * MinvJt[b][c];
      W[a][b] = s_;
    }
  const double mu = model->friction_mu;
  for (int it = 0; it < model->pgs_iterations; ++it) {
    for (int pass = 0; pass < 2; ++pass) { /* joint limits, then normals */
      for (int r_ = 0; r_ < nrows; ++r_) {
        if (kind[r_] != (pass == 0 ? 2 : 0)) continue;
        double wl = 0.0;
        for (int b = 0; b < nrows; ++b) wl += W[r_][b] * lam[b];
        double x = lam[r_] + (rhs[r_] - wl - cfm[r_] * lam[r_]) / (W[r_][r_] + cfm[r_]);
        lam[r_] = x < 0.0 ? 0.0 : x;
      }
    }
    for (int r_ = 0; r_ < nrows; ++r_) { /* the two friction rows of a point together, projected onto the cone */
      if (kind[r_] != 1 || r_ != normal_row[r_] + 1) continue;
      const int r1 = r_, r2 = r_ + 1;
      double w1 = 0.0, w2 = 0.0;
      for (int b = 0; b < nrows; ++b) {
        w1 += W[r1][b] * lam[b];
        w2 += W[r2][b] * lam[b];
      }
      double x1 = lam[r1] + (rhs[r1] - w1) / W[r1][r1], x2 = lam[r2] + (rhs[r2] - w2) / W[r2][r2];
      const double lim = mu * lam[normal_row[r1]], norm = sqrt(x1 * x1 + x2 * x2);
      if (norm > lim) {
        const double sc = norm > 0.0 ? lim / norm : 0.0;
        x1 *= sc;
        x2 *= sc;
      }
      lam[r1] = x1;
      lam[r2] = x2;
    }
  }
  for (int r_ = 0; r_ < nrows; ++r_) {
    if (applied_slot[r_]) *applied_slot[r_] = lam[r_];
    for (int c = 0; c < NV; ++c) nu[c] += MinvJt[r_][c] * lam[r_];
  }
  if (g_contact_sink) { /* get_contact_points: per tire its (first) cached point and the force its points' impulses sum to */
    for (int wheel = 0; wheel < 2; ++wheel) {
      if (!has_point[wheel]) continue;
      g_contact_sink[8 * wheel] = 1.0;
      for (int d = 0; d < 3; ++d) g_contact_sink[8 * wheel + 1 + d] = first_point[wheel][d];
    }
    for (int r_ = 0; r_ < nrows; ++r_)
      if (row_wheel[r_] >= 0)
        for (int d = 0; d < 3; ++d) g_contact_sink[8 * row_wheel[r_] + 4 + d] += lam[r_] * row_dir[r_][d] / h;
  }
  return any_contact;
}

/* The sweeps' warm start across the substeps of ONE env.step() (the product's
 * rule, upkie_amd/csrc/dynamics.hpp SweepWarmStart): the impulses the previous
 * substep's sweeps ended on, per tire row (wheel w: entries 3w .. 3w + 2), and
 * with how many tires touching (0: it did not sweep). Armed by backend_step()
 * for the substeps of a step; single-substep calls start cold. */
static _Thread_local double g_sweep_warm_lam[6];
static _Thread_local int g_sweep_warm_swept = 0, g_sweep_warm_armed = 0;

/* Where oracle_substep_ext() leaves the contact points of the substep it
 * solved when asked (oracle_contact_points): [2][8] = per tire {exists,
 * position in world (3), force in world (3), 0}. */
static _Thread_local double* g_contact_sink = NULL; /* (declared above bullet_like_contacts) */

/* --------------------------------------------------------------- substep */
/* One Bullet-like stepSimulation() (call site pybullet_backend.py:306):
 * free acceleration -> contact/limit rows -> PGS -> velocity update ->
 * position integration (semi-implicit Euler, pinned by
 * upkie/cpp/interfaces/tests/BulletInterfaceTest.cpp:263-285). */
int oracle_substep_ext(const UpkieModel* model, double* s, const double tau[6],
                       double h, const double* body_inertials,
                       const double* ext_forces, const UpkieExternalForces* ext_slots) {
  double* pos = s + UPKIE_S_POS;
  double* quat = s + UPKIE_S_QUAT;
  double* linvel = s + UPKIE_S_LINVEL;
  double* angvel = s + UPKIE_S_ANGVEL;
  double* q = s + UPKIE_S_Q;
  double* qd = s + UPKIE_S_QD;

  oracle_debug_substeps += 1;
  Kin k;
  kinematics(model, body_inertials, pos, quat, q, &k);
  double M[NV * NV], bias[NV], L[NV * NV];
  mass_matrix_and_bias(&k, model->gravity, linvel, angvel, qd, M, bias);

  /* generalised applied forces */
  double Q[NV];
  memset(Q, 0, sizeof(Q));
  for (int j = 0; j < NJ; ++j) Q[6 + j] = tau[j] - model->joint_damping[j] * qd[j];
  {
    /* Bullet-style base damping, force ~ m v (k + k |v|) on the trunk */
    double Jv[3][NV], Jw[3][NV], r[3], vc[3], t[3], F[3], T[3], Iw_w[3];
    point_jacobian(&k, 0, k.c[0], Jv, Jw);
    for (int d = 0; d < 3; ++d) r[d] = k.c[0][d] - k.o[0][d];
    v3_cross(angvel, r, t);
    for (int d = 0; d < 3; ++d) vc[d] = linvel[d] + t[d];
    double vn = sqrt(v3_dot(vc, vc)), wn = sqrt(v3_dot(angvel, angvel));
    double kl = model->base_linear_damping, ka = model->base_angular_damping;
    m3_mulv(k.Iw[0], angvel, Iw_w);
    for (int d = 0; d < 3; ++d) {
      F[d] = -k.m[0] * vc[d] * (kl + kl * vn);
      T[d] = -Iw_w[d] * (ka + ka * wn);
    }
    for (int a = 0; a < NV; ++a)
      for (int d = 0; d < 3; ++d) Q[a] += Jv[d][a] * F[d] + Jw[d][a] * T[d];
  }
  if (ext_forces && ext_slots) {
    /* __apply_external_forces, pybullet_backend.py:625-658: each force acts on
     * one link at a point of that link, given in the world frame
     * (WORLD_FRAME) or in the link frame (LINK_FRAME) */
    for (int i = 0; i < ext_slots->count; ++i) {
      const int b = ext_slots->body[i];
      double Jv[3][NV], Jw[3][NV], r[3], P[3], F[3];
      m3_mulv(k.R[b], ext_slots->point[i], r);
      for (int d = 0; d < 3; ++d) P[d] = k.o[b][d] + r[d];
      if (ext_slots->local[i]) {
        m3_mulv(k.R[b], ext_forces + 3 * i, F);
      } else {
        for (int d = 0; d < 3; ++d) F[d] = ext_forces[3 * i + d];
      }
      point_jacobian(&k, b, P, Jv, Jw);
      for (int a = 0; a < NV; ++a)
        for (int d = 0; d < 3; ++d) Q[a] += Jv[d][a] * F[d];
    }
  }

  cholesky(NV, M, L);
  double rhs[NV], acc[NV], nu[NV];
  for (int a = 0; a < NV; ++a) rhs[a] = Q[a] - bias[a];
  cholesky_solve(NV, L, rhs, acc);
  for (int d = 0; d < 3; ++d) {
    nu[d] = linvel[d] + h * acc[d];
    nu[3 + d] = angvel[d] + h * acc[3 + d];
  }
  for (int j = 0; j < NJ; ++j) nu[6 + j] = qd[j] + h * acc[6 + j];

  int bullet_contact = -1;
  if (g_contact_sink) memset(g_contact_sink, 0, sizeof(double) * 16);
  if (g_bullet_active) bullet_contact = bullet_like_contacts(model, &k, L, q, qd, h, nu);

  /* constraint rows: per wheel (normal, t1, t2), then joint limits */
  double J[MAXROWS][NV], MinvJt[MAXROWS][NV];
  double rhs_c[MAXROWS], cfm[MAXROWS];
  int kind[MAXROWS], normal_row[MAXROWS];
  int nrows = 0, any_contact = 0;
  int contact_row[2] = {-1, -1};
  double contact_dirs[2][3][3];
  double kpc = model->contact_stiffness, kdc = model->contact_damping;
  double denom = h * kpc + kdc;
  double erp = denom > 0 ? h * kpc / denom : 0.2;
  double cfm_n = denom > 0 ? 1.0 / (denom * h) : 0.0;
  for (int wheel = 0; wheel < 2 && bullet_contact < 0; ++wheel) {
    int body = 3 * wheel + 3, joint = 3 * wheel + 2;
    double center[3], r[3];
    m3_mulv(k.R[body], model->wheel_center[wheel], r);
    for (int d = 0; d < 3; ++d) center[d] = k.o[body][d] + r[d];
    const double* a = k.a[joint];
    double n[3] = {0, 0, 1};
    double u[3] = {-a[2] * a[0], -a[2] * a[1], 1.0 - a[2] * a[2]};
    double un = sqrt(v3_dot(u, u));
    if (un < 1e-6) continue; /* wheel lying flat on its side */
    double P[3];
    for (int d = 0; d < 3; ++d) P[d] = center[d] - model->wheel_radius * u[d] / un;
    double dist = P[2];
    /* A contact point exists below Bullet's manifold breaking threshold;
     * that is also what getContactPoints reports (pybullet_backend.py:432). */
    if (dist > model->contact_breaking_threshold) continue;
    any_contact = 1;
    double t1[3], t2[3];
    v3_cross(a, n, t1);
    double t1n = sqrt(v3_dot(t1, t1));
    for (int d = 0; d < 3; ++d) t1[d] /= t1n;
    v3_cross(n, t1, t2);
    double Jv[3][NV], Jw[3][NV];
    point_jacobian(&k, body, P, Jv, Jw);
    const double* dirs[3] = {n, t1, t2};
    contact_row[wheel] = nrows;
    for (int r_ = 0; r_ < 3; ++r_)
      for (int d = 0; d < 3; ++d) contact_dirs[wheel][r_][d] = dirs[r_][d];
    if (g_contact_sink) {
      g_contact_sink[8 * wheel] = 1.0;
      for (int d = 0; d < 3; ++d) g_contact_sink[8 * wheel + 1 + d] = P[d];
    }
    for (int r_ = 0; r_ < 3; ++r_) {
      for (int c = 0; c < NV; ++c)
        J[nrows][c] = dirs[r_][0] * Jv[0][c] + dirs[r_][1] * Jv[1][c] +
                      dirs[r_][2] * Jv[2][c];
      double v = 0.0;
      for (int c = 0; c < NV; ++c) v += J[nrows][c] * nu[c];
      if (r_ == 0) {
        kind[nrows] = 0;
        normal_row[nrows] = nrows;
        cfm[nrows] = cfm_n;
        /* Bullet setupContactConstraint: penetration is pushed out with
         * ERP, a separated point may only close its gap within the step */
        rhs_c[nrows] = dist <= 0.0 ? -v + erp * (-dist) / h : -v - dist / h;
      } else {
        kind[nrows] = 1;
        normal_row[nrows] = nrows - r_;
        cfm[nrows] = model->friction_cfm;
        rhs_c[nrows] = -v;
      }
      ++nrows;
    }
  }
  if (bullet_contact >= 0) any_contact = bullet_contact; /* (contacts AND joint limits were solved there) */
  if (model->enforce_joint_limits && bullet_contact < 0) {
    for (int j = 0; j < NJ; ++j) {
      double sign, bias;
      if (!joint_limit_row(model, j, q[j], qd[j], h, &sign, &bias)) continue;
      memset(J[nrows], 0, sizeof(double) * NV);
      J[nrows][6 + j] = sign;
      kind[nrows] = 2;
      normal_row[nrows] = nrows;
      cfm[nrows] = 0.0;
      rhs_c[nrows] = -sign * nu[6 + j] + bias;
      ++nrows;
    }
  }

  double lam[MAXROWS];
  memset(lam, 0, sizeof(lam));
  if (nrows > 0) {
    /* (aligned explicitly: with rows of 12 doubles gcc 11 -O3 -march=native reads W with ALIGNED 32-byte loads -- the row stride
     * is a multiple of 32 bytes -- from a frame slot that is only 16-byte aligned: a general-protection fault in the sweeps of
     * bench.py's cpu_baseline build, round 6; neither sanitizer sees it. tests/test_oracle_native_build.py runs that build.) */
    double W[MAXROWS][MAXROWS] __attribute__((aligned(64)));
    for (int r_ = 0; r_ < nrows; ++r_) cholesky_solve(NV, L, J[r_], MinvJt[r_]);
    for (int a = 0; a < nrows; ++a)
      for (int b = 0; b < nrows; ++b) {
        double s_ = 0.0;
        for (int c = 0; c < NV; ++c) s_ += J[a][c] * MinvJt[b][c];
        W[a][b] = s_;
      }
    double mu = model->friction_mu;
    /* Direct solve first: in the usual regime (tires loaded, no slip) the
     * unconstrained solution of (W + CFM) lam = rhs already satisfies
     * lam_n >= 0 and |lam_t| <= mu lam_n and IS the solution. Otherwise it is
     * projected and used as the warm start of the projected Gauss-Seidel
     * sweeps (continuous at the stick/slip and lift-off boundaries). */
    int need_pgs = 1;
    {
      double A[MAXROWS * MAXROWS], La[MAXROWS * MAXROWS];
      for (int a = 0; a < nrows; ++a)
        for (int b = 0; b < nrows; ++b) A[a * nrows + b] = W[a][b] + (a == b ? cfm[a] : 0.0);
      if (cholesky(nrows, A, La) == 0) {
        cholesky_solve(nrows, La, rhs_c, lam);
        need_pgs = 0;
        for (int r_ = 0; r_ < nrows; ++r_)
          if (kind[r_] != 1 && lam[r_] < 0.0) {
            lam[r_] = 0.0;
            need_pgs = 1;
          }
        for (int r_ = 0; r_ < nrows; ++r_)
          if (kind[r_] == 1) {
            double lim = mu * lam[normal_row[r_]];
            if (lam[r_] < -lim) { lam[r_] = -lim; need_pgs = 1; }
            if (lam[r_] > lim) { lam[r_] = lim; need_pgs = 1; }
          }
      }
    }
    int sweeps_here = 0;
    /* the product's start of the sweeps (contact_sweeps_warm, dynamics.hpp), for the contact-only systems it applies
     * to there (no joint at its stop): both tires leaving the floor is lam = 0 without a sweep; a substep that follows
     * a swept one with the same tires on the floor starts from the impulses that one ended on */
    const int contact_only = nrows == 3 * ((contact_row[0] >= 0) + (contact_row[1] >= 0)) && nrows > 0;
    const int touching = (contact_row[0] >= 0) + (contact_row[1] >= 0);
    int swept_state = 0;
    if (need_pgs && contact_only) {
      int leaving = 1;
      for (int wheel = 0; wheel < 2; ++wheel)
        if (contact_row[wheel] >= 0 && rhs_c[contact_row[wheel]] > 0.0) leaving = 0;
      if (leaving) {
        memset(lam, 0, sizeof(lam));
        need_pgs = 0;
      } else {
        swept_state = touching;
        if (g_sweep_warm_armed && g_sweep_warm_swept == touching) {
          for (int wheel = 0; wheel < 2; ++wheel)
            if (contact_row[wheel] >= 0)
              for (int r_ = 0; r_ < 3; ++r_) lam[contact_row[wheel] + r_] = g_sweep_warm_lam[3 * wheel + r_];
          for (int r_ = 0; r_ < nrows; ++r_)
            if (kind[r_] == 0 && lam[r_] < 0.0) lam[r_] = 0.0;
          for (int r_ = 0; r_ < nrows; ++r_)
            if (kind[r_] == 1) {
              double lim = mu * lam[normal_row[r_]];
              if (lam[r_] < -lim) lam[r_] = -lim;
              if (lam[r_] > lim) lam[r_] = lim;
            }
        }
      }
    }
    double warm_start[MAXROWS];
    memcpy(warm_start, lam, sizeof(warm_start));
    for (int it = 0; need_pgs && it < model->pgs_iterations; ++it) {
      double change = 0.0, scale = 0.0;
      for (int pass = 0; pass < 3; ++pass) { /* normals, friction, limits */
        if (pass == 2) lateral_pair_sweep(nrows, kind, normal_row, &W[0][0], MAXROWS, cfm, rhs_c, mu, lam, &change, &scale);
        for (int r_ = 0; r_ < nrows; ++r_) {
          if (kind[r_] != pass) continue;
          if (kind[r_] == 1 && r_ == normal_row[r_] + 2 && lateral_pair_exists(nrows, kind, normal_row)) continue;
          double wl = 0.0;
          for (int b = 0; b < nrows; ++b) wl += W[r_][b] * lam[b];
          double delta = (rhs_c[r_] - wl - cfm[r_] * lam[r_]) / (W[r_][r_] + cfm[r_]);
          double x = lam[r_] + delta;
          if (kind[r_] == 1) {
            double lim = mu * lam[normal_row[r_]];
            if (x < -lim) x = -lim;
            if (x > lim) x = lim;
          } else if (x < 0.0) {
            x = 0.0;
          }
          if (fabs(x - lam[r_]) > change) change = fabs(x - lam[r_]);
          if (fabs(x) > scale) scale = fabs(x);
          lam[r_] = x;
        }
      }
      oracle_debug_sweeps += 1;
      sweeps_here += 1;
      /* converged: the sweep moved no impulse by more than pgs_tolerance of
       * the largest one (each env stops on its own criterion) */
      if (change <= model->pgs_tolerance * scale) break;
    }
    if (need_pgs && sweeps_here >= (oracle_debug_capture_threshold > 0 ? oracle_debug_capture_threshold : model->pgs_iterations) && nrows <= 6 &&
        (oracle_debug_capture_rows == 0 || nrows == oracle_debug_capture_rows)) {
      long slot;
#pragma omp atomic capture
      slot = oracle_debug_captured++;
      if (slot >= 0 && slot < ORACLE_CAPTURE_CASES) { /* (a negative start skips the first cases) */
        double* c = oracle_debug_capture[slot];
        c[0] = nrows;
        for (int a = 0; a < nrows; ++a) {
          for (int b = 0; b < nrows; ++b) c[1 + 6 * a + b] = W[a][b] + (a == b ? cfm[a] : 0.0);
          c[37 + a] = rhs_c[a];
          c[43 + a] = warm_start[a];
          c[49 + a] = lam[a];
        }
      }
    }
    if (need_pgs) {
      oracle_debug_fallbacks += 1;
      oracle_debug_sweep_hist[sweeps_here < 63 ? sweeps_here : 63] += 1;
    }
    g_sweep_warm_swept = need_pgs ? swept_state : 0;
    if (g_sweep_warm_swept)
      for (int wheel = 0; wheel < 2; ++wheel)
        for (int r_ = 0; r_ < 3; ++r_) g_sweep_warm_lam[3 * wheel + r_] = contact_row[wheel] >= 0 ? lam[contact_row[wheel] + r_] : 0.0;
    for (int r_ = 0; r_ < nrows; ++r_)
      for (int c = 0; c < NV; ++c) nu[c] += MinvJt[r_][c] * lam[r_];
  }

  if (nrows == 0) g_sweep_warm_swept = 0;
  if (g_contact_sink) {
    /* getContactPoints: normalForce and lateralFriction1/2 are the applied
     * impulses over the time step (pybullet_backend.py:697-708 sums them) */
    for (int wheel = 0; wheel < 2; ++wheel) {
      if (contact_row[wheel] < 0) continue;
      for (int d = 0; d < 3; ++d) {
        double f = 0.0;
        for (int r_ = 0; r_ < 3; ++r_) f += lam[contact_row[wheel] + r_] * contact_dirs[wheel][r_][d];
        g_contact_sink[8 * wheel + 4 + d] = f / h;
      }
    }
  }

  /* Bullet clamps generalised joint speeds to maxCoordinateVelocity */
  for (int j = 0; j < NJ; ++j) {
    if (nu[6 + j] > model->max_joint_velocity) nu[6 + j] = model->max_joint_velocity;
    if (nu[6 + j] < -model->max_joint_velocity) nu[6 + j] = -model->max_joint_velocity;
  }

  /* position integration with the NEW velocities */
  for (int d = 0; d < 3; ++d) {
    linvel[d] = nu[d];
    angvel[d] = nu[3 + d];
    pos[d] += h * nu[d];
  }
  for (int j = 0; j < NJ; ++j) {
    qd[j] = nu[6 + j];
    q[j] += h * nu[6 + j];
  }
  {
    double wn = sqrt(v3_dot(angvel, angvel));
    double half = 0.5 * h * wn;
    double kfac = wn > 1e-9 ? sin(half) / wn : 0.5 * h;
    double dq[4] = {cos(half), kfac * angvel[0], kfac * angvel[1], kfac * angvel[2]};
    double qn[4];
    quat_mul(dq, quat, qn);
    double n = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int d = 0; d < 4; ++d) quat[d] = qn[d] / n;
  }
  s[UPKIE_S_CONTACT] = any_contact ? 1.0 : 0.0;
  return any_contact;
}

/* ------------------------------------------------------- SoA <-> env state */
/* Spine observers attached to the simulation: the batched entry points point
 * this at the observer memory of the env they are stepping (one OpenMP thread
 * per env at a time). */
typedef struct {
  const UpkieObserverConfig* config;
  double st[UPKIE_OBSERVER_STATE_WORDS];
  int active;
} SpineHook;
static _Thread_local SpineHook g_spine;

static void spine_begin(const OracleRandomization* rnd, int B, int e) {
  g_bullet_active = rnd && rnd->bullet_manifold;
  if (g_bullet_active)
    for (int w = 0; w < ORACLE_BULLET_MANIFOLD_WORDS; ++w) g_bullet_manifold[w] = rnd->bullet_manifold[(int64_t)w * B + e];
  g_spine.active = rnd && rnd->observer_config && rnd->observer_state;
  if (!g_spine.active) return;
  g_spine.config = rnd->observer_config;
  for (int w = 0; w < UPKIE_OBSERVER_STATE_WORDS; ++w) g_spine.st[w] = rnd->observer_state[(int64_t)w * B + e];
}
static void spine_end(const OracleRandomization* rnd, int B, int e) {
  if (g_bullet_active) {
    for (int w = 0; w < ORACLE_BULLET_MANIFOLD_WORDS; ++w) rnd->bullet_manifold[(int64_t)w * B + e] = g_bullet_manifold[w];
    g_bullet_active = 0;
  }
  if (!g_spine.active) return;
  for (int w = 0; w < UPKIE_OBSERVER_STATE_WORDS; ++w) rnd->observer_state[(int64_t)w * B + e] = g_spine.st[w];
  g_spine.active = 0;
}
static void spine_reset(void) {
  if (g_bullet_active) memset(g_bullet_manifold, 0, sizeof(g_bullet_manifold)); /* resetBasePositionAndOrientation drops the contact cache */
  if (g_spine.active) memset(g_spine.st, 0, sizeof(g_spine.st));
}
static void spine_cycle(const double* s, const double tau[6], double h) {
  if (g_spine.active) oracle_observers_cycle_env(g_spine.config, h, g_spine.st, s + UPKIE_S_QD, tau);
}

static void load_env(const double* state, int B, int e, double s[NW]) {
  for (int w = 0; w < NW; ++w) s[w] = state[(int64_t)w * B + e];
}
static void store_env(double* state, int B, int e, const double s[NW]) {
  for (int w = 0; w < NW; ++w) state[(int64_t)w * B + e] = s[w];
}
static void env_randomization(const OracleRandomization* rnd, int B, int e,
                              double scale[NB * UPKIE_INERTIAL_WORDS], double force[3 * UPKIE_MAX_EXTERNAL_FORCES],
                              UpkieExternalForces* slots,
                              const double** scale_p, const double** force_p,
                              const UpkieExternalForces** slots_p) {
  *scale_p = NULL;
  *force_p = NULL;
  *slots_p = NULL;
  if (!rnd) return;
  if (rnd->body_inertials) { /* this env's inertial records, [10 * body + word] */
    for (int i = 0; i < NB * UPKIE_INERTIAL_WORDS; ++i) scale[i] = rnd->body_inertials[(int64_t)i * B + e];
    *scale_p = scale;
  }
  if (rnd->ext_force) {
    if (rnd->ext_slots) {
      *slots = *rnd->ext_slots;
    } else { /* one world-frame force on the trunk at ext_point */
      memset(slots, 0, sizeof(*slots));
      slots->count = 1;
      for (int d = 0; d < 3; ++d) slots->point[0][d] = rnd->ext_point[d];
    }
    for (int i = 0; i < slots->count; ++i)
      for (int d = 0; d < 3; ++d) force[3 * i + d] = rnd->ext_force[(int64_t)(3 * i + d) * B + e];
    *force_p = force;
    *slots_p = slots;
  }
}

/* ---------------------------------------------------------- observations */
/* pybullet_backend.py:352 */
static double pitch_from_quat(const double q[4]) {
  double x = 2.0 * (q[0] * q[2] - q[3] * q[1]);
  if (x > 1.0) x = 1.0;
  if (x < -1.0) x = -1.0;
  return asin(x);
}
/* pybullet_backend.py:476-490 */
static void wheel_odometry(const UpkieModel* model, const double s[NW],
                           double* position, double* velocity) {
  double signed_radius = model->left_sign * model->wheel_radius;
  *position = 0.5 * (s[UPKIE_S_Q + 2] - s[UPKIE_S_Q + 5]) * signed_radius;
  *velocity = 0.5 * (s[UPKIE_S_QD + 2] - s[UPKIE_S_QD + 5]) * signed_radius;
}
/* upkie_gyropod.py:186-214 */
static void gyropod_observation(const UpkieModel* model, const double s[NW],
                                double obs[6]) {
  double R[9], wb[3];
  quat_to_matrix(s + UPKIE_S_QUAT, R);
  m3_tmulv(R, s + UPKIE_S_ANGVEL, wb); /* pybullet_backend.py:355-361 */
  wheel_odometry(model, s, &obs[0], &obs[3]);
  obs[1] = pitch_from_quat(s + UPKIE_S_QUAT);
  obs[2] = s[UPKIE_S_YAW];
  obs[4] = wb[1];
  obs[5] = s[UPKIE_S_YAWVEL];
}

/* ------------------------------------------------------------------ reset */
static void euler_zyx_to_quat(double yaw, double pitch, double roll, double q[4]) {
  double qz[4] = {cos(0.5 * yaw), 0, 0, sin(0.5 * yaw)};
  double qy[4] = {cos(0.5 * pitch), 0, sin(0.5 * pitch), 0};
  double qx[4] = {cos(0.5 * roll), sin(0.5 * roll), 0, 0};
  double t[4];
  quat_mul(qz, qy, t);
  quat_mul(t, qx, q);
}
static double uniform(double low, double high, double u) { return low + (high - low) * u; }

static void reset_env(const UpkieModel* model, const UpkieSimConfig* cfg,
                      double s[NW], int64_t env_global, const double* scale,
                      const double* force, const UpkieExternalForces* point) {
  uint32_t episode = (uint32_t)s[UPKIE_S_EPISODE];
  double u0[4], u1[4], u2[4];
  philox_uniform4(cfg->seed, env_global, episode, STREAM_RESET, 0, u0);
  philox_uniform4(cfg->seed, env_global, episode, STREAM_RESET, 1, u1);
  philox_uniform4(cfg->seed, env_global, episode, STREAM_RESET, 2, u2);
  /* draw order of RobotState.sample_state, robot_state.py:182-187:
   * angular velocity, linear velocity, orientation, position */
  double w[3], v[3], ypr[3], p[3];
  w[0] = uniform(-cfg->rand_omega_x, cfg->rand_omega_x, u0[0]);
  w[1] = uniform(-cfg->rand_omega_y, cfg->rand_omega_y, u0[1]);
  w[2] = uniform(0.0, 0.0, u0[2]);
  v[0] = uniform(-cfg->rand_linvel[0], cfg->rand_linvel[0], u0[3]);
  v[1] = uniform(-cfg->rand_linvel[1], cfg->rand_linvel[1], u1[0]);
  v[2] = uniform(-cfg->rand_linvel[2], cfg->rand_linvel[2], u1[1]);
  ypr[0] = uniform(0.0, 0.0, u1[2]);
  ypr[1] = uniform(-cfg->rand_pitch, cfg->rand_pitch, u1[3]);
  ypr[2] = uniform(-cfg->rand_roll, cfg->rand_roll, u2[0]);
  p[0] = uniform(-cfg->rand_x, cfg->rand_x, u2[1]);
  p[1] = uniform(0.0, 0.0, u2[2]);
  p[2] = uniform(0.0, cfg->rand_z, u2[3]);
  double qr[4];
  euler_zyx_to_quat(ypr[0], ypr[1], ypr[2], qr);
  /* robot_state.py:158-160: base orientation * random orientation */
  quat_mul(cfg->init_quat, qr, s + UPKIE_S_QUAT);
  for (int d = 0; d < 3; ++d) {
    s[UPKIE_S_POS + d] = cfg->init_pos[d] + p[d];
    s[UPKIE_S_LINVEL + d] = cfg->init_linvel[d] + v[d];
    /* pybullet_backend.py:253-258: body-frame omega handed to Bullet as a
     * world-frame vector, reproduced as is */
    s[UPKIE_S_ANGVEL + d] = cfg->init_angvel[d] + w[d];
  }
  for (int j = 0; j < NJ; ++j) {
    s[UPKIE_S_Q + j] = cfg->init_joint[j];
    s[UPKIE_S_QD + j] = 0.0; /* resetJointState zeroes velocities, :261-267 */
  }
  /* pybullet_backend.py:228: one stepSimulation() with no motor torque */
  const double zero_tau[6] = {0, 0, 0, 0, 0, 0};
  spine_reset(); /* Observer::reset, then the spine cycles once */
  /* ... and no external force: __apply_external_forces only runs in step() (:303), not in reset (:220-232) */
  (void)force;
  (void)point;
  oracle_substep_ext(model, s, zero_tau, cfg->dt / cfg->nb_substeps, scale, NULL, NULL);
  spine_cycle(s, zero_tau, cfg->dt / cfg->nb_substeps);
  /* upkie_gyropod.py:236-240 */
  s[UPKIE_S_LEGREF + 0] = s[UPKIE_S_Q + 0];
  s[UPKIE_S_LEGREF + 1] = s[UPKIE_S_Q + 1];
  s[UPKIE_S_LEGREF + 2] = s[UPKIE_S_Q + 3];
  s[UPKIE_S_LEGREF + 3] = s[UPKIE_S_Q + 4];
  s[UPKIE_S_YAW] = 0.0;
  s[UPKIE_S_YAWVEL] = 0.0;
  s[UPKIE_S_MPC_V] = 0.0;
  s[UPKIE_S_SE2_X] = 0.0;
  s[UPKIE_S_SE2_Y] = 0.0;
  s[UPKIE_S_EPISODE] = (double)((episode + 1u) & UPKIE_COUNTER_MASK); /* counters live in fp32 words on the device: exact up to 2^24, then wrap */
  s[UPKIE_S_DONE] = 0.0;
  s[UPKIE_S_ELAPSED] = 0.0;
}

/* Push domain randomisation of BASELINE.json configs[4]:
 * the force a user script would hand to PyBulletBackend.set_external_forces
 * (pybullet_backend.py:603-658; examples/pybullet/apply_external_forces.py:37-43)
 * -- world frame, on the torso, norm ~ U(0, max_norm), uniformly random
 * horizontal direction -- for push number `push_index` of every env: one
 * Philox block keyed by (seed, global env id, push_index). force[3][B]. */
void oracle_sample_pushes(const UpkieSimConfig* cfg, uint32_t push_index, double max_norm, double* force) {
  int B = cfg->num_envs;
  for (int e = 0; e < B; ++e) {
    double u[4];
    philox_uniform4(cfg->seed, cfg->env_id_offset + e, push_index, STREAM_PUSH, 0, u);
    double norm = max_norm * u[0], phi = 6.283185307179586 * u[1];
    force[e] = norm * cos(phi);
    force[(int64_t)B + e] = norm * sin(phi);
    force[(int64_t)2 * B + e] = 0.0;
  }
}

/* PyBulletBackend.randomize_inertias, pybullet_backend.py:571-601, for every
 * env: epsilon ~ U(-v, v) per URDF link scales its mass and inertia (:588-594;
 * the root link is not in range(getNumJoints), :563); the links of a composite
 * body are then fused again: records[(10 * body + word)][B]. */
void oracle_sample_body_inertials(const UpkieModel* model, const UpkieSimConfig* cfg,
                                  double inertia_variation, double* records, double* link_scale) {
  int B = cfg->num_envs;
  int n = model->num_links > 0 ? model->num_links : NB;
  for (int e = 0; e < B; ++e) {
    double f[UPKIE_MAX_LINKS];
    for (int blk = 0; blk < UPKIE_MAX_LINKS / 4; ++blk) {
      double u[4];
      philox_uniform4(cfg->seed, cfg->env_id_offset + e, 0, STREAM_INERTIA, blk, u);
      for (int i = 0; i < 4; ++i) {
        int l = 4 * blk + i;
        int randomized = l < n && (model->num_links > 0 ? model->link_randomized[l] != 0 : 1);
        f[l] = randomized ? 1.0 + uniform(-inertia_variation, inertia_variation, u[i]) : 1.0;
        if (link_scale) link_scale[(int64_t)l * B + e] = f[l];
      }
    }
    for (int b = 0; b < NB; ++b) {
      double m = 0.0, mc[3] = {0, 0, 0};
      for (int l = 0; l < n; ++l) {
        int lb = model->num_links > 0 ? model->link_body[l] : l;
        if (lb != b) continue;
        double ml = f[l] * (model->num_links > 0 ? model->link_mass[l] : model->mass[l]);
        const double* c = model->num_links > 0 ? model->link_com[l] : model->com[l];
        m += ml;
        for (int d = 0; d < 3; ++d) mc[d] += ml * c[d];
      }
      double com[3] = {mc[0] / m, mc[1] / m, mc[2] / m};
      double I[6] = {0, 0, 0, 0, 0, 0};
      for (int l = 0; l < n; ++l) {
        int lb = model->num_links > 0 ? model->link_body[l] : l;
        if (lb != b) continue;
        double ml = f[l] * (model->num_links > 0 ? model->link_mass[l] : model->mass[l]);
        const double* c = model->num_links > 0 ? model->link_com[l] : model->com[l];
        const double* Il = model->num_links > 0 ? model->link_inertia[l] : model->inertia[l];
        double d[3] = {c[0] - com[0], c[1] - com[1], c[2] - com[2]};
        /* parallel axis: I + m (|d|^2 1 - d d') */
        I[0] += f[l] * Il[0] + ml * (d[1] * d[1] + d[2] * d[2]);
        I[1] += f[l] * Il[1] + ml * (d[0] * d[0] + d[2] * d[2]);
        I[2] += f[l] * Il[2] + ml * (d[0] * d[0] + d[1] * d[1]);
        I[3] += f[l] * Il[3] - ml * d[0] * d[1];
        I[4] += f[l] * Il[4] - ml * d[0] * d[2];
        I[5] += f[l] * Il[5] - ml * d[1] * d[2];
      }
      double* r = records + (int64_t)(UPKIE_INERTIAL_WORDS * b) * B + e;
      r[0] = m;
      for (int d = 0; d < 3; ++d) r[(int64_t)(1 + d) * B] = com[d];
      for (int d = 0; d < 6; ++d) r[(int64_t)(4 + d) * B] = I[d];
    }
  }
}

void oracle_reset(const UpkieModel* model, const UpkieSimConfig* cfg,
                  double* state, const uint8_t* mask,
                  const OracleRandomization* rnd, double* obs6) {
  int B = cfg->num_envs;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < B; ++e) {
    double s[NW], scale[NB * UPKIE_INERTIAL_WORDS], force[3 * UPKIE_MAX_EXTERNAL_FORCES];
    UpkieExternalForces slots;
    const double *sp, *fp;
    const UpkieExternalForces* pp;
    load_env(state, B, e, s);
    if (!mask || mask[e]) {
      env_randomization(rnd, B, e, scale, force, &slots, &sp, &fp, &pp);
      spine_begin(rnd, B, e);
      reset_env(model, cfg, s, cfg->env_id_offset + e, sp, fp, pp);
      spine_end(rnd, B, e);
      store_env(state, B, e, s);
    }
    if (obs6) gyropod_observation(model, s, obs6 + 6 * (int64_t)e);
  }
}

/* ------------------------------------------------------------------ steps */
/* clamp_and_warn, upkie/utils/clamp.py:42-58 (NaN passes through) */
static double clamp_like_reference(double value, double lower, double upper) {
  if (value < lower) return lower;
  if (value > upper) return upper;
  return value;
}

/* UpkieServos.get_spine_action, upkie_servos.py:316-344 */
static void clamp_servo_commands(const UpkieModel* model,
                                 const UpkieSimConfig* cfg,
                                 OracleServoCommand cmd[NJ]) {
  for (int j = 0; j < NJ; ++j) {
    double eff = model->joint_effort[j], vel = model->joint_velocity[j];
    cmd[j].position = clamp_like_reference(cmd[j].position, model->joint_lower[j], model->joint_upper[j]);
    cmd[j].velocity = clamp_like_reference(cmd[j].velocity, -vel, vel);
    cmd[j].feedforward_torque = clamp_like_reference(cmd[j].feedforward_torque, -eff, eff);
    cmd[j].kp_scale = clamp_like_reference(cmd[j].kp_scale, 0.0, cfg->max_gain_scale);
    cmd[j].kd_scale = clamp_like_reference(cmd[j].kd_scale, 0.0, cfg->max_gain_scale);
    cmd[j].maximum_torque = clamp_like_reference(cmd[j].maximum_torque, 0.0, eff);
    /* Non-finite guard (include/upkie_hip.h, "Non-finite commands and states"; the reference asserts instead,
     * pybullet_backend.py:519): what is still not finite behind the clamp becomes the neutral action's value
     * (upkie_servos.py:255-262) and is counted. NaN IS the neutral position; an infinite one (a wheel has no
     * position limits to clamp it) would make 0 x inf under kp_scale = 0. */
    int replaced = 0;
    if (isinf(cmd[j].position)) { cmd[j].position = NAN; ++replaced; }
    if (!isfinite(cmd[j].velocity)) { cmd[j].velocity = 0.0; ++replaced; }
    if (!isfinite(cmd[j].feedforward_torque)) { cmd[j].feedforward_torque = 0.0; ++replaced; }
    if (!isfinite(cmd[j].kp_scale)) { cmd[j].kp_scale = 1.0; ++replaced; }
    if (!isfinite(cmd[j].kd_scale)) { cmd[j].kd_scale = 1.0; ++replaced; }
    if (!isfinite(cmd[j].maximum_torque)) { cmd[j].maximum_torque = eff; ++replaced; }
    if (replaced) {
#pragma omp atomic
      oracle_guard_counts[0] += replaced;
    }
  }
}

/* Non-finite guard, second half: a state that is not finite behind the substeps (a force, an inertial record or an
 * uploaded state word that was not) is replaced by the configuration's initial state without randomisation, at rest;
 * returns 1 when it did (the step then reports `terminated` and flags the env done, like a fall). */
static int guard_state(const UpkieSimConfig* cfg, double s[NW]) {
  double mag = 0.0;
  for (int w = UPKIE_S_POS; w < UPKIE_S_QD + NJ; ++w) mag += fabs(s[w]);
  if (mag < 3.0e38) return 0; /* (the device's test, on fp32 words: neither NaN nor Inf, nor about to be one) */
  for (int d = 0; d < 3; ++d) {
    s[UPKIE_S_POS + d] = cfg->init_pos[d];
    s[UPKIE_S_LINVEL + d] = cfg->init_linvel[d];
    s[UPKIE_S_ANGVEL + d] = cfg->init_angvel[d];
  }
  for (int d = 0; d < 4; ++d) s[UPKIE_S_QUAT + d] = cfg->init_quat[d];
  for (int j = 0; j < NJ; ++j) {
    s[UPKIE_S_Q + j] = cfg->init_joint[j];
    s[UPKIE_S_QD + j] = 0.0;
    s[UPKIE_S_TORQUE + j] = 0.0;
  }
  s[UPKIE_S_LEGREF + 0] = s[UPKIE_S_Q + 0];
  s[UPKIE_S_LEGREF + 1] = s[UPKIE_S_Q + 1];
  s[UPKIE_S_LEGREF + 2] = s[UPKIE_S_Q + 3];
  s[UPKIE_S_LEGREF + 3] = s[UPKIE_S_Q + 4];
  s[UPKIE_S_CONTACT] = 0.0;
  if (g_bullet_active) memset(g_bullet_manifold, 0, sizeof(g_bullet_manifold)); /* a contact cache of that state means nothing */
#pragma omp atomic
  oracle_guard_counts[1] += 1;
  return 1;
}

/* PyBulletBackend.step, pybullet_backend.py:269-311 */
static int has_noise(const double sigma[NJ]) {
  for (int j = 0; j < NJ; ++j)
    if (sigma[j] > 1e-10) return 1;
  return 0;
}

static void backend_step(const UpkieModel* model, const UpkieSimConfig* cfg,
                         double s[NW], int64_t env_global,
                         const OracleServoCommand cmd[NJ],
                         const double* scale, const double* force,
                         const UpkieExternalForces* point) {
  double h = cfg->dt / cfg->nb_substeps;
  int noisy = has_noise(cfg->torque_control_noise);
  uint32_t step = (uint32_t)s[UPKIE_S_STEP];
  g_sweep_warm_armed = 1; /* the sweeps' warm start spans the substeps of this step */
  g_sweep_warm_swept = 0;
  for (int sub = 0; sub < cfg->nb_substeps; ++sub) {
    double tau[NJ], z[6] = {0, 0, 0, 0, 0, 0};
    /* control noise: one draw per joint per substep, :545-550 */
    if (noisy) philox_normal6(cfg->seed, env_global, step, (uint32_t)sub, z);
    for (int j = 0; j < NJ; ++j) {
      double sigma = cfg->torque_control_noise[j];
      tau[j] = joint_torque_with_noise(s[UPKIE_S_Q + j], s[UPKIE_S_QD + j], &cmd[j],
                                       cfg->torque_control_kp, cfg->torque_control_kd,
                                       cfg->joint_friction[j], sigma > 1e-10 ? sigma * z[j] : 0.0);
      s[UPKIE_S_TORQUE + j] = tau[j]; /* :293 */
    }
    oracle_substep_ext(model, s, tau, h, scale, force, point);
    spine_cycle(s, tau, h);
  }
  g_sweep_warm_armed = 0;
  s[UPKIE_S_STEP] = (double)((step + 1u) & UPKIE_COUNTER_MASK);
}

/* UpkieGyropod.__get_spine_action, upkie_gyropod.py:293-331 */
static void gyropod_commands(const UpkieModel* model, const UpkieSimConfig* cfg,
                             double s[NW], double ground_velocity_action,
                             double yaw_velocity_action,
                             OracleServoCommand cmd[NJ]) {
  double v = clamp_like_reference(ground_velocity_action, -cfg->max_ground_velocity, cfg->max_ground_velocity);
  double yawd = clamp_like_reference(yaw_velocity_action, -cfg->max_yaw_velocity, cfg->max_yaw_velocity);
  double wheel_velocity = v / model->wheel_radius;             /* :316 */
  double left_sign = model->left_sign;                         /* :317 */
  double left = left_sign * wheel_velocity;                    /* :318 */
  double right = -left_sign * wheel_velocity;                  /* :319 */
  double contact_radius = 0.5 * model->wheel_base;             /* :322 */
  double yaw_to_wheel = left_sign * contact_radius / model->wheel_radius;
  left += yaw_to_wheel * yawd;                                 /* :324 */
  right += yaw_to_wheel * yawd;                                /* :325 */
  const int leg_joint[4] = {0, 1, 3, 4};
  double alpha = cfg->dt / 1.0; /* filters.py:77, cutoff_period = 1.0 */
  for (int l = 0; l < 4; ++l) {
    double prev = s[UPKIE_S_LEGREF + l];
    double target = prev + alpha * (0.0 - prev); /* filters.py:80 */
    s[UPKIE_S_LEGREF + l] = target;
    OracleServoCommand* c = &cmd[leg_joint[l]];
    c->position = target;
    c->velocity = 0.0; /* neutral action, upkie_servos.py:255-262 */
    c->feedforward_torque = 0.0;
    c->kp_scale = cfg->leg_gain_scale; /* upkie_gyropod.py:261-266 */
    c->kd_scale = cfg->leg_gain_scale;
    c->maximum_torque = model->joint_effort[leg_joint[l]];
  }
  const int wheel_joint[2] = {2, 5};
  double wheel_cmd[2] = {left, right};
  for (int wi = 0; wi < 2; ++wi) {
    OracleServoCommand* c = &cmd[wheel_joint[wi]];
    c->position = NAN; /* upkie_gyropod.py:280-287 */
    c->velocity = wheel_cmd[wi];
    c->feedforward_torque = 0.0;
    c->kp_scale = 1.0;
    c->kd_scale = 1.0;
    c->maximum_torque = model->joint_effort[wheel_joint[wi]]; /* :289-290 */
  }
}

static void autoreset_or_null(const UpkieModel* model, const UpkieSimConfig* cfg,
                              double s[NW], int64_t env_global, const double* sp,
                              const double* fp, const UpkieExternalForces* pp, int* did_reset) {
  *did_reset = 0;
  if (cfg->autoreset_mode == UPKIE_AUTORESET_NEXT_STEP && s[UPKIE_S_DONE] != 0.0) {
    reset_env(model, cfg, s, env_global, sp, fp, pp);
    *did_reset = 1;
  }
}

/* gymnasium's TimeLimit for the batch (UpkieSimConfig.max_episode_steps): the
 * step that brings an episode to the limit reports `truncated` unless the
 * robot fell in it, and flags the env done like a fall does. An autoreset step
 * is not a step of the new episode. */
static uint8_t time_limit(const UpkieSimConfig* cfg, double s[NW], int did_reset, int terminated) {
  if (cfg->max_episode_steps <= 0 || did_reset) return 0;
  s[UPKIE_S_ELAPSED] += 1.0;
  if (s[UPKIE_S_ELAPSED] >= (double)cfg->max_episode_steps && !terminated) {
    s[UPKIE_S_DONE] = 1.0;
    return 1;
  }
  return 0;
}

static void step_gyropod_env(const UpkieModel* model, const UpkieSimConfig* cfg,
                             double s[NW], int64_t env_global, double a0,
                             double a1, double obs6[6], uint8_t* terminated, uint8_t* truncated,
                             const double* sp, const double* fp, const UpkieExternalForces* pp) {
  int did_reset;
  autoreset_or_null(model, cfg, s, env_global, sp, fp, pp, &did_reset);
  if (did_reset) {
    guard_state(cfg, s);
    gyropod_observation(model, s, obs6);
    *terminated = 0;
    *truncated = 0;
    return;
  }
  OracleServoCommand cmd[NJ];
  /* non-finite guard: a NaN action is the neutral one (zero velocities); an infinite one is clamped below */
  int replaced = 0;
  if (isnan(a0)) { a0 = 0.0; ++replaced; }
  if (isnan(a1)) { a1 = 0.0; ++replaced; }
  if (isinf(a1)) { a1 = a1 > 0 ? cfg->max_yaw_velocity : -cfg->max_yaw_velocity; ++replaced; } /* (the yaw integrates the unclamped action) */
  if (replaced) {
#pragma omp atomic
    oracle_guard_counts[0] += replaced;
  }
  gyropod_commands(model, cfg, s, a0, a1, cmd);
  clamp_servo_commands(model, cfg, cmd);
  backend_step(model, cfg, s, env_global, cmd, sp, fp, pp);
  s[UPKIE_S_YAW] += a1 * cfg->dt; /* upkie_gyropod.py:383-385, unclamped */
  s[UPKIE_S_YAWVEL] = a1;
  const int unsound = guard_state(cfg, s);
  gyropod_observation(model, s, obs6);
  /* __detect_fall, upkie_gyropod.py:344-345 */
  *terminated = (fabs(obs6[1]) > cfg->fall_pitch || unsound) ? 1 : 0;
  if (*terminated) s[UPKIE_S_DONE] = 1.0;
  *truncated = time_limit(cfg, s, 0, *terminated);
}

void oracle_step_gyropod(const UpkieModel* model, const UpkieSimConfig* cfg,
                         double* state, const double* act, double* obs,
                         double* reward, uint8_t* terminated,
                         uint8_t* truncated, const OracleRandomization* rnd) {
  int B = cfg->num_envs;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < B; ++e) {
    double s[NW], scale[NB * UPKIE_INERTIAL_WORDS], force[3 * UPKIE_MAX_EXTERNAL_FORCES];
    UpkieExternalForces slots;
    const double *sp, *fp;
    const UpkieExternalForces* pp;
    load_env(state, B, e, s);
    env_randomization(rnd, B, e, scale, force, &slots, &sp, &fp, &pp);
    spine_begin(rnd, B, e);
    step_gyropod_env(model, cfg, s, cfg->env_id_offset + e, act[2 * e], act[2 * e + 1],
                     obs + 6 * (int64_t)e, &terminated[e], &truncated[e], sp, fp, pp);
    reward[e] = 0.0; /* upkie_env.py:230 */
    spine_end(rnd, B, e);
    store_env(state, B, e, s);
  }
}

/* upkie_pendulum.py:17 */
static const int kPendulumObsIndices[4] = {1, 0, 4, 3};

void oracle_step_pendulum(const UpkieModel* model, const UpkieSimConfig* cfg,
                          double* state, const double* act, double* obs,
                          double* reward, uint8_t* terminated,
                          uint8_t* truncated, const OracleRandomization* rnd) {
  int B = cfg->num_envs;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < B; ++e) {
    double s[NW], scale[NB * UPKIE_INERTIAL_WORDS], force[3 * UPKIE_MAX_EXTERNAL_FORCES], obs6[6];
    UpkieExternalForces slots;
    const double *sp, *fp;
    const UpkieExternalForces* pp;
    load_env(state, B, e, s);
    env_randomization(rnd, B, e, scale, force, &slots, &sp, &fp, &pp);
    spine_begin(rnd, B, e);
    /* upkie_pendulum.py:139: action_2d = [action[0], 0.0] */
    step_gyropod_env(model, cfg, s, cfg->env_id_offset + e, act[e], 0.0, obs6,
                     &terminated[e], &truncated[e], sp, fp, pp);
    for (int i = 0; i < 4; ++i) obs[4 * (int64_t)e + i] = obs6[kPendulumObsIndices[i]];
    reward[e] = 0.0;
    spine_end(rnd, B, e);
    store_env(state, B, e, s);
  }
}

void oracle_step_pendulum_agent(const UpkieModel* model,
                                const UpkieSimConfig* cfg, double* state,
                                double* obs, double* reward,
                                uint8_t* terminated, uint8_t* truncated,
                                const OracleRandomization* rnd) {
  int B = cfg->num_envs;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < B; ++e) {
    double s[NW], scale[NB * UPKIE_INERTIAL_WORDS], force[3 * UPKIE_MAX_EXTERNAL_FORCES], obs6[6];
    UpkieExternalForces slots;
    const double *sp, *fp;
    const UpkieExternalForces* pp;
    load_env(state, B, e, s);
    env_randomization(rnd, B, e, scale, force, &slots, &sp, &fp, &pp);
    spine_begin(rnd, B, e);
    /* README.md:62-64 / examples/pybullet/pd_balancing.py:23-31 */
    double a = 0.0;
    for (int i = 0; i < 4; ++i) a += cfg->agent_gains[i] * obs[4 * (int64_t)e + i];
    a = clamp_like_reference(a, -cfg->agent_clip, cfg->agent_clip);
    step_gyropod_env(model, cfg, s, cfg->env_id_offset + e, a, 0.0, obs6,
                     &terminated[e], &truncated[e], sp, fp, pp);
    for (int i = 0; i < 4; ++i) obs[4 * (int64_t)e + i] = obs6[kPendulumObsIndices[i]];
    reward[e] = 0.0;
    spine_end(rnd, B, e);
    store_env(state, B, e, s);
  }
}

/* bench.py's cpu_baseline: `steps` consecutive env.step() of every env with
 * the README agent, ONE parallel region for the whole rollout. Envs are
 * independent, so each thread carries its chunk of envs through all the steps
 * without meeting the others (no barrier per step, an env's state stays in
 * the thread's cache between its steps); the per-step entry point above pays a
 * fork/join and a strided SoA load/store per step instead. Same arithmetic,
 * same results as `steps` calls of oracle_step_pendulum_agent. `obs` [B][4]
 * in/out; returns the number of terminations seen. */
int64_t oracle_rollout_pendulum_agent(const UpkieModel* model, const UpkieSimConfig* cfg, double* state,
                                      double* obs, int32_t steps, const OracleRandomization* rnd) {
  int B = cfg->num_envs;
  int64_t falls = 0;
#pragma omp parallel for schedule(static) reduction(+ : falls)
  for (int e = 0; e < B; ++e) {
    double s[NW], scale[NB * UPKIE_INERTIAL_WORDS], force[3 * UPKIE_MAX_EXTERNAL_FORCES], obs6[6], o4[4];
    UpkieExternalForces slots;
    const double *sp, *fp;
    const UpkieExternalForces* pp;
    uint8_t terminated, truncated;
    load_env(state, B, e, s);
    env_randomization(rnd, B, e, scale, force, &slots, &sp, &fp, &pp);
    spine_begin(rnd, B, e);
    for (int i = 0; i < 4; ++i) o4[i] = obs[4 * (int64_t)e + i];
    for (int k = 0; k < steps; ++k) {
      double a = 0.0;
      for (int i = 0; i < 4; ++i) a += cfg->agent_gains[i] * o4[i];
      a = clamp_like_reference(a, -cfg->agent_clip, cfg->agent_clip);
      step_gyropod_env(model, cfg, s, cfg->env_id_offset + e, a, 0.0, obs6, &terminated, &truncated, sp, fp, pp);
      for (int i = 0; i < 4; ++i) o4[i] = obs6[kPendulumObsIndices[i]];
      falls += terminated;
    }
    for (int i = 0; i < 4; ++i) obs[4 * (int64_t)e + i] = o4[i];
    spine_end(rnd, B, e);
    store_env(state, B, e, s);
  }
  return falls;
}

/* upkie_servos.py:288-306 with pybullet_backend.py:448-474 */
static void servo_observation(const UpkieSimConfig* cfg, int64_t env_global,
                              const double s[NW], double obs[30]) {
  double z[6] = {0, 0, 0, 0, 0, 0};
  /* measurement noise: one draw per joint per observation, :461-466 */
  if (has_noise(cfg->torque_measurement_noise))
    philox_normal6(cfg->seed, env_global, (uint32_t)s[UPKIE_S_STEP], NOISE_SLOT_MEASUREMENT, z);
  for (int j = 0; j < NJ; ++j) {
    double sigma = cfg->torque_measurement_noise[j];
    obs[5 * j + 0] = s[UPKIE_S_Q + j];
    obs[5 * j + 1] = s[UPKIE_S_QD + j];
    obs[5 * j + 2] = s[UPKIE_S_TORQUE + j] + (sigma > 1e-10 ? sigma * z[j] : 0.0);
    obs[5 * j + 3] = 42.0;
    obs[5 * j + 4] = 18.0;
  }
}

void oracle_step_servos(const UpkieModel* model, const UpkieSimConfig* cfg,
                        double* state, const double* act, double* obs,
                        double* reward, uint8_t* terminated,
                        uint8_t* truncated, const OracleRandomization* rnd) {
  int B = cfg->num_envs;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < B; ++e) {
    double s[NW], scale[NB * UPKIE_INERTIAL_WORDS], force[3 * UPKIE_MAX_EXTERNAL_FORCES];
    UpkieExternalForces slots;
    const double *sp, *fp;
    const UpkieExternalForces* pp;
    int did_reset;
    load_env(state, B, e, s);
    env_randomization(rnd, B, e, scale, force, &slots, &sp, &fp, &pp);
    spine_begin(rnd, B, e);
    autoreset_or_null(model, cfg, s, cfg->env_id_offset + e, sp, fp, pp, &did_reset);
    if (!did_reset) {
      OracleServoCommand cmd[NJ];
      const double* a = act + 36 * (int64_t)e;
      for (int j = 0; j < NJ; ++j) {
        cmd[j].position = a[6 * j + 0];
        cmd[j].velocity = a[6 * j + 1];
        cmd[j].feedforward_torque = a[6 * j + 2];
        cmd[j].kp_scale = a[6 * j + 3];
        cmd[j].kd_scale = a[6 * j + 4];
        cmd[j].maximum_torque = a[6 * j + 5];
      }
      clamp_servo_commands(model, cfg, cmd);
      backend_step(model, cfg, s, cfg->env_id_offset + e, cmd, sp, fp, pp);
    }
    const int unsound = guard_state(cfg, s);
    if (unsound && !did_reset) s[UPKIE_S_DONE] = 1.0;
    servo_observation(cfg, cfg->env_id_offset + e, s, obs + 30 * (int64_t)e);
    reward[e] = 0.0;
    terminated[e] = unsound && !did_reset ? 1 : 0; /* upkie_env.py:231-238: only the joystick ends it -- or the non-finite guard */
    truncated[e] = time_limit(cfg, s, did_reset, terminated[e]);
    spine_end(rnd, B, e);
    store_env(state, B, e, s);
  }
}

/* get_spine_observation, pybullet_backend.py:313-490 */
void oracle_observe(const UpkieModel* model, const UpkieSimConfig* cfg,
                    double* state, const OracleSpineObservation* out,
                    int update_imu) {
  int B = cfg->num_envs;
  for (int e = 0; e < B; ++e) {
    double s[NW], R[9], wb[3];
    load_env(state, B, e, s);
    quat_to_matrix(s + UPKIE_S_QUAT, R);
    m3_tmulv(R, s + UPKIE_S_ANGVEL, wb);
    if (out->pitch) out->pitch[e] = pitch_from_quat(s + UPKIE_S_QUAT);
    for (int d = 0; d < 3; ++d) {
      if (out->angular_velocity) out->angular_velocity[3 * e + d] = wb[d];
      if (out->linear_velocity) out->linear_velocity[3 * e + d] = s[UPKIE_S_LINVEL + d];
    }
    if (out->rotation_base_to_world)
      for (int i = 0; i < 9; ++i) out->rotation_base_to_world[9 * e + i] = R[i];
    if (out->floor_contact) out->floor_contact[e] = s[UPKIE_S_CONTACT] != 0.0;
    /* IMU, pybullet_backend.py:370-430 */
    {
      double Rbi_t[9], Riw[9], r[3], t[3], v_imu[3];
      m3_transpose(model->rot_base_to_imu, Rbi_t); /* imu -> base */
      m3_mul(R, Rbi_t, Riw);                       /* imu -> world */
      m3_mulv(R, model->imu_pos, r);
      v3_cross(s + UPKIE_S_ANGVEL, r, t);
      for (int d = 0; d < 3; ++d) v_imu[d] = s[UPKIE_S_LINVEL + d] + t[d];
      double Rars[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1}, Ria[9], quat[4];
      m3_mul(Rars, Riw, Ria);
      matrix_to_quat_scipy(Ria, quat);
      double a_w[3], a_i[3], proper_w[3], proper_i[3], w_i[3];
      for (int d = 0; d < 3; ++d) a_w[d] = (v_imu[d] - s[UPKIE_S_IMUVEL + d]) / cfg->dt;
      if (update_imu)
        for (int d = 0; d < 3; ++d) s[UPKIE_S_IMUVEL + d] = v_imu[d];
      m3_tmulv(Riw, s + UPKIE_S_ANGVEL, w_i);
      m3_tmulv(Riw, a_w, a_i);
      proper_w[0] = a_w[0];
      proper_w[1] = a_w[1];
      proper_w[2] = a_w[2] + 9.81; /* :418: gravity literal, not model */
      m3_tmulv(Riw, proper_w, proper_i);
      for (int d = 0; d < 4; ++d)
        if (out->imu_orientation) out->imu_orientation[4 * e + d] = quat[d];
      for (int d = 0; d < 3; ++d) {
        if (out->imu_angular_velocity) out->imu_angular_velocity[3 * e + d] = w_i[d];
        if (out->imu_linear_acceleration) out->imu_linear_acceleration[3 * e + d] = a_i[d];
        if (out->imu_raw_linear_acceleration) out->imu_raw_linear_acceleration[3 * e + d] = proper_i[d];
      }
    }
    if (out->servo) servo_observation(cfg, cfg->env_id_offset + e, s, out->servo + 30 * (int64_t)e);
    if (out->wheel_odometry)
      wheel_odometry(model, s, &out->wheel_odometry[2 * e], &out->wheel_odometry[2 * e + 1]);
    if (update_imu) store_env(state, B, e, s);
  }
}

/* ------------------------------------------------ helpers exposed to tests */
/* PyBulletBackend.get_contact_points, pybullet_backend.py:660-716, for the
 * batch: the contact solve of one substep from the current state under the
 * last commanded torques; the state is not modified. out [B][2][8]. */
void oracle_contact_points(const UpkieModel* model, const UpkieSimConfig* cfg,
                           const double* state, const OracleRandomization* rnd, double* out) {
  const int B = cfg->num_envs;
  const double h = cfg->dt / cfg->nb_substeps;
  for (int e = 0; e < B; ++e) {
    double s[NW], scale[NB * UPKIE_INERTIAL_WORDS], force[3 * UPKIE_MAX_EXTERNAL_FORCES];
    UpkieExternalForces slots;
    const double *scale_p, *force_p;
    const UpkieExternalForces* slots_p;
    load_env(state, B, e, s);
    env_randomization(rnd, B, e, scale, force, &slots, &scale_p, &force_p, &slots_p);
    g_contact_sink = out + (int64_t)16 * e;
    if (rnd && rnd->bullet_manifold) { /* a query: the substep runs on a copy of the env's manifold */
      for (int w = 0; w < ORACLE_BULLET_MANIFOLD_WORDS; ++w) g_bullet_manifold[w] = rnd->bullet_manifold[(int64_t)w * B + e];
      g_bullet_active = 1;
    }
    oracle_substep_ext(model, s, s + UPKIE_S_TORQUE, h, scale_p, force_p, slots_p);
    g_bullet_active = 0;
    g_contact_sink = NULL;
  }
}

void oracle_quat_to_matrix(const double quat_wxyz[4], double R[9]) { quat_to_matrix(quat_wxyz, R); }
void oracle_matrix_to_quat(const double R[9], double quat_wxyz[4]) { matrix_to_quat_scipy(R, quat_wxyz); }
void oracle_euler_zyx_compose(const double base_wxyz[4], const double ypr[3], double out_wxyz[4]) {
  double qr[4];
  euler_zyx_to_quat(ypr[0], ypr[1], ypr[2], qr);
  quat_mul(base_wxyz, qr, out_wxyz);
}
double oracle_clamp(double value, double lower, double upper) { return clamp_like_reference(value, lower, upper); }
/* upkie/utils/filters.py:63-80 */
double oracle_low_pass_filter(double prev_output, double cutoff_period, double new_input, double dt) {
  double alpha = dt / cutoff_period;
  return prev_output + alpha * (new_input - prev_output);
}
double oracle_pitch_from_quat(const double quat_wxyz[4]) { return pitch_from_quat(quat_wxyz); }


/* ---- command / observation maps exposed for the golden tests
 * (tests/test_reference_env_goldens.py compares them with the spine actions
 * the reference's own wrappers produced, tests/golden/reference_envs.json) */
void oracle_servo_commands(const UpkieModel* model, const UpkieSimConfig* cfg,
                           const double act[36], OracleServoCommand cmd[6]) {
  for (int j = 0; j < NJ; ++j) {
    cmd[j].position = act[6 * j + 0];
    cmd[j].velocity = act[6 * j + 1];
    cmd[j].feedforward_torque = act[6 * j + 2];
    cmd[j].kp_scale = act[6 * j + 3];
    cmd[j].kd_scale = act[6 * j + 4];
    cmd[j].maximum_torque = act[6 * j + 5];
  }
  clamp_servo_commands(model, cfg, cmd);
}

/* state: one env as an array of UPKIE_STATE_WORDS doubles (leg filter words
 * are advanced as a step would). */
void oracle_gyropod_commands(const UpkieModel* model, const UpkieSimConfig* cfg,
                             double* state, double ground_velocity, double yaw_velocity,
                             OracleServoCommand cmd[6]) {
  gyropod_commands(model, cfg, state, ground_velocity, yaw_velocity, cmd);
  clamp_servo_commands(model, cfg, cmd);
}

void oracle_gyropod_observation(const UpkieModel* model, const double* state, double obs6[6]) {
  gyropod_observation(model, state, obs6);
}

/* One world-frame force at a trunk point (the original entry point). */
/* One substep under the Bullet-like contact specification on a caller-held
 * manifold [ORACLE_BULLET_MANIFOLD_WORDS] (tests of the device's twin,
 * upkie_amd/csrc/bullet_like.hpp, substep by substep). */
int oracle_substep_bullet_like(const UpkieModel* model, double* s, const double tau[6], double h, double* manifold) {
  memcpy(g_bullet_manifold, manifold, sizeof(g_bullet_manifold));
  g_bullet_active = 1;
  const int contact = oracle_substep_ext(model, s, tau, h, NULL, NULL, NULL);
  memcpy(manifold, g_bullet_manifold, sizeof(g_bullet_manifold));
  g_bullet_active = 0;
  return contact;
}

int oracle_substep(const UpkieModel* model, double* s, const double tau[6],
                   double h, const double* body_inertials,
                   const double* ext_force, const double* ext_point) {
  UpkieExternalForces slots;
  memset(&slots, 0, sizeof(slots));
  slots.count = 1;
  if (ext_point)
    for (int d = 0; d < 3; ++d) slots.point[0][d] = ext_point[d];
  return oracle_substep_ext(model, s, tau, h, body_inertials, ext_force, ext_force ? &slots : NULL);
}

/*
 * upkie_oracle_observers.c -- CPU fp64 restatement of the spine's observer
 * pipeline (TEST INFRASTRUCTURE ONLY: tests/, smoke() and bench.py's
 * cpu_baseline leg are the only callers; the product never links this).
 *
 * Follows, statement by statement and in double precision:
 *   upkie/cpp/utils/low_pass_filter.h:17-35      low_pass_filter
 *   upkie/cpp/observers/WheelContact.cpp:19-48   WheelContact::observe
 *   upkie/cpp/observers/FloorContact.cpp:37-91   FloorContact::read
 *   upkie/cpp/observers/WheelOdometry.cpp:16-54  WheelOdometry::read
 *   upkie/cpp/observers/BaseOrientation.h:29-148 orientation helpers
 *   spines/common/observers.h:22-42              pipeline order
 * Pinned by the reference's own observer tests, restated in
 * tests/test_oracle_observers.py (FloorContactTest.cpp,
 * WheelOdometryObserverTest.cpp, BaseOrientationTest.cpp).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "upkie_oracle.h"

/* low_pass_filter.h:17-35; returns NaN where the reference throws FilterError
 * (callers check the configuration up front with oracle_observers_check). */
static double low_pass_filter(double prev_output, double cutoff_period,
                              double new_input, double dt) {
  if (cutoff_period <= 2.0 * dt) return NAN;
  const double alpha = dt / cutoff_period;
  return prev_output + alpha * (new_input - prev_output);
}

/* 0 when every filter of the pipeline satisfies cutoff > 2 dt, -1 otherwise. */
int oracle_observers_check(const UpkieObserverConfig* c) {
  if (!(c->dt > 0.0) || !isfinite(c->dt)) return -1;
  if (c->wheel_cutoff_period >= 1e-6 && c->wheel_cutoff_period <= 2.0 * c->dt) return -1;
  if (0.01 <= 2.0 * c->dt) return -1; /* kTorqueCutoffPeriod, FloorContact.cpp:87 */
  return 0;
}

/* State of one env, same words as enum UpkieObserverStateWord but double. */
typedef struct {
  double velocity, abs_acceleration, abs_torque, inertia, contact;
} WheelContactState;

/* WheelContact::observe, WheelContact.cpp:19-48 */
static void wheel_contact_observe(const UpkieObserverConfig* p,
                                  WheelContactState* s, double torque,
                                  double velocity, double dt) {
  if (p->wheel_cutoff_period < 1e-6) return; /* not configured */
  const double prev_velocity = s->velocity;
  s->velocity = low_pass_filter(s->velocity, p->wheel_cutoff_period, velocity, dt);
  const double new_acceleration = (s->velocity - prev_velocity) / dt;
  s->abs_acceleration = low_pass_filter(s->abs_acceleration, p->wheel_cutoff_period,
                                        fabs(new_acceleration), dt);
  s->abs_torque = low_pass_filter(s->abs_torque, p->wheel_cutoff_period, fabs(torque), dt);
  if (s->contact == 0.0 && (s->abs_acceleration < p->min_touchdown_acceleration ||
                            s->abs_torque < p->min_touchdown_torque)) {
    return;
  }
  s->inertia = s->abs_torque / (s->abs_acceleration + 1e-4);
  if (s->inertia < p->liftoff_inertia) {
    s->contact = 0.0;
  } else if (s->inertia > p->touchdown_inertia) {
    s->contact = 1.0;
  }
}

static void mat3_mul(const double* a, const double* b, double* out) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      out[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

/* compute_pitch_frame_in_parent, BaseOrientation.h:73-93 (Eigen's normalize()
 * leaves a zero vector untouched). R row-major. */
double oracle_pitch_frame_in_parent(const double R[9]) {
  double s[3] = {R[0], R[3], R[6]};
  double n2 = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
  if (n2 > 0.0) {
    const double n = sqrt(n2);
    s[0] /= n; s[1] /= n; s[2] /= n;
  }
  double h[3] = {s[0], s[1], 0.0}; /* sagittal - sagittal.z * e_z */
  n2 = h[0] * h[0] + h[1] * h[1];
  if (n2 > 0.0) {
    const double n = sqrt(n2);
    h[0] /= n; h[1] /= n;
  }
  if (R[8] < 0.0) {
    h[0] = -h[0]; h[1] = -h[1];
  }
  const double sign = (s[2] < 0.0) ? +1.0 : -1.0;
  double cos_pitch = s[0] * h[0] + s[1] * h[1] + s[2] * h[2];
  if (cos_pitch < -1.0) {
    cos_pitch = -1.0;
  } else if (cos_pitch > 1.0) {
    cos_pitch = 1.0;
  }
  return sign * acos(cos_pitch);
}

/* compute_base_orientation_from_imu, BaseOrientation.h:29-37; q = w x y z */
void oracle_base_orientation_from_imu(const double q[4], const double base_to_imu[9],
                                      const double ars_to_world[9], double R[9]) {
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const double I[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                       2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                       2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
  double T[9];
  mat3_mul(I, base_to_imu, T);
  mat3_mul(ars_to_world, T, R);
}

/*
 * One ObserverPipeline::run for B envs. Layouts: state [16][B] (double, words of
 * enum UpkieObserverStateWord), servo [B][6][5], imu_orientation [B][4] or
 * NULL, imu_angular_velocity [B][3], cross_button [B] or NULL; outputs as
 * UpkieObserverOutput but double / uint8, any may be NULL.
 */
void oracle_observers_step(const UpkieObserverConfig* p, double* state, const double* servo,
                           const double* imu_orientation, const double* imu_angular_velocity,
                           const uint8_t* cross_button, double* base_pitch,
                           double* base_angular_velocity, double* rotation_base_to_world,
                           uint8_t* floor_contact, double* upper_leg_torque,
                           double* wheel_contact_out, double* wheel_odometry) {
  const int B = p->num_envs;
  const double dt = p->dt;
  for (int e = 0; e < B; ++e) {
#define OW(w) state[(size_t)(w) * B + e]
    /* BaseOrientation::read + write, BaseOrientation.cpp:16-40 */
    if (imu_orientation) {
      double R[9];
      oracle_base_orientation_from_imu(imu_orientation + 4 * e, p->rotation_base_to_imu,
                                       p->rotation_ars_to_world, R);
      if (base_pitch) base_pitch[e] = oracle_pitch_frame_in_parent(R);
      if (rotation_base_to_world)
        for (int i = 0; i < 9; ++i) rotation_base_to_world[9 * e + i] = R[i];
      if (base_angular_velocity) {
        const double* w = imu_angular_velocity + 3 * e;
        for (int j = 0; j < 3; ++j) /* rotation_base_to_imu^T w, BaseOrientation.h:144-148 */
          base_angular_velocity[3 * e + j] = p->rotation_base_to_imu[j] * w[0] +
                                             p->rotation_base_to_imu[3 + j] * w[1] +
                                             p->rotation_base_to_imu[6 + j] * w[2];
      }
    }
    /* FloorContact::read, FloorContact.cpp:37-50 */
    const double* sv = servo + 30 * e;
    const int cross = cross_button && cross_button[e];
    int at_least_one_contact = 0;
    WheelContactState wc[2];
    double wheel_velocity[2];
    for (int w = 0; w < 2; ++w) { /* check_wheel_contacts, :52-73 */
      const int joint = 3 * w + 2;
      wc[w].velocity = OW(UPKIE_O_WHEEL + 5 * w + 0);
      wc[w].abs_acceleration = OW(UPKIE_O_WHEEL + 5 * w + 1);
      wc[w].abs_torque = OW(UPKIE_O_WHEEL + 5 * w + 2);
      wc[w].inertia = OW(UPKIE_O_WHEEL + 5 * w + 3);
      wc[w].contact = OW(UPKIE_O_WHEEL + 5 * w + 4);
      const double velocity = sv[5 * joint + 1], torque = sv[5 * joint + 2];
      wheel_velocity[w] = velocity;
      wheel_contact_observe(p, &wc[w], torque, velocity, dt);
      if (cross) {
        wc[w].contact = 0.0; /* reset_contact(), WheelContact.h:122 */
      } else if (wc[w].contact != 0.0) {
        at_least_one_contact = 1;
      }
      OW(UPKIE_O_WHEEL + 5 * w + 0) = wc[w].velocity;
      OW(UPKIE_O_WHEEL + 5 * w + 1) = wc[w].abs_acceleration;
      OW(UPKIE_O_WHEEL + 5 * w + 2) = wc[w].abs_torque;
      OW(UPKIE_O_WHEEL + 5 * w + 3) = wc[w].inertia;
      OW(UPKIE_O_WHEEL + 5 * w + 4) = wc[w].contact;
      if (wheel_contact_out) { /* FloorContact::write, :93-104 */
        double* o = wheel_contact_out + 8 * e + 4 * w;
        o[0] = wc[w].abs_acceleration;
        o[1] = wc[w].abs_torque;
        o[2] = wc[w].contact;
        o[3] = wc[w].inertia;
      }
    }
    double squared_torques = 0.0; /* update_upper_leg_torque, :75-91 */
    for (int side = 0; side < 2; ++side)
      for (int k = 0; k < 2; ++k) {
        const double torque = sv[5 * (3 * side + k) + 2];
        squared_torques += torque * torque;
      }
    const double upper = low_pass_filter(OW(UPKIE_O_UPPER_LEG_TORQUE), 0.01, sqrt(squared_torques), dt);
    OW(UPKIE_O_UPPER_LEG_TORQUE) = upper;
    const int contact = at_least_one_contact || (upper > p->upper_leg_torque_threshold);
    OW(UPKIE_O_CONTACT) = contact ? 1.0 : 0.0;
    if (floor_contact) floor_contact[e] = (uint8_t)contact;
    if (upper_leg_torque) upper_leg_torque[e] = upper;
    /* WheelOdometry::read, WheelOdometry.cpp:16-54 */
    if (contact) {
      double velocity_sum = 0.0;
      unsigned nb = 0u;
      for (int w = 0; w < 2; ++w) {
        if (wc[w].contact == 0.0) continue;
        velocity_sum += p->signed_radius[w] * wheel_velocity[w];
        ++nb;
      }
      const double v = nb == 0u ? 0.0 : velocity_sum / nb;
      OW(UPKIE_O_ODOMETRY_VELOCITY) = v;
      OW(UPKIE_O_ODOMETRY_POSITION) += v * dt;
    }
    if (wheel_odometry) {
      wheel_odometry[2 * e] = OW(UPKIE_O_ODOMETRY_POSITION);
      wheel_odometry[2 * e + 1] = OW(UPKIE_O_ODOMETRY_VELOCITY);
    }
#undef OW
  }
}

/* One spine cycle of FloorContact + WheelOdometry for ONE env whose observer
 * memory is st[UPKIE_OBSERVER_STATE_WORDS] (array of words, not SoA): used by
 * the simulator when the observers run inside the step, once per physics
 * substep of duration dt (Spine::simulate, Spine.cpp:119-141). velocity /
 * torque: the six servo readings of this cycle. */
void oracle_observers_cycle_env(const UpkieObserverConfig* p, double dt, double* st,
                                const double velocity[6], const double torque[6]) {
  int at_least_one_contact = 0;
  WheelContactState wc[2];
  for (int w = 0; w < 2; ++w) {
    const int joint = 3 * w + 2;
    wc[w].velocity = st[UPKIE_O_WHEEL + 5 * w + 0];
    wc[w].abs_acceleration = st[UPKIE_O_WHEEL + 5 * w + 1];
    wc[w].abs_torque = st[UPKIE_O_WHEEL + 5 * w + 2];
    wc[w].inertia = st[UPKIE_O_WHEEL + 5 * w + 3];
    wc[w].contact = st[UPKIE_O_WHEEL + 5 * w + 4];
    wheel_contact_observe(p, &wc[w], torque[joint], velocity[joint], dt);
    if (wc[w].contact != 0.0) at_least_one_contact = 1;
    st[UPKIE_O_WHEEL + 5 * w + 0] = wc[w].velocity;
    st[UPKIE_O_WHEEL + 5 * w + 1] = wc[w].abs_acceleration;
    st[UPKIE_O_WHEEL + 5 * w + 2] = wc[w].abs_torque;
    st[UPKIE_O_WHEEL + 5 * w + 3] = wc[w].inertia;
    st[UPKIE_O_WHEEL + 5 * w + 4] = wc[w].contact;
  }
  const double squared = torque[0] * torque[0] + torque[1] * torque[1] + torque[3] * torque[3] + torque[4] * torque[4];
  const double upper = low_pass_filter(st[UPKIE_O_UPPER_LEG_TORQUE], 0.01, sqrt(squared), dt);
  st[UPKIE_O_UPPER_LEG_TORQUE] = upper;
  const int contact = at_least_one_contact || (upper > p->upper_leg_torque_threshold);
  st[UPKIE_O_CONTACT] = contact ? 1.0 : 0.0;
  if (contact) {
    double velocity_sum = 0.0;
    unsigned nb = 0u;
    for (int w = 0; w < 2; ++w) {
      if (wc[w].contact == 0.0) continue;
      velocity_sum += p->signed_radius[w] * velocity[3 * w + 2];
      ++nb;
    }
    const double v = nb == 0u ? 0.0 : velocity_sum / nb;
    st[UPKIE_O_ODOMETRY_VELOCITY] = v;
    st[UPKIE_O_ODOMETRY_POSITION] += v * dt;
  }
}

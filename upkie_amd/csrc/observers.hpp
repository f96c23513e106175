// observers.hpp -- the spine's observer pipeline for a batch of envs
// (spines/common/observers.h:22-42): BaseOrientation -> FloorContact (two
// WheelContact estimators + upper-leg torque) -> WheelOdometry, one env per
// lane. Pure load/filter/store work: ~45 words read, ~16 + outputs written per
// env, HBM bound. State is SoA [UPKIE_OBSERVER_STATE_WORDS][B] so every state
// access of a wave is one coalesced 256-byte transaction; inputs and outputs
// keep the row-major layouts of UpkieSpineObservation and move through LDS as
// contiguous runs (wave_io.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/upkie_hip.h"
#include "wave_io.hpp"

namespace upkie {

struct ObserverDev {
  int num_envs;
  int wheels_configured;  // wheel_cutoff_period >= 1e-6, WheelContact.cpp:21-24
  float dt;
  float wheel_alpha;  // dt / cutoff_period, low_pass_filter.h:33
  float leg_alpha;    // dt / 0.01, FloorContact.cpp:87-90
  float inv_dt;
  float upper_leg_torque_threshold;
  float liftoff_inertia, min_touchdown_acceleration, min_touchdown_torque, touchdown_inertia;
  float signed_radius[2];
  float base_to_imu[9];
  float ars_to_world[9];
};

// upkie/cpp/utils/low_pass_filter.h:32-34 with alpha = dt / cutoff precomputed
__device__ __forceinline__ float obs_low_pass(float prev, float alpha, float input) { return prev + alpha * (input - prev); }

// WheelContact::observe, WheelContact.cpp:19-48 (one spine cycle of one wheel)
__device__ __forceinline__ void wheel_contact_observe(const ObserverDev& P, float torque, float velocity, float& filt_vel, float& abs_acc,
                                                      float& abs_tau, float& inertia, bool& contact) {
  if (!P.wheels_configured) return;
  const float prev = filt_vel;
  filt_vel = obs_low_pass(filt_vel, P.wheel_alpha, velocity);
  const float acc = (filt_vel - prev) * P.inv_dt;
  abs_acc = obs_low_pass(abs_acc, P.wheel_alpha, fabsf(acc));
  abs_tau = obs_low_pass(abs_tau, P.wheel_alpha, fabsf(torque));
  const bool skip = !contact && (abs_acc < P.min_touchdown_acceleration || abs_tau < P.min_touchdown_torque);
  if (!skip) {
    inertia = abs_tau / (abs_acc + 1e-4f);
    if (inertia < P.liftoff_inertia) {
      contact = false;
    } else if (inertia > P.touchdown_inertia) {
      contact = true;
    }
  }
}

#if !defined(UPKIE_STEP_INSTANCES_ONLY)  // (step_instances.hip: the non-template kernels live in the C-ABI's translation unit alone)
__global__ __launch_bounds__(64) void observers_reset_kernel(int B, float* __restrict__ st, const uint8_t* __restrict__ mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  if (mask && !mask[e]) return;
#pragma unroll
  for (int w = 0; w < UPKIE_OBSERVER_STATE_WORDS; ++w) st[(size_t)w * B + e] = 0.f;
}

__global__ __launch_bounds__(64) void observers_step_kernel(ObserverDev P, float* __restrict__ st, UpkieObserverInput in,
                                                            UpkieObserverOutput out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 30];
  const int B = P.num_envs;
  const int e0 = blockIdx.x * blockDim.x;
  const int n_valid = min(64, B - e0);
  // lanes past the batch shadow the last env: they take part in the LDS staging, nothing of theirs is stored
  const bool live = (int)threadIdx.x < n_valid;
  const int e = live ? e0 + (int)threadIdx.x : B - 1;
#define OW(w) st[(size_t)(w) * B + e]

  // ---- BaseOrientation::read (BaseOrientation.cpp:16-34) -------------------
  if (in.imu_orientation) {
    const float4 q = reinterpret_cast<const float4*>(in.imu_orientation)[e];  // w x y z
    const float qw = q.x, qx = q.y, qy = q.z, qz = q.w;
    // Eigen::Quaterniond::toRotationMatrix
    const float I[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qz * qw), 2.f * (qx * qz + qy * qw),
                        2.f * (qx * qy + qz * qw), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qx * qw),
                        2.f * (qx * qz - qy * qw), 2.f * (qy * qz + qx * qw), 1.f - 2.f * (qx * qx + qy * qy)};
    float T[9], R[9];  // R = ars_to_world * I * base_to_imu, BaseOrientation.h:33-36
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        T[3 * i + j] = I[3 * i] * P.base_to_imu[j] + I[3 * i + 1] * P.base_to_imu[3 + j] + I[3 * i + 2] * P.base_to_imu[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        R[3 * i + j] = P.ars_to_world[3 * i] * T[j] + P.ars_to_world[3 * i + 1] * T[3 + j] + P.ars_to_world[3 * i + 2] * T[6 + j];
    // compute_pitch_frame_in_parent, BaseOrientation.h:73-93. With s the
    // normalised first column and heading = +-(sx, sy, 0)/|(sx, sy)|,
    // cos_pitch = +-|(sx, sy)| and acos(cos_pitch) = atan2(|sz|, cos_pitch):
    // same angle, but well conditioned in fp32 around pitch = 0.
    float sx = R[0], sy = R[3], sz = R[6];
    const float n2 = sx * sx + sy * sy + sz * sz;
    if (n2 > 0.f) {
      const float inv = 1.0f / sqrtf(n2);
      sx *= inv; sy *= inv; sz *= inv;
    }
    float hxy = sqrtf(sx * sx + sy * sy);
    if (R[8] < 0.f) hxy = -hxy;
    const float sign = sz < 0.f ? 1.f : -1.f;
    const float pitch = sign * atan2f(fabsf(sz), hxy);
    if (out.base_pitch && live) out.base_pitch[e] = pitch;
    if (out.rotation_base_to_world) wave_store_rows(out.rotation_base_to_world, e0, n_valid, R, lds);
    if (out.base_angular_velocity) {  // base_to_imu^T * omega_imu, BaseOrientation.h:144-148
      float w[3], wb[3];
      wave_load_rows(in.imu_angular_velocity, e0, n_valid, w, lds);
#pragma unroll
      for (int j = 0; j < 3; ++j) wb[j] = P.base_to_imu[j] * w[0] + P.base_to_imu[3 + j] * w[1] + P.base_to_imu[6 + j] * w[2];
      wave_store_rows(out.base_angular_velocity, e0, n_valid, wb, lds);
    }
  }

  // ---- FloorContact::read (FloorContact.cpp:37-50) -------------------------
  if (!in.servo) return;  // no "servo" block: nothing to read, FloorContact.cpp:42-44
  float servo[30];        // [6][5]: position, velocity, torque, ...
  wave_load_rows(in.servo, e0, n_valid, servo, lds);
  if (!live) return;  // (no staging below)
  const bool cross = in.cross_button && in.cross_button[e];
  bool any_wheel = false;
  bool wheel_contact[2];
  float wheel_velocity[2];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const int joint = 3 * w + 2;  // left_wheel, right_wheel
    const float velocity = servo[5 * joint + 1], torque = servo[5 * joint + 2];
    wheel_velocity[w] = velocity;
    float filt_vel = OW(UPKIE_O_WHEEL + 5 * w + 0), abs_acc = OW(UPKIE_O_WHEEL + 5 * w + 1), abs_tau = OW(UPKIE_O_WHEEL + 5 * w + 2),
          inertia = OW(UPKIE_O_WHEEL + 5 * w + 3);
    bool contact = OW(UPKIE_O_WHEEL + 5 * w + 4) != 0.f;
    wheel_contact_observe(P, torque, velocity, filt_vel, abs_acc, abs_tau, inertia, contact);
    if (cross) {  // FloorContact.cpp:62-64
      contact = false;
    } else if (contact) {
      any_wheel = true;
    }
    wheel_contact[w] = contact;
    OW(UPKIE_O_WHEEL + 5 * w + 0) = filt_vel;
    OW(UPKIE_O_WHEEL + 5 * w + 1) = abs_acc;
    OW(UPKIE_O_WHEEL + 5 * w + 2) = abs_tau;
    OW(UPKIE_O_WHEEL + 5 * w + 3) = inertia;
    OW(UPKIE_O_WHEEL + 5 * w + 4) = contact ? 1.f : 0.f;
    if (out.wheel_contact) {
      float4 o = make_float4(abs_acc, abs_tau, contact ? 1.f : 0.f, inertia);
      reinterpret_cast<float4*>(out.wheel_contact)[(size_t)2 * e + w] = o;
    }
  }
  // update_upper_leg_torque, FloorContact.cpp:75-91: hips and knees
  float squared = 0.f;
#pragma unroll
  for (int side = 0; side < 2; ++side)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float t = servo[5 * (3 * side + k) + 2];
      squared += t * t;
    }
  const float upper = obs_low_pass(OW(UPKIE_O_UPPER_LEG_TORQUE), P.leg_alpha, sqrtf(squared));
  OW(UPKIE_O_UPPER_LEG_TORQUE) = upper;
  const bool contact = any_wheel || upper > P.upper_leg_torque_threshold;  // FloorContact.cpp:48-49
  OW(UPKIE_O_CONTACT) = contact ? 1.f : 0.f;
  if (out.floor_contact) out.floor_contact[e] = contact ? 1 : 0;
  if (out.upper_leg_torque) out.upper_leg_torque[e] = upper;

  // ---- WheelOdometry::read (WheelOdometry.cpp:16-54) -----------------------
  float position = OW(UPKIE_O_ODOMETRY_POSITION), velocity = OW(UPKIE_O_ODOMETRY_VELOCITY);
  if (contact) {
    float sum = 0.f;
    int n = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w)
      if (wheel_contact[w]) {
        sum += P.signed_radius[w] * wheel_velocity[w];
        ++n;
      }
    velocity = n == 0 ? 0.f : sum / (float)n;  // :47-50: contact through leg torque only
    position += velocity * P.dt;
    OW(UPKIE_O_ODOMETRY_POSITION) = position;
    OW(UPKIE_O_ODOMETRY_VELOCITY) = velocity;
  }
  if (out.wheel_odometry) reinterpret_cast<float2*>(out.wheel_odometry)[e] = make_float2(position, velocity);
#undef OW
}
#endif  // UPKIE_STEP_INSTANCES_ONLY

}  // namespace upkie

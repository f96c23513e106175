// step_kernels.hpp -- the fused env.step() kernels (one, two and eight lanes per env) and what they share: the
// settings blocks, Philox streams, the servo torque law, the action and observation maps. Included by upkie_hip.hip (the
// C-ABI, which only DECLARES the instantiations it launches: step_instances.hpp) and by step_instances.hip (which
// defines them, one group per translation unit so that the library builds on all cores).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>

#include "dynamics.hpp"
#include "bullet_like.hpp"
#include "mpc.hpp"
#include "observers.hpp"
#include "wave_io.hpp"

namespace upkie {

struct DevConfig {
  int num_envs;
  int nb_substeps;
  float dt;
  float h;  // dt / nb_substeps
  float kp, kd;
  float joint_friction[UPKIE_NJ];
  float control_noise[UPKIE_NJ];      // torque_control_noise std dev
  float measurement_noise[UPKIE_NJ];  // torque_measurement_noise std dev
  int any_control_noise, any_measurement_noise;
  float fall_pitch, max_ground_velocity, max_yaw_velocity, leg_gain_scale, max_gain_scale;
  float init_pos[3], init_quat[4], init_linvel[3], init_angvel[3], init_joint[UPKIE_NJ];
  float rand_roll, rand_pitch, rand_x, rand_z, rand_omega_x, rand_omega_y, rand_linvel[3];
  unsigned seed_lo, seed_hi;
  unsigned env_lo, env_hi;  // env_id_offset
  int autoreset_mode;
  int max_episode_steps;  // 0: no time limit
  float agent_gains[4];
  float agent_clip;
  ExtSlots ext;
  ObserverDev spine;  // in-step spine observers (one cycle per physics substep), used when attached
  unsigned* guard;    // non-finite guard counters of the handle (device, [2]: command words replaced, env states replaced) or null
};

// What the eight-lane step kernels read of a handle's settings, in DEVICE memory (round 3): by value the two structures
// were 1.2 KB of kernel arguments -- a segment the CPU writes over PCIe for every launch and that is not cached in L2
// (a dependent scalar load from it costs ~460 cycles, an L2 hit ~160: profiles/r02_kernarg_latency.txt). The host
// keeps the block current with a small store kernel on the launching stream whenever a setting changed (UpkieSim).
struct DevParams {
  DevLimits limits;
  DevConfig config;
};

// DevConfig::autoreset_mode value of the second launch of a SAME_STEP autoreset
// (upkie_sim_autoreset_done): only the envs whose DONE word is set run, down
// the reset branch of the step that wrote their terminal observation.
enum { AUTORESET_DONE_PASS = 100 };
// words per env of a step mode's observation buffer (0: none)
template <int MODE>
struct ObsWords { static constexpr int value = MODE == 1 || MODE == 2 || MODE == 6 ? 4 : MODE == 3 ? 6 : MODE == 4 ? 30 : MODE == 5 ? 3 : 0; };

enum Mode {
  MODE_RESET = 0,
  MODE_PENDULUM = 1,
  MODE_PENDULUM_AGENT = 2,
  MODE_GYROPOD = 3,
  MODE_SERVOS = 4,
  MODE_BASE_VELOCITY = 5,
  // MODE_PENDULUM_AGENT with several steps per launch (two-lane kernel only): its own
  // instantiation, because the step loop around the body costs the one-step kernel 2.7 %
  MODE_PENDULUM_ROLLOUT = 6
};
constexpr bool fused_agent(int mode) { return mode == MODE_PENDULUM_AGENT || mode == MODE_PENDULUM_ROLLOUT; }

// Extra buffers of the fused UpkieBaseVelocity step (upkie_base_velocity.py:164-202).
struct BaseVelocityPtrs {
  const float* commanded;  // [B] ground velocity out of the MPC balancer
  float* x0;               // [B][4] next MPC state: position, pitch, velocity, pitch rate
  uint8_t* contact;        // [B] next MPC floor-contact flag
  // MPCBalancer.step in front of the step, in the same launch (two-lane kernel, horizon <= 16:
  // upkie_sim_step_base_velocity_mpc): the balancer's constants, its warm start, its velocity state
  int mpc_fused = 0;
  MpcDev mpc{};
  float* mpc_ws = nullptr;
  float* mpc_commanded = nullptr;
};

// ------------------------------------------------------------------ Philox
// Philox4x32-10 (Salmon et al., SC'11): counter = (env id lo/hi, episode,
// stream<<24 | block), key = seed: results do not depend on how envs are sharded.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { STREAM_RESET = 0, STREAM_NOISE = 1, STREAM_INERTIA = 2, STREAM_PUSH = 3 };

template <class ConfigT>
__device__ __forceinline__ void philox_uniform4(const ConfigT& C, unsigned env_local, unsigned episode, unsigned stream,
                                                unsigned block, float (&u)[4]) {
  unsigned lo = C.env_lo + env_local;
  unsigned hi = C.env_hi + (lo < C.env_lo ? 1u : 0u);
  unsigned r[4];
  philox4x32_10(lo, hi, episode, (stream << 24) | block, C.seed_lo, C.seed_hi, r);
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (float)(r[i] >> 8) * (1.0f / 16777216.0f);
}

// Six standard normals for (env, step, slot): Box-Muller on two Philox blocks.
// slot = substep index (control noise) or NOISE_SLOT_MEASUREMENT.
#define NOISE_SLOT_MEASUREMENT 0x7fffu
template <class ConfigT>
__device__ __forceinline__ void philox_normal6(const ConfigT& C, unsigned env_local, unsigned step, unsigned slot, float (&z)[6]) {
  unsigned lo = C.env_lo + env_local;
  unsigned hi = C.env_hi + (lo < C.env_lo ? 1u : 0u);
  unsigned r[8];
#pragma unroll
  for (unsigned k = 0; k < 2; ++k) {
    unsigned q[4];
    philox4x32_10(lo, hi, step, ((unsigned)STREAM_NOISE << 24) | (slot * 2u + k), C.seed_lo, C.seed_hi, q);
    r[4 * k] = q[0]; r[4 * k + 1] = q[1]; r[4 * k + 2] = q[2]; r[4 * k + 3] = q[3];
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    float u1 = ((float)(r[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    float u2 = (float)(r[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
    float radius = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * p] = radius * cs;
    z[2 * p + 1] = radius * sn;
  }
}

__device__ __forceinline__ float uniform(float low, float high, float u) { return fmaf(high - low, u, low); }

// clamp_and_warn (upkie/utils/clamp.py:42-58): NaN passes through.
__device__ __forceinline__ float clamp_ref(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Servo {
  float position, velocity, feedforward_torque, kp_scale, kd_scale, maximum_torque;
};

// moteus-like torque law, pybullet_backend.py:492-553.
__device__ __forceinline__ float joint_torque(float q, float qd, const Servo& c, float kp_gain, float kd_gain, float friction,
                                              float noise) {
  float kp = c.kp_scale * kp_gain;
  float kd = c.kd_scale * kd_gain;
  float torque = c.feedforward_torque;
  torque += kd * (c.velocity - qd);
  if (!isnan(c.position)) torque += kp * (c.position - q);
  if (fabsf(qd) > 1e-3f) torque += qd > 0.f ? -friction : friction;
  torque += noise;  // pybullet_backend.py:545-550
  torque = torque < -c.maximum_torque ? -c.maximum_torque : torque;
  torque = torque > c.maximum_torque ? c.maximum_torque : torque;
  return torque;
}

// ------------------------------------------------------------------ non-finite guard
// (include/upkie_hip.h, "Non-finite commands and states". The reference asserts on a NaN velocity target,
// pybullet_backend.py:519, and has one robot; a batch of thousands cannot stop for one diverged policy output -- and
// without this a NaN velocity or feedforward torque passed both clips of joint_torque, the state went NaN, and
// `fabsf(pitch) > fall_pitch` is false for NaN: never terminated, never reset, NaN observations from then on.)
//  1. Commands: what is still not finite BEHIND the reference's clamp (only NaN is, and an infinite position target of a
//     joint without position limits) becomes the neutral action's value (upkie_servos.py:255-262: velocity 0, feedforward
//     torque 0, gain scales 1, maximum torque = the effort limit, position NaN = no position term).
//  2. State: an env whose state is not finite behind its substeps is put into the configuration's initial state at rest,
//     reports `terminated` and is flagged done: the autoreset treats it like a fall.
// Both are counted (upkie_sim_guard_counts), both cost the sound envs a handful of compares.
__device__ __forceinline__ bool is_finite(float x) { return fabsf(x) < 3.0e38f; }  // (false for NaN)

__device__ __forceinline__ int guard_servo_command(Servo& c, float effort) {
  // (one test for the sound command -- every env but a diverged policy's --, the word-by-word replacement behind it)
  const float magnitude = fabsf(c.velocity) + fabsf(c.feedforward_torque) + fabsf(c.kp_scale) + fabsf(c.kd_scale) + fabsf(c.maximum_torque);
  if (magnitude < 3.0e38f && !(fabsf(c.position) > 3.0e38f)) return 0;
  int replaced = 0;
  if (fabsf(c.position) > 3.0e38f) { c.position = NAN; ++replaced; }  // +-Inf (NaN compares false: it is the neutral value)
  if (!is_finite(c.velocity)) { c.velocity = 0.f; ++replaced; }
  if (!is_finite(c.feedforward_torque)) { c.feedforward_torque = 0.f; ++replaced; }
  if (!is_finite(c.kp_scale)) { c.kp_scale = 1.f; ++replaced; }
  if (!is_finite(c.kd_scale)) { c.kd_scale = 1.f; ++replaced; }
  if (!is_finite(c.maximum_torque)) { c.maximum_torque = effort; ++replaced; }
  return replaced;
}

// ground / yaw velocity actions of the Gyropod family: NaN is the neutral action (0); an infinite yaw velocity would
// integrate into the yaw word unclamped (upkie_gyropod.py:383-385)
__device__ __forceinline__ int guard_velocity_actions(float& a0, float& a1, float max_yaw_velocity) {
  int replaced = 0;
  if (a0 != a0) { a0 = 0.f; ++replaced; }
  if (a1 != a1) { a1 = 0.f; ++replaced; }
  if (fabsf(a1) > 3.0e38f) { a1 = a1 > 0.f ? max_yaw_velocity : -max_yaw_velocity; ++replaced; }
  return replaced;
}

__device__ __forceinline__ void guard_count(unsigned* guard, int which, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (guard && n > 0) atomicAdd(&guard[which], (unsigned)n);
#else
  (void)guard; (void)which; (void)n;
#endif
}

__device__ __forceinline__ void quat_mul(const float (&a)[4], const float (&b)[4], float (&c)[4]) {
  c[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  c[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  c[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  c[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

// Initial-state sampling: RobotState.sample_state draw order
// (robot_state.py:182-187) and _reset_robot_state (pybullet_backend.py:234-267).
template <class ConfigT>
__device__ __forceinline__ void sample_init_state(const ConfigT& C, unsigned env_local, unsigned episode, Phys& s) {
  float u0[4], u1[4], u2[4];
  philox_uniform4(C, env_local, episode, STREAM_RESET, 0, u0);
  philox_uniform4(C, env_local, episode, STREAM_RESET, 1, u1);
  philox_uniform4(C, env_local, episode, STREAM_RESET, 2, u2);
  float wx = uniform(-C.rand_omega_x, C.rand_omega_x, u0[0]);
  float wy = uniform(-C.rand_omega_y, C.rand_omega_y, u0[1]);
  float wz = uniform(0.f, 0.f, u0[2]);
  float vx = uniform(-C.rand_linvel[0], C.rand_linvel[0], u0[3]);
  float vy = uniform(-C.rand_linvel[1], C.rand_linvel[1], u1[0]);
  float vz = uniform(-C.rand_linvel[2], C.rand_linvel[2], u1[1]);
  float yaw = uniform(0.f, 0.f, u1[2]);
  float pitch = uniform(-C.rand_pitch, C.rand_pitch, u1[3]);
  float roll = uniform(-C.rand_roll, C.rand_roll, u2[0]);
  float px = uniform(-C.rand_x, C.rand_x, u2[1]);
  float py = uniform(0.f, 0.f, u2[2]);
  float pz = uniform(0.f, C.rand_z, u2[3]);
  // ScipyRotation.from_euler("ZYX", [yaw, pitch, roll]) = Rz Ry Rx
  float sy, cy, sp, cp, sr, cr;
  sincosf(0.5f * yaw, &sy, &cy);
  sincosf(0.5f * pitch, &sp, &cp);
  sincosf(0.5f * roll, &sr, &cr);
  float qz[4] = {cy, 0.f, 0.f, sy}, qy[4] = {cp, 0.f, sp, 0.f}, qx[4] = {cr, sr, 0.f, 0.f};
  float t[4], qr[4], q0[4] = {C.init_quat[0], C.init_quat[1], C.init_quat[2], C.init_quat[3]}, q[4];
  quat_mul(qz, qy, t);
  quat_mul(t, qx, qr);
  quat_mul(q0, qr, q);  // robot_state.py:158-160
  s.qw = q[0]; s.qx = q[1]; s.qy = q[2]; s.qz = q[3];
  s.pos = v3(C.init_pos[0] + px, C.init_pos[1] + py, C.init_pos[2] + pz);
  s.linvel = v3(C.init_linvel[0] + vx, C.init_linvel[1] + vy, C.init_linvel[2] + vz);
  // body-frame omega handed over as a world-frame vector, pybullet_backend.py:253-258
  s.angvel = v3(C.init_angvel[0] + wx, C.init_angvel[1] + wy, C.init_angvel[2] + wz);
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    s.q[j] = C.init_joint[j];
    s.qd[j] = 0.f;  // resetJointState zeroes velocities, :261-267
  }
}

// Gyropod observation, upkie_gyropod.py:186-214 on top of
// pybullet_backend.py:333-368,476-490.
__device__ __forceinline__ void gyropod_observation(const DevModel& M, const Phys& s, float yaw, float yawvel, float (&obs)[6]) {
  float qw = s.qw, qx = s.qx, qy = s.qy, qz = s.qz;
  float r01 = 2.f * (qx * qy - qz * qw), r11 = 1.f - 2.f * (qx * qx + qz * qz), r21 = 2.f * (qy * qz + qx * qw);
  float x = 2.f * (qw * qy - qz * qx);
  x = fminf(fmaxf(x, -1.f), 1.f);
  float signed_radius = M.left_sign * M.wheel_radius;
  obs[0] = 0.5f * (s.q[2] - s.q[5]) * signed_radius;
  obs[1] = asinf(x);
  obs[2] = yaw;
  obs[3] = 0.5f * (s.qd[2] - s.qd[5]) * signed_radius;
  obs[4] = r01 * s.angvel.x + r11 * s.angvel.y + r21 * s.angvel.z;
  obs[5] = yawvel;
}

// WPS = waves per SIMD the register allocation is capped for: 1 (up to 512
// registers, no spills: lowest latency, small batches) or 2 (256 registers,
// ~90 spilled: +25 % throughput once the batch oversubscribes the chip).
// BULLET_LIKE: contacts by the Bullet-like specification (bullet_like.hpp) on the env's persistent contact manifold
// `manifold` [BL_MANIFOLD_WORDS][B] (upkie_sim_set_contact_manifold) instead of the default one; its own instantiations.
// (WPS = 2 is the register-capped build of very large batches: two wavefronts per SIMD; an experiment may cap it harder,
// tools/build_variant.py -DUPKIE_DENSE_WAVES=3: profiles/r04_dense_waves_per_simd.txt)
#if !defined(UPKIE_DENSE_WAVES)
#define UPKIE_DENSE_WAVES 2
#endif
template <int MODE, bool RAND, int WPS, bool SPINE, bool BULLET_LIKE = false>
__global__ __launch_bounds__(64, WPS == 2 ? UPKIE_DENSE_WAVES : WPS) void step_kernel(const DevModel* __restrict__ Mp, DevLimits Lm, DevConfig C, float* __restrict__ state,
                                                   const float* __restrict__ act, float* __restrict__ obs,
                                                   float* __restrict__ reward, uint8_t* __restrict__ terminated,
                                                   uint8_t* __restrict__ truncated, const uint8_t* __restrict__ mask,
                                                   const float* __restrict__ body_inertials,
                                                   const float* __restrict__ ext_force, int packed, BaseVelocityPtrs bv,
                                                   float* __restrict__ spine_state, float* __restrict__ final_obs,
                                                   float* __restrict__ manifold) {
  warm_kernel_arguments();
  const DevModel& M = *Mp;
  const int B = C.num_envs;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  float* st = state + e;
#define SW(w) st[(size_t)(w) * B]

  // ---- load ----------------------------------------------------------
  Phys s;
  s.pos = v3(SW(UPKIE_S_POS), SW(UPKIE_S_POS + 1), SW(UPKIE_S_POS + 2));
  s.qw = SW(UPKIE_S_QUAT); s.qx = SW(UPKIE_S_QUAT + 1); s.qy = SW(UPKIE_S_QUAT + 2); s.qz = SW(UPKIE_S_QUAT + 3);
  s.linvel = v3(SW(UPKIE_S_LINVEL), SW(UPKIE_S_LINVEL + 1), SW(UPKIE_S_LINVEL + 2));
  s.angvel = v3(SW(UPKIE_S_ANGVEL), SW(UPKIE_S_ANGVEL + 1), SW(UPKIE_S_ANGVEL + 2));
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    s.q[j] = SW(UPKIE_S_Q + j);
    s.qd[j] = SW(UPKIE_S_QD + j);
  }
  float legref[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) legref[l] = SW(UPKIE_S_LEGREF + l);
  constexpr bool YAWING = MODE == MODE_GYROPOD || MODE == MODE_BASE_VELOCITY;
  float yaw = 0.f, yawvel = 0.f;
  if (YAWING) {
    yaw = SW(UPKIE_S_YAW);
    yawvel = SW(UPKIE_S_YAWVEL);
  }
  BodyInertials inertials;
  if (RAND) {
    if (body_inertials) {
      load_body_inertials(body_inertials + e, (size_t)B, inertials);
    } else {
      body_inertials_of_model(M, inertials);
    }
  }
  const ExtForces ext{RAND && ext_force ? ext_force + e : nullptr, (size_t)B, &C.ext};
  // With the state, in ONE memory round trip: the DONE word and this step's
  // action (or, for the fused agent, the previous observation). Loaded inside
  // the branches that use them they would each cost a dependent round trip
  // (0.5-1 us at one wave per SIMD).
  float done_word = MODE != MODE_RESET ? SW(UPKIE_S_DONE) : 0.f;
  asm volatile("" : "+v"(done_word));  // pins the load here: the compiler would sink it into the branch that tests it
  float act0 = 0.f, act1 = 0.f;
  float4 prev_obs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == MODE_PENDULUM) {
    if (act) act0 = act[e];  // (no action buffer in the SAME_STEP reset pass)
  } else if (fused_agent(MODE)) {
    // previous observation: from `act` when the caller double-buffers its records
    const float* prev = act ? act : obs;
    prev_obs = reinterpret_cast<const float4*>(prev)[packed ? 2 * (size_t)e : (size_t)e];
  } else if (MODE == MODE_GYROPOD) {
    if (act) {
      const float2 a = reinterpret_cast<const float2*>(act)[e];
      act0 = a.x;
      act1 = a.y;
    }
  } else if (MODE == MODE_BASE_VELOCITY) {
    act0 = bv.commanded[e];  // MPCBalancer output, upkie_base_velocity.py:185-192
    act1 = act[2 * (size_t)e + 1];
  }

  bool do_reset;
  if (MODE == MODE_RESET) {
    do_reset = mask ? mask[e] != 0 : true;
  } else {
    do_reset = (C.autoreset_mode == UPKIE_AUTORESET_NEXT_STEP || C.autoreset_mode == AUTORESET_DONE_PASS) && done_word != 0.f;
    if (C.autoreset_mode == AUTORESET_DONE_PASS) {
      // SAME_STEP autoreset, second launch: the step has just written this env's
      // terminal observation; keep it aside, then run the reset branch
      if (final_obs) {  // every env: final_obs is the step's observation, obs differs from it where an episode ended
        constexpr int W = ObsWords<MODE>::value;
        const float* last = obs + (size_t)(packed ? 8 : W) * e;
#pragma unroll
        for (int i = 0; i < W; ++i) final_obs[(size_t)W * e + i] = last[i];
      }
      if (!do_reset) return;
    }
  }

  if (MODE == MODE_RESET && !do_reset) {
    // untouched env: only report its current observation
    if (obs) {
      float o6[6];
      gyropod_observation(M, s, SW(UPKIE_S_YAW), SW(UPKIE_S_YAWVEL), o6);
#pragma unroll
      for (int i = 0; i < 6; ++i) obs[(size_t)6 * e + i] = o6[i];
    }
    return;
  }

  // ---- action map ------------------------------------------------------
  Servo cmd[UPKIE_NJ];
  float a0 = 0.f, a1 = 0.f;
  unsigned episode = 0;
  if (do_reset) {
    episode = (unsigned)SW(UPKIE_S_EPISODE);
    sample_init_state(C, (unsigned)e, episode, s);
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) cmd[j] = Servo{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // no motor torque, :228
  } else if (MODE == MODE_SERVOS) {
    // UpkieServos.get_spine_action, upkie_servos.py:316-344
    const float* a = act + (size_t)36 * e;
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) {
      float eff = M.joint_effort[j], vel = M.joint_velocity[j];
      cmd[j].position = clamp_ref(a[6 * j + 0], M.joint_lower[j], M.joint_upper[j]);
      cmd[j].velocity = clamp_ref(a[6 * j + 1], -vel, vel);
      cmd[j].feedforward_torque = clamp_ref(a[6 * j + 2], -eff, eff);
      cmd[j].kp_scale = clamp_ref(a[6 * j + 3], 0.f, C.max_gain_scale);
      cmd[j].kd_scale = clamp_ref(a[6 * j + 4], 0.f, C.max_gain_scale);
      cmd[j].maximum_torque = clamp_ref(a[6 * j + 5], 0.f, eff);
      if (const int replaced = guard_servo_command(cmd[j], eff)) guard_count(C.guard, 0, replaced);
    }
  } else if (MODE != MODE_RESET) {
    if (fused_agent(MODE)) {
      // README.md:62-64: action = gains . observation, clipped
      const float4 o = prev_obs;
      a0 = C.agent_gains[0] * o.x + C.agent_gains[1] * o.y + C.agent_gains[2] * o.z + C.agent_gains[3] * o.w;
      a0 = clamp_ref(a0, -C.agent_clip, C.agent_clip);
    } else {
      a0 = act0;  // Pendulum: [action[0], 0.0], upkie_pendulum.py:139
      a1 = act1;
    }
    if (const int replaced = guard_velocity_actions(a0, a1, C.max_yaw_velocity)) guard_count(C.guard, 0, replaced);
    // UpkieGyropod.__get_spine_action, upkie_gyropod.py:293-331
    float v = clamp_ref(a0, -C.max_ground_velocity, C.max_ground_velocity);
    float yawd = clamp_ref(a1, -C.max_yaw_velocity, C.max_yaw_velocity);
    float inv_radius = fast_rcp(M.wheel_radius);
    float wheel_velocity = v * inv_radius;
    float left = M.left_sign * wheel_velocity, right = -M.left_sign * wheel_velocity;
    float yaw_to_wheel = M.left_sign * (0.5f * M.wheel_base) * inv_radius;
    left = fmaf(yaw_to_wheel, yawd, left);
    right = fmaf(yaw_to_wheel, yawd, right);
    const float alpha = C.dt / 1.0f;  // filters.py:77, cutoff_period = 1 s
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int j = (l < 2) ? l : l + 1;  // lh, lk, rh, rk
      legref[l] = legref[l] + alpha * (0.f - legref[l]);
      // then UpkieServos clamps every field, upkie_servos.py:331-342
      cmd[j].position = clamp_ref(legref[l], M.joint_lower[j], M.joint_upper[j]);
      cmd[j].velocity = 0.f;
      cmd[j].feedforward_torque = 0.f;
      cmd[j].kp_scale = clamp_ref(C.leg_gain_scale, 0.f, C.max_gain_scale);
      cmd[j].kd_scale = cmd[j].kp_scale;
      cmd[j].maximum_torque = M.joint_effort[j];
    }
#pragma unroll
    for (int wi = 0; wi < 2; ++wi) {
      const int j = 3 * wi + 2;
      cmd[j].position = NAN;
      cmd[j].velocity = clamp_ref(wi == 0 ? left : right, -M.joint_velocity[j], M.joint_velocity[j]);
      cmd[j].feedforward_torque = 0.f;
      cmd[j].kp_scale = 1.f;
      cmd[j].kd_scale = 1.f;
      cmd[j].maximum_torque = M.joint_effort[j];
    }
  }

  // the env's persistent contact manifold (Bullet-like contact model): held privately for the step's substeps; a
  // reset drops the contact cache (Bullet: resetBasePositionAndOrientation) before its one torque-free substep
  float contact_manifold[BULLET_LIKE ? BL_MANIFOLD_WORDS : 1];
  if constexpr (BULLET_LIKE) {
    for (int w = 0; w < BL_MANIFOLD_WORDS; ++w) contact_manifold[w] = do_reset ? 0.f : manifold[(size_t)w * B + e];
  }

  // ---- PyBulletBackend.step: substeps of {torques; stepSimulation} -------
  float tau[UPKIE_NJ] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool contact = false;
  const bool any_noise = C.any_control_noise || C.any_measurement_noise;
  unsigned step_count = any_noise ? (unsigned)SW(UPKIE_S_STEP) : 0u;
  const int nsub = do_reset ? 1 : C.nb_substeps;
  SweepWarmStart sweep_warm_start;  // spans the substeps of this env.step() (dynamics.hpp)
  sweep_warm_start.swept = 0;
  for (int sub = 0; sub < C.nb_substeps; ++sub) {
    if (sub >= nsub) break;
    float zn[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // control noise: one draw per joint per substep (uniform branch)
    if (C.any_control_noise && !do_reset) philox_normal6(C, (unsigned)e, step_count, (unsigned)sub, zn);
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j)
      tau[j] = joint_torque(s.q[j], s.qd[j], cmd[j], C.kp, C.kd, C.joint_friction[j], C.control_noise[j] * zn[j]);
    {
      // Re-derive the model pointer every substep: the ~150 model constants are
      // then re-fetched by scalar loads when needed instead of being hoisted
      // out of the loop and spilled to VGPR lanes.
      // (constant address space: uniform loads become s_load, not flat vector loads)
      typedef const __attribute__((address_space(4))) DevModel* ConstModelPtr;
      ConstModelPtr mp = (ConstModelPtr)Mp;
      asm volatile("" : "+s"(mp));
      // PyBulletBackend.reset steps once WITHOUT __apply_external_forces (pybullet_backend.py:220-232 vs :303)
      const ExtForces ext_now{do_reset ? nullptr : ext.force, ext.stride, ext.slots};
      if constexpr (BULLET_LIKE) {
        contact = physics_substep<(WPS > 1), true>(*mp, Lm, s, tau, C.h, RAND ? &inertials : nullptr, ext_now, nullptr, &contact_manifold);
      } else {
        contact = physics_substep<(WPS > 1)>(*mp, Lm, s, tau, C.h, RAND ? &inertials : nullptr, ext_now, nullptr, nullptr, &sweep_warm_start);
      }
    }
    if (SPINE) {
      // one cycle of the spine's observer pipeline (spines/common/observers.h:22-42): it sees the
      // torques commanded for this cycle and the joint velocities the simulator reports after it
#define OM(w) spine_state[(size_t)(w) * B + e]
      bool any_wheel = false, wc[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        float fv = OM(UPKIE_O_WHEEL + 5 * w), aa = OM(UPKIE_O_WHEEL + 5 * w + 1), at = OM(UPKIE_O_WHEEL + 5 * w + 2),
              in = OM(UPKIE_O_WHEEL + 5 * w + 3);
        bool ct = OM(UPKIE_O_WHEEL + 5 * w + 4) != 0.f;
        if (do_reset) { fv = aa = at = in = 0.f; ct = false; }
        wheel_contact_observe(C.spine, tau[3 * w + 2], s.qd[3 * w + 2], fv, aa, at, in, ct);
        OM(UPKIE_O_WHEEL + 5 * w) = fv; OM(UPKIE_O_WHEEL + 5 * w + 1) = aa; OM(UPKIE_O_WHEEL + 5 * w + 2) = at;
        OM(UPKIE_O_WHEEL + 5 * w + 3) = in; OM(UPKIE_O_WHEEL + 5 * w + 4) = ct ? 1.f : 0.f;
        wc[w] = ct;
        any_wheel = any_wheel || ct;
      }
      const float sq = tau[0] * tau[0] + tau[1] * tau[1] + tau[3] * tau[3] + tau[4] * tau[4];
      const float upper = obs_low_pass(do_reset ? 0.f : OM(UPKIE_O_UPPER_LEG_TORQUE), C.spine.leg_alpha, sqrtf(sq));
      OM(UPKIE_O_UPPER_LEG_TORQUE) = upper;
      const bool fc = any_wheel || upper > C.spine.upper_leg_torque_threshold;
      OM(UPKIE_O_CONTACT) = fc ? 1.f : 0.f;
      float op = do_reset ? 0.f : OM(UPKIE_O_ODOMETRY_POSITION), ov = do_reset ? 0.f : OM(UPKIE_O_ODOMETRY_VELOCITY);
      if (fc) {
        float sum = 0.f, n = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w)
          if (wc[w]) {
            sum += C.spine.signed_radius[w] * s.qd[3 * w + 2];
            n += 1.f;
          }
        ov = n > 0.f ? sum / n : 0.f;
        op += ov * C.h;
      }
      OM(UPKIE_O_ODOMETRY_POSITION) = op;
      OM(UPKIE_O_ODOMETRY_VELOCITY) = ov;
#undef OM
    }
  }

  // ---- non-finite guard: the state behind the substeps ---------------------
  bool unsound;
  {
    float mag = fabsf(s.pos.x) + fabsf(s.pos.y) + fabsf(s.pos.z) + fabsf(s.qw) + fabsf(s.qx) + fabsf(s.qy) + fabsf(s.qz);
    mag += fabsf(s.linvel.x) + fabsf(s.linvel.y) + fabsf(s.linvel.z) + fabsf(s.angvel.x) + fabsf(s.angvel.y) + fabsf(s.angvel.z);
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) mag += fabsf(s.q[j]) + fabsf(s.qd[j]);
    if (YAWING) mag += fabsf(yaw);
    unsound = !(mag < 3.0e38f);
  }
  if (unsound) {
    s.pos = v3(C.init_pos[0], C.init_pos[1], C.init_pos[2]);
    s.qw = C.init_quat[0]; s.qx = C.init_quat[1]; s.qy = C.init_quat[2]; s.qz = C.init_quat[3];
    s.linvel = v3(C.init_linvel[0], C.init_linvel[1], C.init_linvel[2]);
    s.angvel = v3(C.init_angvel[0], C.init_angvel[1], C.init_angvel[2]);
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) {
      s.q[j] = C.init_joint[j];
      s.qd[j] = 0.f;
      tau[j] = 0.f;
    }
    legref[0] = s.q[0]; legref[1] = s.q[1]; legref[2] = s.q[3]; legref[3] = s.q[4];
    yaw = 0.f;
    a1 = 0.f;
    contact = false;
    if constexpr (BULLET_LIKE) {
      for (int w = 0; w < BL_MANIFOLD_WORDS; ++w) contact_manifold[w] = 0.f;
    }
    guard_count(C.guard, 1, 1);
  }

  // ---- wrapper post-processing -----------------------------------------
  bool fallen = false, timeout = false;
  float obs6[6];
  if (do_reset) {
    // upkie_gyropod.py:236-240
    legref[0] = s.q[0]; legref[1] = s.q[1]; legref[2] = s.q[3]; legref[3] = s.q[4];
    yaw = 0.f;
    yawvel = 0.f;
    SW(UPKIE_S_YAW) = 0.f;
    SW(UPKIE_S_YAWVEL) = 0.f;
    SW(UPKIE_S_MPC_V) = 0.f;
    SW(UPKIE_S_SE2_X) = 0.f;
    SW(UPKIE_S_SE2_Y) = 0.f;
    SW(UPKIE_S_EPISODE) = (float)((episode + 1u) & UPKIE_COUNTER_MASK);
    SW(UPKIE_S_DONE) = 0.f;
    SW(UPKIE_S_ELAPSED) = 0.f;
    gyropod_observation(M, s, yaw, yawvel, obs6);
  } else {
    if (YAWING) {
      yaw = fmaf(a1, C.dt, yaw);  // upkie_gyropod.py:383-385 (unclamped action)
      yawvel = a1;
      SW(UPKIE_S_YAW) = yaw;
      SW(UPKIE_S_YAWVEL) = yawvel;
    }
    gyropod_observation(M, s, yaw, yawvel, obs6);
    fallen = unsound;  // (every env kind: the non-finite guard ends the episode)
    if (MODE != MODE_SERVOS) fallen = fallen || fabsf(obs6[1]) > C.fall_pitch;  // upkie_gyropod.py:344-345
    if (fallen) SW(UPKIE_S_DONE) = 1.f;
    if (C.max_episode_steps > 0) {
      // gymnasium's TimeLimit: the step that brings the episode to the limit is truncated unless it fell
      const float elapsed = SW(UPKIE_S_ELAPSED) + 1.f;
      SW(UPKIE_S_ELAPSED) = elapsed;
      timeout = elapsed >= (float)C.max_episode_steps && !fallen;
      if (timeout) SW(UPKIE_S_DONE) = 1.f;
    }
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) SW(UPKIE_S_TORQUE + j) = tau[j];  // pybullet_backend.py:293
    if (any_noise) {
      step_count = (step_count + 1u) & UPKIE_COUNTER_MASK;
      SW(UPKIE_S_STEP) = (float)step_count;
    }
  }

  // ---- store -------------------------------------------------------------
  if constexpr (BULLET_LIKE) {
    for (int w = 0; w < BL_MANIFOLD_WORDS; ++w) manifold[(size_t)w * B + e] = contact_manifold[w];
  }
  SW(UPKIE_S_POS) = s.pos.x; SW(UPKIE_S_POS + 1) = s.pos.y; SW(UPKIE_S_POS + 2) = s.pos.z;
  SW(UPKIE_S_QUAT) = s.qw; SW(UPKIE_S_QUAT + 1) = s.qx; SW(UPKIE_S_QUAT + 2) = s.qy; SW(UPKIE_S_QUAT + 3) = s.qz;
  SW(UPKIE_S_LINVEL) = s.linvel.x; SW(UPKIE_S_LINVEL + 1) = s.linvel.y; SW(UPKIE_S_LINVEL + 2) = s.linvel.z;
  SW(UPKIE_S_ANGVEL) = s.angvel.x; SW(UPKIE_S_ANGVEL + 1) = s.angvel.y; SW(UPKIE_S_ANGVEL + 2) = s.angvel.z;
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    SW(UPKIE_S_Q + j) = s.q[j];
    SW(UPKIE_S_QD + j) = s.qd[j];
  }
  if (MODE != MODE_SERVOS) {
#pragma unroll
    for (int l = 0; l < 4; ++l) SW(UPKIE_S_LEGREF + l) = legref[l];
  }
  SW(UPKIE_S_CONTACT) = contact ? 1.f : 0.f;

  if (MODE == MODE_RESET) {
    if (obs) {
#pragma unroll
      for (int i = 0; i < 6; ++i) obs[(size_t)6 * e + i] = obs6[i];
    }
    return;
  }
  if (MODE == MODE_PENDULUM || fused_agent(MODE)) {
    // _PENDULUM_OBS_INDICES = [1, 0, 4, 3], upkie_pendulum.py:17
    const float4 o4 = make_float4(obs6[1], obs6[0], obs6[4], obs6[3]);
    if (packed) {
      // one 32-byte record per env for the rollout gather:
      // [obs(4) | reward, terminated, truncated, 0]
      float4* rec = reinterpret_cast<float4*>(obs) + 2 * (size_t)e;
      rec[0] = o4;
      if (C.autoreset_mode != AUTORESET_DONE_PASS) rec[1] = make_float4(0.f, fallen ? 1.f : 0.f, timeout ? 1.f : 0.f, 0.f);
      return;
    }
    reinterpret_cast<float4*>(obs)[e] = o4;
  } else if (MODE == MODE_BASE_VELOCITY) {
    // dead reckoning with the TARGET linear velocity and the new yaw, :197-199
    float x = 0.f, y = 0.f;
    if (!do_reset) {
      float lin = act[2 * (size_t)e];
      if (!is_finite(lin)) lin = 0.f;  // (non-finite guard: the target velocity the pose integrates)
      float sy, cy;
      sincosf(yaw, &sy, &cy);
      x = fmaf(lin * cy, C.dt, SW(UPKIE_S_SE2_X));
      y = fmaf(lin * sy, C.dt, SW(UPKIE_S_SE2_Y));
      SW(UPKIE_S_SE2_X) = x;
      SW(UPKIE_S_SE2_Y) = y;
    }
    obs[(size_t)3 * e] = x;
    obs[(size_t)3 * e + 1] = y;
    obs[(size_t)3 * e + 2] = yaw;
    // what MPCBalancer.step reads from the spine observation next time, :253-273
    reinterpret_cast<float4*>(bv.x0)[e] = make_float4(obs6[0], obs6[1], obs6[3], obs6[4]);
    bv.contact[e] = contact ? 1 : 0;
  } else if (MODE == MODE_GYROPOD) {
    float2* o2 = reinterpret_cast<float2*>(obs) + (size_t)3 * e;
    o2[0] = make_float2(obs6[0], obs6[1]);
    o2[1] = make_float2(obs6[2], obs6[3]);
    o2[2] = make_float2(obs6[4], obs6[5]);
  } else {
    // upkie_servos.py:288-306 / pybullet_backend.py:448-474
    float* o = obs + (size_t)30 * e;
    float zm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // measurement noise: one draw per joint per observation, :461-466
    if (C.any_measurement_noise) philox_normal6(C, (unsigned)e, step_count, NOISE_SLOT_MEASUREMENT, zm);
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) {
      o[5 * j + 0] = s.q[j];
      o[5 * j + 1] = s.qd[j];
      o[5 * j + 2] = (do_reset ? SW(UPKIE_S_TORQUE + j) : tau[j]) + C.measurement_noise[j] * zm[j];
      o[5 * j + 3] = 42.0f;
      o[5 * j + 4] = 18.0f;
    }
  }
  if (C.autoreset_mode == AUTORESET_DONE_PASS) return;  // reward and flags are those of the terminal step
  reward[e] = 0.f;  // upkie_env.py:230
  terminated[e] = fallen ? 1 : 0;
  truncated[e] = timeout ? 1 : 0;
#undef SW
}

// The servo-level policy of upkie_sim_step_servos_policy reaches the eight-lane Servos kernels BY VALUE, as a kernel
// argument (268 bytes; the other modes carry an empty struct): no device copy to keep coherent with the host's, nothing
// to upload under a hipGraph capture, no race between a policy change on one stream and a step still running on another.
struct NoServoPolicy {};
template <int MODE>
using ServoPolicyArg = std::conditional_t<MODE == MODE_SERVOS, UpkieServoPolicy, NoServoPolicy>;

#include "state_words.hpp"
#include "pair.hpp"
#include "octet.hpp"

// which eight-lane instantiations exist beside <MODE, RAND, false, false> (step_instances.hpp lists them, launch_step
// dispatches on exactly these)
constexpr bool octet_has_default_scalars(int mode) { return mode == MODE_PENDULUM || fused_agent(mode) || mode == MODE_GYROPOD; }
constexpr bool octet_resets_in_place(int mode) { return mode == MODE_PENDULUM || mode == MODE_GYROPOD || mode == MODE_SERVOS; }

}  // namespace upkie

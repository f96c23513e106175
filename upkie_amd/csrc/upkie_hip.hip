// upkie_hip.hip -- fused env.step() kernels for gfx950 and the C-ABI of
// include/upkie_hip.h.
//
// One launch = one env.step() of B environments: each lane loads its env's
// state words (struct-of-arrays, coalesced), runs the action map, the
// nb_substeps x {6 servo torques -> physics substep} loop entirely in
// registers, and writes state + observation + flags back. Model and config
// constants are kernel arguments (scalar loads, SGPR-resident).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>

#include "rollout.hpp"

#include "step_kernels.hpp"
#include "step_instances.hpp"  // (the step kernels this unit launches are compiled elsewhere, by groups: declarations only)
#include "host_setup.hpp"

namespace upkie {

// Full spine observation, pybullet_backend.py:313-490.
struct ObsPtrs {
  float* pitch;
  float* angular_velocity;
  float* linear_velocity;
  float* rotation_base_to_world;
  uint8_t* floor_contact;
  float* imu_orientation;
  float* imu_angular_velocity;
  float* imu_linear_acceleration;
  float* imu_raw_linear_acceleration;
  float* servo;
  float* wheel_odometry;
};

__device__ __forceinline__ void mat3_mul(const float (&A)[9], const float (&Bm)[9], float (&Cm)[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Cm[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
}

__global__ __launch_bounds__(64) void observe_kernel(DevModel M, DevConfig C, float* __restrict__ state, ObsPtrs out,
                                                      int update_imu) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 30];
  const int B = C.num_envs;
  const int e0 = blockIdx.x * blockDim.x;
  const int n_valid = min(64, B - e0);
  // lanes past the batch compute on the last env (they take part in the LDS staging, their rows are not stored)
  const bool live = (int)threadIdx.x < n_valid;
  const int e = live ? e0 + (int)threadIdx.x : B - 1;
  float* st = state + e;
#define SW(w) st[(size_t)(w) * B]
  float qw = SW(UPKIE_S_QUAT), qx = SW(UPKIE_S_QUAT + 1), qy = SW(UPKIE_S_QUAT + 2), qz = SW(UPKIE_S_QUAT + 3);
  float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qz * qw), 2.f * (qw * qy + qx * qz),
                2.f * (qx * qy + qz * qw), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qx * qw),
                2.f * (qx * qz - qy * qw), 2.f * (qy * qz + qx * qw), 1.f - 2.f * (qx * qx + qy * qy)};
  float v[3] = {SW(UPKIE_S_LINVEL), SW(UPKIE_S_LINVEL + 1), SW(UPKIE_S_LINVEL + 2)};
  float w[3] = {SW(UPKIE_S_ANGVEL), SW(UPKIE_S_ANGVEL + 1), SW(UPKIE_S_ANGVEL + 2)};
  float wb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) wb[i] = R[i] * w[0] + R[3 + i] * w[1] + R[6 + i] * w[2];
  if (out.pitch && live) {
    float x = fminf(fmaxf(2.f * (qw * qy - qz * qx), -1.f), 1.f);
    out.pitch[e] = asinf(x);
  }
  if (out.angular_velocity) wave_store_rows(out.angular_velocity, e0, n_valid, wb, lds);
  if (out.linear_velocity) wave_store_rows(out.linear_velocity, e0, n_valid, v, lds);
  if (out.rotation_base_to_world) wave_store_rows(out.rotation_base_to_world, e0, n_valid, R, lds);
  if (out.floor_contact && live) out.floor_contact[e] = SW(UPKIE_S_CONTACT) != 0.f ? 1 : 0;
  {
    // IMU block, pybullet_backend.py:370-430
    float Rbi_t[9], Riw[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rbi_t[3 * i + j] = M.rot_base_to_imu[3 * j + i];
    mat3_mul(R, Rbi_t, Riw);
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = R[3 * i] * M.imu_pos[0] + R[3 * i + 1] * M.imu_pos[1] + R[3 * i + 2] * M.imu_pos[2];
    float v_imu[3] = {v[0] + w[1] * r[2] - w[2] * r[1], v[1] + w[2] * r[0] - w[0] * r[2], v[2] + w[0] * r[1] - w[1] * r[0]};
    // rotation_world_to_ars = diag(1, -1, -1), :385
    float m[9] = {Riw[0], Riw[1], Riw[2], -Riw[3], -Riw[4], -Riw[5], -Riw[6], -Riw[7], -Riw[8]};
    // scipy Rotation.from_matrix -> quaternion (rotations.py:16-33)
    float dec[4] = {m[0], m[4], m[8], m[0] + m[4] + m[8]};
    int choice = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (dec[i] > dec[choice]) choice = i;
    float q[4];  // x y z w
    if (choice == 3) {
      q[0] = m[7] - m[5]; q[1] = m[2] - m[6]; q[2] = m[3] - m[1]; q[3] = 1.f + dec[3];
    } else if (choice == 0) {
      q[0] = 1.f - dec[3] + 2.f * m[0]; q[1] = m[3] + m[1]; q[2] = m[6] + m[2]; q[3] = m[7] - m[5];
    } else if (choice == 1) {
      q[1] = 1.f - dec[3] + 2.f * m[4]; q[2] = m[7] + m[5]; q[0] = m[1] + m[3]; q[3] = m[2] - m[6];
    } else {
      q[2] = 1.f - dec[3] + 2.f * m[8]; q[0] = m[2] + m[6]; q[1] = m[5] + m[7]; q[3] = m[3] - m[1];
    }
    float qn = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float a_w[3], a_i[3], p_i[3], w_i[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) a_w[d] = (v_imu[d] - SW(UPKIE_S_IMUVEL + d)) / C.dt;
    if (update_imu && live) {
#pragma unroll
      for (int d = 0; d < 3; ++d) SW(UPKIE_S_IMUVEL + d) = v_imu[d];
    }
    float pw[3] = {a_w[0], a_w[1], a_w[2] + 9.81f};  // :418
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      w_i[i] = Riw[i] * w[0] + Riw[3 + i] * w[1] + Riw[6 + i] * w[2];
      a_i[i] = Riw[i] * a_w[0] + Riw[3 + i] * a_w[1] + Riw[6 + i] * a_w[2];
      p_i[i] = Riw[i] * pw[0] + Riw[3 + i] * pw[1] + Riw[6 + i] * pw[2];
    }
    if (out.imu_orientation && live)  // 16 bytes per lane: already one contiguous run per wave
      reinterpret_cast<float4*>(out.imu_orientation)[e] = make_float4(q[3] * qn, q[0] * qn, q[1] * qn, q[2] * qn);
    if (out.imu_angular_velocity) wave_store_rows(out.imu_angular_velocity, e0, n_valid, w_i, lds);
    if (out.imu_linear_acceleration) wave_store_rows(out.imu_linear_acceleration, e0, n_valid, a_i, lds);
    if (out.imu_raw_linear_acceleration) wave_store_rows(out.imu_raw_linear_acceleration, e0, n_valid, p_i, lds);
  }
  float ql = SW(UPKIE_S_Q + 2), qr = SW(UPKIE_S_Q + 5), qdl = SW(UPKIE_S_QD + 2), qdr = SW(UPKIE_S_QD + 5);
  if (out.servo) {
    float zm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (C.any_measurement_noise) philox_normal6(C, (unsigned)e, (unsigned)SW(UPKIE_S_STEP), NOISE_SLOT_MEASUREMENT, zm);
    float o[30];
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) {
      o[5 * j + 0] = SW(UPKIE_S_Q + j);
      o[5 * j + 1] = SW(UPKIE_S_QD + j);
      o[5 * j + 2] = SW(UPKIE_S_TORQUE + j) + C.measurement_noise[j] * zm[j];
      o[5 * j + 3] = 42.0f;
      o[5 * j + 4] = 18.0f;
    }
    wave_store_rows(out.servo, e0, n_valid, o, lds);
  }
  if (out.wheel_odometry && live) {
    float sr = M.left_sign * M.wheel_radius;
    reinterpret_cast<float2*>(out.wheel_odometry)[e] = make_float2(0.5f * (ql - qr) * sr, 0.5f * (qdl - qdr) * sr);
  }
#undef SW
}

// PyBulletBackend.get_contact_points (pybullet_backend.py:660-716) for every env:
// the contact solve of one substep from the current state under the last
// commanded torques, run on a register copy (the state is not written).
// out [B][2][8] = per tire {exists, position in world (3), force in world (3), 0}.
// Query path, not the step path: one env per lane, any batch size.
// BULLET_LIKE: the handle's steps run the Bullet-like contact model (upkie_sim_set_contact_manifold): the query solves
// the same model, on a copy of the env's manifold.
template <bool BULLET_LIKE>
__global__ __launch_bounds__(64) void contact_points_kernel(const DevModel* __restrict__ Mp, DevLimits Lm, DevConfig C,
                                                           const float* __restrict__ state, const float* __restrict__ body_inertials,
                                                           const float* __restrict__ ext_force, float* __restrict__ out,
                                                           const float* __restrict__ manifold) {
  const int B = C.num_envs;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  const float* st = state + e;
#define SW(w) st[(size_t)(w) * B]
  Phys s;
  s.pos = v3(SW(UPKIE_S_POS), SW(UPKIE_S_POS + 1), SW(UPKIE_S_POS + 2));
  s.qw = SW(UPKIE_S_QUAT); s.qx = SW(UPKIE_S_QUAT + 1); s.qy = SW(UPKIE_S_QUAT + 2); s.qz = SW(UPKIE_S_QUAT + 3);
  s.linvel = v3(SW(UPKIE_S_LINVEL), SW(UPKIE_S_LINVEL + 1), SW(UPKIE_S_LINVEL + 2));
  s.angvel = v3(SW(UPKIE_S_ANGVEL), SW(UPKIE_S_ANGVEL + 1), SW(UPKIE_S_ANGVEL + 2));
  float tau[UPKIE_NJ];
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    s.q[j] = SW(UPKIE_S_Q + j);
    s.qd[j] = SW(UPKIE_S_QD + j);
    tau[j] = SW(UPKIE_S_TORQUE + j);
  }
#undef SW
  BodyInertials inertials;
  if (body_inertials) load_body_inertials(body_inertials + e, (size_t)B, inertials);
  const ExtForces ext{ext_force ? ext_force + e : nullptr, (size_t)B, &C.ext};
  const float qw = s.qw, qx = s.qx, qy = s.qy, qz = s.qz;
  const V3 origin = s.pos;
  ContactReport rep;
  rep.active[0] = rep.active[1] = false;
  if constexpr (BULLET_LIKE) {
    float mf[BL_MANIFOLD_WORDS];
    for (int w = 0; w < BL_MANIFOLD_WORDS; ++w) mf[w] = manifold[(size_t)w * B + e];
    physics_substep<true, true>(*Mp, Lm, s, tau, C.h, body_inertials ? &inertials : nullptr, ext, &rep, &mf);
  } else {
    physics_substep<true>(*Mp, Lm, s, tau, C.h, body_inertials ? &inertials : nullptr, ext, &rep);
  }
  const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qz * qw), 2.f * (qw * qy + qx * qz),
                      2.f * (qx * qy + qz * qw), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qx * qw),
                      2.f * (qx * qz - qy * qw), 2.f * (qy * qz + qx * qw), 1.f - 2.f * (qx * qx + qy * qy)};
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (rep.active[w]) {
      const V3 P = rep.point[w], F = rep.force[w];
      a = make_float4(1.f, origin.x + R[0] * P.x + R[1] * P.y + R[2] * P.z, origin.y + R[3] * P.x + R[4] * P.y + R[5] * P.z,
                      origin.z + R[6] * P.x + R[7] * P.y + R[8] * P.z);
      b = make_float4(R[0] * F.x + R[1] * F.y + R[2] * F.z, R[3] * F.x + R[4] * F.y + R[5] * F.z, R[6] * F.x + R[7] * F.y + R[8] * F.z, 0.f);
    }
    reinterpret_cast<float4*>(out)[(size_t)4 * e + 2 * w] = a;
    reinterpret_cast<float4*>(out)[(size_t)4 * e + 2 * w + 1] = b;
  }
}

// PyBulletBackend.randomize_inertias (pybullet_backend.py:571-601) for env e:
// epsilon ~ U(-v, v) per link scales that link's mass and inertia; the links
// of each composite body are fused again into records[10 * body + word][env].
__global__ __launch_bounds__(64) void body_inertials_kernel(DevConfig C, DevLinks L, float* __restrict__ records,
                                                            float* __restrict__ link_scale, float variation) {
  const int B = C.num_envs;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  float f[UPKIE_MAX_LINKS];
#pragma unroll
  for (int blk = 0; blk < UPKIE_MAX_LINKS / 4; ++blk) {
    float u[4];
    philox_uniform4(C, (unsigned)e, 0u, STREAM_INERTIA, blk, u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = 4 * blk + i;
      f[l] = (l < L.count && L.randomized[l]) ? 1.f + uniform(-variation, variation, u[i]) : 1.f;
      if (link_scale) link_scale[(size_t)l * B + e] = f[l];
    }
  }
  fuse_links(L, f, records + e, (size_t)B);
}

// Push domain randomisation (BASELINE.json configs[4]): a world-frame force on the trunk per env,
// norm ~ U(0, max_norm), uniformly random horizontal direction; push number `push_index` of env e is one Philox block
// keyed by (seed, global env id, push_index): the same whatever the sharding. force[3][B] is what
// upkie_sim_set_external_forces reads (pybullet_backend.py:603-658 semantics: held until overwritten).
__global__ __launch_bounds__(64) void push_kernel(DevConfig C, float* __restrict__ force, unsigned push_index, float max_norm) {
  const int B = C.num_envs;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  float u[4];
  philox_uniform4(C, (unsigned)e, push_index, STREAM_PUSH, 0u, u);
  const float norm = max_norm * u[0];
  float sn, cs;
  sincosf(6.283185307179586f * u[1], &sn, &cs);
  force[e] = norm * cs;
  force[(size_t)B + e] = norm * sn;
  force[(size_t)2 * B + e] = 0.f;
}

// The projected Gauss-Seidel sweeps of the step kernels (contact_pgs6: the code every lane mapping runs when a
// contact solution leaves its friction cone) on caller-provided systems, one system per lane: A [n][21] (6 x 6, packed
// lower by rows, CFM on the diagonal), rhs [n][6], lam [n][6] in: warm start, out: impulses; sweeps [n] out (may be null).
__global__ __launch_bounds__(64) void contact_sweeps_kernel(const DevModel* __restrict__ Mp, int n, const float* __restrict__ A,
                                                           const float* __restrict__ rhs, float* __restrict__ lam,
                                                           const uint8_t* __restrict__ pair, int* __restrict__ sweeps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a[21], r[6], l[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) a[k] = A[(size_t)21 * i + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    r[k] = rhs[(size_t)6 * i + k];
    l[k] = lam[(size_t)6 * i + k];
  }
  (void)pair;  // (one loop for every system since round 4: a lifted tire's identity rows need no other code, contact_pgs6)
  const int count = contact_pgs6(*Mp, a, r, l);
#pragma unroll
  for (int k = 0; k < 6; ++k) lam[(size_t)6 * i + k] = l[k];
  if (sweeps) sweeps[i] = count;
}

}  // namespace upkie

// =========================================================== C-ABI (host)
using namespace upkie;

struct UpkieSim {
  int lanes_per_env = 0;  // 0 = choose by batch size, 1 / 2 = forced (tests, experiments)
  DevModel model;
  DevLimits limits;
  DevModel* d_model = nullptr;  // device copy read through scalar loads
  DevConfig config;
  const float* body_inertials = nullptr;  // [UPKIE_NB * UPKIE_INERTIAL_WORDS][B], caller-owned
  DevLinks links;
  const float* ext_force = nullptr;
  float* spine_state = nullptr;  // observer memory [16][B] when the spine observers run inside the step
  bool default_scalars = false;  // the model's wheel / floor scalars are the default model's: eight-lane agent steps run the instantiations that hold them as constants (octet.hpp, OctDefaultScalars)
  unsigned* census = nullptr;    // rare-path census of the eight-lane kernel (caller's device buffer) or null
  float* final_obs = nullptr;    // upkie_sim_set_final_observation: SAME_STEP autoreset completed by the step calls themselves
  float* manifold = nullptr;     // upkie_sim_set_contact_manifold: Bullet-like contact model on this persistent manifold (one-lane kernels)
  unsigned* d_guard = nullptr;   // non-finite guard counters [2] (command words replaced, env states replaced): the handle's, DevConfig::guard points here
  // Device copies of {limits, config} for the eight-lane kernels: two blocks, written by a store kernel on the launching
  // stream when a setting changed or the stream did (a launch still running on the other stream keeps its block:
  // up to two streams may step one handle at a time)
  // (include/upkie_hip.h, "Streams and hipGraphs": every hipGraph capture gets a block of its OWN -- UPKIE_MAX_GRAPH_CAPTURES
  // of them, allocated with the handle since a capture cannot allocate -- so that graphs recorded at different settings
  // replay each with its own, in any order, beside eager launches)
  DevParams* d_params[2 + UPKIE_MAX_GRAPH_CAPTURES] = {};  // [0], [1]: eager launches; [2 + c]: the launches recorded into the handle's c-th hipGraph capture
  int params_slot = 0;
  unsigned long long params_version = 1, eager_version = 0, capture_version = 0;  // settings changed / last uploaded
  void* params_stream = nullptr;
  unsigned long long capture_id = ~0ull;  // the hipGraph capture being recorded
  int captures = 0;                       // blocks handed to captures so far
  bool params_refused = false;            // the last current_params() was refused for want of a capture block (not a failed launch)
  std::string error;
};

static thread_local std::string g_create_error;

static int fail(UpkieSim* sim, int status, const std::string& msg) {
  if (sim) sim->error = msg;
  g_create_error = msg;
  return status;
}

static int check_hip(UpkieSim* sim, hipError_t err, const char* what) {
  if (err == hipSuccess) return UPKIE_OK;
  return fail(sim, UPKIE_ERR_HIP, std::string(what) + ": " + hipGetErrorString(err));
}

extern "C" int64_t upkie_hip_struct_bytes(int which) {
  switch (which) {
    case UPKIE_STRUCT_MODEL: return (int64_t)sizeof(UpkieModel);
    case UPKIE_STRUCT_SIM_CONFIG: return (int64_t)sizeof(UpkieSimConfig);
    case UPKIE_STRUCT_EXTERNAL_FORCES: return (int64_t)sizeof(UpkieExternalForces);
    case UPKIE_STRUCT_SERVO_POLICY: return (int64_t)sizeof(UpkieServoPolicy);
    case UPKIE_STRUCT_SPINE_OBSERVATION: return (int64_t)sizeof(UpkieSpineObservation);
    case UPKIE_STRUCT_MPC_CONFIG: return (int64_t)sizeof(UpkieMpcConfig);
    case UPKIE_STRUCT_OBSERVER_CONFIG: return (int64_t)sizeof(UpkieObserverConfig);
    case UPKIE_STRUCT_OBSERVER_INPUT: return (int64_t)sizeof(UpkieObserverInput);
    case UPKIE_STRUCT_OBSERVER_OUTPUT: return (int64_t)sizeof(UpkieObserverOutput);
    default: return -1;
  }
}

extern "C" int upkie_hip_device_count(void) {
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int upkie_sim_create(const UpkieSimConfig* config, const UpkieModel* model, UpkieSim** out) {
  if (!config || !model || !out) return fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (upkie_hip_device_count() <= 0) return fail(nullptr, UPKIE_ERR_NO_DEVICE, "no HIP device visible");
  UpkieSim* sim = new (std::nothrow) UpkieSim();
  if (!sim) return fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "out of host memory");
  std::string why;
  if (!convert_model(model, &sim->model, &why)) {
    delete sim;
    return fail(nullptr, UPKIE_ERR_UNSUPPORTED_MODEL, why);
  }
  if (!convert_config(config, &sim->config, &why)) {
    delete sim;
    return fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, why);
  }
  model_limits(sim->model, &sim->limits);
  if (!convert_links(model, &sim->links, &why)) {
    delete sim;
    return fail(nullptr, UPKIE_ERR_UNSUPPORTED_MODEL, why);
  }
  if (const char* forced = std::getenv("UPKIE_LANES_PER_ENV")) sim->lanes_per_env = std::atoi(forced);
  // (UPKIE_GENERIC_SCALARS=1: the generic instantiations whatever the model, for the test that holds both to the same bits)
  const char* generic = getenv("UPKIE_GENERIC_SCALARS");
  sim->default_scalars = oct_model_has_default_scalars(sim->model) && !(generic && generic[0] == '1');
  hipError_t err = hipMalloc(&sim->d_model, sizeof(DevModel));
  if (err == hipSuccess) err = hipMemcpy(sim->d_model, &sim->model, sizeof(DevModel), hipMemcpyHostToDevice);
  for (int i = 0; i < 2 + UPKIE_MAX_GRAPH_CAPTURES && err == hipSuccess; ++i) err = hipMalloc(&sim->d_params[i], sizeof(DevParams));
  if (err == hipSuccess) err = hipMalloc(&sim->d_guard, 2 * sizeof(unsigned));
  if (err == hipSuccess) err = hipMemset(sim->d_guard, 0, 2 * sizeof(unsigned));
  sim->config.guard = sim->d_guard;
  if (err != hipSuccess) {
    std::string msg = std::string("hipMalloc/hipMemcpy(model): ") + hipGetErrorString(err);
    if (sim->d_model) (void)hipFree(sim->d_model);
    if (sim->d_guard) (void)hipFree(sim->d_guard);
    for (int i = 0; i < 2 + UPKIE_MAX_GRAPH_CAPTURES; ++i)
      if (sim->d_params[i]) (void)hipFree(sim->d_params[i]);
    delete sim;
    return fail(nullptr, UPKIE_ERR_HIP, msg);
  }
  *out = sim;
  return UPKIE_OK;
}

extern "C" int upkie_sim_set_config(UpkieSim* sim, const UpkieSimConfig* config) {
  if (!sim || !config) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (config->num_envs != sim->config.num_envs) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "num_envs cannot change");
  DevConfig next;
  std::string why;
  if (!convert_config(config, &next, &why)) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, why);
  next.ext = sim->config.ext;
  next.spine = sim->config.spine;
  next.guard = sim->config.guard;
  sim->config = next;
  sim->params_version += 1;
  return UPKIE_OK;
}

extern "C" int upkie_sim_destroy(UpkieSim* sim) {
  if (sim && sim->d_model) (void)hipFree(sim->d_model);
  if (sim && sim->d_guard) (void)hipFree(sim->d_guard);
  for (int i = 0; sim && i < 2 + UPKIE_MAX_GRAPH_CAPTURES; ++i)
    if (sim->d_params[i]) (void)hipFree(sim->d_params[i]);
  delete sim;
  return UPKIE_OK;
}

extern "C" const char* upkie_sim_last_error(const UpkieSim* sim) {
  return sim ? sim->error.c_str() : g_create_error.c_str();
}

extern "C" double upkie_sim_pgs_tolerance(const UpkieSim* sim) { return sim ? (double)sim->model.pgs_tolerance : 0.0; }

extern "C" int64_t upkie_sim_state_bytes(const UpkieSim* sim) {
  return sim ? (int64_t)UPKIE_STATE_WORDS * sim->config.num_envs * (int64_t)sizeof(float) : 0;
}

extern "C" int upkie_sim_set_randomization(UpkieSim* sim, const float* body_inertials, const float* ext_force,
                                           const double ext_point[3]) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  sim->body_inertials = body_inertials;
  sim->ext_force = ext_force;
  ExtSlots& x = sim->config.ext;  // one world-frame force on the trunk
  x = ExtSlots{};
  x.count = ext_force ? 1 : 0;
  for (int k = 0; k < 3; ++k) x.point[0][k] = ext_point ? (float)ext_point[k] : 0.f;
  sim->params_version += 1;
  return UPKIE_OK;
}

extern "C" int upkie_sim_set_external_forces(UpkieSim* sim, const float* forces, const UpkieExternalForces* slots) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  ExtSlots x{};
  if (forces && slots && slots->count > 0) {
    if (slots->count > UPKIE_MAX_EXTERNAL_FORCES)
      return fail(sim, UPKIE_ERR_INVALID_ARGUMENT,
                  "at most " + std::to_string(UPKIE_MAX_EXTERNAL_FORCES) + " external forces at a time");
    x.count = slots->count;
    for (int i = 0; i < slots->count; ++i) {
      if (slots->body[i] < 0 || slots->body[i] >= UPKIE_NB) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "external force on an unknown body");
      x.body[i] = slots->body[i];
      x.local[i] = slots->local[i] ? 1 : 0;
      for (int k = 0; k < 3; ++k) x.point[i][k] = (float)slots->point[i][k];
    }
    sim->ext_force = forces;
  } else {
    sim->ext_force = nullptr;
  }
  sim->config.ext = x;
  sim->params_version += 1;
  return UPKIE_OK;
}

static int block_lanes() { return 64; }  // one wavefront per block
static dim3 grid_for(int B) { return dim3((unsigned)((B + block_lanes() - 1) / block_lanes())); }
// eight lanes per env, envs in pairs (a row of 16 lanes steps two)
static dim3 octet_grid_for(int B) { return dim3((unsigned)((8 * B + (B & 1) * 8 + UPKIE_OCTET_BLOCK - 1) / UPKIE_OCTET_BLOCK)); }

extern "C" int upkie_sim_sample_body_inertials(UpkieSim* sim, float* body_inertials, float* link_scale, double inertia_variation,
                                               void* stream) {
  if (!sim || !body_inertials) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  hipLaunchKernelGGL(body_inertials_kernel, grid_for(sim->config.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, sim->config,
                     sim->links, body_inertials, link_scale, (float)inertia_variation);
  return check_hip(sim, hipGetLastError(), "body_inertials_kernel");
}

extern "C" int upkie_sim_contact_sweeps(UpkieSim* sim, int32_t num_systems, const float* A, const float* rhs, float* lam,
                                        const uint8_t* both_tires, int32_t* sweeps, void* stream) {
  if (!sim || !A || !rhs || !lam || !both_tires || num_systems <= 0) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument or no system");
  hipLaunchKernelGGL(contact_sweeps_kernel, grid_for(num_systems), dim3(block_lanes()), 0, (hipStream_t)stream, sim->d_model, (int)num_systems, A,
                     rhs, lam, both_tires, (int*)sweeps);
  return check_hip(sim, hipGetLastError(), "contact_sweeps_kernel");
}

extern "C" int upkie_sim_sample_pushes(UpkieSim* sim, float* force, uint32_t push_index, double max_norm, void* stream) {
  if (!sim || !force) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (!(max_norm >= 0.0)) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "max_norm must be non-negative");
  hipLaunchKernelGGL(push_kernel, grid_for(sim->config.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, sim->config, force,
                     (unsigned)push_index, (float)max_norm);
  return check_hip(sim, hipGetLastError(), "push_kernel");
}

// 256 CUs x 4 SIMDs x 64 lanes x 2 waves
static const int kDenseBatch = 131072;
// up to this many envs two lanes per env still fit one wave per SIMD (1024 SIMDs x 64 lanes / 2)
static const int kPairBatch = 32768;

// up to this many envs the eight-lane kernel (step_kernel_octet) is at least as fast as the two-lane one: one wave per
// SIMD up to 8192 envs (16.5 us), two from there to 16384, where both mappings take 25.3 us -- the two co-resident
// waves do overlap, but the SIMD's issue port is then busy for the whole launch (profiles/r03_two_waves_per_simd_pmc.json)
// -- and the eight-lane kernel is the one that can carry the MPC balancer (upkie_sim_step_base_velocity_mpc) and the
// SAME_STEP autoreset inside its launch
static const int kOctetBatch = 16384;
// ... except the Servos kernels, which take the whole 512-entry register file (joint stops solved in registers): one
// wave per SIMD, 8192 envs
static const int kOctetBatchServos = 8192;

// Lanes per env of a step launch: eight (one quad per leg, one lane per body:
// octet.hpp) while that leaves the chip under-subscribed, two (one lane per
// leg: pair.hpp) up to one wave per SIMD, one beyond. The in-step spine
// observers exist in the one- and two-lane kernels only.
static int mapped_lanes(const UpkieSim* sim) {
  // the eight-lane kernel restates neither the in-step spine observers nor forces on leg links, and addresses the state
  // with 32-bit byte offsets (state_words.hpp): a forced eight-lane mapping yields to the others beyond 2^32 bytes
  bool eight = !sim->spine_state && (unsigned long long)sim->config.num_envs * UPKIE_STATE_WORDS * sizeof(float) < (1ull << 32);
  if (sim->ext_force)
    for (int i = 0; i < sim->config.ext.count; ++i) eight = eight && sim->config.ext.body[i] == 0;
  if (sim->lanes_per_env == 8 || sim->lanes_per_env == 2 || sim->lanes_per_env == 1) {
    if (sim->lanes_per_env == 8 && !eight) return sim->manifold ? 1 : 2;
    if (sim->manifold && sim->lanes_per_env == 2) return 1;
    return sim->lanes_per_env;
  }
  if (sim->config.num_envs <= kOctetBatch && eight) return 8;
  if (sim->manifold) return 1;  // the Bullet-like contact model exists in the one- and eight-lane kernels (bullet_like.hpp, octet.hpp)
  return sim->config.num_envs <= kPairBatch ? 2 : 1;
}
// fewer lanes than envs x 2: several env.step() of the fused agent can share a launch (state in registers)
static bool uses_lane_pairs(const UpkieSim* sim) { return mapped_lanes(sim) >= 2; }

extern "C" int upkie_sim_lanes_per_env(const UpkieSim* sim) { return !sim ? 0 : mapped_lanes(sim); }

// ... of the step kernel a given entry point launches: the Servos kernels leave the eight-lane mapping earlier
static int mapped_lanes_of_mode(const UpkieSim* sim, int mode) {
  int lanes = mapped_lanes(sim);
  // (round 5: the Bullet-like contact model's eight-lane variant serves UpkieServos steps too)
  if (mode == MODE_SERVOS && lanes == 8 && sim->lanes_per_env != 8 && sim->config.num_envs > kOctetBatchServos) lanes = sim->manifold ? 1 : 2;
  // (round 6, ADVICE r5: on the eight-lane Bullet-like kernel a joint within reach of its stop is now a row of the specification's
  // own 50 sweeps -- octet_limit_path_scratch<BULLET_LIKE>, general_constraint_solve_bullet_like -- as on the one-lane kernels, so
  // Servos steps stay on eight lanes up to kOctetBatchServos envs; what the mapping still does not restate is SEVERAL cached
  // points on one tire, a robot lying flat on its side: upkie_sim_set_lanes_per_env(sim, 1) selects the one-lane kernels)
  return lanes;
}
extern "C" int upkie_sim_lanes_per_env_of(const UpkieSim* sim, int observation_layout) {
  if (!sim) return 0;
  return mapped_lanes_of_mode(sim, observation_layout == UPKIE_OBSERVATION_SERVOS ? MODE_SERVOS : MODE_PENDULUM);
}

extern "C" int upkie_sim_set_lanes_per_env(UpkieSim* sim, int lanes) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  if (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 8) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "lanes per env: 0 (automatic), 1, 2 or 8");
  sim->lanes_per_env = lanes;
  return UPKIE_OK;
}

extern "C" int upkie_sim_set_census(UpkieSim* sim, uint32_t* counters) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  sim->census = counters;
  return UPKIE_OK;
}

// Non-finite guard (include/upkie_hip.h): what the step kernels counted so far on this handle; waits for `stream`.
extern "C" int upkie_sim_guard_counts(UpkieSim* sim, uint32_t counts[2], int reset, void* stream) {
  if (!sim || !counts) return UPKIE_ERR_INVALID_ARGUMENT;
  hipError_t err = hipMemcpyAsync(counts, sim->d_guard, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (err == hipSuccess && reset) err = hipMemsetAsync(sim->d_guard, 0, 2 * sizeof(unsigned), (hipStream_t)stream);
  if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
  return check_hip(sim, err, "upkie_sim_guard_counts");
}

extern "C" int upkie_sim_set_contact_manifold(UpkieSim* sim, float* manifold) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  if (manifold && sim->spine_state) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "the Bullet-like contact model does not run the in-step spine observers");
  sim->manifold = manifold;
  return UPKIE_OK;
}

// The caller's word that no hipGraph recorded from this handle so far will be replayed again (they were destroyed, or are
// about to be re-captured): their settings blocks are handed out again, the next capture takes the first one.
extern "C" int upkie_sim_release_graph_captures(UpkieSim* sim) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  sim->captures = 0;
  sim->capture_id = ~0ull;
  sim->capture_version = 0;
  return UPKIE_OK;
}

extern "C" int upkie_sim_set_final_observation(UpkieSim* sim, float* final_obs) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  sim->final_obs = final_obs;
  return UPKIE_OK;
}

__global__ void store_params_kernel(DevParams params, DevParams* out) {
  const unsigned* from = reinterpret_cast<const unsigned*>(&params);
  unsigned* to = reinterpret_cast<unsigned*>(out);
  for (int i = threadIdx.x; i < (int)(sizeof(DevParams) / sizeof(unsigned)); i += blockDim.x) to[i] = from[i];
}

// The device block holding this handle's current {limits, config}, valid for launches on `stream` from here on.
static const DevParams* current_params(UpkieSim* sim, void* stream) {
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  unsigned long long capture_id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &capturing, &capture_id) == hipSuccess && capturing == hipStreamCaptureStatusActive) {
    // a launch recorded into a hipGraph: the graph carries its own store of the settings as they are now into a block
    // no eager launch and no OTHER graph uses (a replay must neither see later settings nor leave stale ones behind
    // for eager launches or for another graph's replay): one block and one store per capture, the store again when a
    // setting changes while capturing
    const bool new_capture = sim->capture_id != capture_id;
    sim->params_refused = false;
    if (new_capture && sim->captures >= UPKIE_MAX_GRAPH_CAPTURES) {  // (launch_step reports it; upkie_sim_release_graph_captures frees the blocks)
      sim->params_refused = true;
      return nullptr;
    }
    if (sim->capture_version != sim->params_version || new_capture) {
      DevParams block;
      block.limits = sim->limits;
      block.config = sim->config;
      DevParams* target = sim->d_params[2 + (new_capture ? sim->captures : sim->captures - 1)];
      hipLaunchKernelGGL(store_params_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, block, target);
      if (hipGetLastError() != hipSuccess) return nullptr;  // (nothing recorded: the step launch behind it is refused too)
      if (new_capture) sim->captures += 1;
      sim->capture_id = capture_id;
      sim->capture_version = sim->params_version;
    }
    return sim->d_params[2 + sim->captures - 1];
  }
  sim->params_refused = false;
  if (sim->eager_version != sim->params_version || stream != sim->params_stream) {
    sim->params_slot ^= 1;
    DevParams block;
    block.limits = sim->limits;
    block.config = sim->config;
    // (as a kernel argument, not a host-to-device copy: ordered on the stream, capturable in a hipGraph, no host buffer to keep alive)
    hipLaunchKernelGGL(store_params_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, block, sim->d_params[sim->params_slot]);
    if (hipGetLastError() != hipSuccess) {  // the block was not refreshed: no step kernel may read it as if it were
      sim->params_slot ^= 1;
      return nullptr;
    }
    sim->eager_version = sim->params_version;
    sim->params_stream = stream;
  }
  return sim->d_params[sim->params_slot];
}

template <int MODE>
static int launch_step(UpkieSim* sim, float* state, const float* act, float* obs, float* reward, uint8_t* terminated,
                       uint8_t* truncated, const uint8_t* mask, void* stream, int packed = 0,
                       BaseVelocityPtrs bv = BaseVelocityPtrs{}, bool done_pass = false,
                       float* final_obs = nullptr, int n_steps = 1, const UpkieServoPolicy* policy = nullptr) {
  if (!sim || !state) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (MODE != MODE_RESET && (!obs || (!packed && !done_pass && (!reward || !terminated || !truncated))))
    return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null output buffer");
  if ((MODE == MODE_PENDULUM || MODE == MODE_GYROPOD || MODE == MODE_SERVOS || MODE == MODE_BASE_VELOCITY) && !act && !done_pass)
    return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null action buffer");
  DevConfig config = sim->config;
  if (done_pass) config.autoreset_mode = AUTORESET_DONE_PASS;
  // SAME_STEP autoreset completed by the step call itself (upkie_sim_set_final_observation): inside the launch on the
  // eight-lane mapping, by a second launch (the DONE pass) behind this one on the others
  constexpr bool RESETS_IN_PLACE = MODE == MODE_PENDULUM || MODE == MODE_GYROPOD || MODE == MODE_SERVOS;
  const bool same_step = RESETS_IN_PLACE && !done_pass && packed != 1 && sim->final_obs != nullptr && config.autoreset_mode == UPKIE_AUTORESET_DISABLED;
  const bool same_step_in_kernel = same_step && mapped_lanes_of_mode(sim, MODE) == 8;  // (round 5: the Bullet-like eight-lane kernels have their IN_PLACE instantiations too)
  if (same_step_in_kernel) final_obs = sim->final_obs;
  // (UPKIE_ALWAYS_RAND_KERNELS=1: launch the randomisation-capable instantiation even without inertial records or forces
  // -- null pointers, tested at run time --: the A/B that decides whether the RAND = false instantiations of the one- and
  // two-lane kernels earn their place in the library, profiles/r05_ab_rand_instantiations.txt)
  static const bool always_rand = [] { const char* v = std::getenv("UPKIE_ALWAYS_RAND_KERNELS"); return v && v[0] == '1'; }();
  const bool rnd = sim->body_inertials || sim->ext_force || (always_rand && mapped_lanes_of_mode(sim, MODE) != 8);
  dim3 grid = grid_for(sim->config.num_envs), block(block_lanes());
  const float* scale = rnd ? sim->body_inertials : nullptr;
  const float* force = rnd ? sim->ext_force : nullptr;
  hipStream_t st = (hipStream_t)stream;
  // more than two waves per SIMD in flight: favour occupancy over registers;
  // fewer lanes than SIMD slots: split every env over two lanes (pair.hpp)
  const bool dense = sim->config.num_envs >= kDenseBatch;
  // SPINE: the spine observers run inside the step (a separate instantiation:
  // compiled in but switched off they would still cost the common path 2 %)
#define UPKIE_LAUNCH_S(R, W, S)                                                                                                   \
  hipLaunchKernelGGL((step_kernel<MODE, R, W, S, false>), grid, block, 0, st, sim->d_model, sim->limits, config, state, act, obs, \
                     reward, terminated, truncated, mask, scale, force, packed, bv, sim->spine_state, final_obs, (float*)nullptr)
#define UPKIE_LAUNCH_BULLET(R)                                                                                                          \
  hipLaunchKernelGGL((step_kernel<MODE, R, 1, false, true>), grid, block, 0, st, sim->d_model, sim->limits, config, state, act, obs, \
                     reward, terminated, truncated, mask, scale, force, packed, bv, (float*)nullptr, final_obs, sim->manifold)
#define UPKIE_LAUNCH(R, W) \
  do { if (spine) UPKIE_LAUNCH_S(R, W, true); else UPKIE_LAUNCH_S(R, W, false); } while (0)
#define UPKIE_LAUNCH_PAIR_S(R, S)                                                                                                 \
  hipLaunchKernelGGL((step_kernel_pair<MODE, R, S>), grid_for(2 * sim->config.num_envs), block, 0, st, sim->d_model, sim->limits, \
                     config, state, act, obs, reward, terminated, truncated, mask, scale, force, packed, bv, sim->spine_state, final_obs, \
                     n_steps)
#define UPKIE_LAUNCH_PAIR(R) \
  do { if (spine) UPKIE_LAUNCH_PAIR_S(R, true); else UPKIE_LAUNCH_PAIR_S(R, false); } while (0)
#define UPKIE_LAUNCH_OCTET_D(R, D, IP)                                                                                          \
  hipLaunchKernelGGL((step_kernel_octet<MODE, R, D, IP, false>), octet_grid_for(sim->config.num_envs), dim3(UPKIE_OCTET_BLOCK), 0, st, \
                     sim->d_model, params, done_pass ? 1 : 0, sim->config.num_envs, state, act, obs, reward, terminated, truncated, mask, scale, \
                     force, packed, bv, final_obs, n_steps, sim->census, policy_arg, (float*)nullptr)
#define UPKIE_LAUNCH_OCTET_BULLET_IP(R, IP)                                                                                     \
  hipLaunchKernelGGL((step_kernel_octet<MODE, R, false, IP, true>), octet_grid_for(sim->config.num_envs), dim3(UPKIE_OCTET_BLOCK), 0, st, \
                     sim->d_model, params, done_pass ? 1 : 0, sim->config.num_envs, state, act, obs, reward, terminated, truncated, mask, scale, \
                     force, packed, bv, final_obs, n_steps, sim->census, policy_arg, sim->manifold)
#define UPKIE_LAUNCH_OCTET_BULLET(R)                                                             \
  do {                                                                                           \
    bool launched = false;                                                                       \
    if constexpr (octet_resets_in_place(MODE)) {                                                 \
      if (same_step_in_kernel) { UPKIE_LAUNCH_OCTET_BULLET_IP(R, true); launched = true; }       \
    }                                                                                            \
    if (!launched) UPKIE_LAUNCH_OCTET_BULLET_IP(R, false);                                       \
  } while (0)
  // which instantiation (step_instances.hpp lists them): the SAME_STEP autoreset inside the launch has its own (the second
  // pass makes the whole step a loop body: spills); the Pendulum / Gyropod steps also exist with the default model's
  // scalars as constants
#define UPKIE_LAUNCH_OCTET(R)                                                                    \
  do {                                                                                           \
    bool launched = false;                                                                       \
    if constexpr (octet_resets_in_place(MODE) && octet_has_default_scalars(MODE)) {              \
      if (same_step_in_kernel && sim->default_scalars) { UPKIE_LAUNCH_OCTET_D(R, true, true); launched = true; } \
    }                                                                                            \
    if constexpr (octet_resets_in_place(MODE)) {                                                 \
      if (!launched && same_step_in_kernel) { UPKIE_LAUNCH_OCTET_D(R, false, true); launched = true; } \
    }                                                                                            \
    if constexpr (octet_has_default_scalars(MODE)) {                                             \
      if (!launched && sim->default_scalars) { UPKIE_LAUNCH_OCTET_D(R, true, false); launched = true; } \
    }                                                                                            \
    if (!launched) UPKIE_LAUNCH_OCTET_D(R, false, false);                                        \
  } while (0)
  const bool spine = sim->spine_state != nullptr;
  const int lanes = mapped_lanes_of_mode(sim, MODE);
  ServoPolicyArg<MODE> policy_arg{};
  if constexpr (MODE == MODE_SERVOS) {
    if (policy) policy_arg = *policy;
  }
  // the eight-lane kernels read limits and config from this handle's device block
  const DevParams* params = lanes == 8 ? current_params(sim, stream) : nullptr;
  if (lanes == 8 && !params)
    return fail(sim, UPKIE_ERR_HIP, sim->params_refused  // (the branch that actually failed: ADVICE r5)
                                        ? "this handle's steps have been recorded into UPKIE_MAX_GRAPH_CAPTURES hipGraph captures already: destroy graphs that are no longer "
                                          "replayed and call upkie_sim_release_graph_captures (include/upkie_hip.h, Streams and hipGraphs)"
                                        : "could not refresh the device block of the handle's settings (the store kernel's launch failed)");
  if (lanes == 8 && sim->manifold) {
    if (rnd) UPKIE_LAUNCH_OCTET_BULLET(true); else UPKIE_LAUNCH_OCTET_BULLET(false);
  } else if (lanes == 8) {
    if (rnd) UPKIE_LAUNCH_OCTET(true); else UPKIE_LAUNCH_OCTET(false);
  } else if (lanes == 2) {
    if (rnd) UPKIE_LAUNCH_PAIR(true); else UPKIE_LAUNCH_PAIR(false);
  } else if constexpr (MODE == MODE_PENDULUM_ROLLOUT) {
    return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "several steps per launch need the two-lane mapping");
  } else if (sim->manifold) {  // the Bullet-like contact model (upkie_sim_set_contact_manifold)
    if (rnd) UPKIE_LAUNCH_BULLET(true); else UPKIE_LAUNCH_BULLET(false);
  } else if (rnd) {
    if (dense) UPKIE_LAUNCH(true, 2); else UPKIE_LAUNCH(true, 1);
  } else {
    if (dense) UPKIE_LAUNCH(false, 2); else UPKIE_LAUNCH(false, 1);
  }
#undef UPKIE_LAUNCH_OCTET
#undef UPKIE_LAUNCH_OCTET_BULLET
#undef UPKIE_LAUNCH_OCTET_BULLET_IP
#undef UPKIE_LAUNCH_BULLET
#undef UPKIE_LAUNCH_OCTET_D
#undef UPKIE_LAUNCH_PAIR_S
#undef UPKIE_LAUNCH_S
#undef UPKIE_LAUNCH_PAIR
#undef UPKIE_LAUNCH
  const int status = check_hip(sim, hipGetLastError(), "step_kernel");
  if constexpr (RESETS_IN_PLACE) {
    if (status == UPKIE_OK && same_step && !same_step_in_kernel)
      return launch_step<MODE>(sim, state, nullptr, obs, nullptr, nullptr, nullptr, nullptr, stream, 0, BaseVelocityPtrs{}, true, sim->final_obs);
  }
  return status;
}

extern "C" int upkie_sim_reset(UpkieSim* sim, float* state, const uint8_t* mask, float* obs6, void* stream) {
  return launch_step<MODE_RESET>(sim, state, nullptr, obs6, nullptr, nullptr, nullptr, mask, stream);
}

extern "C" int upkie_sim_step_pendulum(UpkieSim* sim, float* state, const float* act, float* obs, float* reward,
                                       uint8_t* terminated, uint8_t* truncated, void* stream) {
  return launch_step<MODE_PENDULUM>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream);
}

extern "C" int upkie_sim_step_pendulum_agent(UpkieSim* sim, float* state, float* obs, float* reward, uint8_t* terminated,
                                             uint8_t* truncated, void* stream) {
  return launch_step<MODE_PENDULUM_AGENT>(sim, state, nullptr, obs, reward, terminated, truncated, nullptr, stream);
}

extern "C" int upkie_sim_step_pendulum_packed(UpkieSim* sim, float* state, const float* act, float* records, void* stream) {
  return launch_step<MODE_PENDULUM>(sim, state, act, records, nullptr, nullptr, nullptr, nullptr, stream, 1);
}

extern "C" int upkie_sim_step_pendulum_agent_packed(UpkieSim* sim, float* state, float* records, void* stream) {
  return launch_step<MODE_PENDULUM_AGENT>(sim, state, nullptr, records, nullptr, nullptr, nullptr, nullptr, stream, 1);
}

extern "C" int upkie_sim_step_pendulum_agent_records(UpkieSim* sim, float* state, const float* prev_records, float* records,
                                                    void* stream) {
  if (!prev_records) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  return launch_step<MODE_PENDULUM_AGENT>(sim, state, prev_records, records, nullptr, nullptr, nullptr, nullptr, stream, 1);
}

extern "C" int upkie_sim_step_pendulum_agent_rollout(UpkieSim* sim, float* state, const float* prev_records, float* records,
                                                    int32_t num_steps, void* stream) {
  if (!prev_records || !records || num_steps < 1) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument or num_steps < 1");
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  const BaseVelocityPtrs none{};
  if (uses_lane_pairs(sim))  // one launch: the state stays in registers from step to step
    return launch_step<MODE_PENDULUM_ROLLOUT>(sim, state, prev_records, records, nullptr, nullptr, nullptr, nullptr, stream, 1, none, false,
                                              nullptr, num_steps);
  const size_t stride = (size_t)8 * sim->config.num_envs;  // large batches: launch overhead is already amortised
  for (int32_t k = 0; k < num_steps; ++k) {
    const int status = launch_step<MODE_PENDULUM_AGENT>(sim, state, k ? records + (k - 1) * stride : prev_records, records + k * stride,
                                                        nullptr, nullptr, nullptr, nullptr, stream, 1);
    if (status != UPKIE_OK) return status;
  }
  return UPKIE_OK;
}

extern "C" int upkie_sim_step_gyropod(UpkieSim* sim, float* state, const float* act, float* obs, float* reward,
                                      uint8_t* terminated, uint8_t* truncated, void* stream) {
  return launch_step<MODE_GYROPOD>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream);
}

extern "C" int upkie_sim_step_base_velocity(UpkieSim* sim, float* state, const float* act, const float* commanded_velocity,
                                            float* obs, float* mpc_x0, uint8_t* mpc_contact, float* reward,
                                            uint8_t* terminated, uint8_t* truncated, void* stream) {
  if (!commanded_velocity || !mpc_x0 || !mpc_contact) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null MPC buffer");
  return launch_step<MODE_BASE_VELOCITY>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream, 0,
                                         BaseVelocityPtrs{commanded_velocity, mpc_x0, mpc_contact});
}

extern "C" int upkie_sim_step_servos(UpkieSim* sim, float* state, const float* act, float* obs, float* reward,
                                     uint8_t* terminated, uint8_t* truncated, void* stream) {
  return launch_step<MODE_SERVOS>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream);
}

// Servo-level policy on the device (include/upkie_hip.h, upkie_sim_servo_policy): HBM bound, 12 state words in,
// 36 action words out per env; the row-major action rows leave as 16-byte stores (one env = nine of them).
__global__ __launch_bounds__(64) void servo_policy_kernel(int B, float signed_radius, UpkieServoPolicy P, float* __restrict__ state,
                                                           float* __restrict__ act) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
#define SW(w) state[(size_t)(w) * B + e]
  const float qw = SW(UPKIE_S_QUAT), qx = SW(UPKIE_S_QUAT + 1), qy = SW(UPKIE_S_QUAT + 2), qz = SW(UPKIE_S_QUAT + 3);
  const float pitch = asinf(fminf(fmaxf(2.f * (qw * qy - qz * qx), -1.f), 1.f));
  const float p = 0.5f * (SW(UPKIE_S_Q + 2) - SW(UPKIE_S_Q + 5)) * signed_radius;
  const float pd = 0.5f * (SW(UPKIE_S_QD + 2) - SW(UPKIE_S_QD + 5)) * signed_radius;
  float a[36];
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a[6 * j + k] = P.action[j][k];
    float fb = P.pitch_to_velocity[j] * pitch + P.position_to_velocity[j] * p + P.velocity_to_velocity[j] * pd;
    if (P.velocity_feedback_clip[j] > 0.f) fb = fminf(fmaxf(fb, -P.velocity_feedback_clip[j]), P.velocity_feedback_clip[j]);
    a[6 * j + 1] += fb;
    a[6 * j + 2] += P.pitch_to_torque[j] * pitch;
  }
  float4* out = reinterpret_cast<float4*>(act + (size_t)36 * e);
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
  if (P.fall_pitch > 0.f && fabsf(pitch) > P.fall_pitch) SW(UPKIE_S_DONE) = 1.f;
#undef SW
}

extern "C" int upkie_sim_servo_policy(UpkieSim* sim, float* state, const UpkieServoPolicy* policy, float* act, void* stream) {
  if (!sim || !state || !policy || !act) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  hipLaunchKernelGGL(servo_policy_kernel, grid_for(sim->config.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, sim->config.num_envs,
                     sim->model.left_sign * sim->model.wheel_radius, *policy, state, act);
  return check_hip(sim, hipGetLastError(), "servo_policy_kernel");
}

extern "C" int upkie_sim_step_servos_policy(UpkieSim* sim, float* state, const UpkieServoPolicy* policy, float* act, float* obs, float* reward,
                                            uint8_t* terminated, uint8_t* truncated, void* stream) {
  if (!sim || !state || !policy || !act) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (mapped_lanes_of_mode(sim, MODE_SERVOS) != 8) {  // the other mappings: the policy's own launch into `act`, then the step
    const int status = upkie_sim_servo_policy(sim, state, policy, act, stream);
    if (status != UPKIE_OK) return status;
    return launch_step<MODE_SERVOS>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream);
  }
  // (`packed` == 2 tells the eight-lane Servos kernel to take its actions from the policy argument; `act` is not read)
  return launch_step<MODE_SERVOS>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream, 2, BaseVelocityPtrs{}, false, nullptr, 1, policy);
}

extern "C" int upkie_sim_autoreset_done(UpkieSim* sim, int observation, float* state, float* obs, float* final_obs, void* stream) {
  const BaseVelocityPtrs none{};
  switch (observation) {
    case UPKIE_OBSERVATION_PENDULUM:
      return launch_step<MODE_PENDULUM>(sim, state, nullptr, obs, nullptr, nullptr, nullptr, nullptr, stream, 0, none, true, final_obs);
    case UPKIE_OBSERVATION_PENDULUM_RECORDS:
      return launch_step<MODE_PENDULUM>(sim, state, nullptr, obs, nullptr, nullptr, nullptr, nullptr, stream, 1, none, true, final_obs);
    case UPKIE_OBSERVATION_GYROPOD:
      return launch_step<MODE_GYROPOD>(sim, state, nullptr, obs, nullptr, nullptr, nullptr, nullptr, stream, 0, none, true, final_obs);
    case UPKIE_OBSERVATION_SERVOS:
      return launch_step<MODE_SERVOS>(sim, state, nullptr, obs, nullptr, nullptr, nullptr, nullptr, stream, 0, none, true, final_obs);
    default:
      return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "unknown observation layout");
  }
}

extern "C" int upkie_sim_observe(UpkieSim* sim, float* state, const UpkieSpineObservation* out, int update_imu, void* stream) {
  if (!sim || !state || !out) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  ObsPtrs p;
  p.pitch = out->pitch;
  p.angular_velocity = out->angular_velocity;
  p.linear_velocity = out->linear_velocity;
  p.rotation_base_to_world = out->rotation_base_to_world;
  p.floor_contact = out->floor_contact;
  p.imu_orientation = out->imu_orientation;
  p.imu_angular_velocity = out->imu_angular_velocity;
  p.imu_linear_acceleration = out->imu_linear_acceleration;
  p.imu_raw_linear_acceleration = out->imu_raw_linear_acceleration;
  p.servo = out->servo;
  p.wheel_odometry = out->wheel_odometry;
  hipLaunchKernelGGL(observe_kernel, grid_for(sim->config.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, sim->model, sim->config,
                     state, p, update_imu);
  return check_hip(sim, hipGetLastError(), "observe_kernel");
}

extern "C" int upkie_sim_contact_points(UpkieSim* sim, const float* state, float* out, void* stream) {
  if (!sim || !state || !out) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (sim->manifold)
    hipLaunchKernelGGL(contact_points_kernel<true>, grid_for(sim->config.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, sim->d_model,
                       sim->limits, sim->config, state, sim->body_inertials, sim->ext_force, out, sim->manifold);
  else
    hipLaunchKernelGGL(contact_points_kernel<false>, grid_for(sim->config.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, sim->d_model,
                       sim->limits, sim->config, state, sim->body_inertials, sim->ext_force, out, (const float*)nullptr);
  return check_hip(sim, hipGetLastError(), "contact_points_kernel");
}

// ------------------------------------------------------------------- MPC
struct UpkieMpc {
  MpcDev dev;
  int tiles = 0;
  float* d_minv = nullptr;
  void* d_minv_h = nullptr;
  float* d_gx = nullptr;
  float* d_gv = nullptr;
  float* d_kx = nullptr;
  float* d_kv = nullptr;
  std::string error;
};

static int mpc_fail(UpkieMpc* mpc, int status, const std::string& msg) {
  if (mpc) mpc->error = msg;
  g_create_error = msg;
  return status;
}

extern "C" int upkie_mpc_create(const UpkieMpcConfig* config, UpkieMpc** out) {
  if (!config || !out) return mpc_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (config->num_envs <= 0 || config->nb_timesteps <= 0 || config->admm_iterations <= 0 || !(config->admm_rho > 0.0))
    return mpc_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "num_envs, nb_timesteps, admm_iterations, admm_rho must be positive");
  if (!(config->admm_relaxation >= 0.0 && config->admm_relaxation < 2.0))
    return mpc_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "admm_relaxation must lie in (0, 2) (0: unset, the plain iteration)");
  if (config->nb_timesteps > 64)
    return mpc_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "nb_timesteps > 64 is not supported by the HIP path");
  if (upkie_hip_device_count() <= 0) return mpc_fail(nullptr, UPKIE_ERR_NO_DEVICE, "no HIP device visible");
  UpkieMpc* mpc = new (std::nothrow) UpkieMpc();
  if (!mpc) return mpc_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "out of host memory");
  mpc->tiles = (config->nb_timesteps + 15) / 16;
  const int np = 16 * mpc->tiles;
  std::vector<float> minv, kx, kv;
  std::vector<uint16_t> minv_h;
  std::vector<float> gx, gv;
  std::string why;
  const bool fp16_path = true;  // (every horizon since round 6; the fp32 operands stay for UPKIE_MPC_FP32=1)
  if (!mpc_host_setup(*config, np, &minv, &kx, &kv, &why, fp16_path ? &minv_h : nullptr, fp16_path ? &gx : nullptr, fp16_path ? &gv : nullptr, &mpc->dev.scale)) {
    delete mpc;
    return mpc_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, why);
  }
  hipError_t err = hipMalloc(&mpc->d_minv, minv.size() * sizeof(float));
  if (err == hipSuccess) err = hipMalloc(&mpc->d_kx, kx.size() * sizeof(float));
  if (err == hipSuccess) err = hipMalloc(&mpc->d_kv, kv.size() * sizeof(float));
  if (err == hipSuccess) err = hipMemcpy(mpc->d_minv, minv.data(), minv.size() * sizeof(float), hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(mpc->d_kx, kx.data(), kx.size() * sizeof(float), hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(mpc->d_kv, kv.data(), kv.size() * sizeof(float), hipMemcpyHostToDevice);
  if (err == hipSuccess && !minv_h.empty()) {
    err = hipMalloc(&mpc->d_minv_h, minv_h.size() * sizeof(uint16_t));
    if (err == hipSuccess) err = hipMemcpy(mpc->d_minv_h, minv_h.data(), minv_h.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMalloc(&mpc->d_gx, gx.size() * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&mpc->d_gv, gv.size() * sizeof(float));
    if (err == hipSuccess) err = hipMemcpy(mpc->d_gx, gx.data(), gx.size() * sizeof(float), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(mpc->d_gv, gv.data(), gv.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (err != hipSuccess) {
    std::string msg = std::string("hipMalloc/hipMemcpy: ") + hipGetErrorString(err);
    upkie_mpc_destroy(mpc);
    return mpc_fail(nullptr, UPKIE_ERR_HIP, msg);
  }
  mpc->dev.minv = mpc->d_minv;
  mpc->dev.minv_h = mpc->d_minv_h;
  mpc->dev.gx = mpc->d_gx;
  mpc->dev.gv = mpc->d_gv;
  mpc->dev.kx = mpc->d_kx;
  mpc->dev.kv = mpc->d_kv;
  mpc->dev.num_envs = config->num_envs;
  mpc->dev.n = config->nb_timesteps;
  mpc->dev.iterations = config->admm_iterations;
  mpc->dev.rho = (float)config->admm_rho;
  mpc->dev.alpha = config->admm_relaxation > 0.0 ? (float)config->admm_relaxation : 1.f;
  mpc->dev.bound = (float)config->max_ground_accel;
  mpc->dev.max_ground_velocity = (float)config->max_ground_velocity;
  mpc->dev.fall_pitch = (float)config->fall_pitch;
  *out = mpc;
  return UPKIE_OK;
}

extern "C" int upkie_mpc_destroy(UpkieMpc* mpc) {
  if (!mpc) return UPKIE_OK;
  if (mpc->d_minv) (void)hipFree(mpc->d_minv);
  if (mpc->d_minv_h) (void)hipFree(mpc->d_minv_h);
  if (mpc->d_gx) (void)hipFree(mpc->d_gx);
  if (mpc->d_gv) (void)hipFree(mpc->d_gv);
  if (mpc->d_kx) (void)hipFree(mpc->d_kx);
  if (mpc->d_kv) (void)hipFree(mpc->d_kv);
  delete mpc;
  return UPKIE_OK;
}

extern "C" const char* upkie_mpc_last_error(const UpkieMpc* mpc) { return mpc ? mpc->error.c_str() : g_create_error.c_str(); }

extern "C" int64_t upkie_mpc_workspace_bytes(const UpkieMpc* mpc) {
  return mpc ? (int64_t)2 * mpc->dev.n * mpc->dev.num_envs * (int64_t)sizeof(float) : 0;
}

extern "C" int upkie_mpc_reset(UpkieMpc* mpc, float* workspace, float* commanded_velocity, const uint8_t* mask, void* stream) {
  if (!mpc || !workspace || !commanded_velocity) return mpc_fail(mpc, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  hipLaunchKernelGGL(mpc_reset_kernel, grid_for(mpc->dev.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, mpc->dev.num_envs,
                     mpc->dev.n, workspace, commanded_velocity, mask);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? UPKIE_OK : mpc_fail(mpc, UPKIE_ERR_HIP, hipGetErrorString(err));
}

static int mpc_launch(UpkieMpc* mpc, float* workspace, const float* x0, const float* target_velocity, int target_stride,
                      const uint8_t* contact, const float* done, double dt, float* commanded_velocity, float* first_input,
                      void* stream) {
  if (!mpc || !workspace || !x0 || !target_velocity || !contact || !commanded_velocity)
    return mpc_fail(mpc, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (!(dt / 0.1 < 0.5)) return mpc_fail(mpc, UPKIE_ERR_INVALID_ARGUMENT, "dt too large for the 0.1 s low-pass (filters.py:78-79)");
  dim3 grid((unsigned)((mpc->dev.num_envs + 15) / 16)), block(64);
  hipStream_t st = (hipStream_t)stream;
  // the product on the fp16 matrix path, two terms per operand (mpc_tile_h); UPKIE_MPC_FP32=1 selects the fp32
  // MFMA kernels of rounds 2-6 for this entry point (the balancer inside a step's launch: a build flag, UPKIE_FUSED_MPC_FP32) (A/B, profiles/r06_mpc_f16_split.txt)
  static const bool fp32_product = [] { const char* v = std::getenv("UPKIE_MPC_FP32"); return v && v[0] == '1'; }();
  if (!fp32_product) {
    switch (mpc->tiles) {
      case 1: hipLaunchKernelGGL(mpc_step_h_kernel<1>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input); break;
      case 2: hipLaunchKernelGGL(mpc_step_h_kernel<2>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input); break;
      case 3: hipLaunchKernelGGL(mpc_step_h_kernel<3>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input); break;
      default: hipLaunchKernelGGL(mpc_step_h_kernel<4>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input); break;
    }
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? UPKIE_OK : mpc_fail(mpc, UPKIE_ERR_HIP, hipGetErrorString(err));
  }
  switch (mpc->tiles) {
    case 1: hipLaunchKernelGGL(mpc_step_kernel<1>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input); break;
    case 2: hipLaunchKernelGGL(mpc_step_kernel<2>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input); break;
    case 3: hipLaunchKernelGGL(mpc_step_kernel<3>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input); break;
    default:
      static const bool four_tiles = [] { const char* v = std::getenv("UPKIE_MPC_FOUR_TILES"); return v && v[0] == '1'; }();  // (A/B: the round-5 kernel)
      if (mpc->dev.n >= 49 && mpc->dev.n <= 50 && !four_tiles)  // (round 6: three row tiles on the matrix cores, rows 48 / 49 on the vector unit, mpc.hpp)
        hipLaunchKernelGGL(mpc_step_tail_kernel, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input);
      else if (mpc->dev.n <= 52)  // (k-steps 13-15 of every row tile only read padding, mpc.hpp)
        hipLaunchKernelGGL((mpc_step_kernel<4, 13>), grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input);
      else
        hipLaunchKernelGGL(mpc_step_kernel<4>, grid, block, 0, st, mpc->dev, workspace, x0, target_velocity, target_stride, contact, done, (float)dt, commanded_velocity, first_input);
      break;
  }
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? UPKIE_OK : mpc_fail(mpc, UPKIE_ERR_HIP, hipGetErrorString(err));
}

extern "C" int upkie_mpc_step(UpkieMpc* mpc, float* workspace, const float* x0, const float* target_velocity,
                              const uint8_t* contact, double dt, float* commanded_velocity, float* first_input, void* stream) {
  return mpc_launch(mpc, workspace, x0, target_velocity, 1, contact, nullptr, dt, commanded_velocity, first_input, stream);
}

extern "C" int upkie_mpc_step_env(UpkieMpc* mpc, float* workspace, const float* x0, const float* act, const uint8_t* contact,
                                  const float* done, double dt, float* commanded_velocity, void* stream) {
  return mpc_launch(mpc, workspace, x0, act, 2, contact, done, dt, commanded_velocity, nullptr, stream);
}

// UpkieBaseVelocity's step with its balancer in front, ONE launch where the mapping allows it (two lanes per env,
// horizon <= 16: the wavefront that steps 32 envs first solves their QPs), two launches otherwise: same results.
extern "C" int upkie_sim_step_base_velocity_mpc(UpkieSim* sim, UpkieMpc* mpc, float* state, float* workspace, const float* act,
                                                float* commanded_velocity, float* obs, float* mpc_x0, uint8_t* mpc_contact,
                                                float* reward, uint8_t* terminated, uint8_t* truncated, void* stream) {
  if (!sim || !mpc || !state || !workspace || !act || !commanded_velocity || !mpc_x0 || !mpc_contact)
    return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (mpc->dev.num_envs != sim->config.num_envs) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "balancer and simulation differ in num_envs");
  if (!(sim->config.dt / 0.1 < 0.5)) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "dt too large for the 0.1 s low-pass (filters.py:78-79)");
  if ((mapped_lanes(sim) == 2 || mapped_lanes(sim) == 8) && mpc->tiles == 1) {
    BaseVelocityPtrs bv{commanded_velocity, mpc_x0, mpc_contact};
    bv.mpc_fused = 1;
    bv.mpc = mpc->dev;
    bv.mpc_ws = workspace;
    bv.mpc_commanded = commanded_velocity;
    return launch_step<MODE_BASE_VELOCITY>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream, 0, bv);
  }
  const float* done = sim->config.autoreset_mode != 0 ? state + (size_t)UPKIE_S_DONE * sim->config.num_envs : nullptr;
  const int status = mpc_launch(mpc, workspace, mpc_x0, act, 2, mpc_contact, done, sim->config.dt, commanded_velocity, nullptr, stream);
  if (status != UPKIE_OK) return fail(sim, status, mpc->error);
  return launch_step<MODE_BASE_VELOCITY>(sim, state, act, obs, reward, terminated, truncated, nullptr, stream, 0,
                                         BaseVelocityPtrs{commanded_velocity, mpc_x0, mpc_contact});
}

// =========================================================== observer pipeline
struct UpkieObservers {
  upkie::ObserverDev dev;
  std::string error;
};

static int observers_fail(UpkieObservers* h, int status, const std::string& msg) {
  if (h) h->error = msg;
  g_create_error = msg;
  return status;
}

// UpkieObserverConfig -> device parameters for a spine period `dt`; false (with
// the reference's FilterError message) when a filter violates cutoff > 2 dt.
static bool convert_observer_config(const UpkieObserverConfig* c, double dt, upkie::ObserverDev* d, std::string* why) {
  if (!(dt > 0.0) || !std::isfinite(dt)) {
    *why = "observers are not configured: dt must be a positive number";
    return false;
  }
  const bool wheels = c->wheel_cutoff_period >= 1e-6;  // WheelContact.cpp:21
  auto nyquist = [&](double cutoff, const char* what) {
    if (cutoff <= 2.0 * dt) {  // low_pass_filter.h:22-30
      *why = std::string("[low_pass_filter] ") + what + " cutoff period " + std::to_string(cutoff) + " s is less than 2 * dt = " +
             std::to_string(2.0 * dt) + " s, causing information loss";
      return false;
    }
    return true;
  };
  if (wheels && !nyquist(c->wheel_cutoff_period, "wheel contact")) return false;
  if (!nyquist(0.01, "upper-leg torque")) return false;
  d->num_envs = c->num_envs;
  d->wheels_configured = wheels ? 1 : 0;
  d->dt = (float)dt;
  d->inv_dt = (float)(1.0 / dt);
  d->wheel_alpha = wheels ? (float)(dt / c->wheel_cutoff_period) : 0.f;
  d->leg_alpha = (float)(dt / 0.01);
  d->upper_leg_torque_threshold = (float)c->upper_leg_torque_threshold;
  d->liftoff_inertia = (float)c->liftoff_inertia;
  d->min_touchdown_acceleration = (float)c->min_touchdown_acceleration;
  d->min_touchdown_torque = (float)c->min_touchdown_torque;
  d->touchdown_inertia = (float)c->touchdown_inertia;
  for (int i = 0; i < 2; ++i) d->signed_radius[i] = (float)c->signed_radius[i];
  for (int i = 0; i < 9; ++i) {
    d->base_to_imu[i] = (float)c->rotation_base_to_imu[i];
    d->ars_to_world[i] = (float)c->rotation_ars_to_world[i];
  }
  return true;
}

extern "C" int upkie_observers_create(const UpkieObserverConfig* c, UpkieObservers** out) {
  if (!c || !out) return observers_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (c->num_envs <= 0) return observers_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "num_envs must be positive");
  upkie::ObserverDev dev;
  std::string why;
  if (!convert_observer_config(c, c->dt, &dev, &why)) return observers_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, why);
  if (upkie_hip_device_count() <= 0) return observers_fail(nullptr, UPKIE_ERR_NO_DEVICE, "no HIP device visible");
  UpkieObservers* h = new (std::nothrow) UpkieObservers();
  if (!h) return observers_fail(nullptr, UPKIE_ERR_INVALID_ARGUMENT, "out of host memory");
  h->dev = dev;
  *out = h;
  return UPKIE_OK;
}

// Spine observers inside the step: one observer cycle per physics substep (the
// spine's own rate under a slower agent, spines/bullet_spine.cpp runs
// nb_substeps cycles per action), observer memory [16][B] owned by the caller.
extern "C" int upkie_sim_attach_observers(UpkieSim* sim, const UpkieObserverConfig* config, float* observer_state) {
  if (!sim) return UPKIE_ERR_INVALID_ARGUMENT;
  if (!config || !observer_state) {
    sim->spine_state = nullptr;
    return UPKIE_OK;
  }
  if (sim->manifold) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, "the Bullet-like contact model does not run the in-step spine observers");
  upkie::ObserverDev dev;
  std::string why;
  if (!convert_observer_config(config, (double)sim->config.h, &dev, &why)) return fail(sim, UPKIE_ERR_INVALID_ARGUMENT, why);
  dev.num_envs = sim->config.num_envs;
  sim->config.spine = dev;
  sim->spine_state = observer_state;
  sim->params_version += 1;
  return UPKIE_OK;
}

extern "C" int upkie_observers_destroy(UpkieObservers* h) {
  delete h;
  return UPKIE_OK;
}

extern "C" const char* upkie_observers_last_error(const UpkieObservers* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

extern "C" int64_t upkie_observers_state_bytes(const UpkieObservers* h) {
  return h ? (int64_t)UPKIE_OBSERVER_STATE_WORDS * h->dev.num_envs * (int64_t)sizeof(float) : 0;
}

extern "C" int upkie_observers_reset(UpkieObservers* h, float* state, const uint8_t* mask, void* stream) {
  if (!h || !state) return observers_fail(h, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  hipLaunchKernelGGL(upkie::observers_reset_kernel, grid_for(h->dev.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream,
                     h->dev.num_envs, state, mask);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? UPKIE_OK : observers_fail(h, UPKIE_ERR_HIP, hipGetErrorString(err));
}

extern "C" int upkie_observers_step(UpkieObservers* h, float* state, const UpkieObserverInput* in, const UpkieObserverOutput* out,
                                    void* stream) {
  if (!h || !state || !in || !out) return observers_fail(h, UPKIE_ERR_INVALID_ARGUMENT, "null argument");
  if (!in->servo && !in->imu_orientation) return observers_fail(h, UPKIE_ERR_INVALID_ARGUMENT, "observation has neither a \"servo\" nor an \"imu\" block");
  if (in->imu_orientation && !in->imu_angular_velocity)  // KeyError in BaseOrientation::read, BaseOrientationTest.cpp:93-98
    return observers_fail(h, UPKIE_ERR_INVALID_ARGUMENT, "imu observation has an orientation but no angular_velocity");
  hipLaunchKernelGGL(upkie::observers_step_kernel, grid_for(h->dev.num_envs), dim3(block_lanes()), 0, (hipStream_t)stream, h->dev, state,
                     *in, *out);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? UPKIE_OK : observers_fail(h, UPKIE_ERR_HIP, hipGetErrorString(err));
}

// ============================================================ rollout consumer
extern "C" int upkie_linear_policy(int32_t num_envs, int32_t obs_dim, int32_t act_dim, const float* obs, const float* weights, const float* bias,
                                  double clip, float* act, void* stream) {
  if (num_envs <= 0 || obs_dim <= 0 || act_dim <= 0) {
    g_create_error = "num_envs, obs_dim and act_dim must be positive";
    return UPKIE_ERR_INVALID_ARGUMENT;
  }
  if (!obs || !weights || !act) {
    g_create_error = "null argument";
    return UPKIE_ERR_INVALID_ARGUMENT;
  }
  if (upkie_hip_device_count() <= 0) {
    g_create_error = "no HIP device visible";
    return UPKIE_ERR_NO_DEVICE;
  }
  hipLaunchKernelGGL(upkie::linear_policy_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, num_envs, obs_dim,
                     act_dim, obs, weights, bias, (float)clip, act);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    g_create_error = hipGetErrorString(err);
    return UPKIE_ERR_HIP;
  }
  return UPKIE_OK;
}

extern "C" int upkie_rollout_gae(int32_t num_steps, int32_t num_envs, const float* rewards, const float* values,
                                 const uint8_t* episode_starts, const float* last_values, const uint8_t* last_dones, double gamma,
                                 double gae_lambda, float* advantages, float* returns, void* stream) {
  if (num_steps <= 0 || num_envs <= 0) {
    g_create_error = "num_steps and num_envs must be positive";
    return UPKIE_ERR_INVALID_ARGUMENT;
  }
  if (!rewards || !values || !episode_starts || !last_values || !last_dones || !advantages || !returns) {
    g_create_error = "null argument";
    return UPKIE_ERR_INVALID_ARGUMENT;
  }
  if (upkie_hip_device_count() <= 0) {
    g_create_error = "no HIP device visible";
    return UPKIE_ERR_NO_DEVICE;
  }
  hipLaunchKernelGGL(upkie::gae_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, num_steps, num_envs,
                     rewards, values, episode_starts, last_values, last_dones, (float)gamma, (float)gae_lambda, advantages, returns);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    g_create_error = hipGetErrorString(err);
    return UPKIE_ERR_HIP;
  }
  return UPKIE_OK;
}

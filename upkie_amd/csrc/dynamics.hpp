// dynamics.hpp -- device-side rigid-body step of one Upkie, one env per lane.
//
// Formulation (fp32, everything expressed in the BASE frame about the base
// origin; the legs are planar chains about the lateral axis so each link's
// orientation relative to the base is one rotation about y):
//   1. single-frame Newton-Euler pass for the bias forces,
//   2. composite-rigid-body pass for the joint-space inertia matrix
//        M = [ Mbb  F_L  F_R ]   Mbb 6x6 (base), F 6x3 per leg,
//            [ F_L' H_L   0  ]   H 3x3 per leg (legs decouple given the base),
//            [ F_R'  0   H_R ]
//   3. leg elimination: A = Mbb - sum_leg F H^-1 F' (6x6), LDL' factorisation,
//      reused for the free acceleration and the 6 contact-row solves,
//   4. two tire/floor contacts x (normal, 2 friction) rows, Delassus matrix,
//      fixed-iteration projected Gauss-Seidel with ERP/CFM from the tire's
//      contact stiffness/damping,
//   5. semi-implicit Euler (velocities first, then positions).
// This replaces what the reference delegates to pybullet.stepSimulation()
// (upkie/envs/backends/pybullet_backend.py:306).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/upkie_hip.h"

// Host+device so that tests can single-step the very same arithmetic on the CPU
// (tests/host_harness.hip); the product only ever calls these from kernels.
#define UPKIE_HD __host__ __device__ __forceinline__

namespace upkie {

// Hardware-rate elementary functions (v_rcp_f32, v_sqrt_f32, v_rsq_f32,
// about 1 ulp) in kernels; libm when the same code is compiled for the host harness.
UPKIE_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.f / x;
#endif
}
UPKIE_HD float fast_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}
UPKIE_HD float fast_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.f / sqrtf(x);
#endif
}
// sin and cos of a joint angle: one Cody-Waite reduction step by pi/2 and the
// Cephes single-precision kernels on [-pi/4, pi/4] (about 1 ulp for the
// |x| < 8 rad a hip + knee angle can reach; the absolute error grows like
// 4e-8 |x| beyond, i.e. stays below the fp32 resolution of x itself).
UPKIE_HD void joint_sincos(float x, float* sn, float* cs) {
  const float kf = rintf(x * 0.63661977236758134f);
  float r = fmaf(kf, -1.5707963705062866f, x);
  r = fmaf(kf, 4.3711390001862426e-08f, r);
  const float r2 = r * r;
  const float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
  const float cp = 1.f - 0.5f * r2 + r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
  const int q = (int)kf & 3;
  const float s0 = (q & 1) ? cp : sp;
  const float c0 = (q & 1) ? sp : cp;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}

struct V3 {
  float x, y, z;
};
UPKIE_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
UPKIE_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
UPKIE_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
UPKIE_HD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
UPKIE_HD float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
UPKIE_HD V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Symmetric 3x3: xx yy zz xy xz yz
struct S3 {
  float xx, yy, zz, xy, xz, yz;
};
UPKIE_HD V3 mul(const S3& I, V3 v) {
  return V3{I.xx * v.x + I.xy * v.y + I.xz * v.z, I.xy * v.x + I.yy * v.y + I.yz * v.z,
            I.xz * v.x + I.yz * v.y + I.zz * v.z};
}
UPKIE_HD S3 operator+(const S3& a, const S3& b) {
  return S3{a.xx + b.xx, a.yy + b.yy, a.zz + b.zz, a.xy + b.xy, a.xz + b.xz, a.yz + b.yz};
}
// Inertia about the frame origin of a body of mass m whose com sits at c,
// given its inertia about the com (parallel axis theorem).
UPKIE_HD S3 shift_to_origin(const S3& Ic, float m, V3 c) {
  return S3{Ic.xx + m * (c.y * c.y + c.z * c.z), Ic.yy + m * (c.x * c.x + c.z * c.z),
            Ic.zz + m * (c.x * c.x + c.y * c.y), Ic.xy - m * c.x * c.y, Ic.xz - m * c.x * c.z,
            Ic.yz - m * c.y * c.z};
}
// Rotate a vector / a symmetric tensor by Ry(angle) given (c, s).
UPKIE_HD V3 rot_y(float c, float s, V3 v) { return V3{c * v.x + s * v.z, v.y, c * v.z - s * v.x}; }
UPKIE_HD S3 rot_y(float c, float s, const S3& I) {
  float cc = c * c, ss = s * s, cs = c * s;
  return S3{cc * I.xx + 2.f * cs * I.xz + ss * I.zz,
            I.yy,
            ss * I.xx - 2.f * cs * I.xz + cc * I.zz,
            c * I.xy + s * I.yz,
            (cc - ss) * I.xz + cs * (I.zz - I.xx),
            c * I.yz - s * I.xy};
}

// Model constants, uniform across lanes (kernel argument => scalar loads).
// Per-leg constants in the order the two-lanes-per-env kernel keeps them in
// registers: a lane reads its own leg's row with a handful of 16-byte loads
// instead of selecting every value out of the left / right pair.
enum LegTableWord {
  LT_MASS = 0,      // 3
  LT_SIGN = 3,      // 3
  LT_COM = 6,       // 3 x 3
  LT_POS = 15,      // 3 x 3
  LT_INERTIA = 24,  // 3 x 6
  LT_DAMPING = 42,  // 3
  LT_EFFORT = 45,
  LT_VELOCITY = 48,
  LT_WHEEL_CENTER = 51,
  LT_LOWER = 54,
  LT_UPPER = 57,
  LT_BOUNDED = 60,  // 3, as 0.f / 1.f
  LT_WORDS = 64
};

// Per-lane constants of the eight-lanes-per-env kernel (octet.hpp), one row per
// (leg, lane of the quad): the lane's body and joint, and the 0 / 1 weights
// that stand for its role.
enum OctTableWord {
  OT_MASS = 0,      // (the right quad's trunk lane: 0)
  OT_COM = 1,       // 3
  OT_INERTIA = 4,   // 6 (the right quad's trunk lane: 0)
  OT_POS = 10,      // 3 joint origin in the parent's frame (trunk: 0)
  OT_SIGN = 13,     // (trunk: 0)
  OT_DAMPING = 14,
  OT_LOWER = 15,
  OT_UPPER = 16,
  OT_BOUNDED = 17,  // 0.f / 1.f
  OT_EFFORT = 18,
  OT_VELOCITY = 19,
  OT_WHEEL_CENTER = 20,  // 3
  OT_WJ = 23,       // 1 on the joint lanes
  OT_E = 24,        // 3: one-hot of the joint index
  OT_W0 = 27,       // 1 on the trunk lane
  OT_W0_ONCE = 28,  // 1 on the left quad's trunk lane
  OT_KEEP_PSI = 29, // 0 on the wheel lane of an axisymmetric wheel
  OT_KL = 30,       // base damping on the lane that owns the real trunk
  OT_KA = 31,
  OT_WORDS = 32
};

struct DevModel {
  alignas(16) float leg_table[2][LT_WORDS];
  alignas(16) float oct_table[8][OT_WORDS];
  float mass[UPKIE_NB];
  float com[UPKIE_NB][3];
  float inertia[UPKIE_NB][6];
  float joint_pos[UPKIE_NJ][3];
  float joint_sign[UPKIE_NJ];  // axis = sign * y
  float joint_lower[UPKIE_NJ];
  float joint_upper[UPKIE_NJ];
  float joint_effort[UPKIE_NJ];
  float joint_velocity[UPKIE_NJ];
  float joint_damping[UPKIE_NJ];
  float wheel_radius;
  float wheel_center[2][3];
  float wheel_base;
  float left_sign;
  float imu_pos[3];
  float rot_base_to_imu[9];
  float gravity;
  float contact_stiffness;
  float contact_damping;
  float friction_mu;
  float friction_cfm;
  float contact_breaking_threshold;
  float base_linear_damping;
  float base_angular_damping;
  float max_joint_velocity;
  float pgs_tolerance;
  int pgs_iterations;
  int wheel_axisymmetric;  // wheel inertia invariant under its own rotation
  int enforce_joint_limits;  // hip/knee limit rows in the constraint solve
};

// Joint position limits, passed by value as a kernel argument (loaded once,
// SGPR-resident for the whole launch) rather than re-read every substep.
struct DevLimits {
  float lower[UPKIE_NJ];
  float upper[UPKIE_NJ];
  int bounded[UPKIE_NJ];  // finite lower and upper
  int enforce;
};

// Physics state of one env held in registers.
struct Phys {
  V3 pos;
  float qw, qx, qy, qz;
  V3 linvel;  // world
  V3 angvel;  // world
  float q[UPKIE_NJ];
  float qd[UPKIE_NJ];
};

// A wavefront-uniform value that must stay in a scalar register from here on (the compiler may otherwise reload it
// from the constant model next to each use).
#if defined(__HIP_DEVICE_COMPILE__)
#define UPKIE_KEEP_IN_SGPR(x) asm volatile("" : "+s"(x))
#else
#define UPKIE_KEEP_IN_SGPR(x) (void)0
#endif

// The step kernels take ~1 KB of arguments by value (DevLimits, DevConfig) and the compiler loads them piece by piece,
// with a wait after each piece. The argument block sits in device memory the CPU wrote over PCIe: its lines are not
// in L2, a dependent miss costs 177 ns against 34 ns for a scalar-cache hit (tools/microbench/kernarg_latency.hip).
// First thing in a kernel: touch every line (20 x 64 B covers the 1272-byte segment) with all the misses in flight
// together; the loads that follow hit the scalar cache.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void warm_kernel_arguments() {
  auto args = __builtin_amdgcn_kernarg_segment_ptr();
  int sink;
  asm volatile(
      "s_load_dword %0, %1, 0x0\n s_load_dword %0, %1, 0x40\n s_load_dword %0, %1, 0x80\n s_load_dword %0, %1, 0xc0\n"
      "s_load_dword %0, %1, 0x100\n s_load_dword %0, %1, 0x140\n s_load_dword %0, %1, 0x180\n s_load_dword %0, %1, 0x1c0\n"
      "s_load_dword %0, %1, 0x200\n s_load_dword %0, %1, 0x240\n s_load_dword %0, %1, 0x280\n s_load_dword %0, %1, 0x2c0\n"
      "s_load_dword %0, %1, 0x300\n s_load_dword %0, %1, 0x340\n s_load_dword %0, %1, 0x380\n s_load_dword %0, %1, 0x3c0\n"
      "s_load_dword %0, %1, 0x400\n s_load_dword %0, %1, 0x440\n s_load_dword %0, %1, 0x480\n s_load_dword %0, %1, 0x4c0\n"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(sink)
      : "s"(args)
      : "memory");
}
// The same for a block of BYTES bytes of constants in device memory (the eight-lane kernels' DevParams): one load per
// 64-byte line, all in flight together, one wait.
template <int BYTES, class Ptr>
__device__ __forceinline__ void warm_constant_block(Ptr block) {
  constexpr int LINES = (BYTES + 63) / 64;
  static_assert(LINES <= 24, "block larger than the unrolled touch sequence");
  // ONE asm statement: the loads and the wait for them together, the sink an early-clobber output of the whole block --
  // as separate statements the compiler saw the sink dead after each touch and could hand the register to a live value
  // that a load still in flight would then overwrite. Offsets past the block's last line touch that line again.
#define UPKIE_LINE(i) "n"(64 * ((i) < LINES ? (i) : LINES - 1))
  int sink;
  asm volatile(
      "s_load_dword %0, %1, %2\n s_load_dword %0, %1, %3\n s_load_dword %0, %1, %4\n s_load_dword %0, %1, %5\n"
      "s_load_dword %0, %1, %6\n s_load_dword %0, %1, %7\n s_load_dword %0, %1, %8\n s_load_dword %0, %1, %9\n"
      "s_load_dword %0, %1, %10\n s_load_dword %0, %1, %11\n s_load_dword %0, %1, %12\n s_load_dword %0, %1, %13\n"
      "s_load_dword %0, %1, %14\n s_load_dword %0, %1, %15\n s_load_dword %0, %1, %16\n s_load_dword %0, %1, %17\n"
      "s_load_dword %0, %1, %18\n s_load_dword %0, %1, %19\n s_load_dword %0, %1, %20\n s_load_dword %0, %1, %21\n"
      "s_load_dword %0, %1, %22\n s_load_dword %0, %1, %23\n s_load_dword %0, %1, %24\n s_load_dword %0, %1, %25\n"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(sink)
      : "s"(block), UPKIE_LINE(0), UPKIE_LINE(1), UPKIE_LINE(2), UPKIE_LINE(3), UPKIE_LINE(4), UPKIE_LINE(5), UPKIE_LINE(6), UPKIE_LINE(7),
        UPKIE_LINE(8), UPKIE_LINE(9), UPKIE_LINE(10), UPKIE_LINE(11), UPKIE_LINE(12), UPKIE_LINE(13), UPKIE_LINE(14), UPKIE_LINE(15),
        UPKIE_LINE(16), UPKIE_LINE(17), UPKIE_LINE(18), UPKIE_LINE(19), UPKIE_LINE(20), UPKIE_LINE(21), UPKIE_LINE(22), UPKIE_LINE(23)
      : "memory");
#undef UPKIE_LINE
}
#else
__device__ inline void warm_kernel_arguments() {}
template <int BYTES, class Ptr>
__device__ inline void warm_constant_block(Ptr) {}
#endif

// Base frame of a substep, shared by every lane mapping: rotation base -> world
// from the quaternion (upkie/utils/rotations.py:52-71), the base velocities in
// base coordinates, world z in base coordinates.
struct BaseFrame {
  float r00, r01, r02, r10, r11, r12, r20, r21, r22;
  V3 vB, wB, nB;
};
UPKIE_HD BaseFrame base_frame(float qw, float qx, float qy, float qz, V3 linvel, V3 angvel) {
  BaseFrame f;
  f.r00 = 1.f - 2.f * (qy * qy + qz * qz); f.r01 = 2.f * (qx * qy - qz * qw); f.r02 = 2.f * (qw * qy + qx * qz);
  f.r10 = 2.f * (qx * qy + qz * qw); f.r11 = 1.f - 2.f * (qx * qx + qz * qz); f.r12 = 2.f * (qy * qz - qx * qw);
  f.r20 = 2.f * (qx * qz - qy * qw); f.r21 = 2.f * (qy * qz + qx * qw); f.r22 = 1.f - 2.f * (qx * qx + qy * qy);
  f.vB = v3(f.r00 * linvel.x + f.r10 * linvel.y + f.r20 * linvel.z, f.r01 * linvel.x + f.r11 * linvel.y + f.r21 * linvel.z,
            f.r02 * linvel.x + f.r12 * linvel.y + f.r22 * linvel.z);
  f.wB = v3(f.r00 * angvel.x + f.r10 * angvel.y + f.r20 * angvel.z, f.r01 * angvel.x + f.r11 * angvel.y + f.r21 * angvel.z,
            f.r02 * angvel.x + f.r12 * angvel.y + f.r22 * angvel.z);
  f.nB = v3(f.r20, f.r21, f.r22);
  return f;
}
// End of a substep, shared by every lane mapping (semi-implicit Euler: the new
// velocities move the positions): n = new base velocity in base coordinates
// (linear 0-2, angular 3-5) -> world velocities, position, orientation.
// dq = (cos(half), sin(half) w / |w|) with sin(half) / |w| = h/2 sinc(half); half
// is a few 1e-3 at most in practice: series to x^8 (exact to fp32 below 0.5 rad),
// libm beyond; q <- dq * q (world-frame angular velocity), renormalised.
UPKIE_HD void integrate_base(const BaseFrame& f, float n0, float n1, float n2, float n3, float n4, float n5, float h, V3& pos, float& qw,
                             float& qx, float& qy, float& qz, V3& linvel, V3& angvel) {
  linvel = v3(f.r00 * n0 + f.r01 * n1 + f.r02 * n2, f.r10 * n0 + f.r11 * n1 + f.r12 * n2, f.r20 * n0 + f.r21 * n1 + f.r22 * n2);
  angvel = v3(f.r00 * n3 + f.r01 * n4 + f.r02 * n5, f.r10 * n3 + f.r11 * n4 + f.r12 * n5, f.r20 * n3 + f.r21 * n4 + f.r22 * n5);
  pos = pos + h * linvel;
  const float wn = fast_sqrt(dot(angvel, angvel));
  const float half = 0.5f * h * wn;
  float ch, k;
  if (half < 0.5f) {
    const float x2 = half * half;
    k = 0.5f * h * (1.f + x2 * (-1.f / 6.f + x2 * (1.f / 120.f + x2 * (-1.f / 5040.f + x2 * (1.f / 362880.f)))));
    ch = 1.f + x2 * (-0.5f + x2 * (1.f / 24.f + x2 * (-1.f / 720.f + x2 * (1.f / 40320.f))));
  } else {
    float sh;
    sincosf(half, &sh, &ch);
    k = sh * fast_rcp(wn);
  }
  const float dw = ch, dx = k * angvel.x, dy = k * angvel.y, dz = k * angvel.z;
  const float nw = dw * qw - dx * qx - dy * qy - dz * qz;
  const float nx = dw * qx + dx * qw + dy * qz - dz * qy;
  const float ny = dw * qy - dx * qz + dy * qw + dz * qx;
  const float nz = dw * qz + dx * qy - dy * qx + dz * qw;
  const float inv = fast_rsqrt(nw * nw + nx * nx + ny * ny + nz * nz);
  qw = nw * inv; qx = nx * inv; qy = ny * inv; qz = nz * inv;
}

// 6x6 symmetric LDL' factorisation, unit lower L (15) + inverse pivots (6).
struct Ldl6 {
  float l10, l20, l21, l30, l31, l32, l40, l41, l42, l43, l50, l51, l52, l53, l54;
  float i0, i1, i2, i3, i4, i5;
};

UPKIE_HD void ldl6_factor(const float (&A)[21], Ldl6& f) {
  // A packed lower by rows: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) ...
  float d0 = A[0];
  f.i0 = fast_rcp(d0);
  float a10 = A[1], a20 = A[3], a30 = A[6], a40 = A[10], a50 = A[15];
  f.l10 = a10 * f.i0; f.l20 = a20 * f.i0; f.l30 = a30 * f.i0; f.l40 = a40 * f.i0; f.l50 = a50 * f.i0;
  float d1 = A[2] - f.l10 * a10;
  f.i1 = fast_rcp(d1);
  float a21 = A[4] - f.l20 * a10, a31 = A[7] - f.l30 * a10, a41 = A[11] - f.l40 * a10, a51 = A[16] - f.l50 * a10;
  f.l21 = a21 * f.i1; f.l31 = a31 * f.i1; f.l41 = a41 * f.i1; f.l51 = a51 * f.i1;
  float d2 = A[5] - f.l20 * a20 - f.l21 * a21;
  f.i2 = fast_rcp(d2);
  float a32 = A[8] - f.l30 * a20 - f.l31 * a21, a42 = A[12] - f.l40 * a20 - f.l41 * a21,
        a52 = A[17] - f.l50 * a20 - f.l51 * a21;
  f.l32 = a32 * f.i2; f.l42 = a42 * f.i2; f.l52 = a52 * f.i2;
  float d3 = A[9] - f.l30 * a30 - f.l31 * a31 - f.l32 * a32;
  f.i3 = fast_rcp(d3);
  float a43 = A[13] - f.l40 * a30 - f.l41 * a31 - f.l42 * a32,
        a53 = A[18] - f.l50 * a30 - f.l51 * a31 - f.l52 * a32;
  f.l43 = a43 * f.i3; f.l53 = a53 * f.i3;
  float d4 = A[14] - f.l40 * a40 - f.l41 * a41 - f.l42 * a42 - f.l43 * a43;
  f.i4 = fast_rcp(d4);
  float a54 = A[19] - f.l50 * a40 - f.l51 * a41 - f.l52 * a42 - f.l53 * a43;
  f.l54 = a54 * f.i4;
  float d5 = A[20] - f.l50 * a50 - f.l51 * a51 - f.l52 * a52 - f.l53 * a53 - f.l54 * a54;
  f.i5 = fast_rcp(d5);
}

UPKIE_HD void ldl6_solve(const Ldl6& f, float (&x)[6]) {
  // forward: L y = b
  x[1] -= f.l10 * x[0];
  x[2] -= f.l20 * x[0] + f.l21 * x[1];
  x[3] -= f.l30 * x[0] + f.l31 * x[1] + f.l32 * x[2];
  x[4] -= f.l40 * x[0] + f.l41 * x[1] + f.l42 * x[2] + f.l43 * x[3];
  x[5] -= f.l50 * x[0] + f.l51 * x[1] + f.l52 * x[2] + f.l53 * x[3] + f.l54 * x[4];
  x[0] *= f.i0; x[1] *= f.i1; x[2] *= f.i2; x[3] *= f.i3; x[4] *= f.i4; x[5] *= f.i5;
  // backward: L' x = y
  x[4] -= f.l54 * x[5];
  x[3] -= f.l43 * x[4] + f.l53 * x[5];
  x[2] -= f.l32 * x[3] + f.l42 * x[4] + f.l52 * x[5];
  x[1] -= f.l21 * x[2] + f.l31 * x[3] + f.l41 * x[4] + f.l51 * x[5];
  x[0] -= f.l10 * x[1] + f.l20 * x[2] + f.l30 * x[3] + f.l40 * x[4] + f.l50 * x[5];
}

// Everything one leg contributes to the base-level system.
struct Leg {
  // kinematics (base frame)
  V3 o[3];  // joint origins: hip, knee, wheel
  // columns F_k = I^c_k S_k, k = hip, knee, wheel: linear (f) and angular (n)
  float F[3][6];
  float Hinv[6];  // symmetric inverse of the 3x3 leg block: 00 11 22 01 02 12
  // D = F Hinv (6x3): rows 0-2 linear, 3-5 angular
  float D[6][3];
  float bias[3];  // joint-space bias of the leg
  float sgn[3];   // axis signs
};

// Composite inertia about the base origin: mass, first moment, second moment.
struct Composite {
  float m;
  V3 h;
  S3 I;
};

// One leg: kinematics, single-frame Newton-Euler (bias) and composite pass.
// `body0` is the index of the thigh body, `joint0` of the hip joint.
// Adds the leg's composite inertia to `total` and its bias wrench about the
// base origin to (bias_f, bias_n).
// Parameters of one leg as leg_pass sees them. LegOfModel reads the uniform
// model (scalar loads, leg chosen at compile time); LegRegs is a per-lane copy
// of one leg's constants for the two-lanes-per-env mapping, where the leg a
// lane owns varies across lanes.
// Inertial records of the 7 composite bodies of ONE env (mass, centre of mass
// in the body frame, inertia about it): what randomize_inertias leaves behind
// (pybullet_backend.py:571-601), loaded once per step from the
// body_inertials[UPKIE_NB * UPKIE_INERTIAL_WORDS][B] buffer and held in
// registers. Only the randomised kernel variants carry it.
struct BodyInertials {
  float m[UPKIE_NB];
  float c[UPKIE_NB][3];
  float I[UPKIE_NB][6];
};

template <class ModelT>
UPKIE_HD void body_inertials_of_model(const ModelT& M, BodyInertials& bi) {
#pragma unroll
  for (int b = 0; b < UPKIE_NB; ++b) {
    bi.m[b] = M.mass[b];
#pragma unroll
    for (int d = 0; d < 3; ++d) bi.c[b][d] = M.com[b][d];
#pragma unroll
    for (int d = 0; d < 6; ++d) bi.I[b][d] = M.inertia[b][d];
  }
}

// records: this env's column of the buffer (stride = number of envs)
UPKIE_HD void load_body_inertials(const float* records, size_t stride, BodyInertials& bi) {
#pragma unroll
  for (int b = 0; b < UPKIE_NB; ++b) {
    const float* r = records + (size_t)(UPKIE_INERTIAL_WORDS * b) * stride;
    bi.m[b] = r[0];
#pragma unroll
    for (int d = 0; d < 3; ++d) bi.c[b][d] = r[(size_t)(1 + d) * stride];
#pragma unroll
    for (int d = 0; d < 6; ++d) bi.I[b][d] = r[(size_t)(4 + d) * stride];
  }
}

template <class ModelT>
struct LegOfModel {
  const ModelT& M;
  const BodyInertials* bi;  // inertials of this env or nullptr (the model's)
  int body0, joint0;
  UPKIE_HD float mass(int k) const { return bi ? bi->m[body0 + k] : M.mass[body0 + k]; }
  UPKIE_HD V3 com(int k) const {
    const int b = body0 + k;
    return bi ? v3(bi->c[b][0], bi->c[b][1], bi->c[b][2]) : v3(M.com[b][0], M.com[b][1], M.com[b][2]);
  }
  UPKIE_HD S3 inertia(int k) const {
    const int b = body0 + k;
    return bi ? S3{bi->I[b][0], bi->I[b][1], bi->I[b][2], bi->I[b][3], bi->I[b][4], bi->I[b][5]}
              : S3{M.inertia[b][0], M.inertia[b][1], M.inertia[b][2], M.inertia[b][3], M.inertia[b][4], M.inertia[b][5]};
  }
  UPKIE_HD V3 joint_pos(int k) const { return v3(M.joint_pos[joint0 + k][0], M.joint_pos[joint0 + k][1], M.joint_pos[joint0 + k][2]); }
  UPKIE_HD float sign(int k) const { return M.joint_sign[joint0 + k]; }
};

struct LegRegs {
  float m[3], c[3][3], I[3][6], p[3][3], sg[3];
  UPKIE_HD float mass(int k) const { return m[k]; }
  UPKIE_HD V3 com(int k) const { return v3(c[k][0], c[k][1], c[k][2]); }
  UPKIE_HD S3 inertia(int k) const { return S3{I[k][0], I[k][1], I[k][2], I[k][3], I[k][4], I[k][5]}; }
  UPKIE_HD V3 joint_pos(int k) const { return v3(p[k][0], p[k][1], p[k][2]); }
  UPKIE_HD float sign(int k) const { return sg[k]; }
};

template <class LegParams>
UPKIE_HD void leg_pass(const LegParams& P, bool wheel_axisymmetric, const float (&q)[3], const float (&qd)[3], V3 w0, V3 gn,
                       Leg& L, Composite& total, V3& bias_f, V3& bias_n) {
  float cs[3], sn[3];
  float psi = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    L.sgn[k] = P.sign(k);
    psi = fmaf(L.sgn[k], q[k], psi);
    if (k == 2 && wheel_axisymmetric) {
      cs[k] = 1.f;
      sn[k] = 0.f;
    } else {
      joint_sincos(psi, &sn[k], &cs[k]);
    }
  }
  // joint origins
  L.o[0] = P.joint_pos(0);
  L.o[1] = L.o[0] + rot_y(cs[0], sn[0], P.joint_pos(1));
  L.o[2] = L.o[1] + rot_y(cs[1], sn[1], P.joint_pos(2));

  // outward pass: velocities and velocity-product accelerations
  V3 w = w0, al = v3(0.f, 0.f, 0.f), ao = v3(0.f, 0.f, 0.f), oprev = v3(0.f, 0.f, 0.f);
  V3 Nb[3], fb[3];  // per-body wrench about the base origin
  float mb[3];
  V3 cb[3];
  S3 Ib[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float m = P.mass(k);
    V3 r = L.o[k] - oprev;
    // origin acceleration uses the PARENT's omega/alpha
    ao = ao + cross(al, r) + cross(w, cross(w, r));
    float sq = L.sgn[k] * qd[k];
    // omega_p x a = sign * (omega_p x y) = sign * (-wz, 0, wx)
    al = al + sq * v3(-w.z, 0.f, w.x);
    w.y += sq;
    V3 rc = rot_y(cs[k], sn[k], P.com(k));
    V3 c = L.o[k] + rc;
    S3 Ic = rot_y(cs[k], sn[k], P.inertia(k));
    V3 ac = ao + cross(al, rc) + cross(w, cross(w, rc));
    V3 f = m * (ac + gn);
    V3 n = mul(Ic, al) + cross(w, mul(Ic, w));
    fb[k] = f;
    Nb[k] = n + cross(c, f);
    mb[k] = m;
    cb[k] = c;
    Ib[k] = shift_to_origin(Ic, m, c);
    oprev = L.o[k];
  }
  // inward pass: composite wrench / inertia, joint-space projections
  Composite C{0.f, v3(0.f, 0.f, 0.f), S3{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
  V3 fc = v3(0.f, 0.f, 0.f), Nc = v3(0.f, 0.f, 0.f);
  float H[3][3];
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    fc = fc + fb[k];
    Nc = Nc + Nb[k];
    C.m += mb[k];
    C.h = C.h + mb[k] * cb[k];
    C.I = C.I + Ib[k];
    float s = L.sgn[k];
    V3 o = L.o[k];
    // S = [a; o x a], a = s*y: o x y = (-o.z, 0, o.x)
    V3 oxa = s * v3(-o.z, 0.f, o.x);
    // bias_j = a.N + (o x a).f
    L.bias[k] = s * Nc.y + dot(oxa, fc);
    // F = I^c S: f = m (o x a) + a x h ; n = I a + h x (o x a)
    V3 axh = s * v3(C.h.z, 0.f, -C.h.x);
    V3 Ff = C.m * oxa + axh;
    V3 Fn = s * v3(C.I.xy, C.I.yy, C.I.yz) + cross(C.h, oxa);
    L.F[k][0] = Ff.x; L.F[k][1] = Ff.y; L.F[k][2] = Ff.z;
    L.F[k][3] = Fn.x; L.F[k][4] = Fn.y; L.F[k][5] = Fn.z;
  }
  // H_jk = S_j . F_k for j <= k (ancestor), symmetric
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float s = L.sgn[j];
    V3 o = L.o[j];
    V3 oxa = s * v3(-o.z, 0.f, o.x);
#pragma unroll
    for (int k = j; k < 3; ++k) {
      float hjk = s * L.F[k][4] + oxa.x * L.F[k][0] + oxa.z * L.F[k][2];
      H[j][k] = hjk;
      H[k][j] = hjk;
    }
  }
  // inverse of the SPD 3x3 block through its Cholesky factor
  {
    float i00 = fast_rsqrt(H[0][0]);
    float l10 = H[1][0] * i00, l20 = H[2][0] * i00;
    float i11 = fast_rsqrt(H[1][1] - l10 * l10);
    float l21 = (H[2][1] - l20 * l10) * i11;
    float i22 = fast_rsqrt(H[2][2] - l20 * l20 - l21 * l21);
    // Linv (lower): m00 m10 m11 m20 m21 m22
    float m00 = i00, m11 = i11, m22 = i22;
    float m10 = -l10 * m00 * i11;
    float m21 = -l21 * m11 * i22;
    float m20 = -(l20 * m00 + l21 * m10) * i22;
    // Hinv = Linv' Linv
    L.Hinv[0] = m00 * m00 + m10 * m10 + m20 * m20;
    L.Hinv[1] = m11 * m11 + m21 * m21;
    L.Hinv[2] = m22 * m22;
    L.Hinv[3] = m10 * m11 + m20 * m21;
    L.Hinv[4] = m20 * m22;
    L.Hinv[5] = m21 * m22;
  }
  // D = F Hinv
  {
    const float* h = L.Hinv;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float f0 = L.F[0][r], f1 = L.F[1][r], f2 = L.F[2][r];
      L.D[r][0] = f0 * h[0] + f1 * h[3] + f2 * h[4];
      L.D[r][1] = f0 * h[3] + f1 * h[1] + f2 * h[5];
      L.D[r][2] = f0 * h[4] + f1 * h[5] + f2 * h[2];
    }
  }
  total.m += C.m;
  total.h = total.h + C.h;
  total.I = total.I + C.I;
  bias_f = bias_f + fc;
  bias_n = bias_n + Nc;
}

// Factored system: solve M x = b for b = (base 6, left 3, right 3).
struct System {
  Ldl6 A;
  Leg leg[2];
};

// x overwrites b. `has_leg[l]` false means the leg part of b is zero.
template <bool LEFT, bool RIGHT, class SystemT>
UPKIE_HD void system_solve(const SystemT& S, float (&bb)[6], float (&bl)[3], float (&br)[3]) {
  // y = b_base - D_L b_L - D_R b_R
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    float y = bb[r];
    if (LEFT) y -= S.leg[0].D[r][0] * bl[0] + S.leg[0].D[r][1] * bl[1] + S.leg[0].D[r][2] * bl[2];
    if (RIGHT) y -= S.leg[1].D[r][0] * br[0] + S.leg[1].D[r][1] * br[1] + S.leg[1].D[r][2] * br[2];
    bb[r] = y;
  }
  ldl6_solve(S.A, bb);
  // x_leg = Hinv b_leg - D' x_base
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    float(&b)[3] = l == 0 ? bl : br;
    const auto& G = S.leg[l];
    const float* h = G.Hinv;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if ((l == 0 && LEFT) || (l == 1 && RIGHT)) {
      x0 = h[0] * b[0] + h[3] * b[1] + h[4] * b[2];
      x1 = h[3] * b[0] + h[1] * b[1] + h[5] * b[2];
      x2 = h[4] * b[0] + h[5] * b[1] + h[2] * b[2];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      x0 -= G.D[r][0] * bb[r];
      x1 -= G.D[r][1] * bb[r];
      x2 -= G.D[r][2] * bb[r];
    }
    b[0] = x0; b[1] = x1; b[2] = x2;
  }
}

// The two LATERAL friction rows (row 2 of each tire) of a sweep, together. The tires share one axle direction, so the
// two rows push the robot along the same line (with opposite signs on this model: a25 < 0): their 2 x 2 block
// [[a22, a25], [a25, a55]] is nearly singular -- only friction_cfm and the yaw lever arm of the wheel base separate
// the rows -- and swept one at a time they converge like (a / (a + cfm))^2 per sweep, hundreds of sweeps. Here the
// pair is solved EXACTLY given every other row: the minimiser of 1/2 x'Ax - r'x over the box |x2| <= lim2,
// |x5| <= lim5 -- the unconstrained 2 x 2 solution when it lies inside the box, otherwise the optimal point of the
// four edges (one row on a bound, the other row solved and clamped; corners included). Round 2 solved the pair as a FREE
// pair in sum / difference coordinates and clamped afterwards, and left the weak coordinate out of the convergence
// test: with one row on its bound (a tire sliding sideways, or lifted: lim = 0) that is not the solution of the box
// problem -- the converged impulses violated the complementarity conditions by up to O(1) of the velocity scale on
// the systems of robots tumbling under examples/pybullet/torque_balancing.py's law, 1 % of them ran into the sweep
// cap -- and the oracle did the same. r2, r5: right-hand sides minus the contributions of every other row.
// Returns the largest change of the two impulses.
UPKIE_HD float lateral_pair_sweep(float a22, float a25, float a55, float r2, float r5, float lim2, float lim5, float& l2, float& l5) {
  const float det = a22 * a55 - a25 * a25;
  const float idet = fast_rcp(det), i22 = fast_rcp(a22), i55 = fast_rcp(a55);
  const float u2 = (a55 * r2 - a25 * r5) * idet, u5 = (a22 * r5 - a25 * r2) * idet;
  // (friction_cfm keeps det / (a22 a55) at a few per cent; below fp32's resolution of the difference the free solution is not trusted)
  const bool inside = det > 1e-5f * a22 * a55 && fabsf(u2) <= lim2 && fabsf(u5) <= lim5;
  float best2 = u2, best5 = u5;
  if (!inside) {
    // The solution sits on an edge: one row on a bound, the other row solved for it and clamped. It is the candidate
    // whose bounded row pushes OUTWARD (gradient of 1/2 x'Ax - r'x against the bound): chosen by that sign, the
    // smallest violation winning. (The four candidates used to be compared by the value of the objective: with the
    // solution within 1e-4 of a bound the values differ by 1e-8 of themselves, fp32 cannot tell them apart, and the
    // pair flipped between the bound and the interior point from one sweep to the next, 1.6e-4 apart, for ever --
    // every system that ended at the sweep cap on the device did that: profiles/r03_sweep_cap_systems.txt.)
    float best = 3.4e38f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x2, x5, outward;
      if (e < 2) {
        x2 = e == 0 ? -lim2 : lim2;
        x5 = fminf(fmaxf((r5 - a25 * x2) * i55, -lim5), lim5);
        const float g = (a22 * x2 + a25 * x5 - r2) * i22;  // in impulse: how far a free row 2 would move from here
        outward = e == 0 ? g : -g;                         // (on the lower bound the gradient must not be negative)
      } else {
        x5 = e == 2 ? -lim5 : lim5;
        x2 = fminf(fmaxf((r2 - a25 * x5) * i22, -lim2), lim2);
        const float g = (a55 * x5 + a25 * x2 - r5) * i55;
        outward = e == 2 ? g : -g;
      }
      const float violation = fmaxf(-outward, 0.f);
      if (violation < best) {
        best = violation;
        best2 = x2;
        best5 = x5;
      }
    }
  }
  const float change = fmaxf(fabsf(best2 - l2), fabsf(best5 - l5));
  l2 = best2;
  l5 = best5;
  return change;
}

// Projected Gauss-Seidel on the 6 x 6 contact system of the two tires (rows 0-2
// left: normal, rolling, lateral; rows 3-5 right), shared by every lane
// mapping. A packed lower by rows, `lam` comes in as the projected direct
// solution (the warm start). Normals of both wheels first, then the rolling
// rows, then the two lateral ones together (lateral_pair_sweep; with a tire off
// the floor that is its partner's row alone, see contact_pgs6). Each env stops
// on its own criterion: lanes leave the loop one by one.
template <class ModelT>
UPKIE_HD int contact_pgs6_sweeps(const ModelT& M, const float (&A)[21], const float (&rhs)[6], float (&lam)[6]) {
  const float mu = M.friction_mu;
  // read before the loop and held in a scalar register: left to the compiler, the tolerance is re-fetched from the
  // model inside every sweep, a scalar-cache round trip the lone wavefront waits out each time
  float tolerance = M.pgs_tolerance;
  UPKIE_KEEP_IN_SGPR(tolerance);
  // Rows scaled by their diagonal once, so that a row update is x_r = b_r - sum_{c != r} a_rc lam_c: five multiply-adds
  // (the unscaled form costs eight instructions a row; a launch with many skidding robots is bound by these sweeps).
  // The lateral rows 2 and 5 are swept together from the unscaled entries (lateral_pair_sweep).
  float a[6][6], b[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    if ((r % 3) == 2) continue;
    const float inv = fast_rcp(A[r * (r + 1) / 2 + r]);
    b[r] = rhs[r] * inv;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int hi = r > c ? r : c, lo = r > c ? c : r;
      a[r][c] = A[hi * (hi + 1) / 2 + lo] * inv;
    }
  }
  // A rolling row that slides sits on its bound and moves with the normal impulse, lam_t = +-mu lam_n. Stepping the
  // normal row as if the rolling one stayed put (it follows only in the next pass) makes the pair converge like
  // 0.36^sweeps on a robot that skids (mu = 1: 9-13 sweeps for 1e-5). With the step scaled to the diagonal of the row
  // solved for both, A_nn +- mu A_nt, the same fixed point is reached in about half the sweeps
  // (profiles/r02_sweep_tolerance.txt). slide[tire][0 / 1]: A_nn / that diagonal for a rolling impulse > 0 / < 0.
  float slide[2][2];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const float coupling = mu * a[3 * w][3 * w + 1];
    slide[w][0] = 1.f + coupling > 0.25f ? fast_rcp(1.f + coupling) : 1.f;
    slide[w][1] = 1.f - coupling > 0.25f ? fast_rcp(1.f - coupling) : 1.f;
  }
  int sweeps = 0;
  for (int it = 0; it < M.pgs_iterations; ++it) {
    sweeps = it + 1;
    float change = 0.f, scale = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const bool is_normal = (r % 3) == 0;
        if (is_normal != (pass == 0) || (r % 3) == 2) continue;
        float x = b[r];
#pragma unroll
        for (int c = 0; c < 6; ++c)
          if (c != r) x = fmaf(-a[r][c], lam[c], x);
        if (is_normal) {
          const int w = r / 3, t = r + 1;
          const bool sliding = lam[r] > 0.f && fabsf(lam[t]) >= mu * lam[r];
          const float step = sliding ? (lam[t] > 0.f ? slide[w][0] : slide[w][1]) : 1.f;
          x = fmaxf(fmaf(x - lam[r], step, lam[r]), 0.f);
        } else {
          const float lim = mu * lam[3 * (r / 3)];
          x = fminf(fmaxf(x, -lim), lim);
        }
        change = fmaxf(change, fabsf(x - lam[r]));
        scale = fmaxf(scale, fabsf(x));
        lam[r] = x;
      }
    }
    {  // the two lateral rows together, after the rolling ones (lateral_pair_sweep)
      float r2 = rhs[2], r5 = rhs[5];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (c == 2 || c == 5) continue;
        r2 -= A[(c > 2 ? c * (c + 1) / 2 + 2 : 2 * 3 / 2 + c)] * lam[c];
        r5 -= A[5 * 6 / 2 + c] * lam[c];
      }
      change = fmaxf(change, lateral_pair_sweep(A[5], A[17], A[20], r2, r5, mu * lam[0], mu * lam[3], lam[2], lam[5]));
      scale = fmaxf(scale, fmaxf(fabsf(lam[2]), fabsf(lam[5])));
    }
    if (change <= tolerance * scale) break;
  }
  return sweeps;  // for the census (upkie_sim_set_census)
}
// ONE loop for every env, whether one tire touches or both (round 4). A tire without a contact point has identity rows,
// zero right-hand sides and no coupling (X = 0): the lateral pair's 2 x 2 block is then diagonal, the box problem
// separates, and lateral_pair_sweep returns the touching tire's row solved and clamped -- what the row-by-row update
// of that row gives, at the same place of the sweep (after the rolling row), so the rule of the fp64 checker (a single
// lateral row is swept like any friction row) holds unchanged. Until then envs with one tire on the floor ran a loop of
// their own that swept rows 2 and 5 one by one, and a wavefront that held both kinds of env -- most of them, when robots
// tumble -- ran the two loops one after the other, each to the largest count among its envs: a sweep level cost a
// launch of the C5 share 2.8 us where one loop's 151 instructions account for 1.4 (tools/archive/c5_sweep_cost.py).
template <class ModelT>
UPKIE_HD int contact_pgs6(const ModelT& M, const float (&A)[21], const float (&rhs)[6], float (&lam)[6]) {
  // both tires leaving the floor (neither normal row asks for an impulse): lam = 0 is the solution, what the sweeps
  // return after one pass over the projected zeros (a tire without a contact point has an identity row and rhs 0)
  if (rhs[0] <= 0.f && rhs[3] <= 0.f) {
#pragma unroll
    for (int r = 0; r < 6; ++r) lam[r] = 0.f;
    return 0;
  }
  return contact_pgs6_sweeps(M, A, rhs, lam);
}

// An active-set solve of the same problem, tried before the sweeps (round 5). A robot that skids does so for many
// substeps in a row with the same rows on their bounds, so the warm start -- the previous substep's impulses, or the
// projected direct solution -- usually names the right set: which tires push, which friction rows sit on +-mu lam_n.
// With the set fixed the conditions are LINEAR: a free row keeps its equation, a friction row on a bound is replaced by
// lam_t -+ mu lam_n = 0, the rows of a tire that does not push by lam = 0; one 6 x 6 elimination (no pivoting: the
// normal rows' pivots are A_nn > 0, a bounded row's is the 1 +- mu A_nt / A_nn the sweeps' slide step divides by)
// gives the impulses, and the answer is ACCEPTED only if every condition of the problem holds for it -- equations of
// the free rows to the sweeps' tolerance (which also catches an inaccurate elimination: NaN and infinities fail every
// test), normals >= 0, free friction rows inside their bounds, bounded rows pushed outward, idle tires separating.
// Otherwise the violated conditions name the next set, up to three solves in all; then the sweeps run from the
// unchanged warm start, as before. On 4096 systems of robots skidding and tumbling under torque_balancing.py's law
// (tests/test_contact_active_set.py, the host build of this code): 87.7 % accepted at the first set, 99.2 % by the
// second, 99.9 % by the third (one tire on the floor: 99.5 %); accepted impulses within 3e-6 (p99) of the oracle's
// converged sweeps, contact velocities within 1.6e-6 (worst) -- tighter than the sweeps' own stopping rule.
// THIS function works on the gathered 6 x 6 system: the statement of the method, what the host tests hold to the
// oracle's systems, and the first device version (an attempt: ~390 instructions, 39.5 -> 36.5 us on the C5 share under
// that law). What the eight-lane kernel runs is the same method one row per lane, without gathering anything --
// oct_active_set, octet.hpp: ~150 issue slots an attempt, 31.8 us -- and the one- and two-lane kernels keep the sweeps
// alone (contact_sweeps_warm says why). profiles/r05_active_set.txt. Returns the attempt that was accepted (1 .. 3) or 0.
#if !defined(UPKIE_ACTIVE_SET_ATTEMPTS)  // (an A/B switch: 2 and 4 attempts are in profiles/r05_active_set.txt)
#define UPKIE_ACTIVE_SET_ATTEMPTS 3
#endif
template <class ModelT>
UPKIE_HD int contact_active_set6(const ModelT& M, const float (&A)[21], const float (&rhs)[6], float (&lam)[6]) {
  const float mu = M.friction_mu;
  const float tolerance = M.pgs_tolerance;
  // The set, as numbers (the build and the tests below are arithmetic and selects, no branches: a wavefront runs this for
  // the one env in eight that needs it, and scalar branching cost more issue slots than the elimination itself):
  // push[w] 1 / 0: the tire pushes / all three of its rows are lam = 0; side[r] of a friction row: 0 free, +-1 on the
  // upper / lower bound (0 on the rows of an idle tire).
  float push[2], side[6];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const int n = 3 * w;
    // A warm start without an impulse on this tire (the direct solution pulled there and was projected to zero) while
    // the normal row asks for one: the tire is landing or barely loaded, and what barely loaded tires do under a spinning
    // wheel is slide -- both friction rows on the bound their right-hand side points to.
    const bool landing = !(lam[n] > 0.f) && rhs[n] > 0.f;
    push[w] = lam[n] > 0.f || landing ? 1.f : 0.f;
    side[n] = 0.f;
#pragma unroll
    for (int r = n + 1; r < n + 3; ++r) {
      const float towards = landing ? rhs[r] : lam[r];
      side[r] = !landing && fabsf(lam[r]) < mu * lam[n] ? 0.f : (towards > 0.f ? push[w] : -push[w]);
    }
  }
  const float vs = fmaxf(fmaxf(fmaxf(fabsf(rhs[0]), fabsf(rhs[1])), fmaxf(fabsf(rhs[2]), fabsf(rhs[3]))), fmaxf(fabsf(rhs[4]), fabsf(rhs[5])));
  const float vtol = tolerance * vs;
  for (int attempt = 1; attempt <= UPKIE_ACTIVE_SET_ATTEMPTS; ++attempt) {
    // rows: keep[r] = 1 the row's own equation, 0 replaced by x_r - side mu x_n = 0 (bounded) or x_r = 0 (idle tire)
    float m[6][6], b[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int w = r / 3, n = 3 * w;
      const float keep = r == n ? push[w] : (side[r] == 0.f ? push[w] : 0.f);
#pragma unroll
      for (int c = 0; c < 6; ++c) m[r][c] = keep * A[r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r];
      m[r][r] += 1.f - keep;
      if (r != n) m[r][n] = fmaf(-mu, side[r], m[r][n]);  // (side != 0 only on a replaced row, whose entry here is 0)
      b[r] = keep * rhs[r];
    }
    float inv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      inv[k] = fast_rcp(m[k][k]);
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        const float f = m[i][k] * inv[k];
#pragma unroll
        for (int j = k + 1; j < 6; ++j) m[i][j] = fmaf(-f, m[k][j], m[i][j]);
        b[i] = fmaf(-f, b[k], b[i]);
      }
    }
    float x[6];
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      float acc = b[k];
#pragma unroll
      for (int j = k + 1; j < 6; ++j) acc = fmaf(-m[k][j], x[j], acc);
      x[k] = acc * inv[k];
    }
    // what the impulses do: v = A x - rhs, every row, from the untouched system
    float v[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float acc = -rhs[r];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc = fmaf(A[r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r], x[c], acc);
      v[r] = acc;
    }
    const float xs = fmaxf(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))), fmaxf(fabsf(x[4]), fabsf(x[5])));
    const float xtol = tolerance * xs;
    // Every condition as an excess over what it allows (<= 0: met); `worst` is the largest. The conditions that name the next
    // set are kept apart: `enters` (an idle tire would sink into the floor), `pulls` (a pushing tire pulls), `leaves[r]` (a free
    // friction row outside its bound), `returns[r]` (a bounded row that wants back inside, held ten times tighter than the
    // equations: the two tires' lateral rows are nearly parallel, and a set with the wrong one of them on its bound can
    // meet every equation to 1e-5 with impulses 13 % off).
    float worst = -vtol;  // the kept rows' own equations, all at once: largest |v_r| of a row that kept its equation
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int w = r / 3, n = 3 * w;
      const float keep = r == n ? push[w] : (side[r] == 0.f ? push[w] : 0.f);
      worst = fmaxf(worst, fmaf(keep, fabsf(v[r]), -vtol));
    }
    float push_next[2], side_next[6];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int n = 3 * w;
      const bool pushing = push[w] != 0.f;
      const float enters = pushing ? -1.f : -v[n] - vtol;
      const float pulls = pushing ? -x[n] - xtol : -1.f;
      worst = fmaxf(worst, fmaxf(enters, pulls));
      const bool flips = enters > 0.f || pulls > 0.f;
      push_next[w] = flips ? 1.f - push[w] : push[w];
      side_next[n] = 0.f;
      const float lim = mu * x[n];
#pragma unroll
      for (int r = n + 1; r < n + 3; ++r) {
        const bool free_row = side[r] == 0.f;
        const float leaves = pushing && free_row ? fabsf(x[r]) - lim - xtol : -1.f;
        const float returns = free_row ? -1.f : v[r] * side[r] - 0.1f * vtol;
        worst = fmaxf(worst, fmaxf(leaves, returns));
        const float out = x[r] > 0.f ? 1.f : -1.f;
        side_next[r] = flips ? 0.f : (leaves > 0.f ? out : (returns > 0.f ? 0.f : side[r]));
      }
    }
    // (fmaxf DROPS a NaN operand: `worst` and `xs` alone let a partly NaN elimination through -- a pivot 1 -+ mu A_nt / A_nn that
    // is exactly 0, inf - inf; the sum propagates it: ADVICE r5)
    float every = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) every += fabsf(x[r]) + fabsf(v[r]);
    if (worst <= 0.f && xs <= 3.0e38f && every < 3.0e38f) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int n = 3 * w;
        const float ln = push[w] != 0.f ? fmaxf(x[n], 0.f) : 0.f, lim = mu * ln;
        lam[n] = ln;
        // (a bounded row sits ON its bound: the elimination returns side * mu * x_n to rounding only)
        lam[n + 1] = side[n + 1] != 0.f ? side[n + 1] * lim : fminf(fmaxf(x[n + 1], -lim), lim);
        lam[n + 2] = side[n + 2] != 0.f ? side[n + 2] * lim : fminf(fmaxf(x[n + 2], -lim), lim);
      }
      return attempt;
    }
#pragma unroll
    for (int w = 0; w < 2; ++w) push[w] = push_next[w];
#pragma unroll
    for (int r = 0; r < 6; ++r) side[r] = side_next[r];
  }
  return 0;
}

// The contact impulses of a system whose direct solution is not admissible, from the warm start `lam`: nothing to do when
// both tires unload, else the active-set solve above, else the sweeps. Returns the sweeps run, or -1 / -2: the active
// set that was accepted (for the census).
template <class ModelT>
UPKIE_HD int contact_solve6(const ModelT& M, const float (&A)[21], const float (&rhs)[6], float (&lam)[6]) {
  if (rhs[0] <= 0.f && rhs[3] <= 0.f) {
#pragma unroll
    for (int r = 0; r < 6; ++r) lam[r] = 0.f;
    return 0;
  }
  const int accepted = contact_active_set6(M, A, rhs, lam);
  if (accepted) return -accepted;
  return contact_pgs6_sweeps(M, A, rhs, lam);
}

// What the Gauss-Seidel sweeps of one substep hand to the next substep of the SAME env.step(): the impulses they ended on
// and with which tires on the floor (0: that substep did not sweep, 1 / 2: it did, with one / both tires touching). A
// robot that skids or tumbles does so for many substeps in a row and its contact state changes little from one
// millisecond to the next: started from there the sweeps reach the same fixed point in fewer passes (they stop on their
// own convergence test either way). Every lane mapping and the fp64 checker follow this rule (round 4; round 3: the
// eight-lane kernel only).
struct SweepWarmStart {
  float lam[6];
  int swept;
};

// The contact impulses of a substep whose direct solution `lam` left the friction box / pulls: projected Gauss-Seidel
// sweeps (contact_pgs6) from the projected direct solution, or from the previous substep's impulses (`warm`); both tires
// leaving the floor (neither normal row asks for an impulse) is lam = 0 without a sweep. Returns the sweeps run.
// ACTIVE_SET (round 6): the active-set solve in front of the sweeps, from the same warm start -- what the eight-lane kernel does
// (oct_active_set) -- for the instantiations whose envs skid and tumble as a matter of course: the two-lane UpkieServos kernels
// (batches of 8192 .. 32768 Servos envs). Returns minus the accepted attempt then, as contact_solve6.
template <bool ACTIVE_SET = false, class ModelT>
UPKIE_HD int contact_sweeps_warm(const ModelT& M, const float (&A)[21], const float (&rhs)[6], float (&lam)[6], bool both, SweepWarmStart* warm) {
  if (rhs[0] <= 0.f && rhs[3] <= 0.f) {
#pragma unroll
    for (int r = 0; r < 6; ++r) lam[r] = 0.f;
    if (warm) warm->swept = 0;
    return 0;
  }
  const int state = both ? 2 : 1;
  if (warm && warm->swept == state) {
#pragma unroll
    for (int r = 0; r < 6; ++r) lam[r] = warm->lam[r];
  }
  const float mu = M.friction_mu;
#pragma unroll
  for (int w = 0; w < 2; ++w) lam[3 * w] = fmaxf(lam[3 * w], 0.f);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    if ((r % 3) == 0) continue;
    const float lim = mu * lam[3 * (r / 3)];
    lam[r] = fminf(fmaxf(lam[r], -lim), lim);
  }
  // (the sweeps alone, except where ACTIVE_SET says otherwise: the one- and two-lane kernels that come through here serve batches
  // of 20 000 envs and more, and the active-set solve inlined into them -- 800 instructions of a path one substep in five hundred
  // takes -- cost their COMMON path 3-6 %: register allocation of a body that much larger, 25.9 -> 27.4 us at 32768 envs,
  // 494 -> 523 us at a million; as a call it cost the dense kernels 40 %. The eight-lane kernel, whose common path it leaves
  // alone, has it.)
  if (ACTIVE_SET) {
    const int accepted = contact_active_set6(M, A, rhs, lam);  // (lam: the projected warm start; unchanged when no set is accepted)
    if (accepted) {
      if (warm) {
#pragma unroll
        for (int r = 0; r < 6; ++r) warm->lam[r] = lam[r];
        warm->swept = state;
      }
      return -accepted;
    }
  }
  const int sweeps = contact_pgs6(M, A, rhs, lam);
  if (warm) {
#pragma unroll
    for (int r = 0; r < 6; ++r) warm->lam[r] = lam[r];
    warm->swept = state;
  }
  return sweeps;
}

// Rare path shared by both lane mappings: some hip/knee joint sits at its
// position limit. Contacts and limits are solved together as ONE system of ten
// rows with a fixed layout, so that every index below is a compile-time
// constant and the whole solve lives in registers (no scratch arrays):
//   rows 0-2 left tire (normal, rolling, lateral), rows 3-5 right tire,
//   rows 6-9 limits of left hip, left knee, right hip, right knee.
// A row that does not exist this substep (tire out of range, joint inside its
// limits) is masked: unit diagonal, zero couplings, zero right-hand side, so it
// leaves the other rows' arithmetic untouched and gets lam = 0. Same rows,
// ordering and numerics as the 6-row path: direct LDL' solve, accepted when
// feasible, otherwise projected and used as the warm start of projected
// Gauss-Seidel sweeps (normals, then frictions, then limits). Row data come
// reduced onto the base: Jt (6) and the leg part (3). On return
// (tb, tl, tr) += J' lam.
constexpr int kRows = 10;
// Joint position limits (URDF revolute limits; Bullet: btMultiBodyJointLimitConstraint, SURVEY App. B.1 [third party, restated]):
// one unilateral row per bound, on the joint's velocity towards the free side. A joint still `gap` short of its stop may
// close that gap within the substep (sign v >= -gap / h, what Bullet's row allows a separated pair), a joint `pen` beyond it
// is pushed back with ERP 0.2 (sign v >= 0.2 pen / h); the two meet continuously at the stop. The row is LISTED while the
// joint could reach the stop within the substep: gap <= (|qd| + 0.2 max_joint_velocity) h -- its own speed plus what a substep can
// add to it (20 rad/s with Bullet's default clamp of 100: the servos' full torque adds 0.3-1.6 rad/s per substep, a landing a
// few); rows that cannot bind are left out, and a joint that would need more than that to cross its stop overshoots by the
// difference and is pushed back, as every joint was until round 6. (First stated with the clamp itself, 0.1 rad: exact, but
// tumbling robots then took the ten-row solve for rows that never bound -- 0.8 us per step of the C5 share under
// torque_balancing.py's law, profiles/r06_ab_c5_changes.txt.)
// Until round 6 a row existed only at or beyond the stop (gap <= 0), with the ERP bias alone. A joint RESTING on its stop
// sits within rounding of gap = 0, so whether its row existed in a substep was decided by the last bit of q -- in fp32 the
// joint stayed put, in the fp64 checker it alternated between a substep with the row and a substep of free acceleration
// into the stop (one-step parity test of round 6, joints held at their stops: 1 % of the env-steps off by 1e-3 rad under
// that rule; gpurun_out/parity_windows/one_step_joint_stops.json). With the gap-aware row the answer no longer depends on
// which side of its stop a resting joint is rounded to.
// Returns the row's sign (+1: lower bound, -1: upper bound, 0: no row) and its bias.
UPKIE_HD float joint_limit_row(bool bounded, float q, float lower, float upper, float zone, float ih, float& bias) {
  float sign = 0.f, pen = 0.f;
  if (bounded && q - lower <= zone) {
    sign = 1.f;
    pen = lower - q;
  } else if (bounded && upper - q <= zone) {
    sign = -1.f;
    pen = q - upper;
  }
  bias = (pen > 0.f ? 0.2f : 1.f) * pen * ih;  // Bullet's default ERP beyond the stop; the gap may close within the substep before it
  return sign;
}
UPKIE_HD float joint_limit_reach(float qd, float max_joint_velocity, float h) { return (fabsf(qd) + 0.2f * max_joint_velocity) * h; }
UPKIE_HD bool joint_limit_near(bool bounded, float q, float lower, float upper, float zone) {
  return bounded && (q - lower <= zone || upper - q <= zone);
}

UPKIE_HD constexpr int row_leg(int r) { return r < 3 ? 0 : (r < 6 ? 1 : (r < 8 ? 0 : 1)); }
UPKIE_HD constexpr int row_kind(int r) { return r >= 6 ? 2 : (r % 3 == 0 ? 0 : 1); }  // 0 normal, 1 friction, 2 limit
UPKIE_HD constexpr int sym(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

template <class ModelT>
UPKIE_HD void limit_path(const ModelT& M, const System& S, const float (&lower)[UPKIE_NJ], const float (&upper)[UPKIE_NJ],
                         const int (&bounded)[UPKIE_NJ], const float (&q)[UPKIE_NJ], const float (&qd)[UPKIE_NJ],
                         const float (&Jt6)[6][6], const float (&Jb)[6][6], const float (&Jl6)[6][3], const float (&vnow)[6],
                         const float (&dists)[2], const bool (&active)[2], float cfm, float erp, float ih, float vmax, float h, const float (&rt)[6],
                         float (&tb)[6], float (&tl)[3], float (&tr)[3], float (&contact_lam)[6]) {
  float J[kRows][6], Ll[kRows][3], vn[kRows], bias[kRows], cf[kRows];
  bool on[kRows];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int w = r / 3, k = r % 3;
#pragma unroll
    for (int c = 0; c < 6; ++c) J[r][c] = Jt6[r][c];
#pragma unroll
    for (int j = 0; j < 3; ++j) Ll[r][j] = Jl6[r][j];
    on[r] = active[w];
    vn[r] = vnow[r];
    cf[r] = k == 0 ? cfm : M.friction_cfm;
    bias[r] = k == 0 ? (dists[w] <= 0.f ? erp * (-dists[w]) * ih : -dists[w] * ih) : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = i < 2 ? i : i + 1;  // joints 0, 1, 3, 4
    const int w = j / 3, kk = j % 3, r = 6 + i;
    float row_bias;
    const float sign = joint_limit_row(bounded[j] != 0, q[j], lower[j], upper[j], joint_limit_reach(qd[j], vmax, h), ih, row_bias);
    const Leg& G = S.leg[w];
    // J = sign * e_j: no base part, reduced row = -D_w J_leg
#pragma unroll
    for (int c = 0; c < 6; ++c) J[r][c] = -sign * G.D[c][kk];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) Ll[r][jj] = jj == kk ? sign : 0.f;
    on[r] = sign != 0.f;
    vn[r] = sign * qd[j];
    cf[r] = 0.f;
    bias[r] = row_bias;
  }

  // A = J M^-1 J' + CFM column by column (packed lower), rhs = -(v + J M^-1 t) + bias
  float A[kRows * (kRows + 1) / 2], rhs[kRows], lam[kRows];
#pragma unroll
  for (int b = 0; b < kRows; ++b) {
    float y[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) y[c] = J[b][c];
    ldl6_solve(S.A, y);
    const float* hv = S.leg[row_leg(b)].Hinv;
    const float k0 = hv[0] * Ll[b][0] + hv[3] * Ll[b][1] + hv[4] * Ll[b][2];
    const float k1 = hv[3] * Ll[b][0] + hv[1] * Ll[b][1] + hv[5] * Ll[b][2];
    const float k2 = hv[4] * Ll[b][0] + hv[5] * Ll[b][1] + hv[2] * Ll[b][2];
    float vf = vn[b];
#pragma unroll
    for (int c = 0; c < 6; ++c) vf = fmaf(y[c], rt[c], vf);
    if (row_leg(b) == 0)
      vf += k0 * tl[0] + k1 * tl[1] + k2 * tl[2];
    else
      vf += k0 * tr[0] + k1 * tr[1] + k2 * tr[2];
    rhs[b] = on[b] ? -vf + bias[b] : 0.f;
    lam[b] = 0.f;
#pragma unroll
    for (int a = b; a < kRows; ++a) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc = fmaf(J[a][c], y[c], acc);
      if (row_leg(a) == row_leg(b)) acc += Ll[a][0] * k0 + Ll[a][1] * k1 + Ll[a][2] * k2;
      if (a == b)
        acc = on[a] ? acc + cf[a] : 1.f;
      else
        acc = (on[a] && on[b]) ? acc : 0.f;
      A[sym(a, b)] = acc;
    }
  }

  // LDL' of the 10 x 10 system (unit lower L stored in place of A's strict
  // lower part, reciprocal pivots in idg), then two triangular solves
  float Lf[kRows * (kRows + 1) / 2], idg[kRows];
  bool spd = true;
#pragma unroll
  for (int j = 0; j < kRows; ++j) {
    float d = A[sym(j, j)];
#pragma unroll
    for (int c = 0; c < j; ++c) d -= Lf[sym(j, c)] * Lf[sym(j, c)] * Lf[sym(c, c)];
    spd = spd && d > 0.f;
    Lf[sym(j, j)] = d;
    idg[j] = fast_rcp(fmaxf(d, 1e-30f));
#pragma unroll
    for (int i = j + 1; i < kRows; ++i) {
      float v = A[sym(i, j)];
#pragma unroll
      for (int c = 0; c < j; ++c) v -= Lf[sym(i, c)] * Lf[sym(j, c)] * Lf[sym(c, c)];
      Lf[sym(i, j)] = v * idg[j];
    }
  }
  const float mu = M.friction_mu;
  bool need_pgs = !spd;
  if (spd) {
    float z[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      float v = rhs[i];
#pragma unroll
      for (int c = 0; c < i; ++c) v -= Lf[sym(i, c)] * z[c];
      z[i] = v;
    }
#pragma unroll
    for (int i = kRows - 1; i >= 0; --i) {
      float v = z[i] * idg[i];
#pragma unroll
      for (int c = i + 1; c < kRows; ++c) v -= Lf[sym(c, i)] * lam[c];
      lam[i] = v;
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r)
      if (row_kind(r) != 1 && lam[r] < 0.f) {
        lam[r] = 0.f;
        need_pgs = true;
      }
#pragma unroll
    for (int r = 0; r < kRows; ++r)
      if (row_kind(r) == 1) {
        const float lim = mu * lam[3 * (r / 3)];
        if (lam[r] < -lim) { lam[r] = -lim; need_pgs = true; }
        if (lam[r] > lim) { lam[r] = lim; need_pgs = true; }
      }
  }
  if (need_pgs) {
    float idiag[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) idiag[r] = fast_rcp(A[sym(r, r)]);
    float tolerance = M.pgs_tolerance;  // held in a scalar register across the sweeps (see contact_pgs6_sweeps)
    UPKIE_KEEP_IN_SGPR(tolerance);
    for (int it = 0; it < M.pgs_iterations; ++it) {
      float change = 0.f, scale = 0.f;
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        const bool pair = on[2] && on[5];  // both tires touch: their lateral rows are swept together
        if (pass == 2 && pair) {
          float r2 = rhs[2], r5 = rhs[5];
#pragma unroll
          for (int b = 0; b < kRows; ++b) {
            if (b == 2 || b == 5) continue;
            r2 -= A[sym(2, b)] * lam[b];
            r5 -= A[sym(5, b)] * lam[b];
          }
          change = fmaxf(change, lateral_pair_sweep(A[sym(2, 2)], A[sym(5, 2)], A[sym(5, 5)], r2, r5, mu * lam[0], mu * lam[3], lam[2], lam[5]));
          scale = fmaxf(scale, fmaxf(fabsf(lam[2]), fabsf(lam[5])));
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          if (row_kind(r) != pass || ((r == 2 || r == 5) && pair)) continue;
          float al = 0.f;
#pragma unroll
          for (int b = 0; b < kRows; ++b) al = fmaf(A[sym(r, b)], lam[b], al);
          float x = lam[r] + (rhs[r] - al) * idiag[r];
          if (row_kind(r) == 1) {
            const float lim = mu * lam[3 * (r / 3)];
            x = fminf(fmaxf(x, -lim), lim);
          } else {
            x = fmaxf(x, 0.f);
          }
          x = on[r] ? x : 0.f;
          change = fmaxf(change, fabsf(x - lam[r]));
          scale = fmaxf(scale, fabsf(x));
          lam[r] = x;
        }
      }
      if (change <= tolerance * scale) break;
    }
  }
  // t += J' lam (limit rows have no base part)
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    contact_lam[r] = lam[r];
#pragma unroll
    for (int c = 0; c < 6; ++c) tb[c] = fmaf(Jb[r][c], lam[r], tb[c]);
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (row_leg(r) == 0)
        tl[j] = fmaf(Ll[r][j], lam[r], tl[j]);
      else
        tr[j] = fmaf(Ll[r][j], lam[r], tr[j]);
    }
  }
}

// General constraint solve (contacts + active hip/knee limits, up to 10 rows),
// the rare path: taken only by envs with a joint at its limit. Same rows,
// ordering and numerics as the 6-row path: rows = per touching wheel (normal,
// rolling, lateral), then one row per limited joint; direct Cholesky solve,
// projected Gauss-Seidel warm-started from its projection when infeasible.
// Row data come reduced onto the base: Jt (6), leg index, leg part (3).
// On return (tb, tl, tr) += J' lam. Arrays are indexed dynamically (scratch).
struct GeneralRows {
  float Jt[10][6];
  float Jb[10][6];
  float Jl[10][3];
  float vnow[10];
  float cfm[10];
  float bias[10];  // positional term of the rhs (ERP push / gap closing / limit error)
  int leg[10];
  int kind[10];  // 0 contact normal, 1 friction, 2 joint limit
  int normal_row[10];
  int n;
};

// What the general solve indexes dynamically. Private arrays of this kind live in scratch memory (one-lane kernels of very
// large batches: limit_path_scratch); the eight-lane kernel keeps one per env of the wavefront in LDS (LimitWorkspace,
// octet.hpp): private scratch of a path no env of a Pendulum batch ever takes is still allocated for every wavefront slot,
// and at 135 KB per wavefront the runtime's scratch limit admitted one wavefront per SIMD, no more.
struct GeneralWork {
  float A[10][10], Y[10][6], K[10][3], rhs[10], lam[10];
  float L[55];  // Cholesky factor, lower triangle packed by rows
  float yv[10];
};
UPKIE_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

template <class ModelT, class SystemT>
UPKIE_HD void general_constraint_solve(const ModelT& M, const SystemT& S, const GeneralRows& R, const float (&rt)[6], float (&tb)[6],
                                       float (&tl)[3], float (&tr)[3], float (&lam_out)[10], GeneralWork& W) {
  const int n = R.n;
  auto& A = W.A;
  auto& Y = W.Y;
  auto& K = W.K;
  auto& rhs = W.rhs;
  auto& lam = W.lam;
  for (int b = 0; b < n; ++b) {
    float y[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) y[c] = R.Jt[b][c];
    ldl6_solve(S.A, y);
    const float* hv = R.leg[b] == 0 ? S.leg[0].Hinv : S.leg[1].Hinv;
    const float k0 = hv[0] * R.Jl[b][0] + hv[3] * R.Jl[b][1] + hv[4] * R.Jl[b][2];
    const float k1 = hv[3] * R.Jl[b][0] + hv[1] * R.Jl[b][1] + hv[5] * R.Jl[b][2];
    const float k2 = hv[4] * R.Jl[b][0] + hv[5] * R.Jl[b][1] + hv[2] * R.Jl[b][2];
    float vf = R.vnow[b];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      Y[b][c] = y[c];
      vf = fmaf(y[c], rt[c], vf);
    }
    K[b][0] = k0; K[b][1] = k1; K[b][2] = k2;
    if (R.leg[b] == 0)
      vf += k0 * tl[0] + k1 * tl[1] + k2 * tl[2];
    else
      vf += k0 * tr[0] + k1 * tr[1] + k2 * tr[2];
    rhs[b] = -vf + R.bias[b];
  }
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc = fmaf(R.Jt[a][c], Y[b][c], acc);
      if (R.leg[a] == R.leg[b]) acc += R.Jl[a][0] * K[b][0] + R.Jl[a][1] * K[b][1] + R.Jl[a][2] * K[b][2];
      if (a == b) acc += R.cfm[a];
      A[a][b] = acc;
    }
  // Cholesky A = L L' in place of a copy, then two triangular solves
  auto& L = W.L;
  bool spd = true;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      float s = A[i][j];
      for (int c = 0; c < j; ++c) s -= L[tri(i, c)] * L[tri(j, c)];
      if (i == j) {
        if (!(s > 0.f)) spd = false;
        L[tri(i, i)] = fast_sqrt(fmaxf(s, 1e-30f));
      } else {
        L[tri(i, j)] = s * fast_rcp(L[tri(j, j)]);
      }
    }
  const float mu = M.friction_mu;
  bool need_pgs = !spd;
  for (int i = 0; i < n; ++i) lam[i] = 0.f;
  if (spd) {
    auto& yv = W.yv;
    for (int i = 0; i < n; ++i) {
      float s = rhs[i];
      for (int c = 0; c < i; ++c) s -= L[tri(i, c)] * yv[c];
      yv[i] = s * fast_rcp(L[tri(i, i)]);
    }
    for (int i = n - 1; i >= 0; --i) {
      float s = yv[i];
      for (int c = i + 1; c < n; ++c) s -= L[tri(c, i)] * lam[c];
      lam[i] = s * fast_rcp(L[tri(i, i)]);
    }
    for (int r = 0; r < n; ++r)
      if (R.kind[r] != 1 && lam[r] < 0.f) {
        lam[r] = 0.f;
        need_pgs = true;
      }
    for (int r = 0; r < n; ++r)
      if (R.kind[r] == 1) {
        const float lim = mu * lam[R.normal_row[r]];
        if (lam[r] < -lim) { lam[r] = -lim; need_pgs = true; }
        if (lam[r] > lim) { lam[r] = lim; need_pgs = true; }
      }
  }
  // rows 2 and 5 are the two lateral rows when both tires touch (contact rows come first, in wheel order)
  const bool pair = n >= 6 && R.kind[2] == 1 && R.kind[5] == 1 && R.normal_row[2] == 0 && R.normal_row[5] == 3;
  for (int it = 0; need_pgs && it < M.pgs_iterations; ++it) {
    float change = 0.f, scale = 0.f;
    for (int pass = 0; pass < 3; ++pass) {
      if (pass == 2 && pair) {  // lateral_pair_sweep, as in the register-resident solve
        float r2 = rhs[2], r5 = rhs[5];
        for (int b = 0; b < n; ++b) {
          if (b == 2 || b == 5) continue;
          r2 -= A[2][b] * lam[b];
          r5 -= A[5][b] * lam[b];
        }
        float l2 = lam[2], l5 = lam[5];
        change = fmaxf(change, lateral_pair_sweep(A[2][2], A[5][2], A[5][5], r2, r5, mu * lam[0], mu * lam[3], l2, l5));
        scale = fmaxf(scale, fmaxf(fabsf(l2), fabsf(l5)));
        lam[2] = l2;
        lam[5] = l5;
      }
      for (int r = 0; r < n; ++r) {
        if (R.kind[r] != pass || ((r == 2 || r == 5) && pair)) continue;
        float al = 0.f;
        for (int b = 0; b < n; ++b) al = fmaf(A[r][b], lam[b], al);
        float x = lam[r] + (rhs[r] - al) * fast_rcp(A[r][r]);
        if (R.kind[r] == 1) {
          const float lim = mu * lam[R.normal_row[r]];
          x = fminf(fmaxf(x, -lim), lim);
        } else {
          x = fmaxf(x, 0.f);
        }
        change = fmaxf(change, fabsf(x - lam[r]));
        scale = fmaxf(scale, fabsf(x));
        lam[r] = x;
      }
    }
    if (change <= M.pgs_tolerance * scale) break;
  }
  for (int r = 0; r < n; ++r) {
    lam_out[r] = lam[r];
#pragma unroll
    for (int c = 0; c < 6; ++c) tb[c] = fmaf(R.Jb[r][c], lam[r], tb[c]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (R.leg[r] == 0)
        tl[j] = fmaf(R.Jl[r][j], lam[r], tl[j]);
      else
        tr[j] = fmaf(R.Jl[r][j], lam[r], tr[j]);
    }
  }
}

// The same row list under the BULLET-LIKE contact specification (bullet_like.hpp), for the eight-lane kernel's joint-stop path
// (round 6: until then a Bullet-like substep with a joint at its stop took the default model's solve above on that mapping):
// the rows arrive as the eight-lane path has them -- contact rows of the touching tires in wheel order with the friction
// directions of the DEFAULT basis (rolling, lateral) and no friction CFM, then the limit rows (joint_limit_row) --; the dense
// system is built as above, each tire's friction pair is rotated into the specification's directions (along / across the
// sliding velocity of the point at the free velocity, btPlaneSpace1 when it does not slide: Q, the rotation
// oct_bullet_like_solve applies to its six rows), and the published FIXED number of sequential-impulse sweeps runs on it:
// limit rows, normal rows, then each point's friction pair together, projected onto the cone -- from the normals' warm
// start 0.85 x the impulse applied in the previous step. Row for row bullet_like_contacts' general path (which iterates on
// velocities; the same impulses after every sweep in exact arithmetic). `first_row[w]`: index of tire w's normal row or -1;
// `plane[w]`: the fallback directions (a . t1, a . t2, b . t1, b . t2 of btPlaneSpace1's a, b against the default t1, t2);
// `applied[w]`: the tire's applied normal impulse, in / out. On return (tb, tl, tr) += J' lam.
template <class ModelT, class SystemT>
UPKIE_HD void general_constraint_solve_bullet_like(const ModelT& M, const SystemT& S, const GeneralRows& R, const float (&rt)[6], float (&tb)[6],
                                                   float (&tl)[3], float (&tr)[3], const int (&first_row)[2], const float (&plane)[2][4],
                                                   float (&applied)[2], GeneralWork& W) {
  const int n = R.n;
  auto& A = W.A;
  auto& Y = W.Y;
  auto& K = W.K;
  auto& rhs = W.rhs;
  auto& lam = W.lam;
  for (int b = 0; b < n; ++b) {
    float y[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) y[c] = R.Jt[b][c];
    ldl6_solve(S.A, y);
    const float* hv = R.leg[b] == 0 ? S.leg[0].Hinv : S.leg[1].Hinv;
    const float k0 = hv[0] * R.Jl[b][0] + hv[3] * R.Jl[b][1] + hv[4] * R.Jl[b][2];
    const float k1 = hv[3] * R.Jl[b][0] + hv[1] * R.Jl[b][1] + hv[5] * R.Jl[b][2];
    const float k2 = hv[4] * R.Jl[b][0] + hv[5] * R.Jl[b][1] + hv[2] * R.Jl[b][2];
    float vf = R.vnow[b];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      Y[b][c] = y[c];
      vf = fmaf(y[c], rt[c], vf);
    }
    K[b][0] = k0; K[b][1] = k1; K[b][2] = k2;
    if (R.leg[b] == 0)
      vf += k0 * tl[0] + k1 * tl[1] + k2 * tl[2];
    else
      vf += k0 * tr[0] + k1 * tr[1] + k2 * tr[2];
    rhs[b] = -vf + R.bias[b];
    lam[b] = 0.f;
  }
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc = fmaf(R.Jt[a][c], Y[b][c], acc);
      if (R.leg[a] == R.leg[b]) acc += R.Jl[a][0] * K[b][0] + R.Jl[a][1] * K[b][1] + R.Jl[a][2] * K[b][2];
      if (a == b) acc += R.cfm[a];
      A[a][b] = acc;
    }
  // each tire's friction pair into the specification's directions: A <- Q A Q', rhs <- Q rhs
  float Q[2][4];
  for (int w = 0; w < 2; ++w) {
    const int i = first_row[w] + 1, j = first_row[w] + 2;
    Q[w][0] = 1.f; Q[w][1] = 0.f; Q[w][2] = 0.f; Q[w][3] = 1.f;
    if (first_row[w] < 0) continue;
    const float vt1 = -rhs[i], vt2 = -rhs[j];  // (friction rows carry no bias: -rhs is the point's free velocity along t1 / t2)
    const float lat2 = vt1 * vt1 + vt2 * vt2;
    if (lat2 > 1.1920929e-07f) {  // SIMD_EPSILON
      const float inv = 1.f / sqrtf(lat2);
      const float c = vt1 * inv, sn = vt2 * inv;
      Q[w][0] = c; Q[w][1] = sn; Q[w][2] = sn; Q[w][3] = -c;
    } else {
      Q[w][0] = plane[w][0]; Q[w][1] = plane[w][1]; Q[w][2] = plane[w][2]; Q[w][3] = plane[w][3];
    }
    for (int c = 0; c < n; ++c) {
      const float x = A[i][c], y = A[j][c];
      A[i][c] = Q[w][0] * x + Q[w][1] * y;
      A[j][c] = Q[w][2] * x + Q[w][3] * y;
    }
    for (int r = 0; r < n; ++r) {
      const float x = A[r][i], y = A[r][j];
      A[r][i] = Q[w][0] * x + Q[w][1] * y;
      A[r][j] = Q[w][2] * x + Q[w][3] * y;
    }
    const float x = rhs[i], y = rhs[j];
    rhs[i] = Q[w][0] * x + Q[w][1] * y;
    rhs[j] = Q[w][2] * x + Q[w][3] * y;
    lam[first_row[w]] = 0.85f * applied[w];  // m_warmstartingFactor
  }
  const float mu = M.friction_mu;
  for (int it = 0; it < M.pgs_iterations; ++it) {
    for (int pass = 0; pass < 2; ++pass) {  // joint limits, then normals
      for (int r = 0; r < n; ++r) {
        if (R.kind[r] != (pass == 0 ? 2 : 0)) continue;
        float al = 0.f;
        for (int b = 0; b < n; ++b) al = fmaf(A[r][b], lam[b], al);
        const float x = lam[r] + (rhs[r] - al) * fast_rcp(A[r][r]);
        lam[r] = x < 0.f ? 0.f : x;
      }
    }
    for (int w = 0; w < 2; ++w) {  // the two friction rows of a point together, projected onto the cone
      if (first_row[w] < 0) continue;
      const int i = first_row[w] + 1, j = first_row[w] + 2;
      float a1 = 0.f, a2 = 0.f;
      for (int b = 0; b < n; ++b) {
        a1 = fmaf(A[i][b], lam[b], a1);
        a2 = fmaf(A[j][b], lam[b], a2);
      }
      float x1 = lam[i] + (rhs[i] - a1) * fast_rcp(A[i][i]), x2 = lam[j] + (rhs[j] - a2) * fast_rcp(A[j][j]);
      const float lim = mu * lam[first_row[w]], n2 = x1 * x1 + x2 * x2;
      if (n2 > lim * lim) {
        const float sc = lim * fast_rsqrt(n2);
        x1 *= sc;
        x2 *= sc;
      }
      lam[i] = x1;
      lam[j] = x2;
    }
  }
  // back to the default basis (lam = Q' lam'), the applied normal impulses, and J' lam
  for (int w = 0; w < 2; ++w) {
    if (first_row[w] < 0) {
      applied[w] = 0.f;
      continue;
    }
    const int i = first_row[w] + 1, j = first_row[w] + 2;
    applied[w] = lam[first_row[w]];
    const float x = lam[i], y = lam[j];
    lam[i] = Q[w][0] * x + Q[w][2] * y;
    lam[j] = Q[w][1] * x + Q[w][3] * y;
  }
  for (int r = 0; r < n; ++r) {
#pragma unroll
    for (int c = 0; c < 6; ++c) tb[c] = fmaf(R.Jb[r][c], lam[r], tb[c]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (R.leg[r] == 0)
        tl[j] = fmaf(R.Jl[r][j], lam[r], tl[j]);
      else
        tr[j] = fmaf(R.Jl[r][j], lam[r], tr[j]);
    }
  }
}

// Scratch-memory variant of limit_path() (same rows, same numerics) for the
// register-capped build that runs two waves per SIMD on very large batches:
// there the register-resident solve above would spill into the path every env
// takes. Builds the row list (contact rows of the touching wheels in
// wheel order, then one row per limited joint in joint order) from data of
// BOTH legs and runs the general solver; (tb, tl, tr) += J' lam.
template <class ModelT>
UPKIE_HD void limit_path_scratch(const ModelT& M, const System& S, const float (&lower)[UPKIE_NJ], const float (&upper)[UPKIE_NJ],
                         const int (&bounded)[UPKIE_NJ], const float (&q)[UPKIE_NJ], const float (&qd)[UPKIE_NJ],
                         const float (&Jt)[6][6], const float (&Jb)[6][6], const float (&Jl)[6][3], const float (&vnow)[6],
                         const float (&dists)[2], const bool (&active)[2], float cfm, float erp, float ih, float vmax, float h, const float (&rt)[6],
                         float (&tb)[6], float (&tl)[3], float (&tr)[3], float (&contact_lam)[6]) {
  GeneralRows R;
  R.n = 0;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    if (!active[w]) continue;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = 3 * w + k, i = R.n;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        R.Jt[i][c] = Jt[r][c];
        R.Jb[i][c] = Jb[r][c];
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) R.Jl[i][j] = Jl[r][j];
      R.vnow[i] = vnow[r];
      R.leg[i] = w;
      R.kind[i] = k == 0 ? 0 : 1;
      R.normal_row[i] = i - k;
      R.cfm[i] = k == 0 ? cfm : M.friction_cfm;
      R.bias[i] = k == 0 ? (dists[w] <= 0.f ? erp * (-dists[w]) * ih : -dists[w] * ih) : 0.f;
      R.n = i + 1;
    }
  }
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    float row_bias;
    const float sign = joint_limit_row(bounded[j] != 0, q[j], lower[j], upper[j], joint_limit_reach(qd[j], vmax, h), ih, row_bias);
    if (sign != 0.f) {
      const int i = R.n, w = j / 3, kk = j % 3;
      const Leg& G = S.leg[w];
      // J = sign * e_j: no base part, reduced row = -D_w J_leg
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        R.Jb[i][c] = 0.f;
        R.Jt[i][c] = -sign * G.D[c][kk];
      }
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) R.Jl[i][jj] = jj == kk ? sign : 0.f;
      R.vnow[i] = sign * qd[j];
      R.leg[i] = w;
      R.kind[i] = 2;
      R.normal_row[i] = i;
      R.cfm[i] = 0.f;
      R.bias[i] = row_bias;
      R.n = i + 1;
    }
  }
  float lam_rows[10];
  GeneralWork W;
  general_constraint_solve(M, S, R, rt, tb, tl, tr, lam_rows, W);
  {  // contact impulses back in the fixed (wheel, row) layout
    int i = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int k = 0; k < 3; ++k) contact_lam[3 * w + k] = active[w] ? lam_rows[i++] : 0.f;
  }
}

// External forces (pybullet_backend.py:603-658): up to 4 forces at a time, each
// on one composite body (0 trunk, 1-3 left leg, 4-6 right leg) at a point given
// in the body frame, expressed in the world frame or (local) in the body frame.
struct ExtSlots {
  int count;
  int body[UPKIE_MAX_EXTERNAL_FORCES];
  int local[UPKIE_MAX_EXTERNAL_FORCES];
  float point[UPKIE_MAX_EXTERNAL_FORCES][3];
};
// Forces of ONE env: component d of slot i at force[(3 i + d) * stride];
// force == nullptr: no external force. Read inside the substep (a rare path)
// so that nothing of it stays live in the common path.
struct ExtForces {
  const float* force;
  size_t stride;
  const ExtSlots* slots;
};

// Force on link k (0 thigh, 1 calf, 2 wheel) of a leg. In: Fe = force in the
// base frame (or in the body frame when `local`), `point` in the body frame.
// Out: Fe in the base frame, application point p in the base frame, joint
// torques S_j' [Fe; p x Fe] of the joints above the link.
UPKIE_HD void ext_on_leg(const Leg& G, const float* q3, int k, bool local, V3 point, V3& Fe, V3& p, float (&text)[3]) {
  float psi = 0.f;
  V3 ok = G.o[0];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j <= k) psi = fmaf(G.sgn[j], q3[j], psi);
    if (j == k) ok = G.o[j];
  }
  float sn, cs;
  joint_sincos(psi, &sn, &cs);
  p = ok + rot_y(cs, sn, point);
  if (local) Fe = rot_y(cs, sn, Fe);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    V3 rr = p - G.o[j];
    text[j] = j <= k ? G.sgn[j] * (rr.z * Fe.x - rr.x * Fe.z) : 0.f;
  }
}

// What PyBulletBackend.get_contact_points reads back from Bullet
// (pybullet_backend.py:660-716): per tire, whether a contact point exists, where
// (base frame) and the force the floor exerts on the tire (base frame, N),
// i.e. the impulses of one substep divided by its duration. Filled by
// physics_substep() when asked (never on the step path).
struct ContactReport {
  bool active[2];
  V3 point[2];
  V3 dir[2][3];  // normal, rolling, lateral directions of the rows
  V3 force[2];
};

// One physics substep of duration h. tau: commanded joint torques.
// bi: inertial records of this env's bodies or nullptr (the model's). ext: external forces.
// Returns the floor-contact flag.
// BULLET_LIKE: contacts and joint limits by the Bullet-like specification of bullet_like.hpp on the env's contact
// manifold `manifold` (upkie_sim_set_contact_manifold) instead of the default one.
template <class ModelT>
UPKIE_HD bool bullet_like_contacts(const ModelT& M, const DevLimits& Lm, const System& S, const BaseFrame& bf, const Phys& s, float h,
                                   float (&tb)[6], float (&tl)[3], float (&tr)[3], float (&mf)[64], ContactReport* report);

template <bool SCRATCH_LIMITS = false, bool BULLET_LIKE = false, class ModelT>
UPKIE_HD bool physics_substep(const ModelT& M, const DevLimits& Lm, Phys& s, const float (&tau)[UPKIE_NJ], float h,
                                                const BodyInertials* bi, const ExtForces& ext, ContactReport* report = nullptr,
                                                float (*manifold)[64] = nullptr, SweepWarmStart* warm = nullptr) {
  // hip / knee position limits (URDF revolute limits, enforced by Bullet as
  // unilateral rows with ERP 0.2): rare, handled by the general solver
  bool any_limit = false;
  if (Lm.enforce) {
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j)
      any_limit = any_limit || joint_limit_near(Lm.bounded[j] != 0, s.q[j], Lm.lower[j], Lm.upper[j], joint_limit_reach(s.qd[j], M.max_joint_velocity, h));
  }
  const BaseFrame bf = base_frame(s.qw, s.qx, s.qy, s.qz, s.linvel, s.angvel);
  const float r00 = bf.r00, r01 = bf.r01, r02 = bf.r02, r10 = bf.r10, r11 = bf.r11, r12 = bf.r12, r20 = bf.r20, r21 = bf.r21, r22 = bf.r22;
  const V3 vB = bf.vB, wB = bf.wB, nB = bf.nB;
  V3 gn = M.gravity * nB;

  // trunk
  const LegOfModel<ModelT> trunk{M, bi, 0, 0};
  float m0 = trunk.mass(0);
  V3 c0 = trunk.com(0);
  S3 I0 = trunk.inertia(0);
  V3 I0w = mul(I0, wB);
  V3 bias_f, bias_n;
  {
    V3 ac = cross(wB, cross(wB, c0));
    V3 f = m0 * (ac + gn);
    V3 n = cross(wB, I0w);
    bias_f = f;
    bias_n = n + cross(c0, f);
  }
  Composite total{m0, m0 * c0, shift_to_origin(I0, m0, c0)};

  System S;
  {
    const float ql[3] = {s.q[0], s.q[1], s.q[2]}, qdl[3] = {s.qd[0], s.qd[1], s.qd[2]};
    const float qr[3] = {s.q[3], s.q[4], s.q[5]}, qdr[3] = {s.qd[3], s.qd[4], s.qd[5]};
    leg_pass(LegOfModel<ModelT>{M, bi, 1, 0}, M.wheel_axisymmetric != 0, ql, qdl, wB, gn, S.leg[0], total, bias_f, bias_n);
    leg_pass(LegOfModel<ModelT>{M, bi, 4, 3}, M.wheel_axisymmetric != 0, qr, qdr, wB, gn, S.leg[1], total, bias_f, bias_n);
  }

  // base block (generalised velocity [v, omega]) minus the legs' Schur terms
  {
    float A[21];
    V3 hh = total.h;
    // rows 0-2: m I ; rows 3-5 x cols 0-2: [h]x ; rows 3-5 x cols 3-5: I
    A[0] = total.m;
    A[1] = 0.f; A[2] = total.m;
    A[3] = 0.f; A[4] = 0.f; A[5] = total.m;
    A[6] = 0.f;   A[7] = -hh.z; A[8] = hh.y;  A[9] = total.I.xx;
    A[10] = hh.z; A[11] = 0.f;  A[12] = -hh.x; A[13] = total.I.xy; A[14] = total.I.yy;
    A[15] = -hh.y; A[16] = hh.x; A[17] = 0.f;  A[18] = total.I.xz; A[19] = total.I.yz; A[20] = total.I.zz;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const Leg& G = S.leg[l];
      int idx = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          A[idx] -= G.D[r][0] * G.F[0][c] + G.D[r][1] * G.F[1][c] + G.D[r][2] * G.F[2][c];
          ++idx;
        }
      }
    }
    ldl6_factor(A, S.A);
  }

  // applied generalised forces minus bias
  float bb[6], bl[3], br[3];
  {
    // Bullet-style base damping on the trunk: F = -m v (k + k|v|) at its com
    V3 vc = vB + cross(wB, c0);
    float vn = fast_sqrt(dot(vc, vc)), wn = fast_sqrt(dot(wB, wB));
    float kl = M.base_linear_damping, ka = M.base_angular_damping;
    V3 F = (-m0 * (kl + kl * vn)) * vc;
    V3 T = (-(ka + ka * wn)) * I0w;
    V3 Ntot = T + cross(c0, F);
    float tel[3] = {0.f, 0.f, 0.f}, ter[3] = {0.f, 0.f, 0.f};
    if (ext.force) {
      for (int i = 0; i < ext.slots->count; ++i) {
        const V3 f = v3(ext.force[(size_t)(3 * i) * ext.stride], ext.force[(size_t)(3 * i + 1) * ext.stride],
                        ext.force[(size_t)(3 * i + 2) * ext.stride]);
        const int b = ext.slots->body[i];
        const bool local = ext.slots->local[i] != 0;
        const V3 pt = v3(ext.slots->point[i][0], ext.slots->point[i][1], ext.slots->point[i][2]);
        V3 Fe = local ? f : v3(r00 * f.x + r10 * f.y + r20 * f.z, r01 * f.x + r11 * f.y + r21 * f.z, r02 * f.x + r12 * f.y + r22 * f.z);
        V3 pe = pt;
        float t3[3] = {0.f, 0.f, 0.f};
        if (b >= 4) {
          ext_on_leg(S.leg[1], &s.q[3], b - 4, local, pt, Fe, pe, t3);
#pragma unroll
          for (int j = 0; j < 3; ++j) ter[j] += t3[j];
        } else if (b >= 1) {
          ext_on_leg(S.leg[0], &s.q[0], b - 1, local, pt, Fe, pe, t3);
#pragma unroll
          for (int j = 0; j < 3; ++j) tel[j] += t3[j];
        }
        F = F + Fe;
        Ntot = Ntot + cross(pe, Fe);
      }
    }
    bb[0] = F.x - bias_f.x; bb[1] = F.y - bias_f.y; bb[2] = F.z - bias_f.z;
    bb[3] = Ntot.x - bias_n.x; bb[4] = Ntot.y - bias_n.y; bb[5] = Ntot.z - bias_n.z;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      bl[k] = tau[k] + tel[k] - M.joint_damping[k] * s.qd[k] - S.leg[0].bias[k];
      br[k] = tau[3 + k] + ter[k] - M.joint_damping[3 + k] * s.qd[3 + k] - S.leg[1].bias[k];
    }
  }
  // Generalised impulse so far: t = h (applied - bias). The contact impulses
  // J' lam are added below and ONE solve nu+ = nu + M^-1 t ends the substep.
  // Reduced base right-hand side rt = t_b - D_L t_L - D_R t_R: with the legs
  // eliminated, J M^-1 t = Jt . A^-1 rt + J_leg . Hinv t_leg where
  // Jt = J_b - D_w J_leg is the contact row reduced onto the base (6-vector).
  float tb[6], tl[3], tr[3], rt[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) tb[c] = h * bb[c];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    tl[j] = h * bl[j];
    tr[j] = h * br[j];
  }
#pragma unroll
  for (int c = 0; c < 6; ++c)
    rt[c] = tb[c] - (S.leg[0].D[c][0] * tl[0] + S.leg[0].D[c][1] * tl[1] + S.leg[0].D[c][2] * tl[2]) -
            (S.leg[1].D[c][0] * tr[0] + S.leg[1].D[c][1] * tr[1] + S.leg[1].D[c][2] * tr[2]);

  bool any_contact = false;
  if constexpr (BULLET_LIKE) {
    any_contact = bullet_like_contacts(M, Lm, S, bf, s, h, tb, tl, tr, *manifold, report);  // (leaves the velocity change in tb, tl, tr)
  } else {
  // ---- tire / floor contacts -------------------------------------------
  // Rows 3w+0..2 = (normal, t1, t2) of wheel w. Row r of wheel w only touches
  // the base and leg w: J = [d ; P x d ; leg part (3)].
  float Jb[6][6], Jl[6][3];  // Jacobian rows
  float Jt[6][6];            // rows reduced onto the base
  float rhs[6], vnow[6], dists[2];
  bool active[2];
  float un = fast_sqrt(nB.x * nB.x + nB.z * nB.z);
  float iun = fast_rcp(fmaxf(un, 1e-12f));
  float denom = h * M.contact_stiffness + M.contact_damping;
  float ih = fast_rcp(h);
  float erp = denom > 0.f ? h * M.contact_stiffness * fast_rcp(denom) : 0.2f;
  float cfm = denom > 0.f ? fast_rcp(denom * h) : 0.f;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const Leg& G = S.leg[w];
    float sa = G.sgn[2];  // wheel axis = sa * y
    V3 wc = v3(M.wheel_center[w][0], M.wheel_center[w][1], M.wheel_center[w][2]);
    V3 center = G.o[2] + wc;  // the tire centre lies on the wheel axis (checked at create)
    // lowest point of the tire circle: P = center - r * u / |u|, u = n - (n.a) a
    V3 dlow = v3(-nB.x * iun, 0.f, -nB.z * iun);
    V3 P = center + M.wheel_radius * dlow;
    float dist = s.pos.z + dot(nB, P);
    dists[w] = dist;
    // a contact point exists below the manifold breaking threshold
    active[w] = un >= 1e-6f && dist <= M.contact_breaking_threshold;
    any_contact = any_contact || active[w];
    V3 t1 = (sa * iun) * v3(nB.z, 0.f, -nB.x);  // a x n / |a x n|
    V3 t2 = cross(nB, t1);
    V3 dirs[3] = {nB, t1, t2};
    if (report) {
      report->active[w] = active[w];
      report->point[w] = P;
      report->dir[w][0] = nB; report->dir[w][1] = t1; report->dir[w][2] = t2;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int r = 3 * w + k;
      V3 d = dirs[k];
      V3 Pxd = cross(P, d);
      Jb[r][0] = d.x; Jb[r][1] = d.y; Jb[r][2] = d.z;
      Jb[r][3] = Pxd.x; Jb[r][4] = Pxd.y; Jb[r][5] = Pxd.z;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        V3 rr = P - G.o[j];
        // (a x rr) . d, a = s*y : y x rr = (rr.z, 0, -rr.x)
        Jl[r][j] = G.sgn[j] * (rr.z * d.x - rr.x * d.z);
      }
      // velocity of the contact point before this substep's impulses
      float v = Jb[r][0] * vB.x + Jb[r][1] * vB.y + Jb[r][2] * vB.z + Jb[r][3] * wB.x + Jb[r][4] * wB.y + Jb[r][5] * wB.z;
#pragma unroll
      for (int j = 0; j < 3; ++j) v = fmaf(Jl[r][j], s.qd[3 * w + j], v);
      vnow[r] = v;
#pragma unroll
      for (int c = 0; c < 6; ++c) Jt[r][c] = Jb[r][c] - (G.D[c][0] * Jl[r][0] + G.D[c][1] * Jl[r][1] + G.D[c][2] * Jl[r][2]);
    }
  }
  float lam[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (warm && (any_limit || !(active[0] || active[1]))) warm->swept = 0;
  if (any_limit) {
    if (SCRATCH_LIMITS)
      limit_path_scratch(M, S, Lm.lower, Lm.upper, Lm.bounded, s.q, s.qd, Jt, Jb, Jl, vnow, dists, active, cfm, erp, ih, M.max_joint_velocity, h, rt, tb, tl, tr, lam);
    else
      limit_path(M, S, Lm.lower, Lm.upper, Lm.bounded, s.q, s.qd, Jt, Jb, Jl, vnow, dists, active, cfm, erp, ih, M.max_joint_velocity, h, rt, tb, tl, tr, lam);
  } else if (active[0] || active[1]) {
    // A = J M^-1 J' + CFM (symmetric, packed lower by rows) built column by
    // column from Y_b = A^-1 Jt_b and K_b = Hinv J_leg,b; the same two vectors
    // give the free velocity of row b. Rows of a wheel without a contact point
    // are replaced by identity rows with zero rhs.
    float A[21];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int wb_ = b / 3;
      const Leg& G = S.leg[wb_];
      const float* hv = G.Hinv;
      float Y[6], K[3];
#pragma unroll
      for (int c = 0; c < 6; ++c) Y[c] = Jt[b][c];
      ldl6_solve(S.A, Y);
      K[0] = hv[0] * Jl[b][0] + hv[3] * Jl[b][1] + hv[4] * Jl[b][2];
      K[1] = hv[3] * Jl[b][0] + hv[1] * Jl[b][1] + hv[5] * Jl[b][2];
      K[2] = hv[4] * Jl[b][0] + hv[5] * Jl[b][1] + hv[2] * Jl[b][2];
      // free velocity of row b: v + J M^-1 t
      float vf = vnow[b];
#pragma unroll
      for (int c = 0; c < 6; ++c) vf = fmaf(Y[c], rt[c], vf);
      const float(&tw)[3] = wb_ == 0 ? tl : tr;
      vf += K[0] * tw[0] + K[1] * tw[1] + K[2] * tw[2];
      // penetration is pushed out with ERP; a separated point may only close
      // its gap within the step (continuous at dist = 0)
      const float dist = dists[wb_];
      const float rb = (b % 3 == 0) ? (dist <= 0.f ? -vf + erp * (-dist) * ih : -vf - dist * ih) : -vf;
      rhs[b] = active[wb_] ? rb : 0.f;
#pragma unroll
      for (int a = b; a < 6; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc = fmaf(Jt[a][c], Y[c], acc);
        if (a / 3 == wb_) acc += Jl[a][0] * K[0] + Jl[a][1] * K[1] + Jl[a][2] * K[2];
        if (a == b) acc += (a % 3) == 0 ? cfm : M.friction_cfm;
        const bool live = active[a / 3] && active[wb_];
        A[a * (a + 1) / 2 + b] = live ? acc : (a == b ? 1.f : 0.f);
      }
    }
    // Direct solve first: with both tires loaded and no slip the unconstrained
    // solution already satisfies lam_n >= 0, |lam_t| <= mu lam_n and IS the
    // solution. Otherwise it is projected and warm-starts the projected
    // Gauss-Seidel sweeps (continuous at the stick/slip / lift-off boundaries).
    const float mu = M.friction_mu;
    bool need_pgs = false;
    {
      Ldl6 fac;
      ldl6_factor(A, fac);
#pragma unroll
      for (int r = 0; r < 6; ++r) lam[r] = rhs[r];
      ldl6_solve(fac, lam);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        if (lam[3 * w] < 0.f) {
          lam[3 * w] = 0.f;
          need_pgs = true;
        }
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        if ((r % 3) == 0) continue;
        const float lim = mu * lam[3 * (r / 3)];
        if (lam[r] < -lim) { lam[r] = -lim; need_pgs = true; }
        if (lam[r] > lim) { lam[r] = lim; need_pgs = true; }
      }
    }
    if (need_pgs) {
      contact_sweeps_warm(M, A, rhs, lam, active[0] && active[1], warm);
    } else if (warm) {
      warm->swept = 0;
    }
    // t += J' lam
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      float acc = tb[c];
#pragma unroll
      for (int r = 0; r < 6; ++r) acc = fmaf(Jb[r][c], lam[r], acc);
      tb[c] = acc;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      tl[j] += Jl[0][j] * lam[0] + Jl[1][j] * lam[1] + Jl[2][j] * lam[2];
      tr[j] += Jl[3][j] * lam[3] + Jl[4][j] * lam[4] + Jl[5][j] * lam[5];
    }
  }
  // nu+ = nu + M^-1 t
  system_solve<true, true>(S, tb, tl, tr);
  if (report) {
#pragma unroll
    for (int w = 0; w < 2; ++w)
      report->force[w] = ih * (lam[3 * w] * report->dir[w][0] + lam[3 * w + 1] * report->dir[w][1] + lam[3 * w + 2] * report->dir[w][2]);
  }
  }
  float nu[12];
  nu[0] = vB.x + tb[0]; nu[1] = vB.y + tb[1]; nu[2] = vB.z + tb[2];
  nu[3] = wB.x + tb[3]; nu[4] = wB.y + tb[4]; nu[5] = wB.z + tb[5];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    nu[6 + j] = s.qd[j] + tl[j];
    nu[9 + j] = s.qd[3 + j] + tr[j];
  }

  // ---- integrate (semi-implicit Euler: new velocities move positions) ----
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    float v = fminf(fmaxf(nu[6 + j], -M.max_joint_velocity), M.max_joint_velocity);
    s.qd[j] = v;
    s.q[j] = fmaf(h, v, s.q[j]);
  }
  integrate_base(bf, nu[0], nu[1], nu[2], nu[3], nu[4], nu[5], h, s.pos, s.qw, s.qx, s.qy, s.qz, s.linvel, s.angvel);
  return any_contact;
}

}  // namespace upkie

// mpc.hpp -- batched MPC balancer (upkie/controllers/mpc_balancer.py:168-312).
//
// The reference builds a condensed box-constrained QP once (qpmpc's
// WheeledInvertedPendulum + MPCQP) and re-solves it every step with ProxQP,
// changing only the cost vector. Here:
//   host (fp64, once): Phi/Psi condensing, P, the affine cost map
//     q = Kx x0 + kv v*, and Minv = (P + rho I)^-1;
//   device (fp32, every step): fixed-iteration ADMM
//     U <- Minv (rho (z - y) - q);  z <- clip(U + y, +-a_max);  y <- y + U - z
//   with the dense product done on the matrix cores: one wave owns 16 envs and
//   computes U[Np x 16] = Minv[Np x Np] . R[Np x 16] with
//   v_mfma_f32_16x16x4_f32 (exact fp32). The rows of Minv are permuted per
//   16-row tile (i -> 4 (i % 4) + i / 4) so that the accumulator a lane ends up
//   holding is exactly the B-operand element it must feed to the next
//   iteration: no cross-lane movement at all in the loop.
// Round 6: the kernels that run are mpc_tile_h's below -- the same iteration with the product on the fp16 matrix path
// (v_mfma_f32_16x16x32_f16, two fp16 terms per operand, fp32 accumulation) and the constant part Minv q of the product taken
// out of the loop; mpc_tile / mpc_tile_tail (fp32 MFMA) stay as their A/B partners (UPKIE_MPC_FP32=1, -DUPKIE_FUSED_MPC_FP32).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/upkie_hip.h"

namespace upkie {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

struct MpcDev {
  const float* minv;  // [Np][Np] row-permuted
  const float* gx;    // [Np][4]  Minv Kx  and
  const float* gv;    // [Np]     Minv kv: the constant part of every iteration's product, u_q = Minv q (mpc_tile_h)
  const void* minv_h;  // the same matrix as two fp16 terms (hi + lo) in the A-operand layout of v_mfma_f32_16x16x32_f16 (mpc_tile_h)
  const float* kx;    // [Np][4]
  const float* kv;    // [Np]
  int num_envs;
  int n;       // horizon length N
  int iterations;
  float rho;
  float alpha;  // over-relaxation (1: the plain iteration)
  float scale;  // mpc_tile_h carries scale x r and reads Minv / scale: a power of two that brings max |Minv| / scale into [8, 16)
  float bound;  // max_ground_accel
  float max_ground_velocity;
  float fall_pitch;
};

// logical horizon index held by (tile t, lane group g, register r)
__device__ __forceinline__ int mpc_index(int t, int g, int r) { return 16 * t + 4 * r + g; }

// MPCBalancer.step of the 16 envs env0 .. env0 + 15 by one wavefront (all 64
// lanes must be active). `handover`: 16 floats (LDS) that receive the commanded
// velocities as well, for a step fused behind the solve in the same launch.
// COLUMNS: how many of the tile's 16 columns carry an env (16; 8 when the eight-lane step kernel solves the QPs of
// its wavefront's eight envs in front of their step: the other columns compute on zeros).
// KS: how many of the 4 T k-steps (four horizon indices each) of a row tile's product are issued. The padding beyond the
// horizon N carries zeros for ever (its rows of Minv are the identity / (1 + rho), its right-hand sides start at 0), so the
// k-steps that only read padding, 4 s >= N, contribute nothing: the reference's default N = 50 on four tiles needs 13 of
// its 16 (round 5: the launch is bound by the matrix pipe at this size, 64 -> 52 MFMAs per iteration).
template <int T, int COLUMNS = 16, int KS = 4 * T>
__device__ __forceinline__ void mpc_tile(const MpcDev& P, float* __restrict__ ws, const float* __restrict__ x0,
                                         const float* __restrict__ v_target, int v_target_stride,
                                         const uint8_t* __restrict__ contact, const float* __restrict__ done, float dt,
                                         float* __restrict__ commanded, float* __restrict__ first_input, int env0,
                                         float* handover) {
  constexpr int NP = 16 * T;
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, g = lane >> 4;
  const int B = P.num_envs;
  const int env = env0 + col;
  const bool live = col < COLUMNS && env < B;
  const int N = P.n;

  // A operands: a[t][s] = Minv_perm[16 t + col][4 s + g], stored lane by lane (mpc_host_setup): the 4 T^2 values of a lane
  // are contiguous, T^2 16-byte loads (round 2 read them as 16 T^2 strided words: 43 % of the wave's cycles were memory waits)
  float a[T][4 * T];
  {
    const float4* mine = reinterpret_cast<const float4*>(P.minv) + (size_t)lane * T * T;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s4 = 0; s4 < T; ++s4) {
        const float4 v = mine[t * T + s4];
        a[t][4 * s4] = v.x; a[t][4 * s4 + 1] = v.y; a[t][4 * s4 + 2] = v.z; a[t][4 * s4 + 3] = v.w;
      }
  }

  float4 x = live ? reinterpret_cast<const float4*>(x0)[env] : make_float4(0.f, 0.f, 0.f, 0.f);
  float vt = live ? v_target[(size_t)env * v_target_stride] : 0.f;
  if (!(fabsf(vt) < 3.0e38f)) vt = 0.f;  // non-finite guard (step_kernels.hpp): a NaN target would stay in the warm start for ever
  // envs that the coming env.step() will reset: MPCBalancer.reset() instead of
  // a solve (upkie_base_velocity.py:158): zero warm start, zero velocity
  const bool resetting = live && done != nullptr && done[env] != 0.f;
  // (fetched with everything else: read after the loop, each of the two would be a memory round trip of its own at the end)
  float v_before = live ? commanded[env] : 0.f;
  unsigned touching = live ? contact[env] : 0u;
  asm volatile("" : "+v"(v_before), "+v"(touching));  // pins the loads here
  float q[T][4], z[T][4], y[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = mpc_index(t, g, r);
      const float4 k = reinterpret_cast<const float4*>(P.kx)[n];
      q[t][r] = k.x * x.x + k.y * x.y + k.z * x.z + k.w * x.w + P.kv[n] * vt;
      const bool in = live && n < N && !resetting;
      z[t][r] = in ? ws[(size_t)n * B + env] : 0.f;
      y[t][r] = in ? ws[(size_t)(N + n) * B + env] : 0.f;
    }

  // The ADMM recurrences U <- Minv (rho (z - y) - q); z <- clip(U + y); y <- y + U - z with five VALU instructions per
  // element and iteration between the MFMAs (round 2: eight): the accumulator starts from y, so the matrix product
  // delivers w = U + y itself; z = med3(w, -b, b); y' = w - z; and the next right-hand side needs only
  // z - y' = 2 z - w:  rb' = rho (2 z - w) - q. One dependent chain per wave: what is not MFMA latency is these.
  const float rho = P.rho, bound = P.bound;
  // Per tile the four elements of a lane as two register pairs: the element-wise part of an iteration is packed fp32
  // arithmetic (v_pk_add_f32 / v_pk_fma_f32: two elements per instruction, same rounding as the scalar forms); only the
  // clip is per element (v_med3_f32 has no packed form). y stays a register quadruple: it IS the C operand of the
  // iteration's first MFMA (destination and C are different registers there, which the hardware allows as long as
  // they do not overlap partially), so nothing is copied into the accumulator.
  floatx2 rbp[T][2], qp[T][2];
  floatx4 yv[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    yv[t] = floatx4{y[t][0], y[t][1], y[t][2], y[t][3]};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      qp[t][h] = floatx2{q[t][2 * h], q[t][2 * h + 1]};
      rbp[t][h] = floatx2{fmaf(rho, z[t][2 * h] - y[t][2 * h], -q[t][2 * h]), fmaf(rho, z[t][2 * h + 1] - y[t][2 * h + 1], -q[t][2 * h + 1])};
    }
  }
  floatx2 zp[T][2];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) zp[t][h] = floatx2{z[t][2 * h], z[t][2 * h + 1]};
  const floatx2 rho2 = floatx2{rho, rho}, two = floatx2{2.f, 2.f};
  // Over-relaxation (UpkieMpcConfig.admm_relaxation): x^ = alpha U + (1 - alpha) z takes U's place in the z- and
  // y-updates, i.e. w = U + y becomes alpha w + (1 - alpha)(z + y), and z + y of the previous iterate is its own
  // relaxed w (y' = w - z): one packed multiply and one packed fma per register pair and iteration.
  const floatx2 alpha2 = floatx2{P.alpha, P.alpha}, beta2 = floatx2{1.f - P.alpha, 1.f - P.alpha};
  floatx2 carry[T][2];  // (1 - alpha) x the previous relaxed w
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) carry[t][h] = beta2 * (zp[t][h] + floatx2{y[t][2 * h], y[t][2 * h + 1]});
  for (int it = 0; it < P.iterations; ++it) {
    floatx4 acc0[T], acc1[T];
    // Two accumulator chains per tile. Accumulators in VGPRs (gfx950's register file is
    // unified; hipcc keeps MFMA results in AGPRs and pays twelve v_accvgpr moves per iteration, and its
    // -amdgpu-mfma-vgpr-form option allocates a destination that PARTLY overlaps the C operand, which the hardware
    // does not allow): written as inline asm, destination either tied to C or early-clobber. The wait states the
    // hazard recogniser would insert are written out: VALU write -> MFMA read of rb (s_nop 1), MFMA write -> VALU
    // read (s_nop 12).
    // (round 6: with the k-step as the OUTER loop and the row tile as the inner one -- 2 T accumulator chains in flight instead of
    // two -- the launch takes the same time (N = 48 / 50 / 64 at 16384 envs: 27.1 / 33.1 / 43.1 against 27.2 / 33.1 / 42.8 us,
    // tools/mpc_time.py): the product is bound by the matrix pipe's rate, 46 cycles per 16 x 16 x 4 fp32 MFMA as measured, not by
    // the chains; the tile-by-tile order stays)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(acc0[t]) : "v"(a[t][0]), "v"(rbp[0][0].x), "v"(yv[t]));
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc1[t]) : "v"(a[t][1]), "v"(rbp[0][0].y));
#pragma unroll
      for (int s = 2; s < KS; s += 2) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc0[t]) : "v"(a[t][s]), "v"(rbp[s / 4][(s % 4) / 2].x));
        if (s + 1 < KS)
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc1[t]) : "v"(a[t][s + 1]), "v"(rbp[(s + 1) / 4][((s + 1) % 4) / 2].y));
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) asm volatile("s_nop 12" : "+v"(acc0[t]), "+v"(acc1[t]));
#pragma unroll
    for (int t = 0; t < T; ++t) {
      floatx2 yn[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const floatx2 plain = (h == 0 ? floatx2{acc0[t][0], acc0[t][1]} : floatx2{acc0[t][2], acc0[t][3]}) +
                              (h == 0 ? floatx2{acc1[t][0], acc1[t][1]} : floatx2{acc1[t][2], acc1[t][3]});  // U + y
        const floatx2 w = __builtin_elementwise_fma(alpha2, plain, carry[t][h]);
        carry[t][h] = beta2 * w;
        const floatx2 zi = floatx2{__builtin_amdgcn_fmed3f(w.x, -bound, bound), __builtin_amdgcn_fmed3f(w.y, -bound, bound)};
        yn[h] = w - zi;
        zp[t][h] = zi;
        rbp[t][h] = __builtin_elementwise_fma(rho2, __builtin_elementwise_fma(two, zi, -w), -qp[t][h]);
      }
      yv[t] = floatx4{yn[0].x, yn[0].y, yn[1].x, yn[1].y};
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      y[t][r] = yv[t][r];
      z[t][r] = r < 2 ? zp[t][0][r] : zp[t][1][r - 2];
    }

#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = mpc_index(t, g, r);
      if (live && n < N) {
        ws[(size_t)n * B + env] = resetting ? 0.f : z[t][r];
        ws[(size_t)(N + n) * B + env] = resetting ? 0.f : y[t][r];
      }
    }
  if (live && g == 0) {
    const float u0 = z[0][0];  // plan.first_input, mpc_balancer.py:307
    if (first_input) first_input[env] = u0;
    const bool fallen = fabsf(x.y) > P.fall_pitch;  // :260
    float v = v_before;
    if (resetting) {
      v = 0.f;  // mpc_balancer.py:232
    } else if (fallen || !touching) {
      v = v + (dt / 0.1f) * (0.f - v);  // :295-301
    } else {
      v = v + u0 * dt / 2.0f;  // :305-311
      v = fminf(fmaxf(v, -P.max_ground_velocity), P.max_ground_velocity);
    }
    commanded[env] = v;
    if (handover) handover[col] = v;
  }
}

// The same solve for horizons 49 <= N <= 50 -- the reference's default N = 50 (mpc_balancer.py:174) -- with THREE row tiles on the
// matrix cores (rows 0 .. 47) and the last one or two rows on the vector unit (round 6). Padded to four tiles
// the product issued 4 x 13 = 52 MFMAs an iteration of which the fourth tile's 13 compute two useful rows out of sixteen, and the
// launch is bound by the matrix pipe at 16384 envs. Here rows 48 + g (g = 0, 1) of U = Minv R are dot products:
// lane (g, col) holds R[n][col] for n = g (mod 4) -- its own accumulator elements --, multiplies them with Minv[48 + j][n]
// (13 multiply-adds per row j, issued in the shadow of the MFMAs they do not depend on), and the four lanes of a column add their
// partial sums through two rounds of ds_bpermute (lanes col + 16 g: across the 16-lane rows, where DPP does not reach); 39 MFMAs
// an iteration. The tail's element (horizon index 48 + g) is the B operand of k-step 12, as before. Same layout of Minv in memory
// as mpc_tile<4> (row-permuted, lane by lane): the tail rows' coefficients are what lanes (g, 0) and (g, 4) of the fourth tile hold.
template <int COLUMNS = 16>
__device__ __forceinline__ void mpc_tile_tail(const MpcDev& P, float* __restrict__ ws, const float* __restrict__ x0,
                                              const float* __restrict__ v_target, int v_target_stride,
                                              const uint8_t* __restrict__ contact, const float* __restrict__ done, float dt,
                                              float* __restrict__ commanded, float* __restrict__ first_input, int env0) {
  constexpr int T = 3, KS = 13, LAYOUT_T = 4;  // (Minv is laid out for four tiles: mpc_host_setup)
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, g = lane >> 4;
  const int B = P.num_envs;
  const int env = env0 + col;
  const bool live = col < COLUMNS && env < B;
  const int N = P.n;
  float a[T][KS];
  {
    const float* mine = P.minv + (size_t)lane * LAYOUT_T * LAYOUT_T * 4;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 v = reinterpret_cast<const float4*>(mine)[t * LAYOUT_T + s4];
        if (4 * s4 < KS) a[t][4 * s4] = v.x;
        if (4 * s4 + 1 < KS) a[t][4 * s4 + 1] = v.y;
        if (4 * s4 + 2 < KS) a[t][4 * s4 + 2] = v.z;
        if (4 * s4 + 3 < KS) a[t][4 * s4 + 3] = v.w;
      }
  }
  // rows 48 and 49 of Minv at the columns n = 4 s + g this lane's elements have: Minv_perm rows 48 (i = 0) and 52 (i = 4) of tile 3
  floatx2 mt[KS];  // {Minv[48][4 s + g], Minv[49][4 s + g]}: both tail rows advance with ONE packed multiply-add per k-step
  {
    const float* row48 = P.minv + ((size_t)(16 * g + 0) * LAYOUT_T + 3) * 4 * LAYOUT_T;
    const float* row49 = P.minv + ((size_t)(16 * g + 4) * LAYOUT_T + 3) * 4 * LAYOUT_T;
#pragma unroll
    for (int s = 0; s < KS; ++s) mt[s] = floatx2{row48[s], row49[s]};
  }
  float4 x = live ? reinterpret_cast<const float4*>(x0)[env] : make_float4(0.f, 0.f, 0.f, 0.f);
  float vt = live ? v_target[(size_t)env * v_target_stride] : 0.f;
  if (!(fabsf(vt) < 3.0e38f)) vt = 0.f;  // non-finite guard (step_kernels.hpp)
  const bool resetting = live && done != nullptr && done[env] != 0.f;
  float v_before = live ? commanded[env] : 0.f;
  unsigned touching = live ? contact[env] : 0u;
  asm volatile("" : "+v"(v_before), "+v"(touching));
  float q[T][4], z[T][4], y[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = mpc_index(t, g, r);
      const float4 k = reinterpret_cast<const float4*>(P.kx)[n];
      q[t][r] = k.x * x.x + k.y * x.y + k.z * x.z + k.w * x.w + P.kv[n] * vt;
      const bool in = live && !resetting;
      z[t][r] = in ? ws[(size_t)n * B + env] : 0.f;
      y[t][r] = in ? ws[(size_t)(N + n) * B + env] : 0.f;
    }
  // the tail element of the lane: horizon index 48 + g (real for g < N - 48, padding above)
  const int nt = 48 + g;
  const bool tail_row = nt < N;
  float qt = 0.f, zt = 0.f, yt = 0.f;
  if (tail_row) {
    const float4 k = reinterpret_cast<const float4*>(P.kx)[nt];
    qt = k.x * x.x + k.y * x.y + k.z * x.z + k.w * x.w + P.kv[nt] * vt;
    if (live && !resetting) {
      zt = ws[(size_t)nt * B + env];
      yt = ws[(size_t)(N + nt) * B + env];
    }
  }
  const float rho = P.rho, bound = P.bound;
  floatx2 rbp[T][2], qp[T][2];
  floatx4 yv[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    yv[t] = floatx4{y[t][0], y[t][1], y[t][2], y[t][3]};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      qp[t][h] = floatx2{q[t][2 * h], q[t][2 * h + 1]};
      rbp[t][h] = floatx2{fmaf(rho, z[t][2 * h] - y[t][2 * h], -q[t][2 * h]), fmaf(rho, z[t][2 * h + 1] - y[t][2 * h + 1], -q[t][2 * h + 1])};
    }
  }
  floatx2 zp[T][2];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) zp[t][h] = floatx2{z[t][2 * h], z[t][2 * h + 1]};
  const floatx2 rho2 = floatx2{rho, rho}, two = floatx2{2.f, 2.f};
  const float alpha = P.alpha, beta = 1.f - P.alpha;
  const floatx2 alpha2 = floatx2{alpha, alpha}, beta2 = floatx2{beta, beta};
  floatx2 carry[T][2];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) carry[t][h] = beta2 * (zp[t][h] + floatx2{y[t][2 * h], y[t][2 * h + 1]});
  float rbt = fmaf(rho, zt - yt, -qt), carry_t = beta * (zt + yt);
  const int addr32 = (lane ^ 32) << 2, addr16 = (lane ^ 16) << 2;  // ds_bpermute byte addresses of the lanes 32 / 16 away
  const bool odd = (lane & 16) != 0;
  for (int it = 0; it < P.iterations; ++it) {
    floatx4 acc0[T], acc1[T];
    // the B operand of k-step s: horizon elements 4 s + g -- tile s / 4, register s % 4 -- and, for s = 12, the tail element
    auto rb_of = [&](int s) { return s == 12 ? rbt : ((s % 4) / 2 ? ((s % 2) ? rbp[s / 4][1].y : rbp[s / 4][1].x) : ((s % 2) ? rbp[s / 4][0].y : rbp[s / 4][0].x)); };
    // rows 48 / 49 on the vector unit WHILE the matrix pipe works: a wavefront issues in order and a dependent MFMA holds the
    // issue until its accumulator is ready, so the tail's multiply-adds are written BETWEEN the MFMAs of the first tile (inline
    // asm keeps them there: two per k-step, in the 32 cycles an MFMA occupies the pipe), the column sums behind that tile --
    // their LDS-crossbar latency passes under the second and third tile's MFMAs
    floatx2 p01 = floatx2{0.f, 0.f};  // partial sums of rows 48 / 49 over this lane's horizon elements
    float p0 = 0.f, p1 = 0.f, ut = 0.f;
    int got0 = 0, got1 = 0, got2 = 0;
    float keep = 0.f;
    // (a packed multiply-add p01 += mt[s] * rb broadcast to both halves: the element 4 s + g is one HALF of a register pair of
    // rbp -- op_sel picks it for both results -- or, for s = 12, the tail element itself)
#define UPKIE_TAIL_STEP(S)                                                                                                             \
  do {                                                                                                                                 \
    if ((S) == 12) {                                                                                                                   \
      const floatx2 both = floatx2{rbt, rbt};                                                                                          \
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p01) : "v"(mt[12]), "v"(both));                                                \
    } else if ((S) % 2 == 0) {                                                                                                         \
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p01) : "v"(mt[(S)]), "v"(rbp[(S) / 4][((S) % 4) / 2]));      \
    } else {                                                                                                                           \
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p01) : "v"(mt[(S)]), "v"(rbp[(S) / 4][((S) % 4) / 2]));         \
    }                                                                                                                                  \
  } while (0)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(acc0[t]) : "v"(a[t][0]), "v"(rbp[0][0].x), "v"(yv[t]));
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc1[t]) : "v"(a[t][1]), "v"(rbp[0][0].y));
      if (t == 0) {
        UPKIE_TAIL_STEP(0);
        UPKIE_TAIL_STEP(1);
      }
#pragma unroll
      for (int s = 2; s < KS; s += 2) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc0[t]) : "v"(a[t][s]), "v"(rb_of(s)));
        if (t == 0) UPKIE_TAIL_STEP(s);
        if (s + 1 < KS) {
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc1[t]) : "v"(a[t][s + 1]), "v"(rb_of(s + 1)));
          if (t == 0) UPKIE_TAIL_STEP(s + 1);
        }
      }
      if (t == 0) {
        p0 = p01.x;
        p1 = p01.y;
      }
      // The column sums, two rounds of exchanges, each ISSUED behind one tile's MFMAs and COLLECTED behind the next
      // one's: written as inline asm with their own waits -- left to the compiler the wait lands right behind the exchange and
      // the matrix pipe drains for two LDS round trips an iteration (measured: 37.1 us per launch at 16384 envs, what the
      // thirteen MFMAs of the fourth tile had cost)
      // (round 1: r_j = p_j + p_j of the lane 32 away; round 2: each lane hands the sum its neighbour 16 away wants -- lanes of
      // group 0 want row 48, of group 1 row 49; groups 2 and 3 hold padding and only contribute)
      if (t == 0) {
        asm volatile("ds_bpermute_b32 %0, %2, %3\n\tds_bpermute_b32 %1, %2, %4" : "=&v"(got0), "=&v"(got1) : "v"(addr32), "v"(p0), "v"(p1));
      } else if (t == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(p0), "+v"(p1) : "v"(got0), "v"(got1));
        const float send = odd ? p0 : p1;
        keep = odd ? p1 : p0;
        asm volatile("ds_bpermute_b32 %0, %1, %2" : "=&v"(got2) : "v"(addr16), "v"(send));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %1, %2" : "=v"(ut) : "v"(keep), "v"(got2));
      }
    }
#undef UPKIE_TAIL_STEP
#pragma unroll
    for (int t = 0; t < T; ++t) asm volatile("s_nop 12" : "+v"(acc0[t]), "+v"(acc1[t]));
#pragma unroll
    for (int t = 0; t < T; ++t) {
      floatx2 yn[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const floatx2 plain = (h == 0 ? floatx2{acc0[t][0], acc0[t][1]} : floatx2{acc0[t][2], acc0[t][3]}) +
                              (h == 0 ? floatx2{acc1[t][0], acc1[t][1]} : floatx2{acc1[t][2], acc1[t][3]});  // U + y
        const floatx2 w = __builtin_elementwise_fma(alpha2, plain, carry[t][h]);
        carry[t][h] = beta2 * w;
        const floatx2 zi = floatx2{__builtin_amdgcn_fmed3f(w.x, -bound, bound), __builtin_amdgcn_fmed3f(w.y, -bound, bound)};
        yn[h] = w - zi;
        zp[t][h] = zi;
        rbp[t][h] = __builtin_elementwise_fma(rho2, __builtin_elementwise_fma(two, zi, -w), -qp[t][h]);
      }
      yv[t] = floatx4{yn[0].x, yn[0].y, yn[1].x, yn[1].y};
    }
    {  // the tail element, the same recurrences (padding lanes: everything stays 0)
      const float plain = tail_row ? ut + yt : 0.f;
      const float w = fmaf(alpha, plain, carry_t);
      carry_t = beta * w;
      zt = __builtin_amdgcn_fmed3f(w, -bound, bound);
      yt = w - zt;
      rbt = fmaf(rho, fmaf(2.f, zt, -w), -qt);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = mpc_index(t, g, r);
      if (live) {
        ws[(size_t)n * B + env] = resetting ? 0.f : (r < 2 ? zp[t][0][r] : zp[t][1][r - 2]);
        ws[(size_t)(N + n) * B + env] = resetting ? 0.f : yv[t][r];
      }
    }
  if (live && tail_row) {
    ws[(size_t)nt * B + env] = resetting ? 0.f : zt;
    ws[(size_t)(N + nt) * B + env] = resetting ? 0.f : yt;
  }
  if (live && g == 0) {
    const float u0 = zp[0][0].x;  // plan.first_input, mpc_balancer.py:307
    if (first_input) first_input[env] = u0;
    const bool fallen = fabsf(x.y) > P.fall_pitch;  // :260
    float v = v_before;
    if (resetting) {
      v = 0.f;  // mpc_balancer.py:232
    } else if (fallen || !touching) {
      v = v + (dt / 0.1f) * (0.f - v);  // :295-301
    } else {
      v = v + u0 * dt / 2.0f;  // :305-311
      v = fminf(fmaxf(v, -P.max_ground_velocity), P.max_ground_velocity);
    }
    commanded[env] = v;
  }
}

// The same solve with the product on the fp16 matrix path (round 6; every horizon, worked out on N = 50). v_mfma_f32_16x16x4_f32 runs at the
// vector unit's own fp32 rate -- 32 cycles for 2048 multiply-adds -- and a wavefront that has its SIMD to itself issues nothing
// else in its shadow (tools/microbench/mfma_shadow.hip: 32.4 cycles per MFMA alone, 52 with two vector instructions behind
// each): the N = 50 iteration was 39 of them + 70 vector instructions, one after the other. v_mfma_f32_16x16x32_f16 does eight
// times the multiply-adds in half the time, and two fp16 terms carry 22 of fp32's 24 significand bits:
//   Minv = Ah + Al (split once on the host, from fp64),   r = rh + rl (split every iteration: rh = fp16(r),
//   rl = fp16(r - rh), the difference is exact in fp32; round to nearest),   Minv r ~= Ah rh + Ah rl + Al rh   (fp32 accumulation; the term
//   Al rl, 2^-22 of the product, is dropped).
// Against the fp64 checker this is as close as the fp32 product is (N = 50, 30 iterations: first input within 3e-3 m/s2 on the
// parity test's inputs, 9e-3 on its saturating ones, fp32: 4e-3 / 1e-2 -- the iteration's own conditioning, not the product's:
// profiles/r06_mpc_f16_split.txt; with the re-association below: 2e-6). Range: see kScale below; a solve that leaves it is
// discarded as a whole at the end (tests/test_nonfinite_guard_gpu.py poisons targets with 1e30).
// Layout: K-step j of the 32-wide product reads, from lane (g, col), the eight elements the lane itself holds in row tiles 2 j
// and 2 j + 1 (four registers each) -- as in mpc_tile no element ever changes lanes --, so the permuted column index of element
// (t, g, r) is 32 (t / 2) + 8 g + 4 (t % 2) + r; rows as before (mpc_host_setup).
typedef int intx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mpc_split_pair(floatx2 r, int& hi, int& lo) {
  // round to NEAREST, both terms (v_cvt_pk_f16_f32, new on gfx950): with v_cvt_pkrtz_f16_f32 every term errs towards zero and
  // the errors of a row's fifty products add up instead of averaging out -- 6e-2 m/s2 on the parity test's saturating steps
  // where this gives 9e-3, the fp32 product's figure
  float lx, ly;  // r - float(rh), exact, one instruction each: a mixed-precision multiply-add reads the fp16 half directly
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(r.x), "v"(r.y));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lx) : "v"(hi), "v"(r.x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ly) : "v"(hi), "v"(r.y));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(lx), "v"(ly));
}

// COLUMNS, handover: as mpc_tile's (a step kernel solves its own wavefront's QPs in front of their step).
template <int T, int COLUMNS = 16>
__device__ __forceinline__ void mpc_tile_h(const MpcDev& P, float* __restrict__ ws, const float* __restrict__ x0,
                                           const float* __restrict__ v_target, int v_target_stride,
                                           const uint8_t* __restrict__ contact, const float* __restrict__ done, float dt,
                                           float* __restrict__ commanded, float* __restrict__ first_input, int env0,
                                           float* handover = nullptr) {
  constexpr int KJ = (T + 1) / 2;  // 32-wide K-steps
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, g = lane >> 4;
  const int B = P.num_envs;
  const int env = env0 + col;
  const bool live = col < COLUMNS && env < B;
  const int N = P.n;
  intx4 ah[T][KJ], al[T][KJ];
  {
    const intx4* mine = reinterpret_cast<const intx4*>(P.minv_h) + (size_t)lane * T * KJ * 2;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        ah[t][j] = mine[(t * KJ + j) * 2];
        al[t][j] = mine[(t * KJ + j) * 2 + 1];
      }
  }
  float4 x = live ? reinterpret_cast<const float4*>(x0)[env] : make_float4(0.f, 0.f, 0.f, 0.f);
  float vt = live ? v_target[(size_t)env * v_target_stride] : 0.f;
  if (!(fabsf(vt) < 3.0e38f)) vt = 0.f;  // non-finite guard (step_kernels.hpp)
  const bool resetting = live && done != nullptr && done[env] != 0.f;
  float v_before = live ? commanded[env] : 0.f;
  unsigned touching = live ? contact[env] : 0u;
  asm volatile("" : "+v"(v_before), "+v"(touching));
  // u_q = Minv q = (Minv Kx) x0 + (Minv kv) v*, from the host's fp64 products: see "the iteration" below
  float uq[T][4], z[T][4], y[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = mpc_index(t, g, r);
      const float4 k = reinterpret_cast<const float4*>(P.gx)[n];
      uq[t][r] = k.x * x.x + k.y * x.y + k.z * x.z + k.w * x.w + P.gv[n] * vt;
      const bool in = live && n < N && !resetting;
      z[t][r] = in ? ws[(size_t)n * B + env] : 0.f;
      y[t][r] = in ? ws[(size_t)(N + n) * B + env] : 0.f;
    }
  // The iteration, re-associated (round 6): U = Minv (rho (z - y) - q) is computed as Minv r - u_q with r = rho (z - y) and the
  // constant u_q = Minv q taken out of the loop. In the first form the two parts of the right-hand side cancel to a thousandth
  // of their size in every product (|Minv| reaches 490 at rho = 1e-3, |q| 180, U is ~10) and fp32 keeps three digits of U less
  // than it could: against the fp64 checker the first input of the fp32 kernels above is 4e-3 .. 1e-2 m/s2 off on the parity
  // test's inputs at N = 50 (5e-2 at N = 49). In this form the product's operand is small (|r| < 1) and u_q comes from five
  // multiply-adds on the host's fp64 Minv Kx, Minv kv: 2e-6 with the same number of instructions (the -q of the right-hand side
  // becomes a -alpha u_q in the relaxed update). Same recurrences otherwise (mpc_tile): w = alpha (U + y) + (1 - alpha) w_prev,
  // z = clip(w), y' = w - z, r' = rho (2 z - w).
  // r is carried TIMES P.scale and Minv divided by it on the host (a power of two: nothing rounds; 32 with the default weights,
  // chosen so that the largest entry of Minv / scale lies in [8, 16) whatever rho and the cost weights are): fp16's normal
  // range, 6e-5 .. 65504, then holds both terms of |r| from 4e-3 to 2e3; u_q is clamped once, so that a state far outside anything the balancer is for
  // (|w| follows |u_q|) gives a wrong but finite plan.
  constexpr float kLargest = 1.0e5f;
  const float rho = P.rho * P.scale, bound = P.bound;
  floatx2 rbp[T][2], nauq[T][2], zp[T][2], carry[T][2];
  floatx4 yv[T];
  const floatx2 rho2 = floatx2{rho, rho}, two = floatx2{2.f, 2.f};
  const floatx2 alpha2 = floatx2{P.alpha, P.alpha}, beta2 = floatx2{1.f - P.alpha, 1.f - P.alpha};
#pragma unroll
  for (int t = 0; t < T; ++t) {
    yv[t] = floatx4{y[t][0], y[t][1], y[t][2], y[t][3]};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      nauq[t][h] = floatx2{-P.alpha * __builtin_amdgcn_fmed3f(uq[t][2 * h], -kLargest, kLargest), -P.alpha * __builtin_amdgcn_fmed3f(uq[t][2 * h + 1], -kLargest, kLargest)};
      zp[t][h] = floatx2{z[t][2 * h], z[t][2 * h + 1]};
      const floatx2 yh = floatx2{y[t][2 * h], y[t][2 * h + 1]};
      rbp[t][h] = rho2 * (zp[t][h] - yh);
      carry[t][h] = __builtin_elementwise_fma(beta2, zp[t][h] + yh, nauq[t][h]);  // (1 - alpha) x the previous relaxed w, - alpha u_q
    }
  }
  for (int it = 0; it < P.iterations; ++it) {
    // the right-hand side as two fp16 terms, in B-operand order: register 2 (t % 2) + h of K-step t / 2
    intx4 bh[KJ], bl[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) bh[j] = bl[j] = intx4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int hi, lo;
        mpc_split_pair(rbp[t][h], hi, lo);
        bh[t / 2][2 * (t % 2) + h] = hi;
        bl[t / 2][2 * (t % 2) + h] = lo;
      }
    // six rounds (three with one K-step) over the tiles, row tile innermost: MFMAs on one accumulator are T apart. Accumulators
    // in VGPRs, destination tied to C or early-clobber, wait states written out (see mpc_tile)
    floatx4 acc[T];
    // every term of the split is complete, and two to four wait states old (one is required), before the first MFMA: the compiler does not know that the
    // asm statements below read their operands as MFMAs do, and sinks a conversion right in front of its consumer otherwise
    // (tools/microbench/f16_split_check.hip: a B register written by the instruction in front of the MFMA is read stale)
    if (KJ > 1)
      asm volatile("s_nop 3" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[KJ - 1]), "+v"(bl[KJ - 1]));
    else
      asm volatile("s_nop 1" : "+v"(bh[0]), "+v"(bl[0]));
#pragma unroll
    for (int t = 0; t < T; ++t) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc[t]) : "v"(ah[t][0]), "v"(bh[0]), "v"(yv[t]));
    if (KJ > 1) {
#pragma unroll
      for (int t = 0; t < T; ++t) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(ah[t][KJ - 1]), "v"(bh[KJ - 1]));
    }
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(ah[t][j]), "v"(bl[j]));
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(al[t][j]), "v"(bh[j]));
    // MFMA result -> vector unit: 7 wait states (measured, f16_split_check.hip (4c): 6 read a partly written result); tile 0's
    // last MFMA is T - 1 MFMAs (four wait states each, at least) back, the later tiles' are read behind the updates of the tiles
    // in front of them
    if (T >= 4) {
      asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[T - 1]));
    } else if (T == 3) {
      asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[T - 1]));
    } else if (T == 2) {
      asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[T - 1]));
    } else {
      asm volatile("s_nop 7" : "+v"(acc[0]));
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t > 0) asm volatile("" : "+v"(acc[t]), "+v"(rbp[t - 1][1]));  // (tile t's accumulators are read behind tile t - 1's update)
      floatx2 yn[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const floatx2 plain = h == 0 ? floatx2{acc[t][0], acc[t][1]} : floatx2{acc[t][2], acc[t][3]};  // Minv r + y
        const floatx2 w = __builtin_elementwise_fma(alpha2, plain, carry[t][h]);
        carry[t][h] = __builtin_elementwise_fma(beta2, w, nauq[t][h]);
        const floatx2 zi = floatx2{__builtin_amdgcn_fmed3f(w.x, -bound, bound), __builtin_amdgcn_fmed3f(w.y, -bound, bound)};
        yn[h] = w - zi;
        zp[t][h] = zi;
        rbp[t][h] = rho2 * __builtin_elementwise_fma(two, zi, -w);
      }
      yv[t] = floatx4{yn[0].x, yn[0].y, yn[1].x, yn[1].y};
    }
  }
  // A solve that left fp16's range (a finite but absurd target or state: |64 r| > 65504 turns into an infinity in the split and
  // into NaNs in every row of the next product) is discarded as a whole: zero warm start, the commanded velocity decays as for
  // a fallen robot. Every lane of the column sees it in its own elements (the matrix is dense), so each decides alone.
  bool sound = true;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) sound = sound && fabsf(r < 2 ? zp[t][0][r] : zp[t][1][r - 2]) < 3.0e38f && fabsf(yv[t][r]) < 3.0e38f;
  const bool discard = resetting || !sound;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = mpc_index(t, g, r);
      if (live && n < N) {
        ws[(size_t)n * B + env] = discard ? 0.f : (r < 2 ? zp[t][0][r] : zp[t][1][r - 2]);
        ws[(size_t)(N + n) * B + env] = discard ? 0.f : yv[t][r];
      }
    }
  if (live && g == 0) {
    const float u0 = sound ? zp[0][0].x : 0.f;  // plan.first_input, mpc_balancer.py:307
    if (first_input) first_input[env] = u0;
    const bool fallen = fabsf(x.y) > P.fall_pitch || !sound;  // :260
    float v = v_before;
    if (resetting) {
      v = 0.f;  // mpc_balancer.py:232
    } else if (fallen || !touching) {
      v = v + (dt / 0.1f) * (0.f - v);  // :295-301
    } else {
      v = v + u0 * dt / 2.0f;  // :305-311
      v = fminf(fmaxf(v, -P.max_ground_velocity), P.max_ground_velocity);
    }
    commanded[env] = v;
    if (handover) handover[col] = v;
  }
}

template <int T>
__global__ __launch_bounds__(64) void mpc_step_h_kernel(MpcDev P, float* __restrict__ ws, const float* __restrict__ x0,
                                                         const float* __restrict__ v_target, int v_target_stride,
                                                         const uint8_t* __restrict__ contact, const float* __restrict__ done, float dt,
                                                         float* __restrict__ commanded, float* __restrict__ first_input) {
  unsigned block = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) block = (block & 7u) * (gridDim.x >> 3) + (block >> 3);
  mpc_tile_h<T>(P, ws, x0, v_target, v_target_stride, contact, done, dt, commanded, first_input, (int)block * 16);
}

#if !defined(UPKIE_STEP_INSTANCES_ONLY)  // (a non-template kernel: the C-ABI's translation unit alone defines it)
__global__ __launch_bounds__(64) void mpc_step_tail_kernel(MpcDev P, float* __restrict__ ws, const float* __restrict__ x0,
                                                            const float* __restrict__ v_target, int v_target_stride,
                                                            const uint8_t* __restrict__ contact, const float* __restrict__ done, float dt,
                                                            float* __restrict__ commanded, float* __restrict__ first_input) {
  unsigned block = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) block = (block & 7u) * (gridDim.x >> 3) + (block >> 3);
  mpc_tile_tail(P, ws, x0, v_target, v_target_stride, contact, done, dt, commanded, first_input, (int)block * 16);
}
#endif

template <int T, int KS = 4 * T>
__global__ __launch_bounds__(64) void mpc_step_kernel(MpcDev P, float* __restrict__ ws, const float* __restrict__ x0,
                                                       const float* __restrict__ v_target, int v_target_stride,
                                                       const uint8_t* __restrict__ contact,
                                                       const float* __restrict__ done, float dt,
                                                       float* __restrict__ commanded, float* __restrict__ first_input) {
  // (XCD-aware: the two wavefronts that share a 128-byte line of a workspace row run on one XCD, see step_kernel_octet)
  unsigned block = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) block = (block & 7u) * (gridDim.x >> 3) + (block >> 3);
  static_assert(KS >= 2 && KS <= 4 * T, "k-steps of a row tile");
  mpc_tile<T, 16, KS>(P, ws, x0, v_target, v_target_stride, contact, done, dt, commanded, first_input, (int)block * 16, nullptr);
}


#if !defined(UPKIE_STEP_INSTANCES_ONLY)  // (step_instances.hip: the non-template kernels live in the C-ABI's translation unit alone)
__global__ __launch_bounds__(64) void mpc_reset_kernel(int B, int N, float* __restrict__ ws, float* __restrict__ commanded,
                                                        const uint8_t* __restrict__ mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  if (mask && !mask[e]) return;
  for (int n = 0; n < 2 * N; ++n) ws[(size_t)n * B + e] = 0.f;
  commanded[e] = 0.f;  // mpc_balancer.py:232
}
#endif

// ------------------------------------------------------------ host setup
// Exact zero-order-hold discretisation of the wheeled inverted pendulum and
// condensing, as qpmpc's WheeledInvertedPendulum.build_mpc_problem + MPCQP do
// (third-party, restated from the published algorithm).
inline bool mpc_host_setup(const UpkieMpcConfig& c, int np, std::vector<float>* minv_perm, std::vector<float>* kx,
                           std::vector<float>* kv, std::string* why, std::vector<uint16_t>* minv_h = nullptr, std::vector<float>* gx = nullptr,
                           std::vector<float>* gv = nullptr, float* scale_out = nullptr) {
  const int N = c.nb_timesteps;
  const double T = c.sampling_period, g = 9.81;
  const double omega = std::sqrt(g / c.leg_length);
  const double ch = std::cosh(T * omega), sh = std::sinh(T * omega);
  const double A[4][4] = {{1, 0, T, 0}, {0, ch, 0, sh / omega}, {0, 0, 1, 0}, {0, omega * sh, 0, ch}};
  const double Bv[4] = {T * T / 2.0, (1.0 - ch) / g, T, -omega * sh / g};
  std::vector<double> P((size_t)N * N, 0.0), Kx((size_t)N * 4, 0.0), Kv(N, 0.0);
  std::vector<double> phi(16, 0.0), psi((size_t)4 * N, 0.0);
  for (int i = 0; i < 4; ++i) phi[5 * i] = 1.0;
  for (int i = 0; i < N; ++i) P[(size_t)i * N + i] = c.stage_input_cost_weight;
  for (int k = 0; k <= N; ++k) {
    const double w = k == N ? c.terminal_cost_weight : c.stage_state_cost_weight;
    for (int i = 0; i < N; ++i) {
      for (int j = 0; j < N; ++j) {
        double s = 0;
        for (int l = 0; l < 4; ++l) s += psi[(size_t)N * l + i] * psi[(size_t)N * l + j];
        P[(size_t)i * N + j] += w * s;
      }
      for (int col = 0; col < 4; ++col) {
        double s = 0;
        for (int l = 0; l < 4; ++l) s += psi[(size_t)N * l + i] * (phi[4 * l + col] - ((l == 0 && col == 0) ? 1.0 : 0.0));
        Kx[(size_t)i * 4 + col] += w * s;
      }
      Kv[i] -= w * (psi[i] * (k * T) + psi[(size_t)2 * N + i]);
    }
    if (k == N) break;
    std::vector<double> nphi(16), npsi((size_t)4 * N);
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int l = 0; l < 4; ++l) s += A[i][l] * phi[4 * l + j];
        nphi[4 * i + j] = s;
      }
      for (int j = 0; j < N; ++j) {
        double s = 0;
        for (int l = 0; l < 4; ++l) s += A[i][l] * psi[(size_t)N * l + j];
        npsi[(size_t)N * i + j] = s;
      }
      npsi[(size_t)N * i + k] = Bv[i];
    }
    phi.swap(nphi);
    psi.swap(npsi);
  }
  // Minv = (P + rho I)^-1 through Cholesky, padded with identity / (1 + rho)
  std::vector<double> L((size_t)N * N, 0.0), Minv((size_t)N * N, 0.0);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = P[(size_t)i * N + j] + (i == j ? c.admm_rho : 0.0);
      for (int k = 0; k < j; ++k) s -= L[(size_t)i * N + k] * L[(size_t)j * N + k];
      if (i == j) {
        if (s <= 0) {
          *why = "P + rho I is not positive definite";
          return false;
        }
        L[(size_t)i * N + i] = std::sqrt(s);
      } else {
        L[(size_t)i * N + j] = s / L[(size_t)j * N + j];
      }
    }
  std::vector<double> yv(N), xv(N);
  for (int col = 0; col < N; ++col) {
    for (int i = 0; i < N; ++i) {
      double s = i == col ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[(size_t)i * N + k] * yv[k];
      yv[i] = s / L[(size_t)i * N + i];
    }
    for (int i = N - 1; i >= 0; --i) {
      double s = yv[i];
      for (int k = i + 1; k < N; ++k) s -= L[(size_t)k * N + i] * xv[k];
      xv[i] = s / L[(size_t)i * N + i];
    }
    for (int r = 0; r < N; ++r) Minv[(size_t)r * N + col] = xv[r];
  }
  minv_perm->assign((size_t)np * np, 0.f);
  kx->assign((size_t)np * 4, 0.f);
  kv->assign(np, 0.f);
  // Row-permuted, then laid out LANE BY LANE as the kernel consumes it: lane (g, i) = 16 g + i of the wavefront holds
  // A-operand element a[t][s] = Minv_perm[16 t + i][4 s + g] at [lane][t][s] (4 T^2 contiguous floats per lane)
  const int tiles = np / 16;
  for (int p = 0; p < np; ++p) {
    const int t = p / 16, i = p % 16;
    const int row = 16 * t + 4 * (i % 4) + i / 4;  // logical row behind permuted row p
    for (int col = 0; col < np; ++col) {
      double v;
      if (row < N && col < N)
        v = Minv[(size_t)row * N + col];
      else
        v = row == col ? 1.0 / (1.0 + c.admm_rho) : 0.0;
      const int s_ = col / 4, group = col % 4, lane = 16 * group + i;
      (*minv_perm)[((size_t)lane * tiles + t) * 4 * tiles + s_] = (float)v;
    }
  }
  for (int n = 0; n < N; ++n) {
    for (int col = 0; col < 4; ++col) (*kx)[(size_t)n * 4 + col] = (float)Kx[(size_t)n * 4 + col];
    (*kv)[n] = (float)Kv[n];
  }
  if (gx && gv) {  // Minv Kx, Minv kv (mpc_tile_h's u_q)
    gx->assign((size_t)np * 4, 0.f);
    gv->assign(np, 0.f);
    for (int n = 0; n < N; ++n) {
      double sv = 0.0, sx[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = 0; k < N; ++k) {
        const double m = Minv[(size_t)n * N + k];
        sv += m * Kv[k];
        for (int col = 0; col < 4; ++col) sx[col] += m * Kx[(size_t)k * 4 + col];
      }
      (*gv)[n] = (float)sv;
      for (int col = 0; col < 4; ++col) (*gx)[(size_t)n * 4 + col] = (float)sx[col];
    }
  }
  if (minv_h) {
    // mpc_tile_h's A operands: lane (g, i) holds, for row tile t and K-step j, the eight columns 32 j + 8 g + c of permuted row
    // 16 t + i -- column slot c is element (row tile 2 j + c / 4, group g, register c % 4) -- as fp16 hi and lo terms:
    // [lane][t][j][hi | lo][c], 16 bytes per term
    const int kj = (tiles + 1) / 2;
    minv_h->assign((size_t)64 * tiles * kj * 2 * 8, 0);
    // the power of two that brings the largest entry into [8, 16): fp16 then holds both terms of every entry down to 1e-3 of it
    double largest = 1.0 / (1.0 + c.admm_rho);
    for (double v : Minv) largest = std::max(largest, std::fabs(v));
    const double scale = std::exp2(std::ceil(std::log2(largest / 16.0)));
    if (scale_out) *scale_out = (float)scale;
    auto entry = [&](int row, int col) { return row < N && col < N ? Minv[(size_t)row * N + col] : (row == col ? 1.0 / (1.0 + c.admm_rho) : 0.0); };
    auto bits = [](_Float16 h) {
      uint16_t u;
      std::memcpy(&u, &h, sizeof(u));
      return u;
    };
    for (int lane = 0; lane < 64; ++lane) {
      const int group = lane / 16, i = lane % 16;
      for (int t = 0; t < tiles; ++t) {
        const int row = 16 * t + 4 * (i % 4) + i / 4;
        for (int j = 0; j < kj; ++j)
          for (int slot = 0; slot < 8; ++slot) {
            const int tk = 2 * j + slot / 4;
            const double v = (tk < tiles ? entry(row, 16 * tk + 4 * (slot % 4) + group) : 0.0) / scale;  // (mpc_tile_h carries scale x r)
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (double)hi);
            const size_t at = ((((size_t)lane * tiles + t) * kj + j) * 2) * 8 + slot;
            (*minv_h)[at] = bits(hi);
            (*minv_h)[at + 8] = bits(lo);
          }
      }
    }
  }
  return true;
}

}  // namespace upkie

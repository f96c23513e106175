// octet.hpp -- eight lanes per env: one quad of lanes per leg, one lane per body.
//
// At 4096 envs the two-lane kernel (pair.hpp) fills 128 of the chip's 1024
// SIMDs and the launch lasts as long as one wave's instruction stream
// (~11 k instructions per env.step()). Here an env owns two quads of a row of
// 16 lanes (quads q and q + 2: the row holds two envs), one quad per leg:
//
//     lane of the quad   0            1      2      3
//     body               trunk        thigh  calf   wheel
//     joint              -            hip    knee   wheel
//     contact row        (free vel.)  normal rolling lateral
//
// so that everything the reference does "per body", "per joint" and "per
// contact row" is ONE instruction stream executed by four lanes side by side,
// and the quantities a lane needs from its neighbours arrive through DPP
// (quad_perm inside the quad, row_ror:8 between the legs), mostly folded into
// the consuming multiply-add (v_fmac_f32_dpp). The real trunk lives in the left
// quad's lane 0; the right quad's lane 0 carries a massless copy, so sums over
// the eight lanes count it once. Nothing is staged through LDS.
//
// The algebra is that of physics_substep() (dynamics.hpp), reassociated:
//   * kinematics and the Newton-Euler pass: chain quantities (cumulative joint
//     angle, joint origins, joint-rate sums, origin accelerations) are prefix
//     sums over the quad; the per-body wrench and inertia are lane-local;
//   * composite inertias / wrenches of the subtrees: suffix sums over the quad;
//   * the 3x3 leg block: rows by lane, Cholesky factor in every lane, one
//     column of its inverse per lane; D = F Hinv one column per lane;
//   * the 6x6 base block A = Mbb - sum D F': every lane adds its own body's
//     spatial inertia minus its own column's outer product, one 8-lane sum per
//     entry; LDL' in every lane (the structural zeros of a planar leg, F_y = 0,
//     are skipped);
//   * seven solves with A at once: the six contact rows (lanes 1-3 of both
//     quads) and the free velocity A^-1 rt (lane 0);
//   * the 6x6 contact system: one column per lane, solved by block elimination
//     (each quad eliminates its own tire's 3x3 block by Gauss-Jordan across its
//     lanes, the Schur complement couples the tires through one exchange);
//   * nu+ = A^-1 rt + sum lam_b Y_b: the final solve is a weighted 8-lane sum.
// The rare cases gather their system into every lane of the env and run the
// code the other mappings run, every lane on identical data: a contact solution
// outside its friction cone -> the projected Gauss-Seidel sweeps (contact_pgs6);
// a hip or knee at its stop -> the ten-row solve of the other mappings
// (limit_path in registers for the Servos kernels, limit_path_scratch over
// scratch memory for the others: octet_limit_path). A tire off the floor is
// masked with identity rows. Forces on leg links are not handled
// here: launch_step gives such launches to the two-lane kernel.
//
// Host build (tests/host_harness.hip): the same code, the eight lanes of one env
// run as eight threads in lockstep and every exchange goes through a shared
// slot array between two barriers.
//
// Included by upkie_hip.hip inside namespace upkie, after pair.hpp.
#pragma once

// ---------------------------------------------------------------- lane exchange
#if !defined(__HIP_DEVICE_COMPILE__)
struct OctHostLane {
  int lane;      // 4 * leg + lane of the quad
  float* slots;  // [8], shared by the env's threads
  void (*barrier)(void*);
  void* arg;
};
inline thread_local OctHostLane* g_oct_lane = nullptr;
inline float oct_host_get(float x, int src) {
  OctHostLane* L = g_oct_lane;
  L->slots[L->lane] = x;
  L->barrier(L->arg);
  const float r = L->slots[src];
  L->barrier(L->arg);
  return r;
}
#endif

// value of lane sel[l] of the own quad
template <int S0, int S1, int S2, int S3>
UPKIE_HD float oct_qperm(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), S0 | (S1 << 2) | (S2 << 4) | (S3 << 6), 0xF, 0xF, true));
#else
  const int sel[4] = {S0, S1, S2, S3};
  const int lane = g_oct_lane->lane;
  return oct_host_get(x, (lane & 4) | sel[lane & 3]);
#endif
}
template <int K>
UPKIE_HD float oct_qb(float x) { return oct_qperm<K, K, K, K>(x); }
// the parent's / grandparent's value along the chain trunk -> thigh -> calf -> wheel (the trunk reads itself)
UPKIE_HD float oct_up1(float x) { return oct_qperm<0, 0, 1, 2>(x); }
UPKIE_HD float oct_up2(float x) { return oct_qperm<0, 0, 0, 1>(x); }
// same lane of the other leg's quad
UPKIE_HD float oct_swp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));  // row_ror:8
#else
  return oct_host_get(x, g_oct_lane->lane ^ 4);
#endif
}
// sum over the quad / over the env's eight lanes, in every lane
UPKIE_HD float oct_qsum(float x) {
#pragma clang fp contract(off)  // keep x + dpp(x) one v_add_f32_dpp (a contracted multiply-add costs a separate DPP move)
  x += oct_qperm<1, 0, 3, 2>(x);
  x += oct_qperm<2, 3, 0, 1>(x);
  return x;
}
UPKIE_HD float oct_esum(float x) {
#pragma clang fp contract(off)  // keep x + dpp(x) one v_add_f32_dpp (a contracted multiply-add costs a separate DPP move)
  x = oct_qsum(x);
  return x + oct_swp(x);
}
template <int N>
UPKIE_HD void oct_esum(float (&x)[N]) {  // step by step over all values: independent chains between dependent DPP reads
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] += oct_qperm<1, 0, 3, 2>(x[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] += oct_qperm<2, 3, 0, 1>(x[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] += oct_swp(x[i]);
}
UPKIE_HD bool oct_env_any(bool b) { return oct_esum(b ? 1.f : 0.f) != 0.f; }
// largest value among the env's eight lanes, in every lane
UPKIE_HD float oct_emax(float x) {
  x = fmaxf(x, oct_qperm<1, 0, 3, 2>(x));
  x = fmaxf(x, oct_qperm<2, 3, 0, 1>(x));
  return fmaxf(x, oct_swp(x));
}
// some lane of the wavefront (device) / of the env (host): a uniform branch guards the rare paths
UPKIE_HD bool oct_wave_any(bool b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(b) != 0;
#else
  return oct_env_any(b);
#endif
}
// inclusive prefix sum along the chain; the trunk lane's value must be 0
UPKIE_HD float oct_chain(float x) {
#pragma clang fp contract(off)  // keep x + dpp(x) one v_add_f32_dpp (a contracted multiply-add costs a separate DPP move)
  x += oct_up1(x);
  x += oct_up2(x);
  return x;
}
// the same for several values at once, step by step over all of them (no wait states between dependent DPP reads)
template <int N>
UPKIE_HD void oct_chain(float (&x)[N]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] += oct_up1(x[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] += oct_up2(x[i]);
}
// sum over the subtree of the own joint (lanes >= own, own >= 1) of a value that is 0 on the trunk lane
// (DPP bank masks select quads of a row, not lanes of a quad: the zero on the trunk lane is what keeps the wheel and
// calf lanes' sums clean)
template <int N>
UPKIE_HD void oct_subtree(float (&xm)[N]) {
#pragma clang fp contract(off)
  float s1[N];
#pragma unroll
  for (int i = 0; i < N; ++i) s1[i] = xm[i] + oct_qperm<1, 2, 3, 0>(xm[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) xm[i] = s1[i] + oct_qperm<0, 3, 0, 0>(xm[i]);
}
UPKIE_HD float oct_subtree(float xm) {
#pragma clang fp contract(off)  // keep x + dpp(x) one v_add_f32_dpp (a contracted multiply-add costs a separate DPP move)
  const float s1 = xm + oct_qperm<1, 2, 3, 0>(xm);
  return s1 + oct_qperm<0, 3, 0, 0>(xm);
}

// Multiply-accumulate blocks with the broadcast folded into the instruction
// (v_fmac_f32_dpp: hipcc fuses DPP moves into adds and multiplies but not into
// fused multiply-adds). The block opens with the two wait states a DPP read of
// a freshly written VGPR needs; inside it only the accumulator is written.
#define OCT_Q(k) " quad_perm:[" #k "," #k "," #k "," #k "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// init + qb<1>(x) y1 + qb<2>(x) y2 + qb<3>(x) y3: x as held by the three joint lanes
UPKIE_HD float oct_sumj(float init, float x, float y1, float y2, float y3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %1, %2" OCT_Q(1) "v_fmac_f32_dpp %0, %1, %3" OCT_Q(2) "v_fmac_f32_dpp %0, %1, %4" OCT_Q(3)
      : "+v"(init)
      : "v"(x), "v"(y1), "v"(y2), "v"(y3));
  return init;
#else
  return fmaf(oct_qb<3>(x), y3, fmaf(oct_qb<2>(x), y2, fmaf(oct_qb<1>(x), y1, init)));
#endif
}
UPKIE_HD float oct_sumj_neg(float init, float x, float y1, float y2, float y3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, -%1, %2" OCT_Q(1) "v_fmac_f32_dpp %0, -%1, %3" OCT_Q(2) "v_fmac_f32_dpp %0, -%1, %4" OCT_Q(3)
      : "+v"(init)
      : "v"(x), "v"(y1), "v"(y2), "v"(y3));
  return init;
#else
  return fmaf(-oct_qb<3>(x), y3, fmaf(-oct_qb<2>(x), y2, fmaf(-oct_qb<1>(x), y1, init)));
#endif
}
// the same for five registers at once (one wait, fifteen multiply-adds): acc[i] -= sum_j qb<j>(x[i]) y_j
UPKIE_HD void oct_sumj_neg5(float (&acc)[5], const float (&x)[5], float y1, float y2, float y3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, -%5, %10" OCT_Q(1) "v_fmac_f32_dpp %1, -%6, %10" OCT_Q(1) "v_fmac_f32_dpp %2, -%7, %10" OCT_Q(1)
      "v_fmac_f32_dpp %3, -%8, %10" OCT_Q(1) "v_fmac_f32_dpp %4, -%9, %10" OCT_Q(1)
      "v_fmac_f32_dpp %0, -%5, %11" OCT_Q(2) "v_fmac_f32_dpp %1, -%6, %11" OCT_Q(2) "v_fmac_f32_dpp %2, -%7, %11" OCT_Q(2)
      "v_fmac_f32_dpp %3, -%8, %11" OCT_Q(2) "v_fmac_f32_dpp %4, -%9, %11" OCT_Q(2)
      "v_fmac_f32_dpp %0, -%5, %12" OCT_Q(3) "v_fmac_f32_dpp %1, -%6, %12" OCT_Q(3) "v_fmac_f32_dpp %2, -%7, %12" OCT_Q(3)
      "v_fmac_f32_dpp %3, -%8, %12" OCT_Q(3) "v_fmac_f32_dpp %4, -%9, %12" OCT_Q(3)
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(y1), "v"(y2), "v"(y3));
#else
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = fmaf(-oct_qb<3>(x[i]), y3, fmaf(-oct_qb<2>(x[i]), y2, fmaf(-oct_qb<1>(x[i]), y1, acc[i])));
#endif
}
// init + sum_i qb<K>(x[i]) y[i]: a vector held by lane K of the quad against an own vector
template <int K>
UPKIE_HD float oct_dot3(float init, float x0, float x1, float x2, float y0, float y1, float y2) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %1, %4 quad_perm:[%7,%7,%7,%7] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %2, %5 quad_perm:[%7,%7,%7,%7] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %3, %6 quad_perm:[%7,%7,%7,%7] row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(init)
      : "v"(x0), "v"(x1), "v"(x2), "v"(y0), "v"(y1), "v"(y2), "i"(K));
  return init;
#else
  return fmaf(oct_qb<K>(x2), y2, fmaf(oct_qb<K>(x1), y1, fmaf(oct_qb<K>(x0), y0, init)));
#endif
}
template <int K>
UPKIE_HD float oct_dot3_neg(float init, float x0, float x1, float x2, float y0, float y1, float y2) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, -%1, %4 quad_perm:[%7,%7,%7,%7] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, -%2, %5 quad_perm:[%7,%7,%7,%7] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, -%3, %6 quad_perm:[%7,%7,%7,%7] row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(init)
      : "v"(x0), "v"(x1), "v"(x2), "v"(y0), "v"(y1), "v"(y2), "i"(K));
  return init;
#else
  return fmaf(-oct_qb<K>(x2), y2, fmaf(-oct_qb<K>(x1), y1, fmaf(-oct_qb<K>(x0), y0, init)));
#endif
}
template <int K>
UPKIE_HD float oct_dot6(float init, const float (&x)[6], const float (&y)[6]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %1, %7 quad_perm:[%13,%13,%13,%13] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %2, %8 quad_perm:[%13,%13,%13,%13] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %3, %9 quad_perm:[%13,%13,%13,%13] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %4, %10 quad_perm:[%13,%13,%13,%13] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %5, %11 quad_perm:[%13,%13,%13,%13] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %6, %12 quad_perm:[%13,%13,%13,%13] row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(init)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]),
        "v"(y[5]), "i"(K));
  return init;
#else
  float acc = init;
#pragma unroll
  for (int i = 0; i < 6; ++i) acc = fmaf(oct_qb<K>(x[i]), y[i], acc);
  return acc;
#endif
}

// h_k = qb<k>(x0) y0 + qb<k>(x1) y1 + qb<k>(x2) y2 for k = 1, 2, 3: the own vector against the three joint lanes' vectors
UPKIE_HD void oct_dot3_all(float x0, float x1, float x2, float y0, float y1, float y2, float& h1, float& h2, float& h3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %0, %3, %6" OCT_Q(1) "v_mul_f32_dpp %1, %3, %6" OCT_Q(2) "v_mul_f32_dpp %2, %3, %6" OCT_Q(3)
      "v_fmac_f32_dpp %0, %4, %7" OCT_Q(1) "v_fmac_f32_dpp %1, %4, %7" OCT_Q(2) "v_fmac_f32_dpp %2, %4, %7" OCT_Q(3)
      "v_fmac_f32_dpp %0, %5, %8" OCT_Q(1) "v_fmac_f32_dpp %1, %5, %8" OCT_Q(2) "v_fmac_f32_dpp %2, %5, %8" OCT_Q(3)
      : "=&v"(h1), "=&v"(h2), "=&v"(h3)
      : "v"(x0), "v"(x1), "v"(x2), "v"(y0), "v"(y1), "v"(y2));
#else
  h1 = fmaf(oct_qb<1>(x2), y2, fmaf(oct_qb<1>(x1), y1, oct_qb<1>(x0) * y0));
  h2 = fmaf(oct_qb<2>(x2), y2, fmaf(oct_qb<2>(x1), y1, oct_qb<2>(x0) * y0));
  h3 = fmaf(oct_qb<3>(x2), y2, fmaf(oct_qb<3>(x1), y1, oct_qb<3>(x0) * y0));
#endif
}
// out[i] = sum_j qb<j>(x[i]) y_j for five registers (one wait, fifteen instructions)
UPKIE_HD void oct_sumj5(float (&out)[5], const float (&x)[5], float y1, float y2, float y3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %0, %5, %10" OCT_Q(1) "v_mul_f32_dpp %1, %6, %10" OCT_Q(1) "v_mul_f32_dpp %2, %7, %10" OCT_Q(1)
      "v_mul_f32_dpp %3, %8, %10" OCT_Q(1) "v_mul_f32_dpp %4, %9, %10" OCT_Q(1)
      "v_fmac_f32_dpp %0, %5, %11" OCT_Q(2) "v_fmac_f32_dpp %1, %6, %11" OCT_Q(2) "v_fmac_f32_dpp %2, %7, %11" OCT_Q(2)
      "v_fmac_f32_dpp %3, %8, %11" OCT_Q(2) "v_fmac_f32_dpp %4, %9, %11" OCT_Q(2)
      "v_fmac_f32_dpp %0, %5, %12" OCT_Q(3) "v_fmac_f32_dpp %1, %6, %12" OCT_Q(3) "v_fmac_f32_dpp %2, %7, %12" OCT_Q(3)
      "v_fmac_f32_dpp %3, %8, %12" OCT_Q(3) "v_fmac_f32_dpp %4, %9, %12" OCT_Q(3)
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(y1), "v"(y2), "v"(y3));
#else
#pragma unroll
  for (int i = 0; i < 5; ++i) out[i] = fmaf(oct_qb<3>(x[i]), y3, fmaf(oct_qb<2>(x[i]), y2, oct_qb<1>(x[i]) * y1));
#endif
}
// out[a - 1] = sum_i qb<a>(x[i]) y[i] (+ sum_j qb<a>(u[j]) v[j]) for a = 1, 2, 3: a six-vector (and a three-vector)
// held by each joint lane against own ones -- a 3-row slice of the contact matrix in one block
#define OCT_ROW6(o, k) \
  "v_mul_f32_dpp " o ", %3, %9" OCT_Q(k) "v_fmac_f32_dpp " o ", %4, %10" OCT_Q(k) "v_fmac_f32_dpp " o ", %5, %11" OCT_Q(k) \
  "v_fmac_f32_dpp " o ", %6, %12" OCT_Q(k) "v_fmac_f32_dpp " o ", %7, %13" OCT_Q(k) "v_fmac_f32_dpp " o ", %8, %14" OCT_Q(k)
#define OCT_ROW3(o, k) "v_fmac_f32_dpp " o ", %15, %18" OCT_Q(k) "v_fmac_f32_dpp " o ", %16, %19" OCT_Q(k) "v_fmac_f32_dpp " o ", %17, %20" OCT_Q(k)
UPKIE_HD void oct_rows6(float (&out)[3], const float (&x)[6], const float (&y)[6]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t" OCT_ROW6("%0", 1) OCT_ROW6("%1", 2) OCT_ROW6("%2", 3)
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]));
#else
  out[0] = oct_dot6<1>(0.f, x, y);
  out[1] = oct_dot6<2>(0.f, x, y);
  out[2] = oct_dot6<3>(0.f, x, y);
#endif
}
UPKIE_HD void oct_rows9(float (&out)[3], const float (&x)[6], const float (&y)[6], const float (&u)[3], const float (&v)[3]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t" OCT_ROW6("%0", 1) OCT_ROW3("%0", 1) OCT_ROW6("%1", 2) OCT_ROW3("%1", 2) OCT_ROW6("%2", 3) OCT_ROW3("%2", 3)
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]),
        "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(v[0]), "v"(v[1]), "v"(v[2]));
#else
  out[0] = oct_dot3<1>(oct_dot6<1>(0.f, x, y), u[0], u[1], u[2], v[0], v[1], v[2]);
  out[1] = oct_dot3<2>(oct_dot6<2>(0.f, x, y), u[0], u[1], u[2], v[0], v[1], v[2]);
  out[2] = oct_dot3<3>(oct_dot6<3>(0.f, x, y), u[0], u[1], u[2], v[0], v[1], v[2]);
#endif
}
// acc[i] -= sum_k qb<k>(x[i]) y_k for four registers
UPKIE_HD void oct_sumj_neg4(float (&acc)[4], const float (&x)[4], float y1, float y2, float y3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, -%4, %8" OCT_Q(1) "v_fmac_f32_dpp %1, -%5, %8" OCT_Q(1) "v_fmac_f32_dpp %2, -%6, %8" OCT_Q(1) "v_fmac_f32_dpp %3, -%7, %8" OCT_Q(1)
      "v_fmac_f32_dpp %0, -%4, %9" OCT_Q(2) "v_fmac_f32_dpp %1, -%5, %9" OCT_Q(2) "v_fmac_f32_dpp %2, -%6, %9" OCT_Q(2) "v_fmac_f32_dpp %3, -%7, %9" OCT_Q(2)
      "v_fmac_f32_dpp %0, -%4, %10" OCT_Q(3) "v_fmac_f32_dpp %1, -%5, %10" OCT_Q(3) "v_fmac_f32_dpp %2, -%6, %10" OCT_Q(3) "v_fmac_f32_dpp %3, -%7, %10" OCT_Q(3)
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(y1), "v"(y2), "v"(y3));
#else
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = fmaf(-oct_qb<3>(x[i]), y3, fmaf(-oct_qb<2>(x[i]), y2, fmaf(-oct_qb<1>(x[i]), y1, acc[i])));
#endif
}

// ---------------------------------------------------------------- per-lane constants
// What a lane keeps in registers for the whole launch: the inertial record and
// the joint of ITS body, and the 0 / 1 weights that stand for "if this lane is
// ..." in a stream every lane executes.
struct OctLane {
  int l, leg;         // lane of the quad (0 trunk, 1 thigh, 2 calf, 3 wheel), leg (0 left, 1 right)
  float m, c[3], I[6];  // mass, centre of mass in the body frame, inertia about it (xx yy zz xy xz yz); the right quad's trunk: 0
  float p[3];         // joint origin in the parent's frame (trunk: 0)
  float sg;           // joint axis = sg * y (trunk: 0)
  float damping, lower, upper, effort, velocity, friction, control_noise, measurement_noise;
  bool bounded;
  float wheel_center[3];  // tire centre relative to the wheel joint of this leg
  float wj;           // 1 on the joint lanes, 0 on the trunk lane
  float e[3];         // one-hot of the joint index l - 1 (all 0 on the trunk lane)
  float w0;           // 1 on the trunk lane
  float w0_once;      // 1 on the trunk lane of the left quad
  float keep_psi;     // 0 on the wheel lane when the wheel is axisymmetric
  float kl, ka;       // Bullet-style base damping, on the lane that owns the real trunk
  // the same in every lane and every substep of a launch
  float inv_h, erp, cfm;  // 1 / h, the normal rows' error reduction and constraint force mixing (from the contact stiffness / damping)
  float total_mass;       // sum of the env's link masses: entry (1,1) of the base block
};

// the lane's row of DevModel::oct_table: eight 16-byte loads issued together (cached: eight rows for the whole grid).
// Its own function so that a kernel can issue them with its first loads, before the settings they are combined with arrive.
template <class ModelT>
UPKIE_HD void load_oct_table_row(const ModelT& M, int l, int leg, float (&t)[OT_WORDS]) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float Vec4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) Vec4* GlobalVec;
  GlobalVec row = (GlobalVec)(const void*)&M.oct_table[4 * leg + l][0];
#pragma unroll
  for (int i = 0; i < OT_WORDS / 4; ++i) {
    const Vec4 v = row[i];
    t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w;
  }
#else
  for (int i = 0; i < OT_WORDS; ++i) t[i] = M.oct_table[4 * leg + l][i];
#endif
}

template <class ModelT, class LimitsT, class ConfigT>
UPKIE_HD OctLane load_oct_lane(const ModelT& M, const LimitsT& Lm, const ConfigT& C, int l, int leg, const float* records, size_t stride,
                               const float (*preloaded_row)[OT_WORDS] = nullptr) {
  float t[OT_WORDS];
  if (preloaded_row) {
#pragma unroll
    for (int i = 0; i < OT_WORDS; ++i) t[i] = (*preloaded_row)[i];
  } else {
    load_oct_table_row(M, l, leg, t);
  }
  OctLane L;
  L.l = l;
  L.leg = leg;
  const bool trunk = l == 0, real = !(trunk && leg == 1);
  const int k = trunk ? 0 : l - 1, body = trunk ? 0 : 1 + 3 * leg + k, joint = 3 * leg + k;
  L.m = t[OT_MASS];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    L.c[d] = t[OT_COM + d];
    L.p[d] = t[OT_POS + d];
    L.wheel_center[d] = t[OT_WHEEL_CENTER + d];
    L.e[d] = t[OT_E + d];
  }
#pragma unroll
  for (int d = 0; d < 6; ++d) L.I[d] = t[OT_INERTIA + d];
  if (records) {  // this env's inertial record of the lane's body (randomize_inertias)
    const float* r = records + (size_t)(UPKIE_INERTIAL_WORDS * body) * stride;
    L.m = real ? r[0] : 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) L.c[d] = r[(size_t)(1 + d) * stride];
#pragma unroll
    for (int d = 0; d < 6; ++d) L.I[d] = real ? r[(size_t)(4 + d) * stride] : 0.f;
  }
  L.sg = t[OT_SIGN];
  L.damping = t[OT_DAMPING];
  L.lower = t[OT_LOWER];
  L.upper = t[OT_UPPER];
  L.bounded = t[OT_BOUNDED] != 0.f && Lm.enforce != 0;  // (the handle's switch folded in here, once per launch: the substep reads no setting)
  L.effort = t[OT_EFFORT];
  L.velocity = t[OT_VELOCITY];
  L.wj = t[OT_WJ];
  L.w0 = t[OT_W0];
  L.w0_once = t[OT_W0_ONCE];
  L.keep_psi = t[OT_KEEP_PSI];
  L.kl = t[OT_KL];
  L.ka = t[OT_KA];
  // per-joint settings of the config (kernel arguments)
  float fr = 0.f, cn = 0.f, mn = 0.f;
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    fr = j == joint ? C.joint_friction[j] : fr;
    cn = j == joint ? C.control_noise[j] : cn;
    mn = j == joint ? C.measurement_noise[j] : mn;
  }
  L.friction = trunk ? 0.f : fr;
  L.control_noise = trunk ? 0.f : cn;
  L.measurement_noise = trunk ? 0.f : mn;
  {
    const float h = C.h, denom = h * M.contact_stiffness + M.contact_damping;
    L.inv_h = fast_rcp(h);
    L.erp = denom > 0.f ? h * M.contact_stiffness * fast_rcp(denom) : 0.2f;
    L.cfm = denom > 0.f ? fast_rcp(denom * h) : 0.f;
  }
  L.total_mass = oct_esum(L.m);
  (void)Lm;
  return L;
}

// Physics state as a lane holds it: the base in every lane, the own joint.
struct OctPhys {
  V3 pos;
  float qw, qx, qy, qz;
  V3 linvel, angvel;
  float q, qd;  // own joint (trunk lane: 0)
  // Gauss-Seidel warm start across the substeps of a launch (not part of the stored state): the impulse of the own
  // contact row the sweeps of the PREVIOUS substep ended on, and whether they ran with both tires on the floor
  float lam_prev = 0.f;
  int swept_prev = 0;  // 0: the previous substep did not sweep; 1 / 2: it did, with one / both tires touching
  // Bullet-like contact model (BULLET_LIKE instantiations): the applied normal impulse of the own leg's cached contact
  // point, the same in the four lanes of the quad; persists from step to step in the env's contact manifold
  float bl_applied = 0.f;
};

// 6x6 LDL' of the base block of a robot whose legs move in the sagittal plane:
// F_y = 0 for every joint, so A(1,0) = A(2,1) = A(4,1) = 0 and with them
// l10 = l21 = l41 = 0. A packed lower by rows as in ldl6_factor.
struct Ldl6Planar {
  float l20, l30, l31, l32, l40, l42, l43, l50, l51, l52, l53, l54;
  float i0, i1, i2, i3, i4, i5;
};
UPKIE_HD void ldl6_factor_planar(const float (&A)[21], Ldl6Planar& f) {
  f.i0 = fast_rcp(A[0]);
  const float a20 = A[3], a30 = A[6], a40 = A[10], a50 = A[15];
  f.l20 = a20 * f.i0; f.l30 = a30 * f.i0; f.l40 = a40 * f.i0; f.l50 = a50 * f.i0;
  f.i1 = fast_rcp(A[2]);
  const float a31 = A[7], a51 = A[16];
  f.l31 = a31 * f.i1; f.l51 = a51 * f.i1;
  const float d2 = A[5] - f.l20 * a20;
  f.i2 = fast_rcp(d2);
  const float a32 = A[8] - f.l30 * a20, a42 = A[12] - f.l40 * a20, a52 = A[17] - f.l50 * a20;
  f.l32 = a32 * f.i2; f.l42 = a42 * f.i2; f.l52 = a52 * f.i2;
  const float d3 = A[9] - f.l30 * a30 - f.l31 * a31 - f.l32 * a32;
  f.i3 = fast_rcp(d3);
  const float a43 = A[13] - f.l40 * a30 - f.l42 * a32, a53 = A[18] - f.l50 * a30 - f.l51 * a31 - f.l52 * a32;
  f.l43 = a43 * f.i3; f.l53 = a53 * f.i3;
  const float d4 = A[14] - f.l40 * a40 - f.l42 * a42 - f.l43 * a43;
  f.i4 = fast_rcp(d4);
  const float a54 = A[19] - f.l50 * a40 - f.l52 * a42 - f.l53 * a43;
  f.l54 = a54 * f.i4;
  const float d5 = A[20] - f.l50 * a50 - f.l51 * a51 - f.l52 * a52 - f.l53 * a53 - f.l54 * a54;
  f.i5 = fast_rcp(d5);
}
UPKIE_HD void ldl6_solve_planar(const Ldl6Planar& f, float (&x)[6]) {
  x[2] -= f.l20 * x[0];
  x[3] -= f.l30 * x[0] + f.l31 * x[1] + f.l32 * x[2];
  x[4] -= f.l40 * x[0] + f.l42 * x[2] + f.l43 * x[3];
  x[5] -= f.l50 * x[0] + f.l51 * x[1] + f.l52 * x[2] + f.l53 * x[3] + f.l54 * x[4];
  x[0] *= f.i0; x[1] *= f.i1; x[2] *= f.i2; x[3] *= f.i3; x[4] *= f.i4; x[5] *= f.i5;
  x[4] -= f.l54 * x[5];
  x[3] -= f.l43 * x[4] + f.l53 * x[5];
  x[2] -= f.l32 * x[3] + f.l42 * x[4] + f.l52 * x[5];
  x[1] -= f.l31 * x[3] + f.l51 * x[5];
  x[0] -= f.l20 * x[2] + f.l30 * x[3] + f.l40 * x[4] + f.l50 * x[5];
}

// Gauss-Jordan across the three joint lanes of a quad: lane j + 1 holds row j of
// [D (3x3) | E (NE extra columns)]; on return E holds row j of D^-1 E. D is
// symmetric positive definite (a tire's own Delassus block or its Schur
// complement): no pivoting. The trunk lane carries zeros along (its reciprocal
// pivot is that of a unit diagonal).
template <int NE>
UPKIE_HD void oct_gauss_jordan(const OctLane& L, float (&D)[3], float (&E)[NE]) {
  // pivot 0 (lane 1)
  {
    const float ip = fast_rcp(oct_qb<1>(D[0]));
    const float f = (1.f - L.e[0]) * (D[0] * ip);  // 0 on the pivot lane itself
    D[1] -= f * oct_qb<1>(D[1]);
    D[2] -= f * oct_qb<1>(D[2]);
#pragma unroll
    for (int i = 0; i < NE; ++i) E[i] -= f * oct_qb<1>(E[i]);
  }
  {
    const float ip = fast_rcp(oct_qb<2>(D[1]));
    const float f = (1.f - L.e[1]) * (D[1] * ip);
    D[2] -= f * oct_qb<2>(D[2]);
#pragma unroll
    for (int i = 0; i < NE; ++i) E[i] -= f * oct_qb<2>(E[i]);
  }
  {
    const float ip = fast_rcp(oct_qb<3>(D[2]));
    const float f = (1.f - L.e[2]) * (D[2] * ip);
#pragma unroll
    for (int i = 0; i < NE; ++i) E[i] -= f * oct_qb<3>(E[i]);
  }
  const float diag = L.e[0] * D[0] + L.e[1] * D[1] + L.e[2] * D[2] + L.w0;
  const float id = fast_rcp(diag);
#pragma unroll
  for (int i = 0; i < NE; ++i) E[i] *= id;
}

// A hip or knee at its stop (rare): contacts and joint-limit rows are solved
// together, as in the other mappings. The system is gathered into every lane of
// the env -- both legs' D and Hinv, the six contact rows, joint angles and
// rates -- and every lane runs the shared solve on identical data, then keeps
// the base velocity change and its own joint's:
//   REGISTERS (the Servos instantiations, whose agents may drive joints into
//   their stops all the time): limit_path, the ten-row solve in registers of
//   the two-lane kernel (about 8 us per substep; the kernel then needs 512
//   registers, its common path pays a few AGPR moves);
//   otherwise octet_limit_path_scratch / general_constraint_solve: rows listed
//   in scratch memory (about ten times slower, costs the kernel no registers:
//   the Pendulum / Gyropod kernels, whose legs are held by the servos, keep a
//   clean 256-register common path).
template <int K>
UPKIE_HD float oct_from_joint(float x) { return oct_qb<K + 1>(x); }

template <class ModelT, class LimitsT>
UPKIE_HD void octet_limit_path_registers(const ModelT& M, const LimitsT& Lm_, const OctLane& L, const Ldl6Planar& fac, const float (&Dc)[6], float hv0,
                               float hv1, float hv2, V3 o, V3 Pc, V3 nB, float iun, V3 vB, V3 wB, float dist, bool active,
                               bool active_partner, float q, float qd, float tl, const float (&rt)[6], float cfm, float erp, float ih, float vmax, float h,
                               float (&xb)[6], float& xl) {
  const bool left = L.leg == 0;
  DevLimits Lm;  // (a register copy: the limits may sit in the constant address space, limit_path takes plain arrays)
#pragma unroll
  for (int j = 0; j < UPKIE_NJ; ++j) {
    Lm.lower[j] = Lm_.lower[j];
    Lm.upper[j] = Lm_.upper[j];
    Lm.bounded[j] = Lm_.bounded[j];
  }
  System S;
  S.A.l10 = 0.f; S.A.l20 = fac.l20; S.A.l21 = 0.f; S.A.l30 = fac.l30; S.A.l31 = fac.l31; S.A.l32 = fac.l32;
  S.A.l40 = fac.l40; S.A.l41 = 0.f; S.A.l42 = fac.l42; S.A.l43 = fac.l43;
  S.A.l50 = fac.l50; S.A.l51 = fac.l51; S.A.l52 = fac.l52; S.A.l53 = fac.l53; S.A.l54 = fac.l54;
  S.A.i0 = fac.i0; S.A.i1 = fac.i1; S.A.i2 = fac.i2; S.A.i3 = fac.i3; S.A.i4 = fac.i4; S.A.i5 = fac.i5;
  // contact rows of the own tire (lanes 1-3 of the quad), as in the eight-lane path
  const float sa = oct_qb<3>(L.sg);
  const V3 t1 = (sa * iun) * v3(nB.z, 0.f, -nB.x);
  const V3 t2 = cross(nB, t1);
  const V3 d = L.e[0] * nB + L.e[1] * t1 + L.e[2] * t2;
  const V3 Pxd = cross(Pc, d);
  const float Jb[6] = {d.x, d.y, d.z, Pxd.x, Pxd.y, Pxd.z};
  const float srz = L.sg * (Pc.z - o.z), srx = L.sg * (Pc.x - o.x);
  const float Jl[3] = {oct_qb<1>(srz) * d.x - oct_qb<1>(srx) * d.z, oct_qb<2>(srz) * d.x - oct_qb<2>(srx) * d.z,
                       oct_qb<3>(srz) * d.x - oct_qb<3>(srx) * d.z};
  float vnow = Jb[0] * vB.x + Jb[1] * vB.y + Jb[2] * vB.z + Jb[3] * wB.x + Jb[4] * wB.y + Jb[5] * wB.z;
  vnow = oct_sumj(vnow, qd, Jl[0], Jl[1], Jl[2]);
  float Jt[6];
  {
    float acc[5] = {Jb[0], Jb[2], Jb[3], Jb[4], Jb[5]};
    const float dc[5] = {Dc[0], Dc[2], Dc[3], Dc[4], Dc[5]};
    oct_sumj_neg5(acc, dc, Jl[0], Jl[1], Jl[2]);
    Jt[0] = acc[0]; Jt[1] = Jb[1]; Jt[2] = acc[1]; Jt[3] = acc[2]; Jt[4] = acc[3]; Jt[5] = acc[4];
  }
  // everything of both legs, in the fixed (left, right) order of the shared solve
  float Jt6[6][6], Jb6[6][6], Jl6[6][3], vn6[6], q6[6], qd6[6], tl6[2][3], d2[2];
  bool act2[2];
  {
    const float hv[3] = {hv0, hv1, hv2};
    const float other_dist = oct_swp(dist);
    d2[0] = left ? dist : other_dist;
    d2[1] = left ? other_dist : dist;
    act2[0] = left ? active : active_partner;
    act2[1] = left ? active_partner : active;
    auto joint = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      auto both = [&](float x, float& l_, float& r_) {
        const float own = oct_from_joint<k>(x), other = oct_swp(own);
        l_ = left ? own : other;
        r_ = left ? other : own;
      };
#pragma unroll
      for (int r = 0; r < 6; ++r) both(Dc[r], S.leg[0].D[r][k], S.leg[1].D[r][k]);
#pragma unroll
      for (int a = 0; a <= k; ++a) {  // Hinv[a][k], a <= k (symmetric storage 00 11 22 01 02 12)
        const int idx = a == k ? a : (a == 0 ? (k == 1 ? 3 : 4) : 5);
        both(hv[a], S.leg[0].Hinv[idx], S.leg[1].Hinv[idx]);
      }
      both(tl, tl6[0][k], tl6[1][k]);
      both(q, q6[k], q6[3 + k]);
      both(qd, qd6[k], qd6[3 + k]);
      both(vnow, vn6[k], vn6[3 + k]);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        both(Jt[c], Jt6[k][c], Jt6[3 + k][c]);
        both(Jb[c], Jb6[k][c], Jb6[3 + k][c]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) both(Jl[j], Jl6[k][j], Jl6[3 + k][j]);
    };
    joint(std::integral_constant<int, 0>{});
    joint(std::integral_constant<int, 1>{});
    joint(std::integral_constant<int, 2>{});
  }
  // base impulse before the constraints: rt is tb reduced by the legs' impulses
  float tb[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float v = rt[c];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int k = 0; k < 3; ++k) v = fmaf(S.leg[w].D[c][k], tl6[w][k], v);
    tb[c] = v;
  }
  float contact_lam[6];
  limit_path(M, S, Lm.lower, Lm.upper, Lm.bounded, q6, qd6, Jt6, Jb6, Jl6, vn6, d2, act2, cfm, erp, ih, vmax, h, rt, tb, tl6[0], tl6[1], contact_lam);
  system_solve<true, true>(S, tb, tl6[0], tl6[1]);
#pragma unroll
  for (int c = 0; c < 6; ++c) xb[c] = tb[c];
  const float mine_l = L.l == 1 ? tl6[0][0] : (L.l == 2 ? tl6[0][1] : tl6[0][2]);
  const float mine_r = L.l == 1 ? tl6[1][0] : (L.l == 2 ? tl6[1][1] : tl6[1][2]);
  xl = L.l == 0 ? 0.f : (left ? mine_l : mine_r);
}

// The same path with the rows written straight into the general solver's structures (no register arrays in between:
// gathered into registers first, they cost the common path of a 256-register kernel spills). Those structures are
// indexed dynamically; they live in the env's LimitWorkspace, which the kernel keeps in LDS (one per env of the
// wavefront, 2.2 KB each): as private arrays they were 2112 B of scratch memory per lane = 135 KB per wavefront for a
// path no robot of a Pendulum batch takes, and the runtime sizes the scratch ring for every wavefront slot -- under its
// default limit that left room for ONE wavefront per SIMD, so a second wavefront of this 256-register kernel never
// became resident (16384 envs took as long as two launches of 8192). The eight lanes of an env write the same values to
// the same words and read them back: every lane on identical data, as before.
struct LimitLeg {
  float Hinv[6];  // 00 11 22 01 02 12
  float D[6][3];
};
struct LimitWorkspace {
  LimitLeg leg[2];
  GeneralRows R;
  GeneralWork W;
};
struct LimitSystemRef {  // what system_solve / general_constraint_solve read of a System
  Ldl6 A;                // registers (named fields)
  LimitLeg* leg;         // the workspace's
};
// which of the wavefront's eight envs this lane belongs to (the workspace slot is worked out here, inside the rare
// branch: computed in the kernel's prologue the address stayed live through the common path and cost it spills)
UPKIE_HD int oct_env_slot() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(((threadIdx.x >> 4) << 1) | ((threadIdx.x >> 2) & 1));
#else
  return 0;
#endif
}
// BULLET_LIKE (round 6): the same rows under the Bullet-like specification -- no friction CFM, and the sweeps of
// general_constraint_solve_bullet_like (dynamics.hpp) instead of the default model's solve: a joint within reach of its stop is
// solved INSIDE the specification's 50 sweeps on this mapping too; `bf`: the base frame (btPlaneSpace1's directions), `applied`:
// the own tire's applied normal impulse, in / out.
template <bool BULLET_LIKE = false, class ModelT>
UPKIE_HD void octet_limit_path_scratch(const ModelT& M, const OctLane& L, const Ldl6Planar& fac, const float (&Dc)[6], float hv0, float hv1, float hv2,
                               V3 o, V3 Pc, V3 nB, float iun, V3 vB, V3 wB, float dist, bool active, bool active_partner, float qd,
                               float tl, const float (&rt)[6], float cfm, float erp, float ih, float lim_sign, float lim_bias,
                               float (&xb)[6], float& xl, LimitWorkspace& ws, const BaseFrame* bf = nullptr, float* applied = nullptr) {
  const bool left = L.leg == 0;
  LimitSystemRef S;
  S.leg = ws.leg;
  S.A.l10 = 0.f; S.A.l20 = fac.l20; S.A.l21 = 0.f; S.A.l30 = fac.l30; S.A.l31 = fac.l31; S.A.l32 = fac.l32;
  S.A.l40 = fac.l40; S.A.l41 = 0.f; S.A.l42 = fac.l42; S.A.l43 = fac.l43;
  S.A.l50 = fac.l50; S.A.l51 = fac.l51; S.A.l52 = fac.l52; S.A.l53 = fac.l53; S.A.l54 = fac.l54;
  S.A.i0 = fac.i0; S.A.i1 = fac.i1; S.A.i2 = fac.i2; S.A.i3 = fac.i3; S.A.i4 = fac.i4; S.A.i5 = fac.i5;
  // both legs' D (6 x 3) and Hinv (symmetric: 00 11 22 01 02 12), column k from joint lane k
  float tl6[2][3];
  {
    const float hv[3] = {hv0, hv1, hv2};
    auto column = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const float own = oct_from_joint<k>(Dc[r]), other = oct_swp(own);
        S.leg[0].D[r][k] = left ? own : other;
        S.leg[1].D[r][k] = left ? other : own;
      }
#pragma unroll
      for (int a = 0; a <= k; ++a) {  // Hinv[a][k], a <= k
        const float own = oct_from_joint<k>(hv[a]), other = oct_swp(own);
        const int idx = a == k ? a : (a == 0 ? (k == 1 ? 3 : 4) : 5);
        S.leg[0].Hinv[idx] = left ? own : other;
        S.leg[1].Hinv[idx] = left ? other : own;
      }
      const float own = oct_from_joint<k>(tl), other = oct_swp(own);
      tl6[0][k] = left ? own : other;
      tl6[1][k] = left ? other : own;
    };
    column(std::integral_constant<int, 0>{});
    column(std::integral_constant<int, 1>{});
    column(std::integral_constant<int, 2>{});
  }
  // contact rows of the own tire (lanes 1-3 of the quad), as in the eight-lane path
  const float sa = oct_qb<3>(L.sg);
  const V3 t1 = (sa * iun) * v3(nB.z, 0.f, -nB.x);
  const V3 t2 = cross(nB, t1);
  const V3 d = L.e[0] * nB + L.e[1] * t1 + L.e[2] * t2;
  const V3 Pxd = cross(Pc, d);
  const float Jb[6] = {d.x, d.y, d.z, Pxd.x, Pxd.y, Pxd.z};
  const float srz = L.sg * (Pc.z - o.z), srx = L.sg * (Pc.x - o.x);
  const float Jl[3] = {oct_qb<1>(srz) * d.x - oct_qb<1>(srx) * d.z, oct_qb<2>(srz) * d.x - oct_qb<2>(srx) * d.z,
                       oct_qb<3>(srz) * d.x - oct_qb<3>(srx) * d.z};
  float vnow = Jb[0] * vB.x + Jb[1] * vB.y + Jb[2] * vB.z + Jb[3] * wB.x + Jb[4] * wB.y + Jb[5] * wB.z;
  vnow = oct_sumj(vnow, qd, Jl[0], Jl[1], Jl[2]);
  float Jt[6];
  {
    float acc[5] = {Jb[0], Jb[2], Jb[3], Jb[4], Jb[5]};
    const float dc[5] = {Dc[0], Dc[2], Dc[3], Dc[4], Dc[5]};
    oct_sumj_neg5(acc, dc, Jl[0], Jl[1], Jl[2]);
    Jt[0] = acc[0]; Jt[1] = Jb[1]; Jt[2] = acc[1]; Jt[3] = acc[2]; Jt[4] = acc[3]; Jt[5] = acc[4];
  }
  const float dist_other = oct_swp(dist);
  GeneralRows& R = ws.R;
  R.n = 0;
  int first_row[2] = {-1, -1};
  // rows of the touching tires in wheel order (left, right); each tire's rows come from its own quad
  auto tire = [&](int w) {
    const bool mine = (w == 0) == left;  // the quad this lane sits in owns tire w
    const bool touching = mine ? active : active_partner;
    const float dw = mine ? dist : dist_other;
    auto row = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const int i = R.n;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const float own_t = oct_from_joint<k>(Jt[c]), other_t = oct_swp(own_t);
        const float own_b = oct_from_joint<k>(Jb[c]), other_b = oct_swp(own_b);
        if (touching) {
          R.Jt[i][c] = mine ? own_t : other_t;
          R.Jb[i][c] = mine ? own_b : other_b;
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float own = oct_from_joint<k>(Jl[j]), other = oct_swp(own);
        if (touching) R.Jl[i][j] = mine ? own : other;
      }
      const float own_v = oct_from_joint<k>(vnow), other_v = oct_swp(own_v);
      if (touching) {
        R.vnow[i] = mine ? own_v : other_v;
        R.leg[i] = w;
        R.kind[i] = k == 0 ? 0 : 1;
        R.normal_row[i] = i - k;
        R.cfm[i] = k == 0 ? cfm : (BULLET_LIKE ? 0.f : M.friction_cfm);  // (Bullet: no friction CFM)
        R.bias[i] = k == 0 ? (dw <= 0.f ? erp * (-dw) * ih : -dw * ih) : 0.f;
        if (k == 0) first_row[w] = i;
        R.n = i + 1;
      }
    };
    row(std::integral_constant<int, 0>{});
    row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{});
  };
  tire(0);
  tire(1);
  // one row per limited joint in joint order (left hip, left knee, right hip, right knee)
  auto limit = [&](int w, auto kc) {
    constexpr int k = decltype(kc)::value;
    const bool mine = (w == 0) == left;
    const float own_s = oct_from_joint<k>(lim_sign), other_s = oct_swp(own_s);
    const float own_e = oct_from_joint<k>(lim_bias), other_e = oct_swp(own_e);
    const float own_q = oct_from_joint<k>(qd), other_q = oct_swp(own_q);
    const float sign = mine ? own_s : other_s, bias = mine ? own_e : other_e, qdj = mine ? own_q : other_q;
    if (sign != 0.f) {
      const int i = R.n;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        R.Jb[i][c] = 0.f;
        R.Jt[i][c] = -sign * S.leg[w].D[c][k];
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) R.Jl[i][j] = j == k ? sign : 0.f;
      R.vnow[i] = sign * qdj;
      R.leg[i] = w;
      R.kind[i] = 2;
      R.normal_row[i] = i;
      R.cfm[i] = 0.f;
      R.bias[i] = bias;  // (joint_limit_row, dynamics.hpp)
      R.n = i + 1;
    }
  };
  limit(0, std::integral_constant<int, 0>{});
  limit(0, std::integral_constant<int, 1>{});
  limit(1, std::integral_constant<int, 0>{});
  limit(1, std::integral_constant<int, 1>{});
  // base impulse before the constraints: rt is tb reduced by the legs' impulses
  float tb[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float v = rt[c];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int k = 0; k < 3; ++k) v = fmaf(S.leg[w].D[c][k], tl6[w][k], v);
    tb[c] = v;
  }
  if constexpr (BULLET_LIKE) {
    // btPlaneSpace1(n) for n = world z -- world (0, -1, 0) and (1, 0, 0) -- against each tire's default directions, and the tires'
    // applied normal impulses, in the (left, right) order of the shared solve
    const V3 a = v3(-bf->r10, -bf->r11, -bf->r12), b = v3(bf->r00, bf->r01, bf->r02);
    const float p4[4] = {dot(a, t1), dot(a, t2), dot(b, t1), dot(b, t2)};
    float plane[2][4], applied2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float other = oct_swp(p4[i]);
      plane[0][i] = left ? p4[i] : other;
      plane[1][i] = left ? other : p4[i];
    }
    const float other_applied = oct_swp(*applied);
    applied2[0] = left ? *applied : other_applied;
    applied2[1] = left ? other_applied : *applied;
    general_constraint_solve_bullet_like(M, S, R, rt, tb, tl6[0], tl6[1], first_row, plane, applied2, ws.W);
    *applied = left ? applied2[0] : applied2[1];
  } else {
    float lam_rows[10];
    general_constraint_solve(M, S, R, rt, tb, tl6[0], tl6[1], lam_rows, ws.W);
  }
  system_solve<true, true>(S, tb, tl6[0], tl6[1]);
#pragma unroll
  for (int c = 0; c < 6; ++c) xb[c] = tb[c];
  const float mine_l = L.l == 1 ? tl6[0][0] : (L.l == 2 ? tl6[0][1] : tl6[0][2]);
  const float mine_r = L.l == 1 ? tl6[1][0] : (L.l == 2 ? tl6[1][1] : tl6[1][2]);
  xl = L.l == 0 ? 0.f : (left ? mine_l : mine_r);
}

// The active-set solve of a substep whose direct contact solution is not admissible (contact_active_set6, dynamics.hpp, says
// what that is), WITHOUT gathering the system: one row per lane, as the direct solve has it. The block elimination of the
// direct solve (oct_gauss_jordan on the own tire's block, exchange, Schur complement, oct_gauss_jordan again) computes
// lam' N = b' for ANY matrix N stored one column per lane -- nothing in it uses symmetry --, i.e. it solves M lam = b for M = N'
// stored one ROW per lane. A row of the contact matrix A is its column (A is symmetric): what the lane already holds. And
// the rows an active set replaces -- a friction row on its bound: lam_t -+ mu lam_n = 0; the rows of a tire that does not
// push: lam = 0 -- are each one lane's own (Dg, X, rhs). So an attempt is: every lane rewrites its own row or keeps it, the
// same elimination runs again, and every lane checks ITS row's condition on the result. About 150 issue slots an attempt
// against 150 for the gather plus 390 for the gathered solve (profiles/r05_active_set.txt). `start`: the own row's warm
// start (previous substep's impulse or the direct solution, not projected). Returns the attempt accepted (1 .. 3; `lam` =
// the own row's impulse then) or 0 (`lam` untouched): the same rules, tolerances and first guess as contact_active_set6.
UPKIE_HD void oct_block_solve(const OctLane& L, const float (&D)[3], const float (&Xc)[3], float b, float& x) {
  float Dw[3] = {D[0], D[1], D[2]};
  float W[4] = {Xc[0], Xc[1], Xc[2], b};
  oct_gauss_jordan<4>(L, Dw, W);
  float WP[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) WP[i] = oct_swp(W[i]);
  float acc[4] = {D[0], D[1], D[2], b};
  oct_sumj_neg4(acc, WP, Xc[0], Xc[1], Xc[2]);
  float Sd[3] = {acc[0], acc[1], acc[2]}, rr[1] = {acc[3]};
  oct_gauss_jordan<1>(L, Sd, rr);
  x = rr[0];
}
UPKIE_HD int oct_active_set(const OctLane& L, const float (&Dg)[3], const float (&X)[3], float rhs, float start, float mu, float tolerance, float& lam) {
  const bool normal = L.l == 1, friction = L.l >= 2;
  // the first set, from the projected warm start (tire-level facts come from the quad's normal lane)
  const float start_n = fmaxf(oct_qb<1>(start), 0.f);
  const float rhs_n = oct_qb<1>(rhs);
  const bool landing = !(start_n > 0.f) && rhs_n > 0.f;
  bool push = start_n > 0.f || landing;
  const float towards = landing ? rhs : start;
  float side = friction && push && (landing || !(fabsf(start) < mu * start_n)) ? (towards > 0.f ? 1.f : -1.f) : 0.f;
  const float vtol = tolerance * oct_emax(L.wj * fabsf(rhs));
#pragma nounroll  // (one copy of the body: three would be 1.3 k instructions in each of fifty kernels, for a path that loops rarely)
  for (int attempt = 1; attempt <= UPKIE_ACTIVE_SET_ATTEMPTS; ++attempt) {
    const bool kept = L.l == 0 || (push && side == 0.f);  // (the trunk lane has no row: its by-product column stays as it is)
    float Dm[3], Xm[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      Dm[a] = kept ? Dg[a] : L.e[a];
      Xm[a] = kept ? X[a] : 0.f;
    }
    Dm[0] = fmaf(-mu, side, Dm[0]);  // (side != 0 only on a replaced friction row, whose entry here is 0)
    float x;
    oct_block_solve(L, Dm, Xm, kept ? rhs : 0.f, x);
    // what the impulses do in the own row: v = (A x - rhs)_own, from the untouched row (= column) of A
    const float x1 = oct_qb<1>(x), x2 = oct_qb<2>(x), x3 = oct_qb<3>(x);
    float v = -rhs;
    v = fmaf(Dg[0], x1, v); v = fmaf(Dg[1], x2, v); v = fmaf(Dg[2], x3, v);
    v = fmaf(X[0], oct_swp(x1), v); v = fmaf(X[1], oct_swp(x2), v); v = fmaf(X[2], oct_swp(x3), v);
    const float xs = oct_emax(L.wj * fabsf(x));
    const float xtol = tolerance * xs;
    // the own row's conditions as excesses (<= 0: met), as in contact_active_set6
    const float equation = kept && L.l != 0 ? fabsf(v) - vtol : -1.f;
    const float enters = normal && !push ? -v - vtol : -1.f;
    const float pulls = normal && push ? -x - xtol : -1.f;
    const float lim = mu * x1;
    const float leaves = friction && push && side == 0.f ? fabsf(x) - lim - xtol : -1.f;
    const float returns = side != 0.f ? v * side - 0.1f * vtol : -1.f;
    const float worst = fmaxf(fmaxf(fmaxf(equation, enters), fmaxf(pulls, leaves)), returns);
    // (fmaxf DROPS a NaN operand: a partly NaN elimination -- a pivot 1 -+ mu A_nt / A_nn that is exactly 0, inf - inf -- would
    // pass every test above; the own row's impulse and velocity are looked at directly: ADVICE r5)
#if defined(UPKIE_AB_NO_NANCHECK)
    const bool bad = !(worst <= 0.f) || !(xs <= 3.0e38f);
#else
    const bool bad = !(worst <= 0.f) || !(xs <= 3.0e38f) || !(fabsf(x) + fabsf(v) < 3.0e38f);
#endif
    if (!oct_env_any(bad)) {
      const float ln = push ? fmaxf(x1, 0.f) : 0.f, bound = mu * ln;
      lam = normal ? ln : (side != 0.f ? side * bound : fminf(fmaxf(x, -bound), bound));
      if (L.l == 0) lam = x;  // (the trunk lane's by-product, as after the direct solve)
      return attempt;
    }
    // the next set: the tire flips when its normal lane says so, else each friction lane moves its own row
    const bool flips = oct_qb<1>(enters > 0.f || pulls > 0.f ? 1.f : 0.f) != 0.f;
    side = flips || !friction ? 0.f : (leaves > 0.f ? (x > 0.f ? 1.f : -1.f) : (returns > 0.f ? 0.f : side));
    push = flips ? !push : push;
  }
  return 0;
}

// The env's 6 x 6 contact system gathered into EVERY lane of the env from its one-column-per-lane form (rows 0-2: the
// left tire's normal / rolling / lateral row, 3-5: the right tire's): A packed lower by rows, right-hand sides.
// Dg[a]: entry (a, own row) of the own tire's block, X[a]: entry (other tire's row a, own row) of the coupling block.
UPKIE_HD void oct_gather_system(const OctLane& L, const float (&Dg)[3], const float (&X)[3], float rhs, float (&A6)[21], float (&rhs6)[6]) {
  const bool left = L.leg == 0;
  // diagonal blocks: entry (a, b) of the own tire's block sits in lane b + 1 of the own quad
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const float own = b == 0 ? oct_qb<1>(Dg[a]) : (b == 1 ? oct_qb<2>(Dg[a]) : oct_qb<3>(Dg[a]));
      const float other = oct_swp(own);
      A6[a * (a + 1) / 2 + b] = left ? own : other;
      A6[(3 + a) * (4 + a) / 2 + 3 + b] = left ? other : own;
    }
  }
  // coupling block (right tire's row a, left tire's column b): the left quad's lane b + 1 holds it as X[a]
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float own = b == 0 ? oct_qb<1>(X[a]) : (b == 1 ? oct_qb<2>(X[a]) : oct_qb<3>(X[a]));
      const float other = oct_swp(own);
      A6[(3 + a) * (4 + a) / 2 + b] = left ? own : other;
    }
  }
  const float r1 = oct_qb<1>(rhs), r2 = oct_qb<2>(rhs), r3 = oct_qb<3>(rhs);
  const float p1 = oct_swp(r1), p2 = oct_swp(r2), p3 = oct_swp(r3);
  rhs6[0] = left ? r1 : p1; rhs6[1] = left ? r2 : p2; rhs6[2] = left ? r3 : p3;
  rhs6[3] = left ? p1 : r1; rhs6[4] = left ? p2 : r2; rhs6[5] = left ? p3 : r3;
}

// Contacts of one substep under the Bullet-like specification (bullet_like.hpp: what the one-lane kernels run), in the
// case a rolling wheel produces -- at most ONE cached point per tire, which the tire's deepest point replaces every
// substep (it moves <= 5 mm per substep in the wheel's frame at the joint-speed limit, the replacement threshold is
// 2 cm; the exception is a robot lying FLAT ON ITS SIDE: with the wheel plane within a few degrees of horizontal the
// deepest point of the tire circle is ill-defined, jumps along the tire when the robot rocks, and Bullet's rule -- the
// one-lane kernels' -- caches up to four points there; this variant keeps the deepest one), so the point IS the
// default specification's contact point and only the rows differ: friction directions
// along / across the point's sliding velocity -- a rotation (or reflection) of the default rolling / lateral rows within
// the tangent plane, per tire --, no friction CFM, a FIXED number of Gauss-Seidel sweeps over the dense 6 x 6 system
// (normals, then each point's friction pair projected onto the cone), normals warm-started with 0.85 x the last applied
// impulse. Every lane of the env sweeps the same system in lockstep; returns the own row's impulse in the DEFAULT
// basis (what the rest of the substep consumes). vt1 / vt2: free velocity of the own tire's point along the default
// rolling / lateral directions t1 / t2; `applied`: the own tire's applied normal impulse (in / out).
template <class ModelT>
UPKIE_HD float oct_bullet_like_solve(const ModelT& M, const OctLane& L, const BaseFrame& bf, const float (&Dg)[3], const float (&X)[3], float rhs,
                                     V3 t1, V3 t2, float vt1, float vt2, bool active, bool active_partner, float& applied) {
  const bool left = L.leg == 0;
  float A6[21], rhs6[6];
  oct_gather_system(L, Dg, X, rhs, A6, rhs6);
  // rows of the Bullet-like basis in the default one, own tire: [t1'; t2'] = R [t1; t2]
  float R00, R01, R10, R11;
  {
    const float lat2 = vt1 * vt1 + vt2 * vt2;
    if (lat2 > 1.1920929e-07f) {  // SIMD_EPSILON: t1' along the sliding velocity, t2' = t1' x n (= s t1 - c t2: t2 = n x t1)
      const float inv = 1.f / sqrtf(lat2);
      const float c = vt1 * inv, sn = vt2 * inv;
      R00 = c; R01 = sn; R10 = sn; R11 = -c;
    } else {  // btPlaneSpace1(n) for n = world z: world (0, -1, 0) and (1, 0, 0)
      const V3 a = v3(-bf.r10, -bf.r11, -bf.r12), b = v3(bf.r00, bf.r01, bf.r02);
      R00 = dot(a, t1); R01 = dot(a, t2); R10 = dot(b, t1); R11 = dot(b, t2);
    }
  }
  float Q[2][4];  // [tire] R00 R01 R10 R11
  {
    const float o0 = oct_swp(R00), o1 = oct_swp(R01), o2 = oct_swp(R10), o3 = oct_swp(R11);
    Q[0][0] = left ? R00 : o0; Q[0][1] = left ? R01 : o1; Q[0][2] = left ? R10 : o2; Q[0][3] = left ? R11 : o3;
    Q[1][0] = left ? o0 : R00; Q[1][1] = left ? o1 : R01; Q[1][2] = left ? o2 : R10; Q[1][3] = left ? o3 : R11;
  }
  float W[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) W[a][b] = a >= b ? A6[a * (a + 1) / 2 + b] : A6[b * (b + 1) / 2 + a];
#pragma unroll
  for (int w = 0; w < 2; ++w) {  // W <- Q W Q', rhs <- Q rhs: the friction rows / columns of each tire
    const int i = 3 * w + 1, j = 3 * w + 2;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float x = W[i][c], y = W[j][c];
      W[i][c] = Q[w][0] * x + Q[w][1] * y;
      W[j][c] = Q[w][2] * x + Q[w][3] * y;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const float x = W[r][i], y = W[r][j];
      W[r][i] = Q[w][0] * x + Q[w][1] * y;
      W[r][j] = Q[w][2] * x + Q[w][3] * y;
    }
    const float x = rhs6[i], y = rhs6[j];
    rhs6[i] = Q[w][0] * x + Q[w][1] * y;
    rhs6[j] = Q[w][2] * x + Q[w][3] * y;
  }
  const bool on[2] = {left ? active : active_partner, left ? active_partner : active};
  const float other_applied = oct_swp(applied);
  float lam[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  lam[0] = on[0] ? 0.85f * (left ? applied : other_applied) : 0.f;  // m_warmstartingFactor
  lam[3] = on[1] ? 0.85f * (left ? other_applied : applied) : 0.f;
#if !defined(__HIP_DEVICE_COMPILE__)
  BulletLikeProbe* const probe = L.l == 1 && L.leg == 0 ? g_bullet_like_probe : nullptr;
  if (probe) {
    float* out = probe->system;
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) *out++ = W[a][b];
    for (int a = 0; a < 6; ++a) *out++ = rhs6[a];
    for (int a = 0; a < 6; ++a) *out++ = lam[a];
    *out++ = on[0] ? 1.f : 0.f;
    *out++ = on[1] ? 1.f : 0.f;
    probe->sweeps = 0;
  }
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
  if (probe) {  // (host tests: sweep by sweep, to record every sweep's impulses -- the same arithmetic as the one call below)
    for (int it = 0; it < M.pgs_iterations; ++it) {
      float before[6];
      for (int a = 0; a < 6; ++a) before[a] = lam[a];
      bullet_like_sweeps6(W, rhs6, lam, on, 0.f, M.friction_mu, 1);
      if (it < 64) {
        float change = 0.f;
        for (int a = 0; a < 6; ++a) {
          change = fmaxf(change, fabsf(lam[a] - before[a]));
          probe->lam[it][a] = lam[a];
        }
        probe->change[it] = change;
        probe->sweeps = it + 1;
      }
    }
  } else
#endif
    bullet_like_sweeps6(W, rhs6, lam, on, 0.f, M.friction_mu, M.pgs_iterations);  // (the normal rows' CFM sits on the diagonal of the gathered system)
  applied = left ? lam[0] : lam[3];
  // back to the default basis: lam = Q' lam'
  float out[6];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    out[3 * w] = lam[3 * w];
    out[3 * w + 1] = Q[w][0] * lam[3 * w + 1] + Q[w][2] * lam[3 * w + 2];
    out[3 * w + 2] = Q[w][1] * lam[3 * w + 1] + Q[w][3] * lam[3 * w + 2];
  }
  const float mine_l = L.l == 1 ? out[0] : (L.l == 2 ? out[1] : out[2]);
  const float mine_r = L.l == 1 ? out[3] : (L.l == 2 ? out[4] : out[5]);
  return left ? mine_l : mine_r;
}

// The six model scalars the common path of a substep reads, at the values of the default model (upkie_amd/model/
// default_model.py; what the reference's wheel / floor settings come to). A handle whose model carries exactly these
// values runs instantiations in which they are compile-time constants (DEFAULT_SCALARS): no scalar load in the common
// path of the substep loop, multiplications by mu = 1 folded away -- the same arithmetic on the same values, bit for
// bit the same results (tests/test_default_scalars_gpu.py), 1.6 % less time per launch (profiles/r03_ab_model_scalars_as_constants.txt).
struct OctDefaultScalars {
  static constexpr float gravity = 9.81f, wheel_radius = 0.05f, contact_breaking_threshold = 0.02f, friction_cfm = 0.01f, friction_mu = 1.0f,
                         max_joint_velocity = 100.0f;
};
template <class ModelT>
inline bool oct_model_has_default_scalars(const ModelT& M) {
  typedef OctDefaultScalars D;
  return M.gravity == D::gravity && M.wheel_radius == D::wheel_radius && M.contact_breaking_threshold == D::contact_breaking_threshold &&
         M.friction_cfm == D::friction_cfm && M.friction_mu == D::friction_mu && M.max_joint_velocity == D::max_joint_velocity;
}
#define OCT_HOT(field) (DEFAULT_SCALARS ? OctDefaultScalars::field : M.field)
// Substep outcomes (returned) and rare paths taken (reported through `census`).
enum { OCT_NOT_MINE_INFEASIBLE = -3, OCT_NOT_MINE_LIMIT = -1, OCT_NO_CONTACT = 0, OCT_CONTACT = 1 };
struct OctRare {  // which rare path the env took this substep, Gauss-Seidel sweeps it ran (two registers, never memory)
  int path, sweeps;
};

// One physics substep, eight lanes per env. tau: commanded torque of the own
// joint (trunk lane: 0). trunk_forces: sum of the external forces on the trunk
// in the BASE frame and their moment about the base origin, or nullptr.
// Returns OCT_CONTACT / OCT_NO_CONTACT (same answer in the env's eight lanes).
// BULLET_LIKE: contacts by the Bullet-like specification (oct_bullet_like_solve above) instead of the default one; a
// joint at its stop (which the Pendulum / Gyropod / BaseVelocity envs these instantiations serve do not reach: their
// legs are held at zero by the servos) still takes the default model's joint-stop path for that substep.
template <bool LIMITS_IN_REGISTERS = false, bool DEFAULT_SCALARS = false, bool BULLET_LIKE = false, class ModelT, class LimitsT>
UPKIE_HD int physics_substep_octet(const ModelT& M, const LimitsT& Lm, const OctLane& L, OctPhys& s, float tau, float h,
                                   const float* trunk_wrench, LimitWorkspace* ws, OctRare* census = nullptr, float* manifold_out = nullptr,
                                   size_t manifold_stride = 0) {
  // ---- a joint at its stop (rare): its row joins the contact rows in the general solve below
  // (L.bounded carries the handle's `enforce` switch: read from the settings block here, `if (Lm.enforce)` was a scalar
  // load and a wait for it in every substep, 86-160 cycles of a lone wavefront each)
  bool at_a_stop = false;
  {
#if defined(UPKIE_AB_OLD_LIMITS)
    const bool own_limit = L.bounded && (s.q <= L.lower || s.q >= L.upper);
#else
    const bool own_limit = joint_limit_near(L.bounded, s.q, L.lower, L.upper, joint_limit_reach(s.qd, OCT_HOT(max_joint_velocity), h));
#endif
    if (__builtin_expect(oct_wave_any(own_limit), 0)) at_a_stop = oct_env_any(own_limit);
  }

  // ---- base frame ----------------------------------------------------------
  const BaseFrame bf = base_frame(s.qw, s.qx, s.qy, s.qz, s.linvel, s.angvel);
  const V3 vB = bf.vB, wB = bf.wB, nB = bf.nB;
  const V3 gn = OCT_HOT(gravity) * nB;

  // ---- kinematics along the chain (prefix sums over the quad) -----------------
  const float psi = L.keep_psi * oct_chain(L.sg * s.q);
  float sn, cs;
  joint_sincos(psi, &sn, &cs);
  const float pcs = oct_up1(cs), psn = oct_up1(sn);  // the parent's frame (trunk: identity)
  const V3 r = v3(pcs * L.p[0] + psn * L.p[2], L.p[1], pcs * L.p[2] - psn * L.p[0]);  // joint origin minus the parent's
  const float sq = L.sg * s.qd;
  float along[4] = {r.x, r.y, r.z, sq};
  oct_chain(along);
  const V3 o = v3(along[0], along[1], along[2]);
  const float S = along[3], Sp = S - sq;  // joint rates summed down to this body / to its parent
  // parent's omega = wB + Sp y, alpha = Sp (wB x y): origin acceleration term of this joint offset
  const V3 wp = v3(wB.x, wB.y + Sp, wB.z);
  const V3 alp = v3(-Sp * wB.z, 0.f, Sp * wB.x);
  const V3 e = cross(alp, r) + cross(wp, cross(wp, r));
  float ao3[3] = {e.x, e.y, e.z};
  oct_chain(ao3);
  const V3 ao = v3(ao3[0], ao3[1], ao3[2]);

  // ---- the own body: wrench and inertia about the base origin -----------------
  const V3 w = v3(wB.x, wB.y + S, wB.z);
  const V3 al = v3(-S * wB.z, 0.f, S * wB.x);
  const V3 rc = rot_y(cs, sn, v3(L.c[0], L.c[1], L.c[2]));
  const V3 c = o + rc;
  const S3 Ic = rot_y(cs, sn, S3{L.I[0], L.I[1], L.I[2], L.I[3], L.I[4], L.I[5]});
  const V3 ac = ao + cross(al, rc) + cross(w, cross(w, rc));
  const V3 f = L.m * (ac + gn);
  const V3 Iw = mul(Ic, w);
  const V3 N = mul(Ic, al) + cross(w, Iw) + cross(c, f);
  const V3 mc = L.m * c;
  const S3 Ib = shift_to_origin(Ic, L.m, c);
  // applied wrench on the body: Bullet-style damping of the trunk (kl, ka are 0 elsewhere), external forces on the trunk
  V3 Fa, Na;
  {
    const V3 vc = vB + cross(w, rc);
    const float vn = fast_sqrt(dot(vc, vc)), wn = fast_sqrt(dot(w, w));
    Fa = (-L.m * (L.kl + L.kl * vn)) * vc;
    Na = (-(L.ka + L.ka * wn)) * Iw + cross(rc, Fa);
    if (trunk_wrench) {
      Fa = Fa + L.w0_once * v3(trunk_wrench[0], trunk_wrench[1], trunk_wrench[2]);
      Na = Na + L.w0_once * v3(trunk_wrench[3], trunk_wrench[4], trunk_wrench[5]);
    }
  }

  // ---- subtree sums: composite wrench / inertia seen by the own joint -------
  float sub[10] = {L.wj * f.x, L.wj * f.z, L.wj * N.y, L.wj * L.m, L.wj * mc.x, L.wj * mc.y, L.wj * mc.z, L.wj * Ib.xy, L.wj * Ib.yy, L.wj * Ib.yz};
  oct_subtree(sub);
  const float fcx = sub[0], fcz = sub[1], Ncy = sub[2], Cm = sub[3];
  const V3 Ch = v3(sub[4], sub[5], sub[6]);
  const float CIxy = sub[7], CIyy = sub[8], CIyz = sub[9];
  // S = [a; o x a], a = sg y
  const float oxa_x = -L.sg * o.z, oxa_z = L.sg * o.x;
  const float bias = L.sg * Ncy + oxa_x * fcx + oxa_z * fcz;
  // F = I^c S (F[1] = 0): force (x, z), moment (x, y, z)
  float F[6];
  F[0] = Cm * oxa_x + L.sg * Ch.z;
  F[1] = 0.f;
  F[2] = Cm * oxa_z - L.sg * Ch.x;
  F[3] = L.sg * CIxy + Ch.y * oxa_z;
  F[4] = L.sg * CIyy + (Ch.z * oxa_x - Ch.x * oxa_z);
  F[5] = L.sg * CIyz - Ch.y * oxa_x;

  // ---- leg block H (3x3): row of the own joint, factor in every lane, own column of the inverse
  // H[j][k] = S_j . F_k for k >= j
  float h1, h2, h3;
  oct_dot3_all(F[4], F[0], F[2], L.sg, oxa_x, oxa_z, h1, h2, h3);
  const float H00 = oct_qb<1>(h1), H01 = oct_qb<1>(h2), H02 = oct_qb<1>(h3), H11 = oct_qb<2>(h2), H12 = oct_qb<2>(h3), H22 = oct_qb<3>(h3);
  const float i00 = fast_rsqrt(H00);
  const float l10 = H01 * i00, l20 = H02 * i00;
  const float i11 = fast_rsqrt(H11 - l10 * l10);
  const float l21 = (H12 - l20 * l10) * i11;
  const float i22 = fast_rsqrt(H22 - l20 * l20 - l21 * l21);
  // x = H^-1 b through the factor (L y = b, L' x = y)
  auto leg_solve = [&](float b0, float b1, float b2, float& x0, float& x1, float& x2) {
    const float y0 = b0 * i00;
    const float y1 = (b1 - l10 * y0) * i11;
    const float y2 = (b2 - l20 * y0 - l21 * y1) * i22;
    x2 = y2 * i22;
    x1 = (y1 - l21 * x2) * i11;
    x0 = (y0 - l10 * x1 - l20 * x2) * i00;
  };
  float hv0, hv1, hv2;  // column (= row) of Hinv of the own joint; the trunk lane: 0
  leg_solve(L.e[0], L.e[1], L.e[2], hv0, hv1, hv2);
  // D = F Hinv, own column: Dc[r] = sum_k F_k[r] Hinv[k][own]
  float Dc[6];
  {
    const float f5[5] = {F[0], F[2], F[3], F[4], F[5]};
    float d5[5];
    oct_sumj5(d5, f5, hv0, hv1, hv2);
    Dc[0] = d5[0]; Dc[1] = 0.f; Dc[2] = d5[1]; Dc[3] = d5[2]; Dc[4] = d5[3]; Dc[5] = d5[4];
  }

  // ---- base block: own body's spatial inertia minus own column's outer product, summed over the env
  Ldl6Planar fac;
  {
    // the 17 configuration-dependent entries that are not structurally zero, packed ((1,1) is the total mass)
    float P[17];
    P[0] = L.m - Dc[0] * F[0];          // (0,0)
    P[1] = -Dc[2] * F[0];               // (2,0)
    P[2] = L.m - Dc[2] * F[2];          // (2,2)
    P[3] = -Dc[3] * F[0];               // (3,0)
    P[4] = -mc.z;                       // (3,1)
    P[5] = mc.y - Dc[3] * F[2];         // (3,2)
    P[6] = Ib.xx - Dc[3] * F[3];        // (3,3)
    P[7] = mc.z - Dc[4] * F[0];         // (4,0)
    P[8] = -mc.x - Dc[4] * F[2];        // (4,2)
    P[9] = Ib.xy - Dc[4] * F[3];        // (4,3)
    P[10] = Ib.yy - Dc[4] * F[4];       // (4,4)
    P[11] = -mc.y - Dc[5] * F[0];       // (5,0)
    P[12] = mc.x;                       // (5,1)
    P[13] = -Dc[5] * F[2];              // (5,2)
    P[14] = Ib.xz - Dc[5] * F[3];       // (5,3)
    P[15] = Ib.yz - Dc[5] * F[4];       // (5,4)
    P[16] = Ib.zz - Dc[5] * F[5];       // (5,5)
    oct_esum(P);
    float A[21];
    A[0] = P[0];
    A[1] = 0.f; A[2] = L.total_mass;
    A[3] = P[1]; A[4] = 0.f; A[5] = P[2];
    A[6] = P[3]; A[7] = P[4]; A[8] = P[5]; A[9] = P[6];
    A[10] = P[7]; A[11] = 0.f; A[12] = P[8]; A[13] = P[9]; A[14] = P[10];
    A[15] = P[11]; A[16] = P[12]; A[17] = P[13]; A[18] = P[14]; A[19] = P[15]; A[20] = P[16];
    ldl6_factor_planar(A, fac);
  }

  // ---- impulses so far: own joint, and the base right-hand side reduced over the env
  const float tl = h * (tau - L.damping * s.qd - bias);
  float rt[6];
  rt[0] = h * (Fa.x - f.x) - Dc[0] * tl;
  rt[1] = h * (Fa.y - f.y);
  rt[2] = h * (Fa.z - f.z) - Dc[2] * tl;
  rt[3] = h * (Na.x - N.x) - Dc[3] * tl;
  rt[4] = h * (Na.y - N.y) - Dc[4] * tl;
  rt[5] = h * (Na.z - N.z) - Dc[5] * tl;
  oct_esum(rt);

  // ---- the tire of this leg: contact point, directions (every lane of the quad) --
  const float un = fast_sqrt(nB.x * nB.x + nB.z * nB.z);
  const float iun = fast_rcp(fmaxf(un, 1e-12f));
  const float ih = L.inv_h, erp = L.erp, cfm = L.cfm;  // of this launch's h (load_oct_lane)
  const V3 ow = v3(oct_qb<3>(o.x), oct_qb<3>(o.y), oct_qb<3>(o.z));
  const V3 center = ow + v3(L.wheel_center[0], L.wheel_center[1], L.wheel_center[2]);
  const V3 Pc = center + OCT_HOT(wheel_radius) * v3(-nB.x * iun, 0.f, -nB.z * iun);
  const float dist = s.pos.z + dot(nB, Pc);
  const bool active = un >= 1e-6f && dist <= OCT_HOT(contact_breaking_threshold);
  const bool active_partner = oct_swp(active ? 1.f : 0.f) != 0.f;
  const bool both = active && active_partner;

  float xb[6];   // base velocity change
  float tlc = tl;  // own joint impulse incl. contacts
  float xl = 0.f;  // own joint velocity change
  int swept_now = 0;
  float lam_now = 0.f;
  if (__builtin_expect(at_a_stop, 0)) {
    if (census) census->path = OCT_NOT_MINE_LIMIT;
    if (LIMITS_IN_REGISTERS && !BULLET_LIKE) {
      octet_limit_path_registers(M, Lm, L, fac, Dc, hv0, hv1, hv2, o, Pc, nB, iun, vB, wB, dist, active, active_partner, s.q, s.qd, tl, rt, cfm,
                                 erp, ih, OCT_HOT(max_joint_velocity), h, xb, xl);
    } else {
      float lim_bias;
      const float lim_sign = joint_limit_row(L.bounded, s.q, L.lower, L.upper, joint_limit_reach(s.qd, OCT_HOT(max_joint_velocity), h), ih, lim_bias);
      if (BULLET_LIKE && !active) s.bl_applied = 0.f;  // (no point cached: nothing to warm-start from)
      octet_limit_path_scratch<BULLET_LIKE>(M, L, fac, Dc, hv0, hv1, hv2, o, Pc, nB, iun, vB, wB, dist, active, active_partner, s.qd, tl, rt, cfm, erp, ih,
                                            lim_sign, lim_bias, xb, xl, ws[oct_env_slot()], &bf, &s.bl_applied);
    }
  } else if (__builtin_expect(active || active_partner, 1)) {
    const float sa = oct_qb<3>(L.sg);
    const V3 t1 = (sa * iun) * v3(nB.z, 0.f, -nB.x);
    const V3 t2 = cross(nB, t1);
    // row of the lane: normal / rolling / lateral (the trunk lane: none, it solves for the free velocity)
    const V3 d = L.e[0] * nB + L.e[1] * t1 + L.e[2] * t2;
    const V3 Pxd = cross(Pc, d);
    float Jb[6] = {d.x, d.y, d.z, Pxd.x, Pxd.y, Pxd.z};
    // joint parts: Jl[row][j] = sg_j ((P - o_j) x d) . y-ish; lane j prepares its two factors, the row lanes combine
    const float srz = L.sg * (Pc.z - o.z), srx = L.sg * (Pc.x - o.x);
    const float Jl1 = oct_qb<1>(srz) * d.x - oct_qb<1>(srx) * d.z;
    const float Jl2 = oct_qb<2>(srz) * d.x - oct_qb<2>(srx) * d.z;
    const float Jl3 = oct_qb<3>(srz) * d.x - oct_qb<3>(srx) * d.z;
    float vnow = Jb[0] * vB.x + Jb[1] * vB.y + Jb[2] * vB.z + Jb[3] * wB.x + Jb[4] * wB.y + Jb[5] * wB.z;
    vnow = oct_sumj(vnow, s.qd, Jl1, Jl2, Jl3);
    // row reduced onto the base: Jt = Jb - sum_j D[:, j] Jl[j]; the trunk lane takes rt instead
    float Y[6];
    {
      float acc[5] = {Jb[0], Jb[2], Jb[3], Jb[4], Jb[5]};
      const float dc[5] = {Dc[0], Dc[2], Dc[3], Dc[4], Dc[5]};
      oct_sumj_neg5(acc, dc, Jl1, Jl2, Jl3);
      Y[0] = acc[0]; Y[1] = Jb[1]; Y[2] = acc[1]; Y[3] = acc[2]; Y[4] = acc[3]; Y[5] = acc[4];
    }
    float Jt[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      Jt[i] = Y[i];
      Y[i] = fmaf(L.w0, rt[i], Y[i]);
    }
    ldl6_solve_planar(fac, Y);  // contact rows: A^-1 Jt; trunk lane: A^-1 rt
    float K1, K2, K3;
    leg_solve(Jl1, Jl2, Jl3, K1, K2, K3);
    float vf = vnow;
#pragma unroll
    for (int i = 0; i < 6; ++i) vf = fmaf(Y[i], rt[i], vf);
    vf = oct_sumj(vf, tl, K1, K2, K3);
    const float push = dist <= 0.f ? erp * (-dist) * ih : -dist * ih;
    // (a tire without a contact point: identity rows, zero right-hand side, no coupling, as in physics_substep)
    const float rhs = active ? L.e[0] * push - vf : 0.f;
    // own tire's block and the coupling to the other tire, one column per lane:
    // Dg[a] = Jt_a . Y + Jl_a . K for the rows a of this quad, X[a] for the rows of the other one
    float Dg[3], X[3];
    {
      const float Jl[3] = {Jl1, Jl2, Jl3}, Kv[3] = {K1, K2, K3};
      oct_rows9(Dg, Jt, Y, Jl, Kv);
    }
    Dg[0] = active ? fmaf(L.e[0], cfm, Dg[0]) : L.e[0];
    Dg[1] = active ? (BULLET_LIKE ? Dg[1] : fmaf(L.e[1], OCT_HOT(friction_cfm), Dg[1])) : L.e[1];  // (Bullet: no friction CFM)
    Dg[2] = active ? (BULLET_LIKE ? Dg[2] : fmaf(L.e[2], OCT_HOT(friction_cfm), Dg[2])) : L.e[2];
    float JtP[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) JtP[i] = oct_swp(Jt[i]);
    oct_rows6(X, JtP, Y);
#pragma unroll
    for (int i = 0; i < 3; ++i) X[i] = both ? X[i] : 0.f;
    // block elimination: W = Dg^-1 [X | rhs] in this quad, exchanged; Schur complement of the other tire
    float lam;
    if constexpr (BULLET_LIKE) {
      // (vf of the rolling / lateral lanes: the free velocity of the tire's point along t1 / t2)
      if (!active) s.bl_applied = 0.f;  // no point cached: nothing to warm-start from when the tire comes back
      lam = oct_bullet_like_solve(M, L, bf, Dg, X, rhs, t1, t2, oct_qb<2>(vf), oct_qb<3>(vf), active, active_partner, s.bl_applied);
    } else {
    {
      float Dw[3] = {Dg[0], Dg[1], Dg[2]};
      float W[4] = {X[0], X[1], X[2], rhs};
      oct_gauss_jordan<4>(L, Dw, W);
      float WP[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) WP[i] = oct_swp(W[i]);
      // S[a] = Dg[a] - sum_k X[k] WP_k[a] ; r' = rhs - sum_k X[k] WP_k[3]   (WP_k: row k of the other quad's W)
      float Sd[3], rr[1];
      {
        float acc[4] = {Dg[0], Dg[1], Dg[2], rhs};
        oct_sumj_neg4(acc, WP, X[0], X[1], X[2]);
        Sd[0] = acc[0]; Sd[1] = acc[1]; Sd[2] = acc[2]; rr[0] = acc[3];
      }
      oct_gauss_jordan<1>(L, Sd, rr);
      lam = rr[0];
    }
    // admissible? normal >= 0, friction inside the cone; otherwise the direct solution is projected and warm-starts
    // projected Gauss-Seidel sweeps: the system is gathered into every lane and the sweeps of the other mappings run
    // on it (contact_pgs6: same rows, same order), every lane of the env in lockstep on identical data
    {
      const float lam_n = oct_qb<1>(lam);
      const bool bad = L.l == 1 ? lam < 0.f : (L.l != 0 && fabsf(lam) > OCT_HOT(friction_mu) * lam_n);
      if (__builtin_expect(oct_wave_any(bad), 0)) {
        if (oct_env_any(bad)) {
          if (census) census->path = OCT_NOT_MINE_INFEASIBLE;
          // Both tires leaving the floor (neither normal row asks for an impulse, rhs_n <= 0): lam = 0 is the solution
          // -- what the sweeps return after one pass over the projected zeros; 38 % of the infeasible env-substeps of
          // the C5 workload, 95 % while robots land after a reset -- known without gathering the system.
          const float rhs_n = oct_qb<1>(rhs), rhs_n_other = oct_swp(rhs_n);  // (both exchanged before the test: every lane of the env takes part)
          if (rhs_n <= 0.f && rhs_n_other <= 0.f) {
            lam = 0.f;
          } else {
          // Warm start: the projected direct solution, or -- when the previous substep came through here too, with the
          // same tires on the floor -- the impulses it ended on: a robot that skids or tumbles does so for many substeps
          // in a row and its contact state changes little from one millisecond to the next
          const bool from_previous = s.swept_prev == (both ? 2 : 1);
          const float start = from_previous ? s.lam_prev : lam;
          // round 5: an active-set solve first, one row per lane as the direct solve (oct_active_set); the gathered
          // system and the sweeps only when none of its three sets is accepted
          const int accepted = oct_active_set(L, Dg, X, rhs, start, OCT_HOT(friction_mu), M.pgs_tolerance, lam);
          if (accepted) {
            if (census) census->sweeps = -accepted;
            swept_now = both ? 2 : 1;
          } else {
          const bool left = L.leg == 0;
          float A6[21], rhs6[6], lam6[6];
          oct_gather_system(L, Dg, X, rhs, A6, rhs6);
          {
            const float l1 = oct_qb<1>(start), l2 = oct_qb<2>(start), l3 = oct_qb<3>(start);
            const float q1 = oct_swp(l1), q2 = oct_swp(l2), q3 = oct_swp(l3);
            lam6[0] = left ? l1 : q1; lam6[1] = left ? l2 : q2; lam6[2] = left ? l3 : q3;
            lam6[3] = left ? q1 : l1; lam6[4] = left ? q2 : l2; lam6[5] = left ? q3 : l3;
          }
          // projection of the direct solution (the warm start), as in physics_substep
#pragma unroll
          for (int w = 0; w < 2; ++w) lam6[3 * w] = fmaxf(lam6[3 * w], 0.f);
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            if ((r % 3) == 0) continue;
            const float lim = OCT_HOT(friction_mu) * lam6[3 * (r / 3)];
            lam6[r] = fminf(fmaxf(lam6[r], -lim), lim);
          }
          const int sweeps = contact_pgs6(M, A6, rhs6, lam6);
          if (census) census->sweeps = sweeps;
          const float mine_l = L.l == 1 ? lam6[0] : (L.l == 2 ? lam6[1] : lam6[2]);
          const float mine_r = L.l == 1 ? lam6[3] : (L.l == 2 ? lam6[4] : lam6[5]);
          lam = left ? mine_l : mine_r;
          swept_now = both ? 2 : 1;
          }
          }
        }
      }
    }
    }
    lam_now = lam;
    // nu+ = A^-1 rt + sum_b lam_b Y_b: the trunk lane's Y is the free part (counted once)
    const float wgt = L.l == 0 ? L.w0_once : lam;  // (the trunk lane's own `lam` is the by-product of rows it does not have)
#pragma unroll
    for (int i = 0; i < 6; ++i) xb[i] = wgt * Y[i];
    oct_esum(xb);
    // own joint: t += sum_b Jl_b[own] lam_b, with Jl_b[own] rebuilt from the own factors (bit-identical to the row lanes')
    const float JlT1 = srz * nB.x - srx * nB.z, JlT2 = srz * t1.x - srx * t1.z, JlT3 = srz * t2.x - srx * t2.z;
    tlc = oct_sumj(tl, lam, JlT1, JlT2, JlT3);
  } else {
    // no contact: nu+ = A^-1 rt, every lane solves it
#pragma unroll
    for (int i = 0; i < 6; ++i) xb[i] = rt[i];
    ldl6_solve_planar(fac, xb);
    if (BULLET_LIKE) s.bl_applied = 0.f;
  }

  // ---- joint velocity change: Hinv t - D' xb ------------------------------------
  if (__builtin_expect(!at_a_stop, 1)) {
    xl = oct_sumj(0.f, tlc, hv0, hv1, hv2);
    xl -= Dc[0] * xb[0] + Dc[2] * xb[2] + Dc[3] * xb[3] + Dc[4] * xb[4] + Dc[5] * xb[5];
  }

  // ---- integrate --------------------------------------------------------------------
  {
    const float v = fminf(fmaxf(s.qd + xl, -OCT_HOT(max_joint_velocity)), OCT_HOT(max_joint_velocity));
    s.qd = L.wj * v;
    s.q = fmaf(h, s.qd, s.q);
  }
  s.lam_prev = lam_now;
  s.swept_prev = swept_now;
  if constexpr (BULLET_LIKE) {
    // The env's contact manifold as the one-lane kernels keep it (bullet_like.hpp: per tire four records of point in the
    // wheel frame (3), on the plane in world coordinates (3), applied normal impulse, live), written by the last
    // substep of a step so that both kernels can continue from it: record 0 = this substep's point (the tire's
    // deepest point: it replaced the cached one), records 1-3 empty. The normal-row lane of each quad writes its tire.
    if (manifold_out) {
      // the wheel body's orientation at the START of this substep (where Pc and ow were taken): the chain's summed joint
      // angles, the wheel's own spin included; the joints have been integrated above: q - h qd is what they were
      float wsn, wcs;
      joint_sincos(oct_qb<3>(oct_chain(L.sg * fmaf(-h, s.qd, s.q))), &wsn, &wcs);
      const V3 local = rot_y(wcs, -wsn, Pc - ow);
      const float x = s.pos.x + bf.r00 * Pc.x + bf.r01 * Pc.y + bf.r02 * Pc.z, y = s.pos.y + bf.r10 * Pc.x + bf.r11 * Pc.y + bf.r12 * Pc.z;
      if (L.l == 1) {
        float* rec = manifold_out + (size_t)(L.leg * 4 * 8) * manifold_stride;
        const float live = active ? 1.f : 0.f;
        rec[0] = live * local.x; rec[manifold_stride] = live * local.y; rec[2 * manifold_stride] = live * local.z;
        rec[3 * manifold_stride] = live * x; rec[4 * manifold_stride] = live * y; rec[5 * manifold_stride] = 0.f;
        rec[6 * manifold_stride] = active ? s.bl_applied : 0.f;
        rec[7 * manifold_stride] = live;
#pragma unroll
        for (int p = 1; p < 4; ++p) rec[(size_t)(8 * p + 7) * manifold_stride] = 0.f;
      }
    }
  }
  const float n0 = vB.x + xb[0], n1 = vB.y + xb[1], n2 = vB.z + xb[2];
  const float n3 = wB.x + xb[3], n4 = wB.y + xb[4], n5 = wB.z + xb[5];
  integrate_base(bf, n0, n1, n2, n3, n4, n5, h, s.pos, s.qw, s.qx, s.qy, s.qz, s.linvel, s.angvel);
  return (active || active_partner) ? OCT_CONTACT : OCT_NO_CONTACT;
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// A value every lane of the wavefront computed identically, moved to a scalar register
__device__ __forceinline__ float oct_uniform(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// Which instantiations address the state through the descriptor: those whose step is a loop body -- the multi-step kernels
// (-0.2 us per step: no spill left) and the ones that complete a SAME_STEP autoreset inside the launch (IN_PLACE: 80-112 B of
// scratch and 19-21 spilled registers -> none; 23.2 -> 22.2 us per env.step() of the public loop); the plain one-step
// kernels measured 0.1-0.2 us per launch SLOWER that way on the bench workload (profiles/r04_ab_state_addressing.txt)
#if defined(UPKIE_OCTET_BUFFERED_STATE)
constexpr bool octet_state_through_descriptor(int, bool) { return UPKIE_OCTET_BUFFERED_STATE != 0; }
#else
constexpr bool octet_state_through_descriptor(int mode, bool in_place) { return mode == MODE_PENDULUM_ROLLOUT || in_place; }
#endif

template <class T>
__device__ __forceinline__ T pick6(int j, const T (&a)[6]) {
  return j == 0 ? a[0] : (j == 1 ? a[1] : (j == 2 ? a[2] : (j == 3 ? a[3] : (j == 4 ? a[4] : a[5]))));
}

// One env.step() of B envs on 8 B lanes. Same contract as step_kernel /
// step_kernel_pair (the in-step spine observers are not restated here: launches
// with observers attached use the two-lane kernel).
// Joint stops of the Servos kernels solved in registers (512-entry register file, one wave per SIMD) or, like the other
// modes', over the LDS workspace (256 registers): a build-time choice while it is being measured
#if defined(UPKIE_SERVOS_LIMITS_IN_LDS)
constexpr bool kServosLimitsInRegisters = false;
#else
constexpr bool kServosLimitsInRegisters = true;
#endif
// BULLET_LIKE: contacts by the Bullet-like specification on the env's persistent contact manifold `manifold`
// [BL_MANIFOLD_WORDS][B] (upkie_sim_set_contact_manifold) -- the eight-lane variant of what the one-lane kernels run
// (oct_bullet_like_solve); since round 5 for every mode: a Servos env whose joint sits at its stop takes the default model's
// joint-stop solve for that substep (the one-lane kernels keep the limit row inside the same sweeps).
// Wavefronts per SIMD the register budget allows: two (256 registers; up to 16384 envs put two on every SIMD). An ISA probe
// may set one (tools/isa_probe.sh -DUPKIE_PROBE_OCTET_WAVES=1: spills then go to AGPRs instead of scratch memory).
#if !defined(UPKIE_PROBE_OCTET_WAVES)
#define UPKIE_PROBE_OCTET_WAVES 2
#endif
// Lanes per workgroup of the eight-lane kernels: one wavefront. (An A/B probe may set 128 or 256 -- wavefronts of one
// workgroup share a CU --: tools/archive/ab_octet_block.py, profiles/r05_ab_octet_block.txt: nothing to gain. Not for
// MODE_BASE_VELOCITY, whose balancer tile is per workgroup.)
#if !defined(UPKIE_OCTET_BLOCK)
#define UPKIE_OCTET_BLOCK 64
#endif
// UPKIE_STAMPS (a probe build, never the shipped one: tools/build_variant.py ab/x.so -DUPKIE_STAMPS; tools/stamps.py): the first
// lane of every wavefront reads the shader clock (s_memtime: 2.4 GHz on gfx950, profiles/r02_kernarg_latency.txt) at the
// marks below, keeps the readings in scalar registers and writes them behind the census words ([UPKIE_CENSUS_WORDS ..]:
// 16 words per wavefront) when it leaves -- where the time of a launch goes that is not substeps (profiles/r06_fixed_cost.txt).
#if defined(UPKIE_STAMPS)
#define UPKIE_STAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#define UPKIE_STAMPS_FLUSH()                                                                                                      \
  do {                                                                                                                            \
    UPKIE_STAMP(5); /* stores issued */                                                                                           \
    __builtin_amdgcn_s_waitcnt(0);                                                                                                \
    UPKIE_STAMP(6); /* stores acknowledged (a wave does not wait for this before it ends) */                                      \
    if (census && (threadIdx.x & 63) == 0) {                                                                                      \
      unsigned long long* out_ = reinterpret_cast<unsigned long long*>(census + UPKIE_CENSUS_WORDS) +                             \
                                 (size_t)8 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));                               \
      for (int i_ = 0; i_ < 7; ++i_) out_[i_] = stamp[i_];                                                                        \
      out_[7] = __builtin_amdgcn_s_memrealtime(); /* the 100 MHz wall clock at exit: orders the wavefronts of a launch */          \
    }                                                                                                                             \
  } while (0)
#else
#define UPKIE_STAMP(i) (void)0
#define UPKIE_STAMPS_FLUSH() (void)0
#endif
template <int MODE, bool RAND, bool DEFAULT_SCALARS = false, bool IN_PLACE = false, bool BULLET_LIKE = false>
__global__ __launch_bounds__(UPKIE_OCTET_BLOCK, MODE == MODE_SERVOS && kServosLimitsInRegisters ? 1 : UPKIE_PROBE_OCTET_WAVES) void step_kernel_octet(const DevModel* __restrict__ Mp, const DevParams* __restrict__ Pp,
                                                         int done_pass, int num_envs, float* __restrict__ state, const float* __restrict__ act,
                                                         float* __restrict__ obs, float* __restrict__ reward,
                                                         uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                                                         const uint8_t* __restrict__ mask, const float* __restrict__ body_inertials,
                                                         const float* __restrict__ ext_force, int packed, BaseVelocityPtrs bv,
                                                         float* __restrict__ final_obs, int n_steps, unsigned* __restrict__ census,
                                                         ServoPolicyArg<MODE> policy_arg, float* __restrict__ manifold) {
  // Which eight envs this wavefront steps. Workgroup b runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md) and
  // every XCD has its own L2: with envs handed out in launch order the four wavefronts that share a 128-byte line of a
  // state row (32 envs) sit on four different XCDs and each L2 fetches -- and writes back -- the whole line for its
  // 32 bytes. Swizzled, XCD x owns one contiguous eighth of the batch: a line lives in one L2. (A speed matter only:
  // any placement gives the same results.)
#if defined(UPKIE_STAMPS)
  unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  UPKIE_STAMP(0);  // kernel entry
  const int B = num_envs;  // (a kernel argument of its own: the state loads below wait for nothing but the argument segment)
  unsigned block = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) block = (block & 7u) * (gridDim.x >> 3) + (block >> 3);
  const int tid = block * blockDim.x + threadIdx.x;
  // row of 16 lanes = quads [env 2r left, env 2r+1 left, env 2r right, env 2r+1 right]
  const int l = tid & 3, quad = (tid >> 2) & 3, leg = quad >> 1;
  const int e = (tid >> 4) * 2 + (quad & 1);
  const bool in_batch = e < B;  // the eight lanes of an env leave together (below, once the whole wavefront has solved its MPC tile)
  const bool lead = l == 0 && leg == 0;  // the lane that writes per-env words
  const bool jointed = l != 0;
  const int k = jointed ? l - 1 : 0;
  const int joint = 3 * leg + k;  // the own joint's index in the state / action / observation layouts
  // State word w of env e: through 64-bit lane addresses (`st[w * B]`) or -- BUFFERED -- the state's buffer descriptor
  // (state_words.hpp)
  constexpr bool BUFFERED = octet_state_through_descriptor(MODE, IN_PLACE);
  const unsigned row_bytes = (unsigned)B * 4u;
  const __amdgpu_buffer_rsrc_t state_rsrc = state_descriptor(state, B);
  float* const st = state + (in_batch ? e : 0);
  const unsigned env_off = (unsigned)(in_batch ? e : 0) * 4u;
  const unsigned joint_off = env_off + (unsigned)joint * row_bytes;           // the own joint's row of a per-joint block
  const unsigned legref_off = env_off + (unsigned)(2 * leg + k) * row_bytes;  // the own low-pass target (hip and knee lanes)
  // word w + i of the env (i: a lane-dependent index into a block of words); off: the byte offset that goes with i
#define SWI(w, i, off) state_word<BUFFERED>(state_rsrc, st, (size_t)((w) + (i)) * B, (off), (unsigned)(w) * row_bytes)
#define SW(w) SWI(w, 0, env_off)

  // ---- load: issued first, in flight while the settings below arrive ------------------
  OctPhys s;
  s.pos = v3(SW(UPKIE_S_POS), SW(UPKIE_S_POS + 1), SW(UPKIE_S_POS + 2));
  s.qw = SW(UPKIE_S_QUAT); s.qx = SW(UPKIE_S_QUAT + 1); s.qy = SW(UPKIE_S_QUAT + 2); s.qz = SW(UPKIE_S_QUAT + 3);
  s.linvel = v3(SW(UPKIE_S_LINVEL), SW(UPKIE_S_LINVEL + 1), SW(UPKIE_S_LINVEL + 2));
  s.angvel = v3(SW(UPKIE_S_ANGVEL), SW(UPKIE_S_ANGVEL + 1), SW(UPKIE_S_ANGVEL + 2));
  s.q = jointed ? SWI(UPKIE_S_Q, joint, joint_off) : 0.f;
  s.qd = jointed ? SWI(UPKIE_S_QD, joint, joint_off) : 0.f;
  const bool legged = l == 1 || l == 2;  // hip and knee lanes carry their low-pass target
  float legref = legged ? SWI(UPKIE_S_LEGREF, 2 * leg + k, legref_off) : 0.f;
  constexpr bool YAWING = MODE == MODE_GYROPOD || MODE == MODE_BASE_VELOCITY;
  float yaw = 0.f, yawvel = 0.f;
  if (YAWING) {
    yaw = SW(UPKIE_S_YAW);
    yawvel = SW(UPKIE_S_YAWVEL);
  }
  float done_word = MODE != MODE_RESET ? SW(UPKIE_S_DONE) : 0.f;
  if constexpr (BULLET_LIKE) {  // the applied normal impulse of the own leg's cached contact point (at most one is live)
    const float* rec = manifold + (size_t)(leg * 4 * 8) * B + (in_batch ? e : 0);
    float applied = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) applied = fmaf(rec[(size_t)(8 * p + 6) * B], rec[(size_t)(8 * p + 7) * B], applied);
    s.bl_applied = applied;
  }
  // (and the lane's row of constants: eight more loads that wait for nothing but the model pointer)
  // (not in front of the balancer's tile, which wants the registers and has its own loads to wait for: measured, C3)
#if defined(UPKIE_AB_ROW_UP_FRONT)  // (A/B, round 6, with the balancer's tile on the fp16 path: C3 27.1 -> 27.7 us, still not)
  constexpr bool ROW_UP_FRONT = true;
#else
  constexpr bool ROW_UP_FRONT = MODE != MODE_BASE_VELOCITY;
#endif
  float lane_row[OT_WORDS];
  if (ROW_UP_FRONT) load_oct_table_row(*(const __attribute__((address_space(4))) DevModel*)Mp, l, leg, lane_row);
  // (they stay in front of the settings block: its touch sequence is a memory barrier to the compiler -- and nothing
  // here makes the wave WAIT for them before that sequence is issued)

  // the handle's limits and config: a block in device memory (L2 hits), read through scalar loads; every line touched up front
  typedef const __attribute__((address_space(4))) DevParams* ConstParamsPtr;
  warm_constant_block<sizeof(DevParams)>((ConstParamsPtr)Pp);
  const auto& Lm = ((ConstParamsPtr)Pp)->limits;
  const auto& C = ((ConstParamsPtr)Pp)->config;
  const int autoreset_mode = done_pass ? (int)AUTORESET_DONE_PASS : C.autoreset_mode;  // (the second launch of a SAME_STEP autoreset)
  typedef const __attribute__((address_space(4))) DevModel* ConstModelPtr;
  // what the substep loop reads of the config, fetched once, here, where control flow is still uniform (left to the
  // compiler the constant block is re-loaded -- and waited for -- in every iteration)
  int nb_substeps = C.nb_substeps, any_control_noise = C.any_control_noise;
  float kp_gain = C.kp, kd_gain = C.kd, substep_h = C.h;
  UPKIE_KEEP_IN_SGPR(nb_substeps);
  UPKIE_KEEP_IN_SGPR(any_control_noise);
  UPKIE_KEEP_IN_SGPR(kp_gain);
  UPKIE_KEEP_IN_SGPR(kd_gain);
  UPKIE_KEEP_IN_SGPR(substep_h);
  // ... and what a step reads of it OUTSIDE its substeps (action map, fall and time-limit tests), as one batch of scalar
  // loads: read where they are used they are a dozen load-and-wait pairs one after the other (86-160 cycles each for a
  // lone wavefront) -- once per launch in front of the first substep, but once per env.step() in the multi-step kernels
  constexpr bool WHEELED = MODE != MODE_SERVOS && MODE != MODE_RESET;  // the modes that map a ground velocity to the wheels
  float agent_gain0 = 0.f, agent_gain1 = 0.f, agent_gain2 = 0.f, agent_gain3 = 0.f, agent_clip = 0.f;
  float max_ground_velocity = 0.f, max_yaw_velocity = 0.f, leg_gain_scale = 0.f, max_gain_scale = 0.f, fall_pitch_limit = 0.f, step_dt = 0.f;
  int max_episode_steps = C.max_episode_steps;
  if (fused_agent(MODE)) {
    agent_gain0 = C.agent_gains[0]; agent_gain1 = C.agent_gains[1]; agent_gain2 = C.agent_gains[2]; agent_gain3 = C.agent_gains[3];
    agent_clip = C.agent_clip;
  }
  if (WHEELED) {
    max_ground_velocity = C.max_ground_velocity; max_yaw_velocity = C.max_yaw_velocity; leg_gain_scale = C.leg_gain_scale;
    fall_pitch_limit = C.fall_pitch; step_dt = C.dt;
  }
  if (MODE != MODE_RESET) max_gain_scale = C.max_gain_scale;
  if (fused_agent(MODE)) {
    UPKIE_KEEP_IN_SGPR(agent_gain0); UPKIE_KEEP_IN_SGPR(agent_gain1); UPKIE_KEEP_IN_SGPR(agent_gain2); UPKIE_KEEP_IN_SGPR(agent_gain3);
    UPKIE_KEEP_IN_SGPR(agent_clip);
  }
  if (WHEELED) {
    UPKIE_KEEP_IN_SGPR(max_ground_velocity); UPKIE_KEEP_IN_SGPR(max_yaw_velocity); UPKIE_KEEP_IN_SGPR(leg_gain_scale);
    UPKIE_KEEP_IN_SGPR(fall_pitch_limit); UPKIE_KEEP_IN_SGPR(step_dt);
  }
  if (MODE != MODE_RESET) UPKIE_KEEP_IN_SGPR(max_gain_scale);
  UPKIE_KEEP_IN_SGPR(max_episode_steps);
  // UpkieBaseVelocity with its MPC balancer in the same launch (upkie_sim_step_base_velocity_mpc): the wavefront first
  // solves the condensed QPs of its eight envs on the matrix cores -- columns 0-7 of one 16-column MFMA tile, all 64
  // lanes at work under the tile's own lane mapping (mpc_tile) -- and hands the commanded velocities to the lanes that
  // step those envs through LDS. (One launch instead of two: no second dispatch, no second prologue; the solve is a
  // 30-iteration dependent chain, so its half-empty tile costs the wavefront nothing.)
  __shared__ float mpc_velocity[MODE == MODE_BASE_VELOCITY ? 16 : 1];
  if (MODE == MODE_BASE_VELOCITY && bv.mpc_fused) {
    const float* done_row = autoreset_mode != 0 ? state + (size_t)UPKIE_S_DONE * B : nullptr;
#if defined(UPKIE_FUSED_MPC_FP32)  // (A/B build: the fp32 MFMA tile of rounds 4-5)
    mpc_tile<1, 8>(bv.mpc, bv.mpc_ws, bv.x0, act, 2, bv.contact, done_row, C.dt, bv.mpc_commanded, nullptr, 8 * (int)block, mpc_velocity);
#else  // round 6: the fp16 matrix path, whose MFMAs leave the vector unit to the wavefront this one shares its SIMD with
    mpc_tile_h<1, 8>(bv.mpc, bv.mpc_ws, bv.x0, act, 2, bv.contact, done_row, C.dt, bv.mpc_commanded, nullptr, 8 * (int)block, mpc_velocity);
#endif
    __syncthreads();
  }
  if (!in_batch) return;
  // joint stops of the kernels that do not solve them in registers: one workspace per env of the wavefront, in LDS
  // (octet_limit_path_scratch: 17.8 KB per wavefront, eight wavefronts per CU fit the 160 KB)
  __shared__ LimitWorkspace limit_workspaces[UPKIE_OCTET_BLOCK / 8];
  LimitWorkspace* const limit_ws = MODE == MODE_SERVOS && kServosLimitsInRegisters && !BULLET_LIKE ? nullptr : limit_workspaces;
  const float* records = RAND && body_inertials ? body_inertials + e : nullptr;
  const OctLane L = load_oct_lane(*(ConstModelPtr)Mp, Lm, C, l, leg, records, (size_t)B, ROW_UP_FRONT ? &lane_row : nullptr);
  const auto& M = *(ConstModelPtr)Mp;
  // external forces: those on the trunk enter the substep as one wrench (launches with a force on a leg link use the
  // two-lane kernel: launch_step)
  const ExtForces ext{RAND && ext_force ? ext_force + e : nullptr, (size_t)B, nullptr};  // (the slots are read from C.ext below)

  float act0 = 0.f, act1 = 0.f;
  float4 prev_obs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == MODE_PENDULUM) {
    if (act) act0 = act[e];
  } else if (fused_agent(MODE)) {
    const float* prev = act ? act : obs;
    prev_obs = reinterpret_cast<const float4*>(prev)[packed ? 2 * (size_t)e : (size_t)e];
  } else if (MODE == MODE_GYROPOD) {
    if (act) {
      const float2 a = reinterpret_cast<const float2*>(act)[e];
      act0 = a.x;
      act1 = a.y;
    }
  } else if (MODE == MODE_BASE_VELOCITY) {
    act0 = bv.mpc_fused ? mpc_velocity[e & 7] : bv.commanded[e];
    act1 = act[2 * (size_t)e + 1];
  }

  constexpr bool ROLLOUT = MODE == MODE_PENDULUM_ROLLOUT;
  int steps_left = ROLLOUT && packed && n_steps > 1 ? n_steps : 1;
  float* records_out = obs;
  float episode_word = ROLLOUT ? SW(UPKIE_S_EPISODE) : 0.f;
  float elapsed_word = ROLLOUT && max_episode_steps > 0 ? SW(UPKIE_S_ELAPSED) : 0.f;
  const bool any_noise = C.any_control_noise || C.any_measurement_noise;
  unsigned step_count = any_noise ? (unsigned)SW(UPKIE_S_STEP) : 0u;
  const float signed_radius = M.left_sign * M.wheel_radius;
  // (the wheel map's constants, upkie_gyropod.py:216-234, once per launch as well)
  float left_sign = M.left_sign;
  const float inv_radius_lanes = fast_rcp(M.wheel_radius);
  const float inv_radius = oct_uniform(inv_radius_lanes);  // (vector-unit results, the same in every lane: back to scalar registers)
  const float yaw_to_wheel = oct_uniform(M.left_sign * (0.5f * M.wheel_base) * inv_radius_lanes);
  if (WHEELED) UPKIE_KEEP_IN_SGPR(left_sign);

  // Gyropod observation (upkie_gyropod.py:186-214): the wheel lanes hold what it needs
  auto observe6 = [&](float yaw_, float yawvel_, float (&o6)[6]) {
    const float qo = oct_qb<3>(s.q), qdo = oct_qb<3>(s.qd);
    const float qp = oct_swp(qo), qdp = oct_swp(qdo);
    const float ql = leg ? qp : qo, qr = leg ? qo : qp;
    const float qdl = leg ? qdp : qdo, qdr = leg ? qdo : qdp;
    const float r01 = 2.f * (s.qx * s.qy - s.qz * s.qw), r11 = 1.f - 2.f * (s.qx * s.qx + s.qz * s.qz), r21 = 2.f * (s.qy * s.qz + s.qx * s.qw);
    const float x = fminf(fmaxf(2.f * (s.qw * s.qy - s.qz * s.qx), -1.f), 1.f);
    o6[0] = 0.5f * (ql - qr) * signed_radius;
    o6[1] = asinf(x);
    o6[2] = yaw_;
    o6[3] = 0.5f * (qdl - qdr) * signed_radius;
    o6[4] = r01 * s.angvel.x + r11 * s.angvel.y + r21 * s.angvel.z;
    o6[5] = yawvel_;
  };

  // UpkieServos with the servo-level policy evaluated here instead of by a launch of its own in front of this one
  // (upkie_sim_step_servos_policy: `packed` == 2, the policy is the kernel argument `policy_arg`): what
  // servo_policy_kernel computes from the state the step starts from, the own joint's row only.
  const UpkieServoPolicy* policy = nullptr;
  if constexpr (MODE == MODE_SERVOS) policy = packed == 2 ? &policy_arg : nullptr;
  float policy_pitch = 0.f, policy_position = 0.f, policy_velocity = 0.f;
  if (MODE == MODE_SERVOS && policy) {
    policy_pitch = asinf(fminf(fmaxf(2.f * (s.qw * s.qy - s.qz * s.qx), -1.f), 1.f));
    const float qo = oct_qb<3>(s.q), qdo = oct_qb<3>(s.qd);
    const float qp = oct_swp(qo), qdp = oct_swp(qdo);
    policy_position = 0.5f * ((leg ? qp : qo) - (leg ? qo : qp)) * signed_radius;
    policy_velocity = 0.5f * ((leg ? qdp : qdo) - (leg ? qdo : qdp)) * signed_radius;
    const float fall_pitch = policy->fall_pitch;
    if (fall_pitch > 0.f && fabsf(policy_pitch) > fall_pitch) {  // flagged for the NEXT_STEP autoreset: this very launch
      done_word = 1.f;
      if (lead) SW(UPKIE_S_DONE) = 1.f;
    }
  }
  // gymnasium's SAME_STEP autoreset inside the launch (upkie_sim_set_final_observation): an env that finishes keeps its
  // last observation in `final_obs` and goes through the reset branch once more before the kernel returns -- its eight
  // lanes together, the other envs of the wavefront wait masked. Its own instantiations (IN_PLACE, round 4): the jump back
  // to `next_step` makes the whole step a loop body, and with it in every kernel the NEXT_STEP / disabled launches -- what
  // `VecEnv.step(actions)` runs by default -- carried 19-21 spilled VGPRs and 80-112 B of scratch for a pass they never take.
  constexpr bool CAN_RESET_IN_PLACE = IN_PLACE && (MODE == MODE_PENDULUM || MODE == MODE_GYROPOD || MODE == MODE_SERVOS);
  const bool same_step = CAN_RESET_IN_PLACE && autoreset_mode == UPKIE_AUTORESET_DISABLED && final_obs != nullptr && packed != 1;
  bool second_pass = false;
  // Several steps in a launch (ROLLOUT): nothing reads the state words back before the launch ends, so the per-step ones
  // (DONE, ELAPSED, EPISODE, STEP, the applied torque, the words a reset zeroes) go to memory ONCE, behind the last step,
  // from the registers that mirror them -- the same memory image as one launch per step, and no store address is live
  // across the step loop (hoisted there, thirteen of them cost the kernel 24 spilled registers reloaded every step).
  bool reset_seen = false, step_seen = false;
  float tau_stepped = 0.f;
  // Every load of the prologue lands BEFORE the step loop. Left pending, the compiler waits for them where the loop first
  // reads their registers -- the same instructions in every iteration, and from the second step on those waits
  // (`s_waitcnt vmcnt(0)`: the counter is in order) are waits for the record stores the previous step has just issued
  // (measured: 0.12 us per env.step(), tools/ab_step.py --rollout).
  if (ROLLOUT) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
next_step:
  s.swept_prev = 0;  // the sweeps' warm start spans the substeps of ONE env.step(): several steps in a launch = as many launches, bit for bit
  bool do_reset;
  if (MODE == MODE_RESET) {
    do_reset = mask ? mask[e] != 0 : true;
  } else {
    do_reset = second_pass || ((autoreset_mode == UPKIE_AUTORESET_NEXT_STEP || autoreset_mode == AUTORESET_DONE_PASS) && done_word != 0.f);
    if (autoreset_mode == AUTORESET_DONE_PASS) {
      if (final_obs) {  // every env keeps its last observation (see step_kernel)
        constexpr int W = ObsWords<MODE>::value;
        if (MODE == MODE_SERVOS) {
          if (jointed) {
#pragma unroll
            for (int i = 0; i < 5; ++i) final_obs[(size_t)30 * e + 5 * joint + i] = obs[(size_t)30 * e + 5 * joint + i];
          }
        } else if (lead) {
          const float* last = obs + (size_t)(packed ? 8 : W) * e;
#pragma unroll
          for (int i = 0; i < W; ++i) final_obs[(size_t)W * e + i] = last[i];
        }
      }
      if (!do_reset) return;
    }
  }

  if (MODE == MODE_RESET && !do_reset) {
    if (obs) {
      float o6[6];
      observe6(SW(UPKIE_S_YAW), SW(UPKIE_S_YAWVEL), o6);
      if (lead) {
#pragma unroll
        for (int i = 0; i < 6; ++i) obs[(size_t)6 * e + i] = o6[i];
      }
    }
    return;
  }

  // ---- action map: the own joint's servo target -----------------------------
  Servo cmd{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float a0 = 0.f, a1 = 0.f;
  unsigned episode = 0;
  if (do_reset) {
    episode = ROLLOUT ? (unsigned)episode_word : (unsigned)SW(UPKIE_S_EPISODE);
    Phys full;
    sample_init_state(C, (unsigned)e, episode, full);  // same draws in the eight lanes
    s.pos = full.pos; s.qw = full.qw; s.qx = full.qx; s.qy = full.qy; s.qz = full.qz;
    s.linvel = full.linvel; s.angvel = full.angvel;
    s.q = jointed ? pick6(joint, full.q) : 0.f;
    s.qd = 0.f;
    s.bl_applied = 0.f;  // (a reset drops the contact cache)
  } else if (MODE == MODE_SERVOS) {
    if (jointed) {
      float a[6];
      if (policy) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] = policy->action[joint][i];
        float fb = policy->pitch_to_velocity[joint] * policy_pitch + policy->position_to_velocity[joint] * policy_position +
                   policy->velocity_to_velocity[joint] * policy_velocity;
        const float clip = policy->velocity_feedback_clip[joint];
        if (clip > 0.f) fb = fminf(fmaxf(fb, -clip), clip);
        a[1] += fb;
        a[2] += policy->pitch_to_torque[joint] * policy_pitch;
      } else {
        const float* given = act + (size_t)36 * e + 6 * joint;
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] = given[i];
      }
      const float eff = L.effort, vel = L.velocity;
      cmd.position = clamp_ref(a[0], L.lower, L.upper);
      cmd.velocity = clamp_ref(a[1], -vel, vel);
      cmd.feedforward_torque = clamp_ref(a[2], -eff, eff);
      cmd.kp_scale = clamp_ref(a[3], 0.f, max_gain_scale);
      cmd.kd_scale = clamp_ref(a[4], 0.f, max_gain_scale);
      cmd.maximum_torque = clamp_ref(a[5], 0.f, eff);
#if !defined(UPKIE_AB_NO_GUARD)
      if (const int replaced = guard_servo_command(cmd, eff)) guard_count(C.guard, 0, replaced);  // non-finite guard (step_kernels.hpp)
#endif
    }
  } else if (MODE != MODE_RESET) {
    if (fused_agent(MODE)) {
      const float4 o = prev_obs;
      a0 = agent_gain0 * o.x + agent_gain1 * o.y + agent_gain2 * o.z + agent_gain3 * o.w;
      a0 = clamp_ref(a0, -agent_clip, agent_clip);
    } else {
      a0 = act0;
      a1 = act1;
    }
    {
      const int replaced = guard_velocity_actions(a0, a1, max_yaw_velocity);
      if (lead && replaced) guard_count(C.guard, 0, replaced);
    }
    const float v = clamp_ref(a0, -max_ground_velocity, max_ground_velocity);
    const float yawd = clamp_ref(a1, -max_yaw_velocity, max_yaw_velocity);
    const float wheel_velocity = v * inv_radius;
    float left = left_sign * wheel_velocity, right = -left_sign * wheel_velocity;
    left = fmaf(yaw_to_wheel, yawd, left);
    right = fmaf(yaw_to_wheel, yawd, right);
    const float alpha = step_dt / 1.0f;
    legref = legref + alpha * (0.f - legref);
    if (legged) {
      cmd.position = clamp_ref(legref, L.lower, L.upper);
      cmd.kp_scale = clamp_ref(leg_gain_scale, 0.f, max_gain_scale);
      cmd.kd_scale = cmd.kp_scale;
      cmd.maximum_torque = L.effort;
    } else if (l == 3) {
      cmd.position = NAN;
      cmd.velocity = clamp_ref(leg ? right : left, -L.velocity, L.velocity);
      cmd.kp_scale = 1.f;
      cmd.kd_scale = 1.f;
      cmd.maximum_torque = L.effort;
    }
  }

#if defined(UPKIE_STAMPS)
  __builtin_amdgcn_s_waitcnt(0);  // every load of the prologue has landed: what follows is arithmetic
#endif
  UPKIE_STAMP(1);  // prologue done: settings, lane constants, state, action map
  // ---- substeps ------------------------------------------------------------
  float tau = 0.f;
  bool contact = false;
  const int nsub = do_reset ? 1 : nb_substeps;
  for (int sub = 0; sub < nb_substeps; ++sub) {
    if (sub >= nsub) break;
    float noise = 0.f;
    if (any_control_noise && !do_reset) {
      float zn[6];
      philox_normal6(C, (unsigned)e, step_count, (unsigned)sub, zn);
      noise = L.control_noise * pick6(joint, zn);
    }
    tau = jointed ? joint_torque(s.q, s.qd, cmd, kp_gain, kd_gain, L.friction, noise) : 0.f;
    ConstModelPtr mp = (ConstModelPtr)Mp;
    asm volatile("" : "+s"(mp));
    const bool forces = RAND && ext.force && !do_reset;  // the reset substep runs without external forces
    float wrench[6];
    if (forces) {  // forces on the trunk: one wrench about the base origin, base frame
      const BaseFrame bf = base_frame(s.qw, s.qx, s.qy, s.qz, s.linvel, s.angvel);
      const float r00 = bf.r00, r01 = bf.r01, r02 = bf.r02, r10 = bf.r10, r11 = bf.r11, r12 = bf.r12, r20 = bf.r20, r21 = bf.r21, r22 = bf.r22;
      V3 Fs = v3(0.f, 0.f, 0.f), Ns = v3(0.f, 0.f, 0.f);
      for (int i = 0; i < C.ext.count; ++i) {
        const V3 f = v3(ext.force[(size_t)(3 * i) * ext.stride], ext.force[(size_t)(3 * i + 1) * ext.stride], ext.force[(size_t)(3 * i + 2) * ext.stride]);
        const V3 pt = v3(C.ext.point[i][0], C.ext.point[i][1], C.ext.point[i][2]);
        const V3 Fe = C.ext.local[i] != 0 ? f : v3(r00 * f.x + r10 * f.y + r20 * f.z, r01 * f.x + r11 * f.y + r21 * f.z, r02 * f.x + r12 * f.y + r22 * f.z);
        Fs = Fs + Fe;
        Ns = Ns + cross(pt, Fe);
      }
      wrench[0] = Fs.x; wrench[1] = Fs.y; wrench[2] = Fs.z; wrench[3] = Ns.x; wrench[4] = Ns.y; wrench[5] = Ns.z;
    }
    OctRare rare_path{0, 0};
    // (always handed over: a pointer that is null without a census put the two words in scratch memory, stored every substep)
    const int status = physics_substep_octet<MODE == MODE_SERVOS && kServosLimitsInRegisters && !BULLET_LIKE, DEFAULT_SCALARS, BULLET_LIKE>(
        *mp, Lm, L, s, tau, substep_h, forces ? wrench : nullptr, limit_ws, &rare_path, BULLET_LIKE && sub == nsub - 1 ? manifold + e : nullptr, (size_t)B);
    const int rare = rare_path.path;
    if (census) {  // rare-path census (upkie_sim_set_census): ONE atomic per wavefront, substep and path (per-env atomics on two
                   // addresses serialise: 14 k of them per launch cost 140 us when 70 % of the substeps sweep)
      const unsigned long long limited = __builtin_amdgcn_ballot_w64(lead && rare == OCT_NOT_MINE_LIMIT);
      const unsigned long long swept = __builtin_amdgcn_ballot_w64(lead && rare == OCT_NOT_MINE_INFEASIBLE);
      const bool first = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
      if (first && limited) {
        atomicAdd(&census[0], (unsigned)__builtin_popcountll(limited));
        atomicAdd(&census[4], 1u);
      }
      if (first && swept) atomicAdd(&census[2], (unsigned)__builtin_popcountll(swept));
      if (swept) {  // the sweeps those envs ran, summed over the wavefront first (at most eight envs: a scalar loop over their lead lanes)
        unsigned total = 0, most = 0, capped = 0, direct = 0;
        for (unsigned long long left = swept; left; left &= left - 1) {
          const int code = __builtin_amdgcn_readlane(rare_path.sweeps, __builtin_ctzll(left));
          const unsigned count = code > 0 ? (unsigned)code : 0u;  // (negative: solved by an active set, no sweep)
          direct += code < 0 ? 1u : 0u;
          total += count;
          most = count > most ? count : most;
          capped += count >= (unsigned)M.pgs_iterations ? 1u : 0u;
        }
        if (first) {
          atomicAdd(&census[6], total);
          if (direct) atomicAdd(&census[3], direct);
          atomicMax(&census[7], most);
          if (capped) atomicAdd(&census[1], capped);
          if (total) {  // (a wavefront whose envs were all answered by an active set ran no sweep: not in [5], not in the histogram -- ADVICE r5)
            atomicAdd(&census[5], 1u);
            atomicAdd(&census[8 + (most < 63u ? most : 63u)], 1u);  // what the wavefront waited for in this substep
          }
        }
      }
    }
    contact = status == OCT_CONTACT;
#if defined(UPKIE_STAMPS)
    if (sub == 0) UPKIE_STAMP(2);  // first substep done (it carries the cold instruction cache)
#endif
  }
  UPKIE_STAMP(3);  // substeps done

  // ---- non-finite guard: the state behind the substeps (step_kernels.hpp) -------
  bool unsound;
#if defined(UPKIE_AB_NO_GUARD)
  unsound = false;
  if (false)
#endif
  {
    float mag = oct_esum(fabsf(s.q) + fabsf(s.qd));  // (the trunk lanes hold zeros)
    mag += fabsf(s.pos.x) + fabsf(s.pos.y) + fabsf(s.pos.z) + fabsf(s.qw) + fabsf(s.qx) + fabsf(s.qy) + fabsf(s.qz);
    mag += fabsf(s.linvel.x) + fabsf(s.linvel.y) + fabsf(s.linvel.z) + fabsf(s.angvel.x) + fabsf(s.angvel.y) + fabsf(s.angvel.z);
    if (YAWING) mag += fabsf(yaw);
    unsound = !(mag < 3.0e38f);
  }
  if (unsound) {  // the eight lanes of the env together (the sum is the same in all of them)
    s.pos = v3(C.init_pos[0], C.init_pos[1], C.init_pos[2]);
    s.qw = C.init_quat[0]; s.qx = C.init_quat[1]; s.qy = C.init_quat[2]; s.qz = C.init_quat[3];
    s.linvel = v3(C.init_linvel[0], C.init_linvel[1], C.init_linvel[2]);
    s.angvel = v3(C.init_angvel[0], C.init_angvel[1], C.init_angvel[2]);
    float init_q[UPKIE_NJ];
#pragma unroll
    for (int j = 0; j < UPKIE_NJ; ++j) init_q[j] = C.init_joint[j];
    s.q = jointed ? pick6(joint, init_q) : 0.f;
    s.qd = 0.f;
    s.bl_applied = 0.f;
    tau = 0.f;
    legref = s.q;
    yaw = 0.f;
    a1 = 0.f;
    contact = false;
    if constexpr (BULLET_LIKE) {  // a contact cache of that state means nothing: the own leg's cached point is dropped
      if (l == 1) {  // (the lane that wrote the record in the last substep)
        manifold[(size_t)(leg * 4 * 8 + 6) * B + e] = 0.f;
        manifold[(size_t)(leg * 4 * 8 + 7) * B + e] = 0.f;
      }
    }
    if (lead) guard_count(C.guard, 1, 1);
  }

  // ---- wrapper post-processing ---------------------------------------------
  bool fallen = false, timeout = false;
  float obs6[6];
  if (ROLLOUT) {
    reset_seen = reset_seen || do_reset;
    if (!do_reset) {
      step_seen = true;
      tau_stepped = tau;
    }
  }
  if (do_reset) {
    legref = s.q;
    yaw = 0.f;
    yawvel = 0.f;
    if (lead && !ROLLOUT) {
      SW(UPKIE_S_YAW) = 0.f;
      SW(UPKIE_S_YAWVEL) = 0.f;
      SW(UPKIE_S_MPC_V) = 0.f;
      SW(UPKIE_S_SE2_X) = 0.f;
      SW(UPKIE_S_SE2_Y) = 0.f;
      SW(UPKIE_S_EPISODE) = (float)((episode + 1u) & UPKIE_COUNTER_MASK);
      SW(UPKIE_S_DONE) = 0.f;
      SW(UPKIE_S_ELAPSED) = 0.f;
    }
    episode_word = (float)((episode + 1u) & UPKIE_COUNTER_MASK);
    done_word = 0.f;
    elapsed_word = 0.f;
    observe6(yaw, yawvel, obs6);
  } else {
    if (YAWING) {
      yaw = fmaf(a1, step_dt, yaw);
      yawvel = a1;
      if (lead) {
        SW(UPKIE_S_YAW) = yaw;
        SW(UPKIE_S_YAWVEL) = yawvel;
      }
    }
    observe6(yaw, yawvel, obs6);
    fallen = unsound;  // (every env kind: the non-finite guard ends the episode)
    if (MODE != MODE_SERVOS) fallen = fallen || fabsf(obs6[1]) > fall_pitch_limit;
    if (fallen && lead && !ROLLOUT) SW(UPKIE_S_DONE) = 1.f;
    if (max_episode_steps > 0) {
      const float elapsed = (ROLLOUT ? elapsed_word : SW(UPKIE_S_ELAPSED)) + 1.f;
      timeout = elapsed >= (float)max_episode_steps && !fallen;
      elapsed_word = elapsed;
      if (lead && !ROLLOUT) {
        SW(UPKIE_S_ELAPSED) = elapsed;
        if (timeout) SW(UPKIE_S_DONE) = 1.f;
      }
    }
    if (fallen || timeout) done_word = 1.f;
    if (jointed && !ROLLOUT) SWI(UPKIE_S_TORQUE, joint, joint_off) = tau;
    if (any_noise) {
      step_count = (step_count + 1u) & UPKIE_COUNTER_MASK;
      if (lead && !ROLLOUT) SW(UPKIE_S_STEP) = (float)step_count;
    }
  }

  UPKIE_STAMP(4);  // guard, observation, flags computed
  // ---- store (last step of the launch) --------------------------------------
  if (steps_left > 1) {
    const float4 o4 = make_float4(obs6[1], obs6[0], obs6[4], obs6[3]);
    if (lead) {
      float4* rec = reinterpret_cast<float4*>(records_out) + 2 * (size_t)e;
      rec[0] = o4;
      rec[1] = make_float4(0.f, fallen ? 1.f : 0.f, timeout ? 1.f : 0.f, 0.f);
    }
    prev_obs = o4;
    records_out += (size_t)8 * B;
    steps_left -= 1;
    goto next_step;
  }
  if (ROLLOUT) {  // the per-step words of the steps of this launch (above)
    if (lead) {
      if (reset_seen) {
        SW(UPKIE_S_YAW) = 0.f;
        SW(UPKIE_S_YAWVEL) = 0.f;
        SW(UPKIE_S_MPC_V) = 0.f;
        SW(UPKIE_S_SE2_X) = 0.f;
        SW(UPKIE_S_SE2_Y) = 0.f;
        SW(UPKIE_S_EPISODE) = episode_word;
      }
      SW(UPKIE_S_DONE) = done_word;
      if (reset_seen || max_episode_steps > 0) SW(UPKIE_S_ELAPSED) = elapsed_word;
      if (any_noise && step_seen) SW(UPKIE_S_STEP) = (float)step_count;
    }
    if (jointed && step_seen) SWI(UPKIE_S_TORQUE, joint, joint_off) = tau_stepped;
  }
  if (lead) {
    SW(UPKIE_S_POS) = s.pos.x; SW(UPKIE_S_POS + 1) = s.pos.y; SW(UPKIE_S_POS + 2) = s.pos.z;
    SW(UPKIE_S_QUAT) = s.qw; SW(UPKIE_S_QUAT + 1) = s.qx; SW(UPKIE_S_QUAT + 2) = s.qy; SW(UPKIE_S_QUAT + 3) = s.qz;
    SW(UPKIE_S_LINVEL) = s.linvel.x; SW(UPKIE_S_LINVEL + 1) = s.linvel.y; SW(UPKIE_S_LINVEL + 2) = s.linvel.z;
    SW(UPKIE_S_ANGVEL) = s.angvel.x; SW(UPKIE_S_ANGVEL + 1) = s.angvel.y; SW(UPKIE_S_ANGVEL + 2) = s.angvel.z;
    SW(UPKIE_S_CONTACT) = contact ? 1.f : 0.f;
  }
  if (jointed) {
    SWI(UPKIE_S_Q, joint, joint_off) = s.q;
    SWI(UPKIE_S_QD, joint, joint_off) = s.qd;
  }
  if (MODE != MODE_SERVOS && legged) SWI(UPKIE_S_LEGREF, 2 * leg + k, legref_off) = legref;

  if (MODE == MODE_RESET) {
    if (obs && lead) {
#pragma unroll
      for (int i = 0; i < 6; ++i) obs[(size_t)6 * e + i] = obs6[i];
    }
    return;
  }
  const bool keep_last = same_step && !second_pass;  // this pass's observation is also the env's `final_obs` row
  if (MODE == MODE_SERVOS) {
    // each joint lane reports its servo (upkie_servos.py:288-306)
    float zm = 0.f;
    if (C.any_measurement_noise) {
      float z6[6];
      philox_normal6(C, (unsigned)e, step_count, NOISE_SLOT_MEASUREMENT, z6);
      zm = pick6(joint, z6);
    }
    if (jointed) {
      const float o2 = (do_reset ? SWI(UPKIE_S_TORQUE, joint, joint_off) : tau) + L.measurement_noise * zm;
      float* o = obs + (size_t)30 * e + 5 * joint;
      o[0] = s.q;
      o[1] = s.qd;
      o[2] = o2;
      o[3] = 42.0f;
      o[4] = 18.0f;
      if (keep_last) {
        float* f = final_obs + (size_t)30 * e + 5 * joint;
        f[0] = s.q;
        f[1] = s.qd;
        f[2] = o2;
        f[3] = 42.0f;
        f[4] = 18.0f;
      }
    }
  }
  if (lead) {
    if (MODE == MODE_PENDULUM || fused_agent(MODE)) {
      const float4 o4 = make_float4(obs6[1], obs6[0], obs6[4], obs6[3]);
      if (packed) {
        float4* rec = reinterpret_cast<float4*>(records_out) + 2 * (size_t)e;
        rec[0] = o4;
        if (autoreset_mode != AUTORESET_DONE_PASS) rec[1] = make_float4(0.f, fallen ? 1.f : 0.f, timeout ? 1.f : 0.f, 0.f);
#if defined(UPKIE_STAMPS)
        goto stamps_flush;
#else
        return;
#endif
      }
      reinterpret_cast<float4*>(obs)[e] = o4;
      if (keep_last) reinterpret_cast<float4*>(final_obs)[e] = o4;
    } else if (MODE == MODE_BASE_VELOCITY) {
      float x = 0.f, y = 0.f;
      if (!do_reset) {
        float lin = act[2 * (size_t)e];
        if (!is_finite(lin)) lin = 0.f;  // (non-finite guard)
        float sy, cy;
        sincosf(yaw, &sy, &cy);
        x = fmaf(lin * cy, step_dt, SW(UPKIE_S_SE2_X));
        y = fmaf(lin * sy, step_dt, SW(UPKIE_S_SE2_Y));
        SW(UPKIE_S_SE2_X) = x;
        SW(UPKIE_S_SE2_Y) = y;
      }
      obs[(size_t)3 * e] = x;
      obs[(size_t)3 * e + 1] = y;
      obs[(size_t)3 * e + 2] = yaw;
      reinterpret_cast<float4*>(bv.x0)[e] = make_float4(obs6[0], obs6[1], obs6[3], obs6[4]);
      bv.contact[e] = contact ? 1 : 0;
    } else if (MODE == MODE_GYROPOD) {
      float2* o2 = reinterpret_cast<float2*>(obs) + (size_t)3 * e;
      o2[0] = make_float2(obs6[0], obs6[1]);
      o2[1] = make_float2(obs6[2], obs6[3]);
      o2[2] = make_float2(obs6[4], obs6[5]);
      if (keep_last) {
        float2* f2 = reinterpret_cast<float2*>(final_obs) + (size_t)3 * e;
        f2[0] = make_float2(obs6[0], obs6[1]);
        f2[1] = make_float2(obs6[2], obs6[3]);
        f2[2] = make_float2(obs6[4], obs6[5]);
      }
    }
    if (autoreset_mode != AUTORESET_DONE_PASS && !second_pass) {  // (the flags are those of the step, not of the reset behind it)
      reward[e] = 0.f;
      terminated[e] = fallen ? 1 : 0;
      truncated[e] = timeout ? 1 : 0;
    }
  }
  if (keep_last && done_word != 0.f) {
    second_pass = true;
    goto next_step;
  }
#if defined(UPKIE_STAMPS)
stamps_flush:
#endif
  UPKIE_STAMPS_FLUSH();
#undef SW
#undef SWI
}
#endif

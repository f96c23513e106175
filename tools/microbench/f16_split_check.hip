// f16_split_check.hip -- hardware truths behind mpc_tile_h (round 6): (1) the fp16 hi / lo split of an fp32 value
// (v_cvt_pkrtz_f16_f32 + v_fma_mix_f32): hi + lo against the value; (2) the operand layout of v_mfma_f32_16x16x32_f16 as the
// kernel assumes it: A[i][k] in lane 16 (k / 8) + i, half k % 8; B[k][n] in lane 16 (k / 8) + n, half k % 8; D[4 g + r][n] in
// lane 16 g + n, register r; (3) a 16 x 32 x 16 product of fp32 data through the three-term split against fp64.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/f16_split_check.hip -o /tmp/f16_split_check && /tmp/f16_split_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair(float x, float y, int& hi, int& lo) {
  const auto h = __builtin_amdgcn_cvt_pkrtz(x, y);
  hi = __builtin_bit_cast(int, h);
  float lx, ly;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lx) : "v"(hi), "v"(x));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ly) : "v"(hi), "v"(y));
  lo = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(lx, ly));
}

__global__ void k_split(const float* in, int* hi, int* lo, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 < n) split_pair(in[2 * i], in[2 * i + 1], hi[i], lo[i]);
}

// A [16][32], B [32][16] fp32 in memory; D [16][16]; terms: 1 = hi x hi only, 3 = the kernel's three
__global__ __launch_bounds__(64) void k_product(const float* A, const float* B, float* D, int terms) {
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  i4 ah, al, bh, bl;
  for (int p = 0; p < 4; ++p) {
    int h, l;
    split_pair(A[i * 32 + 8 * g + 2 * p], A[i * 32 + 8 * g + 2 * p + 1], h, l);
    ah[p] = h;
    al[p] = l;
    split_pair(B[(8 * g + 2 * p) * 16 + i], B[(8 * g + 2 * p + 1) * 16 + i], h, l);
    bh[p] = h;
    bl[p] = l;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bh));
  if (terms == 3) {
    asm volatile("s_nop 15\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah), "v"(bl));
    asm volatile("s_nop 15\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(al), "v"(bh));
  }
  asm volatile("s_nop 15" : "+v"(acc));
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}


// (4) wait states between DEPENDENT MFMAs (destination = C of the next one) and between a vector instruction that writes a
// B register and the MFMA that reads it: inline asm is opaque to hipcc's hazard recogniser, the kernel has to space them itself.
// NOPS < 0: -NOPS independent MFMAs (another accumulator) between the dependent ones instead of s_nop.
template <int NOPS>
__global__ __launch_bounds__(64) void k_dependent(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  i4 ah, al, bh, bl;
  for (int p = 0; p < 4; ++p) {
    int h, l;
    split_pair(A[i * 32 + 8 * g + 2 * p], A[i * 32 + 8 * g + 2 * p + 1], h, l);
    ah[p] = h;
    al[p] = l;
    split_pair(B[(8 * g + 2 * p) * 16 + i], B[(8 * g + 2 * p + 1) * 16 + i], h, l);
    bh[p] = h;
    bl[p] = l;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f}, other = {0.f, 0.f, 0.f, 0.f};
  asm volatile("s_nop 7" : "+v"(ah), "+v"(al), "+v"(bh), "+v"(bl), "+v"(acc), "+v"(other));
  if (NOPS >= 0) {
    asm volatile(
        "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop %5\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\ts_nop %5\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %4, %2, %0\n\ts_nop 15"
        : "+v"(acc)
        : "v"(ah), "v"(bh), "v"(bl), "v"(al), "n"(NOPS > 0 ? NOPS - 1 : 0));
  } else {
    f4 o = other;
#define FILL "v_mfma_f32_16x16x32_f16 %5, %1, %2, %5\n\t"
    if (NOPS == -1)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t" FILL "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t" FILL
                   "v_mfma_f32_16x16x32_f16 %0, %4, %2, %0\n\ts_nop 15"
                   : "+v"(acc) : "v"(ah), "v"(bh), "v"(bl), "v"(al), "v"(o));
    else if (NOPS == -2)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t" FILL FILL "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t" FILL FILL
                   "v_mfma_f32_16x16x32_f16 %0, %4, %2, %0\n\ts_nop 15"
                   : "+v"(acc) : "v"(ah), "v"(bh), "v"(bl), "v"(al), "v"(o));
    else
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t" FILL FILL FILL "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t" FILL FILL FILL
                   "v_mfma_f32_16x16x32_f16 %0, %4, %2, %0\n\ts_nop 15"
                   : "+v"(acc) : "v"(ah), "v"(bh), "v"(bl), "v"(al), "v"(o));
#undef FILL
  }
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}

// the last B register written by a vector instruction NOPS wait states in front of the MFMA (NOPS = 0: right in front); fixed
// registers, since inline asm cannot name one register of a four-register operand
template <int NOPS>
__global__ __launch_bounds__(64) void k_valu_to_mfma(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  i4 ah, bh;
  for (int p = 0; p < 4; ++p) {
    ah[p] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(A[i * 32 + 8 * g + 2 * p], A[i * 32 + 8 * g + 2 * p + 1]));
    bh[p] = p < 3 ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(B[(8 * g + 2 * p) * 16 + i], B[(8 * g + 2 * p + 1) * 16 + i])) : 0x7e007e00;  // (NaNs until written)
  }
  const float x = B[(8 * g + 6) * 16 + i], y = B[(8 * g + 7) * 16 + i];
  f4 acc;
#define V2M_HEAD                                                                                                               \
  "v_mov_b32 v32, %1\n\tv_mov_b32 v33, %2\n\tv_mov_b32 v34, %3\n\tv_mov_b32 v35, %4\n\t"                                       \
  "v_mov_b32 v36, %5\n\tv_mov_b32 v37, %6\n\tv_mov_b32 v38, %7\n\tv_mov_b32 v39, %8\n\ts_nop 7\n\t"                            \
  "v_cvt_pkrtz_f16_f32 v39, %9, %10\n\t"
#define V2M_TAIL                                                                                                               \
  "v_mfma_f32_16x16x32_f16 v[40:43], v[32:35], v[36:39], 0\n\ts_nop 15\n\t"                                                     \
  "v_mov_b32 %0, v40\n\tv_mov_b32 %11, v41\n\tv_mov_b32 %12, v42\n\tv_mov_b32 %13, v43"
#define V2M_OPS                                                                                                                \
  : "=&v"(acc[0]), "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]),     \
    "+v"(*const_cast<float*>(&x)), "+v"(*const_cast<float*>(&y)), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])                  \
  :                                                                                                                            \
  : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43"
  if (NOPS == 0) asm volatile(V2M_HEAD V2M_TAIL V2M_OPS);
  if (NOPS == 1) asm volatile(V2M_HEAD "s_nop 0\n\t" V2M_TAIL V2M_OPS);
  if (NOPS == 2) asm volatile(V2M_HEAD "s_nop 1\n\t" V2M_TAIL V2M_OPS);
  if (NOPS == 3) asm volatile(V2M_HEAD "s_nop 2\n\t" V2M_TAIL V2M_OPS);
  if (NOPS == 4) asm volatile(V2M_HEAD "s_nop 3\n\t" V2M_TAIL V2M_OPS);
  if (NOPS == 6) asm volatile(V2M_HEAD "s_nop 5\n\t" V2M_TAIL V2M_OPS);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}

// (4c) the MFMA's result read by a vector instruction NOPS wait states behind it (fixed registers as above)
template <int NOPS>
__global__ __launch_bounds__(64) void k_mfma_to_valu(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  i4 ah, bh;
  for (int p = 0; p < 4; ++p) {
    ah[p] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(A[i * 32 + 8 * g + 2 * p], A[i * 32 + 8 * g + 2 * p + 1]));
    bh[p] = __builtin_bit_cast(int, __builtin_amdgcn_cvt_pkrtz(B[(8 * g + 2 * p) * 16 + i], B[(8 * g + 2 * p + 1) * 16 + i]));
  }
  f4 acc;
#define M2V_HEAD                                                                                                               \
  "v_mov_b32 v32, %1\n\tv_mov_b32 v33, %2\n\tv_mov_b32 v34, %3\n\tv_mov_b32 v35, %4\n\t"                                       \
  "v_mov_b32 v36, %5\n\tv_mov_b32 v37, %6\n\tv_mov_b32 v38, %7\n\tv_mov_b32 v39, %8\n\t"                                       \
  "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\ts_nop 7\n\t"                                \
  "v_mfma_f32_16x16x32_f16 v[40:43], v[32:35], v[36:39], v[40:43]\n\t"
#define M2V_TAIL "v_mov_b32 %0, v40\n\tv_mov_b32 %9, v41\n\tv_mov_b32 %10, v42\n\tv_mov_b32 %11, v43\n\ts_nop 15"
#define M2V_OPS                                                                                                                \
  : "=&v"(acc[0]), "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]),     \
    "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])                                                                                \
  :                                                                                                                            \
  : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43"
  if (NOPS == 0) asm volatile(M2V_HEAD M2V_TAIL M2V_OPS);
  if (NOPS == 1) asm volatile(M2V_HEAD "s_nop 0\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 2) asm volatile(M2V_HEAD "s_nop 1\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 3) asm volatile(M2V_HEAD "s_nop 2\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 4) asm volatile(M2V_HEAD "s_nop 3\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 5) asm volatile(M2V_HEAD "s_nop 4\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 6) asm volatile(M2V_HEAD "s_nop 5\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 7) asm volatile(M2V_HEAD "s_nop 6\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 8) asm volatile(M2V_HEAD "s_nop 7\n\t" M2V_TAIL M2V_OPS);
  if (NOPS == 12) asm volatile(M2V_HEAD "s_nop 11\n\t" M2V_TAIL M2V_OPS);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}

static float half_bits(unsigned short u) {
  _Float16 h;
  __builtin_memcpy(&h, &u, 2);
  return (float)h;
}

int main() {
  const int n = 4096;
  std::vector<float> in(n);
  srand(1);
  for (int i = 0; i < n; ++i) in[i] = (float)((rand() / (double)RAND_MAX - 0.5) * pow(10.0, (rand() % 9) - 4));
  float* d_in;
  int *d_hi, *d_lo;
  hipMalloc(&d_in, n * 4);
  hipMalloc(&d_hi, n * 2);
  hipMalloc(&d_lo, n * 2);
  hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_split, dim3(n / 2 / 64), dim3(64), 0, 0, d_in, d_hi, d_lo, n);
  std::vector<int> hi(n / 2), lo(n / 2);
  hipMemcpy(hi.data(), d_hi, n * 2, hipMemcpyDeviceToHost);
  hipMemcpy(lo.data(), d_lo, n * 2, hipMemcpyDeviceToHost);
  double worst = 0, worst_abs = 0, at = 0, at_hi = 0, at_lo = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned short h = (unsigned short)((unsigned)hi[i / 2] >> (16 * (i % 2))), l = (unsigned short)((unsigned)lo[i / 2] >> (16 * (i % 2)));
    const double e = fabs((double)half_bits(h) + (double)half_bits(l) - in[i]) / (fabs(in[i]) + 1e-30);
    if (e > worst) worst = e, at = in[i], at_hi = half_bits(h), at_lo = half_bits(l);
    if (fabs(in[i]) > 1e-3) worst_abs = fmax(worst_abs, e);
    if (i < 4) printf("  %g = %g + %g\n", in[i], half_bits(h), half_bits(l));
  }
  printf("(1) split: worst |hi + lo - x| / |x| over %d values of 1e-4 .. 1e4: %.3g at %g = %g + %g; over the values above 1e-3: %.3g (2^-22 = 2.4e-7)\n", n, worst, at, at_hi, at_lo, worst_abs);

  std::vector<float> A(16 * 32), B(32 * 16), D(256);
  for (auto& v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * 100.0);
  for (auto& v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 10.0);
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4);
  hipMalloc(&dB, B.size() * 4);
  hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  for (int terms : {1, 3}) {
    hipLaunchKernelGGL(k_product, dim3(1), dim3(64), 0, 0, dA, dB, dD, terms);
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    double werr = 0, scale = 0;
    for (int i = 0; i < 16; ++i)
      for (int nn = 0; nn < 16; ++nn) {
        double s = 0;
        for (int k = 0; k < 32; ++k) s += (double)A[i * 32 + k] * (double)B[k * 16 + nn];
        werr = fmax(werr, fabs(D[i * 16 + nn] - s));
        scale = fmax(scale, fabs(s));
      }
    printf("(%d) product, %d term(s): worst |D - fp64| = %.3g on entries up to %.3g (relative %.2g)\n", terms == 1 ? 2 : 3, terms, werr, scale, werr / scale);
  }

  auto worst_of = [&](int terms_hi_only) {
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    double werr = 0;
    for (int i = 0; i < 16; ++i)
      for (int nn = 0; nn < 16; ++nn) {
        double s = 0;
        for (int k = 0; k < 32; ++k) s += (double)A[i * 32 + k] * (double)B[k * 16 + nn];
        const double e = fabs(D[i * 16 + nn] - s);
        werr = (e != e) ? 1e30 : fmax(werr, e);
      }
    (void)terms_hi_only;
    return werr;
  };
  printf("(4a) three DEPENDENT MFMAs, wait states between them -> worst |D - fp64| (3-term product: 5e-4 when right)\n");
#define DEP(N) hipLaunchKernelGGL(k_dependent<N>, dim3(1), dim3(64), 0, 0, dA, dB, dD); printf("   %s %d: %.3g\n", N >= 0 ? "s_nop states" : "independent MFMAs between", N >= 0 ? N : -N, worst_of(0));
  DEP(0) DEP(1) DEP(2) DEP(3) DEP(4) DEP(5) DEP(6) DEP(8) DEP(-1) DEP(-2) DEP(-3)
  printf("(4b) v_cvt_pkrtz_f16_f32 writes a B register, wait states, MFMA reads it -> worst |D - fp64| (1-term product: 1 when right)\n");
#define V2M(N) hipLaunchKernelGGL(k_valu_to_mfma<N>, dim3(1), dim3(64), 0, 0, dA, dB, dD); printf("   wait states %d: %.3g\n", N, worst_of(1));
  V2M(0) V2M(1) V2M(2) V2M(3) V2M(4) V2M(6)
  printf("(4c) MFMA, wait states, vector instruction reads the result -> worst |D - fp64| (1-term product: 1 when right)\n");
#define M2V(N) hipLaunchKernelGGL(k_mfma_to_valu<N>, dim3(1), dim3(64), 0, 0, dA, dB, dD); printf("   wait states %d: %.3g\n", N, worst_of(1));
  M2V(0) M2V(1) M2V(2) M2V(3) M2V(4) M2V(5) M2V(6) M2V(7) M2V(8) M2V(12)
  return 0;
}

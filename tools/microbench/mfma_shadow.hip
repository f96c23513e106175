// mfma_shadow.hip -- how many vector instructions a LONE wavefront issues in the shadow of its own fp32 MFMAs (round 6).
// The N = 50 balancer's iteration is 39 v_mfma_f32_16x16x4_f32 (8 passes: 32 cycles of the matrix pipe each) and ~70 packed
// element-wise instructions; counters say the pipe is busy 48 % of the launch. Before / after software-pipelining the
// iteration (csrc/mpc.hpp) this asks the hardware directly: a loop of 8 independent MFMAs (four accumulators), each followed
// by K independent vector instructions (K = 0 .. 6), plain or packed; and the same with two accumulator chains only.
// Second part: the same question for v_mfma_f32_16x16x32_f16, which mpc_tile_h uses since.
// Wall clock per MFMA from hipEvents (the shader clock under MFMA load is part of the answer), one wavefront per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_shadow.hip -o /tmp/mfma_shadow && /tmp/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MF(i) "v_mfma_f32_16x16x4_f32 %" #i ", %12, %13, %" #i "\n"
#define V0 ""
#define PK1 "v_pk_fma_f32 %4, %14, %15, %4\n"
#define PK2 PK1 "v_pk_fma_f32 %5, %14, %15, %5\n"
#define PK3 PK2 "v_pk_fma_f32 %6, %14, %15, %6\n"
#define PK4 PK3 "v_pk_fma_f32 %7, %14, %15, %7\n"
#define PK5 PK4 "v_pk_fma_f32 %8, %14, %15, %8\n"
#define PK6 PK5 "v_pk_fma_f32 %9, %14, %15, %9\n"
#define S1 "v_fma_f32 %10, %12, %13, %10\n"
#define S2 S1 "v_fma_f32 %11, %12, %13, %11\n"
#define S4 S2 "v_mul_f32 %10, %12, %10\n" "v_mul_f32 %11, %13, %11\n"
#define S6 S4 "v_add_f32 %10, %12, %10\n" "v_add_f32 %11, %13, %11\n"

// CH = number of accumulator chains in use (4: every MFMA independent of the previous three; 2: of the previous one; 1: a dependent chain)
#define BODY4(V) MF(0) V MF(1) V MF(2) V MF(3) V MF(0) V MF(1) V MF(2) V MF(3) V
#define BODY2(V) MF(0) V MF(1) V MF(0) V MF(1) V MF(0) V MF(1) V MF(0) V MF(1) V
#define BODY1(V) MF(0) V MF(0) V MF(0) V MF(0) V MF(0) V MF(0) V MF(0) V MF(0) V

#define KERNEL(NAME, BODY)                                                                                                   \
  __global__ __launch_bounds__(64) void NAME(float* out, int iters, float x, float y) {                                      \
    f4 c[4];                                                                                                                 \
    f2 p[6];                                                                                                                 \
    float s0 = x, s1 = y;                                                                                                    \
    const f2 b2 = {x, y}, c2 = {y, x};                                                                                       \
    for (int i = 0; i < 4; ++i) c[i] = f4{x, y, x, y};                                                                       \
    for (int i = 0; i < 6; ++i) p[i] = f2{x + i, y};                                                                         \
    for (int it = 0; it < iters; ++it)                                                                                       \
      asm volatile(BODY : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]),    \
                   "+v"(p[4]), "+v"(p[5]), "+v"(s0), "+v"(s1)                                                                \
                   : "v"(x), "v"(y), "v"(b2), "v"(c2));                                                                      \
    float s = s0 + s1;                                                                                                       \
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];                                                  \
    for (int i = 0; i < 6; ++i) s += p[i].x + p[i].y;                                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                          \
  }

KERNEL(k4_0, BODY4(V0))
KERNEL(k4_pk1, BODY4(PK1))
KERNEL(k4_pk2, BODY4(PK2))
KERNEL(k4_pk3, BODY4(PK3))
KERNEL(k4_pk4, BODY4(PK4))
KERNEL(k4_pk6, BODY4(PK6))
KERNEL(k4_s2, BODY4(S2))
KERNEL(k4_s4, BODY4(S4))
KERNEL(k4_s6, BODY4(S6))
KERNEL(k2_0, BODY2(V0))
KERNEL(k2_pk2, BODY2(PK2))
KERNEL(k1_0, BODY1(V0))
KERNEL(k1_pk2, BODY1(PK2))

// the fp16 matrix path: v_mfma_f32_16x16x32_f16, eight times the multiply-adds of the fp32 form per instruction (A = B = the
// four-register operands %16, %17)
#define MH(i) "v_mfma_f32_16x16x32_f16 %" #i ", %16, %17, %" #i "\n"
#define BODYH(V) MH(0) V MH(1) V MH(2) V MH(3) V MH(0) V MH(1) V MH(2) V MH(3) V
#define S3 S2 "v_mul_f32 %10, %12, %10\n"
#define KERNELH(NAME, BODY)                                                                                                  \
  __global__ __launch_bounds__(64) void NAME(float* out, int iters, float x, float y) {                                      \
    f4 c[4];                                                                                                                 \
    f2 p[6];                                                                                                                 \
    float s0 = x, s1 = y;                                                                                                    \
    const f2 b2 = {x, y}, c2 = {y, x};                                                                                       \
    const f4 ha = {x, y, x, y}, hb = {y, x, y, x};                                                                           \
    for (int i = 0; i < 4; ++i) c[i] = f4{x, y, x, y};                                                                       \
    for (int i = 0; i < 6; ++i) p[i] = f2{x + i, y};                                                                         \
    for (int it = 0; it < iters; ++it)                                                                                       \
      asm volatile(BODY : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]),    \
                   "+v"(p[4]), "+v"(p[5]), "+v"(s0), "+v"(s1)                                                                \
                   : "v"(x), "v"(y), "v"(b2), "v"(c2), "v"(ha), "v"(hb));                                                    \
    float s = s0 + s1;                                                                                                       \
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];                                                  \
    for (int i = 0; i < 6; ++i) s += p[i].x + p[i].y;                                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                          \
  }
KERNELH(h_0, BODYH(V0))
KERNELH(h_s1, BODYH(S1))
KERNELH(h_s2, BODYH(S2))
KERNELH(h_s3, BODYH(S3))
KERNELH(h_s4, BODYH(S4))
KERNELH(h_s6, BODYH(S6))
KERNELH(h_pk1, BODYH(PK1))
KERNELH(h_pk2, BODYH(PK2))
KERNELH(h_pk3, BODYH(PK3))
KERNELH(h_pk4, BODYH(PK4))

typedef void (*kern_t)(float*, int, float, float);

int main() {
  float* out;
  const int waves = 1024;  // one per SIMD
  hipMalloc(&out, waves * 64 * sizeof(float));
  struct { const char* name; kern_t k; } ks[] = {
      {"4 chains, MFMA only", k4_0}, {"4 chains, +1 v_pk_fma_f32 per MFMA", k4_pk1}, {"4 chains, +2 packed", k4_pk2}, {"4 chains, +3 packed", k4_pk3},
      {"4 chains, +4 packed", k4_pk4}, {"4 chains, +6 packed", k4_pk6}, {"4 chains, +2 plain fp32", k4_s2}, {"4 chains, +4 plain fp32", k4_s4},
      {"4 chains, +6 plain fp32", k4_s6}, {"2 chains, MFMA only", k2_0}, {"2 chains, +2 packed", k2_pk2}, {"1 chain (dependent), MFMA only", k1_0},
      {"1 chain, +2 packed", k1_pk2},
      {"fp16 16x16x32, MFMA only", h_0}, {"fp16 16x16x32, +1 plain fp32", h_s1}, {"fp16 16x16x32, +2 plain fp32", h_s2}, {"fp16 16x16x32, +3 plain fp32", h_s3},
      {"fp16 16x16x32, +4 plain fp32", h_s4}, {"fp16 16x16x32, +6 plain fp32", h_s6}, {"fp16 16x16x32, +1 packed", h_pk1}, {"fp16 16x16x32, +2 packed", h_pk2},
      {"fp16 16x16x32, +3 packed", h_pk3}, {"fp16 16x16x32, +4 packed", h_pk4}};
  const int iters = 4000;
  for (int grid : {128, 1024}) {
    printf("%d wavefronts (%s)\n", grid, grid == 1024 ? "one per SIMD, the whole chip" : "one per SIMD on an eighth of the chip");
    for (auto& e : ks) {
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(64), 0, 0, out, 100, 1.0f, 0.5f);
      hipDeviceSynchronize();
      hipEventRecord(a, 0);
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(64), 0, 0, out, iters, 1.0f, 0.5f);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, a, b);
      const double ns = ms * 1e6 / (iters * 8.0);
      printf("  %-40s %6.2f ns per MFMA (+ its vector instructions) = %5.1f cycles at 2.4 GHz\n", e.name, ns, ns * 2.4);
    }
  }
  return 0;
}

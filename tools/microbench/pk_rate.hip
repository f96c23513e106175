// pk_rate.hip -- what a PACKED fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 operations per
// lane and instruction) costs a wavefront that has a SIMD to itself -- the regime of the step kernels up to 8192 envs --
// and two wavefronts that share one. VERDICT r4 item 3a asked for it before any pair-native rewrite: if a lone wave
// issues a packed instruction in the ~4.5-5 cycles it issues any other one, an instruction stream that pairs up is
// half as long. Same method as issue_rate.hip: s_memtime around an unrolled loop, 64 instructions per iteration.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OUT8 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define BODY(STR) asm volatile(R8(STR) R8(STR) : OUT8 : "v"(b), "v"(c))

#define PK_FMA(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define PK_FMA_BCAST(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " op_sel_hi:[1,0,1]\n"
#define PK_FMA_SWAP(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
#define PK_MUL(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n"
#define PK_ADD(i) "v_pk_add_f32 %" #i ", %8, %" #i "\n"
#define PK_FMA_DEP(i) "v_pk_fma_f32 %0, %8, %9, %0\n"
#define PK_FMA_DEP_NOP(i) "v_pk_fma_f32 %0, %8, %9, %0\ns_nop 0\n"

#define KERNEL(NAME, STR)                                                          \
  __global__ void NAME(float* out, long long* cyc, int iters, float x, float y) { \
    f2 a[8];                                                                       \
    const f2 b = {x, x * 0.999f}, c = {y, y * 1.001f};                            \
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)(threadIdx.x + i), (float)i};     \
    long long t0 = __builtin_amdgcn_s_memtime();                                   \
    for (int it = 0; it < iters; ++it) {                                           \
      BODY(STR);                                                                   \
      BODY(STR);                                                                   \
      BODY(STR);                                                                   \
      BODY(STR);                                                                   \
    }                                                                              \
    long long t1 = __builtin_amdgcn_s_memtime();                                   \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                               \
    float s = 0.f;                                                                 \
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
  }

KERNEL(k_pk_fma, PK_FMA)
KERNEL(k_pk_fma_bcast, PK_FMA_BCAST)
KERNEL(k_pk_fma_swap, PK_FMA_SWAP)
KERNEL(k_pk_mul, PK_MUL)
KERNEL(k_pk_add, PK_ADD)
KERNEL(k_pk_fma_dep, PK_FMA_DEP)
KERNEL(k_pk_fma_dep_nop, PK_FMA_DEP_NOP)

// the scalar reference: v_fma_f32, one fp32 operation per lane and instruction
#define S_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define S_FMA_DEP(i) "v_fma_f32 %0, %8, %9, %0\n"
#define KERNEL_S(NAME, STR)                                                        \
  __global__ void NAME(float* out, long long* cyc, int iters, float x, float y) { \
    float a[8];                                                                    \
    const float b = x, c = y;                                                      \
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i);                   \
    long long t0 = __builtin_amdgcn_s_memtime();                                   \
    for (int it = 0; it < iters; ++it) {                                           \
      BODY(STR);                                                                   \
      BODY(STR);                                                                   \
      BODY(STR);                                                                   \
      BODY(STR);                                                                   \
    }                                                                              \
    long long t1 = __builtin_amdgcn_s_memtime();                                   \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                               \
    float s = 0.f;                                                                 \
    for (int i = 0; i < 8; ++i) s += a[i];                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
  }
KERNEL_S(k_s_fma, S_FMA)
KERNEL_S(k_s_fma_dep, S_FMA_DEP)

typedef void (*kern_t)(float*, long long*, int, float, float);

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 1024 * 1024);
  (void)hipMalloc(&cyc, sizeof(long long) * 4096);
  const int iters = 20000;
  struct { const char* name; kern_t k; int per_iter; } ks[] = {
      {"v_fma_f32 (one fp32 op per lane), independent", k_s_fma, 64},
      {"v_fma_f32, dependent chain", k_s_fma_dep, 64},
      {"v_pk_fma_f32 (two fp32 ops per lane), independent", k_pk_fma, 64},
      {"v_pk_fma_f32 op_sel_hi:[1,0,1] (low half broadcast), independent", k_pk_fma_bcast, 64},
      {"v_pk_fma_f32 op_sel swap of the halves, independent", k_pk_fma_swap, 64},
      {"v_pk_mul_f32, independent", k_pk_mul, 64},
      {"v_pk_add_f32, independent", k_pk_add, 64},
      {"v_pk_fma_f32, dependent chain (hardware interlock only)", k_pk_fma_dep, 64},
      {"v_pk_fma_f32 + s_nop 0, dependent chain (what hipcc emits; per pair)", k_pk_fma_dep_nop, 64},
  };
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k_s_fma, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  for (int wps : {1, 2}) {
    for (auto& k : ks) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k.k, dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(256);
      (void)hipMemcpy(h.data(), cyc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
      printf("%d wave(s)/SIMD  %-72s %.2f s_memtime ticks per instruction per wave; wall clock %.2f ns per instruction per wave\n", wps, k.name,
             h[0] / ((double)k.per_iter * iters), ms * 1e6 / ((double)k.per_iter * iters));
    }
  }
  return 0;
}

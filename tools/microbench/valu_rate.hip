// valu_rate.hip -- how fast does one SIMD of gfx950 retire fp32 VALU work?
// Measures cycles per wave-instruction for v_fma_f32 and v_pk_fma_f32 with
// 1, 2, 4 waves per SIMD, with 16 independent accumulators (throughput) and with
// a single dependent chain (latency). Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ void fma_kernel(float* out, long long* cyc, int iters, float b, float c) {
  float a[CHAINS];
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / CHAINS; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void pk_fma_kernel(float* out, long long* cyc, int iters, float b, float c) {
  f2 a[CHAINS];
  long long t0 = __builtin_amdgcn_s_memtime();
  f2 bb = {b, b + 1.f}, cc = {c, c * 0.5f};
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / CHAINS; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(bb), "v"(cc));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double time_kernel(K launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 / reps;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double clk = prop.clockRate * 1e3;  // Hz
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, clk / 1e6);
  float* out;
  long long* cyc;
  hipMalloc(&cyc, sizeof(long long) * 4096);
  std::vector<long long> hc(4096);
  hipMalloc(&out, sizeof(float) * 1024 * 1024 * 4);
  const int iters = 200000;
  // warm the clocks up
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fma_kernel<16>, dim3(cus), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  const double instr = 16.0 * iters;
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps;  // 4 SIMDs x wps waves, one block per CU
    const int blocks = cus;
    double t;
    t = time_kernel([&] { hipLaunchKernelGGL(fma_kernel<16>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f); }, 3);
    hipMemcpy(hc.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("  [s_memtime: %.2f cycles per instr per wave] ", hc[0] / instr);
    printf("v_fma_f32     16 chains, %d waves/SIMD: %.2f cycles per wave-instr per SIMD (%.2f per wave)\n", wps,
           t * clk / (instr * wps), t * clk / instr);
    t = time_kernel([&] { hipLaunchKernelGGL(pk_fma_kernel<8>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f); }, 3);
    hipMemcpy(hc.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("  [s_memtime: %.2f cycles per instr per wave] ", hc[0] / instr);
    printf("v_pk_fma_f32   8 chains, %d waves/SIMD: %.2f cycles per wave-instr per SIMD (%.2f per wave)\n", wps,
           t * clk / (instr * wps), t * clk / instr);
    t = time_kernel([&] { hipLaunchKernelGGL(fma_kernel<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f); }, 3);
    hipMemcpy(hc.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("  [s_memtime: %.2f cycles per instr per wave] ", hc[0] / instr);
    printf("v_fma_f32      1 chain,  %d waves/SIMD: %.2f cycles per wave-instr per SIMD (%.2f per wave)\n", wps,
           t * clk / (instr * wps), t * clk / instr);
    t = time_kernel([&] { hipLaunchKernelGGL(pk_fma_kernel<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f); }, 3);
    hipMemcpy(hc.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("  [s_memtime: %.2f cycles per instr per wave] ", hc[0] / instr);
    printf("v_pk_fma_f32   1 chain,  %d waves/SIMD: %.2f cycles per wave-instr per SIMD (%.2f per wave)\n", wps,
           t * clk / (instr * wps), t * clk / instr);
    t = time_kernel([&] { hipLaunchKernelGGL(fma_kernel<2>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.5f); }, 3);
    hipMemcpy(hc.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("  [s_memtime: %.2f cycles per instr per wave] ", hc[0] / instr);
    printf("v_fma_f32      2 chains, %d waves/SIMD: %.2f cycles per wave-instr per SIMD (%.2f per wave)\n", wps,
           t * clk / (instr * wps), t * clk / instr);
  }
  hipFree(out);
  return 0;
}

// any_order.hip -- do two kernels launched back to back on ONE stream with hipExtAnyOrderLaunch (no barrier between their AQL
// packets) run concurrently on gfx950? (round 6: the step kernel of env.step() k + 1 depends on step k's workgroup of the same
// index only; without the barrier the next launch's dispatch and prologue could hide behind the previous launch's tail, with a
// per-workgroup counter in memory for the dependency.) Each kernel: 256 workgroups of one wavefront spinning ~40 us on s_memtime.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void spin(long long cycles, long long* out) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  long long t = t0;
  while (t - t0 < cycles) t = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t - t0;
}

int main() {
  long long* out;
  hipMalloc(&out, 4096 * sizeof(long long));
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const long long cycles = 100000;  // s_memtime ticks (shader clock): ~40 us
  for (int flags : {0, 1}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(st);
      hipEventRecord(a, st);
      for (int k = 0; k < 4; ++k) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, nullptr, nullptr, flags, cycles, out + 256 * k);
      hipEventRecord(b, st);
      hipEventSynchronize(b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, a, b);
      printf("%s: four kernels of ~40 us each, back to back on one stream: %.1f us (%s)\n", flags ? "hipExtAnyOrderLaunch" : "ordered launches   ", ms * 1e3,
             hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}

// Latency of a dependent chain of scalar loads, one cache line apart: from the kernel
// argument segment (a 1 KB by-value struct, like DevLimits + DevConfig of the step kernels)
// and from a device buffer, in s_memtime ticks (2.4 GHz on gfx950: `calibrate` below measures it against the
// 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/kernarg_latency.hip -o /tmp/kl && /tmp/kl
//   HIP_FORCE_DEV_KERNARG=0 /tmp/kl ; HIP_FORCE_DEV_KERNARG=1 /tmp/kl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Big {
  int w[256];
};

__global__ void chain(Big b, const int* __restrict__ dev, unsigned long long* out) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int x = b.w[0];
#pragma unroll
  for (int i = 0; i < 7; ++i) x = b.w[x & 255];
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(x));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int y = dev[x & 255];
#pragma unroll
  for (int i = 0; i < 7; ++i) y = dev[y & 255];
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(y));
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  // second pass over the same lines: scalar-cache hits
  int z = b.w[y & 255];
#pragma unroll
  for (int i = 0; i < 7; ++i) z = b.w[z & 255];
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(z));
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = t1 - t0;
    out[4 * blockIdx.x + 1] = t2 - t1;
    out[4 * blockIdx.x + 2] = t3 - t2;
    out[4 * blockIdx.x + 3] = z;
  }
}

// s_memtime against the 100 MHz s_memrealtime: which clock the tick counts are in
__global__ void calibrate(float* sink, unsigned long long* out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float a = threadIdx.x;
  for (int i = 0; i < 200000; ++i) asm volatile("v_fmac_f32_e32 %0, %0, %0" : "+v"(a));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  sink[threadIdx.x] = a;
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
}

__global__ void empty_plain(unsigned long long* out) {
  if (out == nullptr) asm volatile("s_nop 0");
}
__global__ void empty_big(Big b, unsigned long long* out) {
  if (b.w[255] == -1) out[0] = 1;  // one argument line read
}

static void launch_interval() {
  hipEvent_t a, z;
  hipEventCreate(&a);
  hipEventCreate(&z);
  unsigned long long* out;
  hipMalloc(&out, 64);
  Big b;
  for (int i = 0; i < 256; ++i) b.w[i] = 0;
  const int K = 2000;
  for (int which = 0; which < 2; ++which) {
    for (int blocks : {1, 512}) {
      for (int i = 0; i < 200; ++i) {
        if (which == 0) hipLaunchKernelGGL(empty_plain, dim3(blocks), dim3(64), 0, 0, out);
        else hipLaunchKernelGGL(empty_big, dim3(blocks), dim3(64), 0, 0, b, out);
      }
      hipDeviceSynchronize();
      hipEventRecord(a, 0);
      for (int i = 0; i < K; ++i) {
        if (which == 0) hipLaunchKernelGGL(empty_plain, dim3(blocks), dim3(64), 0, 0, out);
        else hipLaunchKernelGGL(empty_big, dim3(blocks), dim3(64), 0, 0, b, out);
      }
      hipEventRecord(z, 0);
      hipEventSynchronize(z);
      float ms = 0.f;
      hipEventElapsedTime(&ms, a, z);
      printf("back-to-back launches of an empty kernel, %s, %d blocks of 64: %.2f us per launch\n", which ? "1 KB of arguments by value" : "one pointer argument", blocks, ms * 1e3 / K);
    }
  }
}

int main() {
  {
    float* sink;
    unsigned long long* out;
    unsigned long long host[2];
    hipMalloc(&sink, 64 * sizeof(float));
    hipMalloc(&out, 16);
    hipLaunchKernelGGL(calibrate, dim3(1), dim3(64), 0, 0, sink, out);
    hipDeviceSynchronize();
    hipMemcpy(host, out, 16, hipMemcpyDeviceToHost);
    printf("s_memtime: %llu ticks in %llu ticks of the 100 MHz clock: %.1f MHz; 200000 dependent v_fmac_f32: %.2f ticks each\n", host[0], host[1],
           100.0 * host[0] / host[1], host[0] / 200000.0);
  }
  launch_interval();
  Big b;
  for (int i = 0; i < 256; ++i) b.w[i] = 0;
  for (int k = 0; k < 8; ++k) b.w[(k * 16) & 255] = ((k + 1) * 16) & 255;  // a chain over 8 cache lines
  int* dev;
  unsigned long long* out;
  const int blocks = 512;
  hipMalloc(&dev, sizeof(b));
  hipMalloc(&out, blocks * 4 * sizeof(unsigned long long));
  hipMemcpy(dev, b.w, sizeof(b), hipMemcpyHostToDevice);
  unsigned long long* host = (unsigned long long*)malloc(blocks * 4 * sizeof(unsigned long long));
  const char* env = getenv("HIP_FORCE_DEV_KERNARG");
  for (int rep = 0; rep < 5; ++rep) {
    b.w[255] = rep;  // a new argument block every launch
    hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, b, dev, out);
    hipDeviceSynchronize();
    hipMemcpy(host, out, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int i = 0; i < blocks; ++i)
      for (int j = 0; j < 3; ++j) {
        s[j] += host[4 * i + j];
        if (host[4 * i + j] > mx[j]) mx[j] = host[4 * i + j];
      }
    printf("HIP_FORCE_DEV_KERNARG=%s launch %d: chain of 8 dependent scalar loads, cycles per load, mean over %d waves (slowest wave): "
           "kernel arguments %.0f (%.0f)  device buffer %.0f (%.0f)  kernel arguments again (scalar-cache hits) %.0f (%.0f)\n",
           env ? env : "unset", rep, blocks, s[0] / blocks / 8, mx[0] / 8, s[1] / blocks / 8, mx[1] / 8, s[2] / blocks / 8, mx[2] / 8);
  }
  return 0;
}

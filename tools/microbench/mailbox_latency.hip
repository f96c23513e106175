// mailbox_latency.hip -- what handing work to a RESIDENT kernel costs on MI355X.
//
// The launch boundary is 3.9 us of the 15.2 us headline launch; 32 steps per launch (the on-device agent) remove it:
// 11.3-11.6 us per step. For a HOST-DRIVEN loop -- env.step(policy(obs)) with the policy's own kernels in between -- the
// same could come from a persistent step kernel that polls a mailbox: the policy's last kernel is followed, in stream
// order, by a write of a sequence word the resident kernel spins on; the step's results are followed by a sequence word
// the policy's stream waits for (hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory). Whether that beats one
// launch per step is a matter of four latencies, measured here with the GPU's constant 100 MHz clock (s_memrealtime,
// the same counter on every CU):
//   gap      end of kernel A -> start of dependent kernel B on the same stream (what the public loop pays today, x3)
//   post     end of kernel A -> [hipStreamWriteValue32] -> a resident kernel's polling load sees the value
//   signal   a resident kernel's store (system scope) -> [hipStreamWaitValue32 satisfied] -> start of kernel B
//   poll     a store by one resident kernel -> seen by another resident kernel (the floor: memory only)
// Every spin is BOUNDED (20 ms on the 100 MHz clock): a lost signal ends the kernel, it cannot hang the box.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/mailbox_latency.hip -o /tmp/mailbox && /tmp/mailbox
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz
__device__ __forceinline__ unsigned load_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_sys(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
constexpr unsigned long long kTimeoutTicks = 2000000ull;  // 20 ms

// a short kernel that stamps the clock when it starts and when it ends (~1 us of dependent work in between)
__global__ void stamp(unsigned long long* start, unsigned long long* end, float* sink) {
  const unsigned long long t0 = now();
  float x = (float)threadIdx.x;
  for (int i = 0; i < 200; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  if (x == 123.f) sink[0] = x;
  if (threadIdx.x == 0) {
    if (start) *start = t0;
    if (end) *end = now();
  }
}
// resident: waits until *flag == value, stamps the clock, then (optionally) raises *raise to value
__global__ void wait_for(const unsigned* flag, unsigned value, unsigned long long* seen, unsigned* raise, unsigned long long* raised, int* timed_out) {
  const unsigned long long t0 = now();
  bool ok = false;
  while (now() - t0 < kTimeoutTicks) {
    if (load_sys(flag) == value) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  const unsigned long long t = now();
  if (threadIdx.x == 0) {
    *seen = t;
    if (!ok) *timed_out = 1;
    if (raise) {
      if (raised) *raised = now();
      store_sys(raise, value);
    }
  }
}
// resident: raises *flag to value after a short delay (so that the waiter is already spinning), stamping the clock
__global__ void raise_after(unsigned* flag, unsigned value, unsigned long long* raised, unsigned delay_ticks) {
  const unsigned long long t0 = now();
  while (now() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(1);
  if (threadIdx.x == 0) {
    *raised = now();
    store_sys(flag, value);
  }
}

static void report(const char* name, std::vector<double>& us) {
  std::sort(us.begin(), us.end());
  printf("%-92s median %6.2f us   p10 %6.2f   p90 %6.2f   (n = %zu)\n", name, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], us.size());
}

int main() {
  int can_wait = 0;
  CHECK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned *flag_a, *flag_b;
  CHECK(hipExtMallocWithFlags((void**)&flag_a, 8, hipMallocSignalMemory));
  CHECK(hipExtMallocWithFlags((void**)&flag_b, 8, hipMallocSignalMemory));
  CHECK(hipMemset(flag_a, 0, 8));
  CHECK(hipMemset(flag_b, 0, 8));
  unsigned long long* t;  // device timestamps
  CHECK(hipMalloc(&t, 16 * sizeof(unsigned long long)));
  float* sink;
  CHECK(hipMalloc(&sink, 64));
  int* timed_out;
  CHECK(hipMalloc(&timed_out, sizeof(int)));
  CHECK(hipMemset(timed_out, 0, sizeof(int)));
  unsigned long long h[16];
  const int reps = 200;
  std::vector<double> gap, post, signal, poll, round_trip;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s1, t, t + 1, sink);  // warm up
  CHECK(hipStreamSynchronize(s1));
  for (int r = 1; r <= reps; ++r) {
    const unsigned v = (unsigned)r;
    // gap: A then B on one stream
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s1, (unsigned long long*)nullptr, t + 0, sink);
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s1, t + 1, (unsigned long long*)nullptr, sink);
    CHECK(hipStreamSynchronize(s1));
    // post: resident waiter on s2; A then hipStreamWriteValue32 on s1
    hipLaunchKernelGGL(wait_for, dim3(1), dim3(64), 0, s2, flag_a, v, t + 3, (unsigned*)nullptr, (unsigned long long*)nullptr, timed_out);
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s1, (unsigned long long*)nullptr, t + 2, sink);
    CHECK(hipStreamWriteValue32(s1, flag_a, v, 0));
    CHECK(hipStreamSynchronize(s1));
    CHECK(hipStreamSynchronize(s2));
    // signal: resident raiser on s2 (raises flag_b after 30 us); s1 waits for the value, then B
    if (can_wait) {
      hipLaunchKernelGGL(raise_after, dim3(1), dim3(64), 0, s2, flag_b, v, t + 4, 3000u);
      CHECK(hipStreamWaitValue32(s1, flag_b, v, hipStreamWaitValueEq, 0xFFFFFFFFu));
      hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s1, t + 5, (unsigned long long*)nullptr, sink);
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipStreamSynchronize(s2));
    }
    // poll: two resident kernels, device store -> device polling load (flag_a reused with a value no write op used)
    hipLaunchKernelGGL(wait_for, dim3(1), dim3(64), 0, s1, flag_a, v + 100000u, t + 7, (unsigned*)nullptr, (unsigned long long*)nullptr, timed_out);
    hipLaunchKernelGGL(raise_after, dim3(1), dim3(64), 0, s2, flag_a, v + 100000u, t + 6, 3000u);
    CHECK(hipStreamSynchronize(s1));
    CHECK(hipStreamSynchronize(s2));
    CHECK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
    if (r > 10) {  // (the first rounds load code objects)
      gap.push_back((double)(h[1] - h[0]) * 0.01);
      post.push_back((double)(h[3] - h[2]) * 0.01);
      if (can_wait) signal.push_back((double)(h[5] - h[4]) * 0.01);
      poll.push_back((double)(h[7] - h[6]) * 0.01);
    }
  }
  int lost = 0;
  CHECK(hipMemcpy(&lost, timed_out, sizeof(int), hipMemcpyDeviceToHost));
  printf("bounded spins that ran into their 20 ms limit: %d\n", lost);
  report("gap:    end of kernel A -> start of dependent kernel B, same stream", gap);
  report("post:   end of kernel A -> hipStreamWriteValue32 -> seen by a resident kernel's polling load", post);
  if (can_wait) report("signal: resident kernel's store -> hipStreamWaitValue32 satisfied -> start of kernel B", signal);
  report("poll:   resident kernel's store -> seen by another resident kernel's polling load", poll);
  return 0;
}

// rollout.hpp -- consumer side of a rollout (BASELINE.json configs[3] names a
// "PPO rollout consumer"): generalized advantage
// estimation over a [T][N] rollout that already sits in HBM, one env column per
// lane, walking the T steps backwards. Pure streaming: 3 words read and 2
// written per (step, env), coalesced over envs -> HBM bound.
//
// Recurrence (Schulman et al. 2016, as every PPO implementation states it):
//   delta_t = r_t + gamma * V_{t+1} * (1 - start_{t+1}) - V_t
//   A_t     = delta_t + gamma * lambda * (1 - start_{t+1}) * A_{t+1}
//   R_t     = A_t + V_t
// with V_T = last_values, start_T = last_dones; start_t = 1 when step t is the
// first of an episode.
#pragma once

#include <hip/hip_runtime.h>

namespace upkie {

__global__ __launch_bounds__(256) void gae_kernel(int T, int N, const float* __restrict__ rewards, const float* __restrict__ values,
                                                  const uint8_t* __restrict__ episode_starts, const float* __restrict__ last_values,
                                                  const uint8_t* __restrict__ last_dones, float gamma, float lambda,
                                                  float* __restrict__ advantages, float* __restrict__ returns) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float next_value = last_values[n];
  float next_non_terminal = last_dones[n] ? 0.f : 1.f;
  float gae = 0.f;
  // (the loads do not depend on the recurrence: unrolled, eight time steps are in flight per lane)
#pragma unroll 8
  for (int t = T - 1; t >= 0; --t) {
    const size_t i = (size_t)t * N + n;
    const float v = values[i];
    const float delta = rewards[i] + gamma * next_value * next_non_terminal - v;
    gae = delta + gamma * lambda * next_non_terminal * gae;
    advantages[i] = gae;
    returns[i] = gae + v;
    next_value = v;
    next_non_terminal = episode_starts[i] ? 0.f : 1.f;
  }
}

// A linear policy in ONE launch: act[n][a] = clamp(sum_d obs[n][d] W[d][a] + bias[a], -clip, clip) (clip <= 0: no clamp).
// What an RL loop `env.step(policy(obs))` puts between two steps: as torch ops (`obs @ W`, `+ b`, `.clamp()`) it is two or
// three launches of 2-5 us each (rocBLAS's gemv for a 4-column matrix: 4.9 us) behind a 14.5 us step; as this kernel, one
// launch of about 2 us. One env per lane; W and bias are wave-uniform reads, the observation row of a lane is contiguous.
__global__ __launch_bounds__(256) void linear_policy_kernel(int N, int D, int A, const float* __restrict__ obs, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float clip, float* __restrict__ act) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* row = obs + (size_t)n * D;
  for (int a = 0; a < A; ++a) {
    float x = bias ? bias[a] : 0.f;
    for (int d = 0; d < D; ++d) x = fmaf(row[d], W[(size_t)d * A + a], x);
    if (clip > 0.f) x = fminf(fmaxf(x, -clip), clip);
    act[(size_t)n * A + a] = x;
  }
}

}  // namespace upkie

// wave_io.hpp -- [env][W] row-major arrays (the layouts of the spine
// observation: servo [B][6][5], rotation [B][9], vectors [B][3]) moved between
// HBM and the lanes of a one-wave block as contiguous runs of 16-byte accesses.
#pragma once

#include <hip/hip_runtime.h>

namespace upkie {

// W words per env for the (up to) 64 consecutive envs of this one-wave block,
// stored as ONE contiguous run of 16-byte coalesced stores: the lanes first
// lay their rows out in LDS the way HBM wants them ([env][W] row-major), then
// the wave streams that image out. Per-lane stores of W scattered dwords
// (stride 4 W bytes across lanes) fill every cache line in W separate pieces.
template <int W>
__device__ __forceinline__ void wave_store_rows(float* __restrict__ dst, int e0, int n_valid, const float (&v)[W], float* lds) {
  const int lane = threadIdx.x;
  __syncthreads();  // the previous image has been streamed out
#pragma unroll
  for (int k = 0; k < W; ++k) lds[lane * W + k] = v[k];
  __syncthreads();
  float* row = dst + (size_t)e0 * W;  // 64 * W * 4 bytes per wave: 16-byte aligned
  const int total = n_valid * W;
  const int total4 = total >> 2;
  for (int i = lane; i < total4; i += 64) reinterpret_cast<float4*>(row)[i] = reinterpret_cast<const float4*>(lds)[i];
  for (int i = 4 * total4 + lane; i < total; i += 64) row[i] = lds[i];
}

// The reverse: the wave streams the rows of its envs into LDS with coalesced
// 16-byte loads, then every lane picks its own row.
template <int W>
__device__ __forceinline__ void wave_load_rows(const float* __restrict__ src, int e0, int n_valid, float (&v)[W], float* lds) {
  const int lane = threadIdx.x;
  __syncthreads();
  const float* row = src + (size_t)e0 * W;
  const int total = n_valid * W;
  const int total4 = total >> 2;
  for (int i = lane; i < total4; i += 64) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(row)[i];
  for (int i = 4 * total4 + lane; i < total; i += 64) lds[i] = row[i];
  __syncthreads();
  const int r = lane < n_valid ? lane : n_valid - 1;
#pragma unroll
  for (int k = 0; k < W; ++k) v[k] = lds[r * W + k];
}

}  // namespace upkie

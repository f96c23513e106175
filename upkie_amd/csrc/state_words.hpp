// state_words.hpp -- one 4-byte word of the state [UPKIE_STATE_WORDS][B] as an lvalue, two ways (device code only).
// Word w of env e lives at state[w * B + e]. The step kernels reach it through 64-bit lane addresses, or through ONE buffer
// descriptor over the whole state (four scalar registers), the word's row as the instruction's scalar offset and the lane's
// 32-bit byte offset, which every word of an env shares: `buffer_load/store_dword v, voffset, s[descriptor], soffset offen`.
// The lane addresses of the thirty-odd words a step loads and stores hold some fifty vector registers between them for the
// whole launch; kernels whose step is a loop body (several steps per launch, SAME_STEP autoreset inside the launch) spilled
// them and reloaded them every step. Which instantiations use which: octet.hpp (octet_state_through_descriptor), pair.hpp.
#pragma once

// (included from step_kernels.hpp inside namespace upkie, like pair.hpp and octet.hpp)
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

struct StateWord {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned lane_offset, row_offset;  // bytes: per lane (vector register), per word (scalar register)
  __device__ __forceinline__ operator float() const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_offset, (int)row_offset, 0));
  }
  __device__ __forceinline__ void operator=(float value) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, value), rsrc, (int)lane_offset, (int)row_offset, 0);
  }
};
struct StateWordAt {  // ... or through the lane's own 64-bit address
  float* word;
  __device__ __forceinline__ operator float() const { return *word; }
  __device__ __forceinline__ void operator=(float value) const { *word = value; }
};
template <bool BUFFERED>
__device__ __forceinline__ auto state_word(__amdgpu_buffer_rsrc_t rsrc, float* st, size_t word_index, unsigned lane_offset, unsigned row_offset) {
  if constexpr (BUFFERED) {
    return StateWord{rsrc, lane_offset, row_offset};
  } else {
    return StateWordAt{st + word_index};
  }
}
// the descriptor over a handle's state (byte offsets stay below 2^32: launch_step keeps these mappings to batches that fit)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t state_descriptor(float* state, int num_envs) {
  return __builtin_amdgcn_make_buffer_rsrc(state, 0, (int)((unsigned)UPKIE_STATE_WORDS * (unsigned)num_envs * 4u), 0x00020000);
}

#endif

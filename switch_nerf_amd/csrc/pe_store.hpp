// Write-out of one encoded row per thread (the positional-encoding kernels): shared by elementwise.hip and bounds.hip.
#pragma once
#include "common.hpp"

namespace swn {

template <typename T>
__device__ __forceinline__ void store_vals(T* dst, const float* v, int n_pad) {
  if constexpr (sizeof(T) == 2) {
    for (int c = 0; c < n_pad; c += 8) {
      uint4 u;
      u.x = pack_bf16x2(v[c + 0], v[c + 1]);
      u.y = pack_bf16x2(v[c + 2], v[c + 3]);
      u.z = pack_bf16x2(v[c + 4], v[c + 5]);
      u.w = pack_bf16x2(v[c + 6], v[c + 7]);
      *(uint4*)(dst + c) = u;
    }
  } else {
    for (int c = 0; c < n_pad; c += 4) *(float4*)(dst + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
  }
}

// A thread owns a whole row v[0, used) (zero-padded to pe_stride columns; pe_stride * sizeof(T) bytes, i.e. a 256 B stride
// between lanes): stage the block's rows in LDS and write them out as one contiguous, fully coalesced region (the block's
// rows are consecutive in memory).  Launch with 128 threads (bf16) / 64 threads (fp32) per block, row p = global thread id;
// surplus threads of the last block pass live = false.
template <typename T, int VN>
__device__ __forceinline__ void pe_store_rows(const float (&v)[VN], int used, T* __restrict__ pe, int pe_stride, long p, bool live,
                                              long total_rows) {
  constexpr int step = 16 / (int)sizeof(T);
  constexpr int NT = sizeof(T) == 2 ? 128 : 64;                  // threads per block (see the launchers)
  constexpr int ROWB = 128 * (int)sizeof(T) + 16;                // LDS row stride: <= 128 columns, +16 B against bank conflicts
  __shared__ __attribute__((aligned(16))) char stage[NT * ROWB];
  const bool staged = pe_stride <= 128;                          // (uniform) wider rows fall back to direct stores
  T* dst = staged ? (T*)(stage + threadIdx.x * ROWB) : pe + p * pe_stride;
  if (live) {
    // v[] is indexed with COMPILE-TIME indices only (a runtime index would put the whole row into scratch memory: 368 bytes per lane
    // and two waves per SIMD for the positional-encoding kernels)
    constexpr int VPAD = (VN + step - 1) / step * step;
#pragma unroll
    for (int c0 = 0; c0 < VPAD; c0 += step) {
      if (c0 < pe_stride) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = 0.f;
#pragma unroll
        for (int j = 0; j < step; ++j) {
          const int c = c0 + j;
          tmp[j] = (c < VN && c < used) ? v[c < VN ? c : 0] : 0.f;
        }
        store_vals<T>(dst + c0, tmp, step);
      }
    }
    for (int c0 = VPAD; c0 < pe_stride; c0 += step) {            // zero padding beyond the row buffer
      float tmp[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) tmp[j] = 0.f;
      store_vals<T>(dst + c0, tmp, step);
    }
  }
  if (staged) {
    __syncthreads();
    const int cpr = pe_stride / step;                            // 16-byte chunks per row
    const long row0 = (long)blockIdx.x * NT;
    const long rows = min((long)NT, total_rows - row0);
    char* out = (char*)(pe + row0 * pe_stride);
    for (int c = threadIdx.x; c < rows * cpr; c += NT) {
      const int row = c / cpr, ch = c - row * cpr;
      *(uint4*)(out + (long)c * 16) = *(const uint4*)(stage + row * ROWB + ch * 16);
    }
  }
}

}  // namespace swn

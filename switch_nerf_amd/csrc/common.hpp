// Shared helpers for libswn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/swn.h"

namespace swn {

extern thread_local char g_err[512];
int set_error(const char* fmt, ...);

#define SWN_CHECK(cond, ...)                 \
  do {                                       \
    if (!(cond)) return swn::set_error(__VA_ARGS__); \
  } while (0)

#define SWN_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) return swn::set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
  } while (0)

typedef uint16_t bf16_t;  // raw bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even (NaN payloads are not preserved; inputs are finite)
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace swn

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

// The 16-bit compute type.  The library is built twice from the same sources (build.sh): libswn_hip.so with bfloat16 (dtype code
// SWN_BF16, the reference's amp_use_bfloat16 recipes) and libswn_hip_f16.so (-DSWN_HALF_F16) with IEEE half (dtype code SWN_F16,
// the reference's default fp16 autocast + GradScaler, runner.py:483, 679; BASELINE configs[4]).  Everything below the three
// conversion helpers, the MFMA macro and the constants is type-agnostic: `bf16_t` reads "the 16-bit compute type, raw bits".
typedef uint16_t bf16_t;  // raw bits

typedef float f32x2_t __attribute__((ext_vector_type(2)));
#ifdef SWN_HALF_F16
#define SWN_HALF SWN_F16
#define SWN_HALF_ONE_X2 0x3C003C00u        /* two 1.0 */
typedef _Float16 hwf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 swn_mfma16_t __attribute__((ext_vector_type(8)));
#define SWN_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(swn_mfma16_t, a), __builtin_bit_cast(swn_mfma16_t, b), c, 0, 0, 0)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {      // round to nearest even (v_cvt_f16_f32 x 2 + pack)
  const hwf16x2_t v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
#else
#define SWN_HALF SWN_BF16
#define SWN_HALF_ONE_X2 0x3F803F80u
typedef __bf16 hwbf16x2_t __attribute__((ext_vector_type(2)));
typedef short swn_mfma16_t __attribute__((ext_vector_type(8)));
#define SWN_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(swn_mfma16_t, a), __builtin_bit_cast(swn_mfma16_t, b), c, 0, 0, 0)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even conversions: gfx950 has a packed hardware convert (v_cvt_pk_bf16_f32); going through the
// __bf16 vector type lets hipcc emit it (one VALU op per two values instead of ~10 of integer rounding code).
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hwbf16x2_t));
}
#endif
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xFFFFu); }

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

int chain_wide_launch(const swn_chain_desc& d, void* stream);   // chain.hip compiled with -DSWN_WIDE=1
int chain_wide_tile_rows(int dtype);
int chain_wide2_launch(const swn_chain_desc& d, void* stream);  // chain.hip compiled with -DSWN_WIDE=2 (16-bit types: 128-row tiles, 8 waves)
int chain_wide2_tile_rows();
int chain_concat_launch(const swn_chain_desc& d, void* stream); // chain.hip compiled with -DSWN_CONCAT=1 (concat-skip layer mode)
int gate_fwd_mfma_launch(const void* g, const float* ln_w, const float* ln_b, const float* wg, int n_tokens, int n_experts, float* gates,
                         int32_t* idx, float* gmax, float* stats, void* stream);
int gate_bwd_mfma_launch(const void* g, const float* ln_w, const float* wg, const float* gates, const int32_t* idx, const float* d_gmax,
                         const float* stats, const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int n_experts,
                         void* dg, float* dlogits, float* partial, void* stream);
int gate_bwd_mfma_blocks(int n_tokens);
// gate_mfma.hip, the 512-feature router (<= 16 experts): forward, and the backward's data path (dlogits, dg)
int gate_fwd_wide_launch(const void* g, const float* ln_w, const float* ln_b, const float* wg, int n_tokens, int n_experts, float* gates,
                         int32_t* idx, float* gmax, float* stats, void* stream);
int gate_bwd_wide_launch(const void* g, const float* ln_w, const float* wg, const float* gates, const int32_t* idx, const float* d_gmax,
                         const float* stats, const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int n_experts,
                         void* dg, float* dlogits, void* stream);
int gate_dwg_wide_blocks(int n_tokens);
int gate_dwg_wide_launch(const void* g, bool layer_norm, const float* stats, const float* dlogits, int n_tokens, int n_experts, float* partial,
                         void* stream);
bool chain_big_eligible(const swn_chain_desc& d);                // chain_big.hip: the 256-row geometry
bool chain_persistent_eligible(const swn_chain_desc& d);         // chain_big.hip: geometries 6 / 7 (persistent; also the dense front chains)
int chain_big_launch(const swn_chain_desc& d, void* stream);
int chain_big_tile_rows(int geometry);
int chain_big_mask_words_per_tile(int geometry);

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Device fills are KERNELS in this library, never hipMemsetAsync: captured into a hipGraph a hipMemsetAsync becomes a memset NODE, and
// with ROCm 7.2 such a node leaves wrong memory contents from the SECOND replay of the graph on (scripts/memset_graph_repro.py: torch +
// libamdhip64 only; first replay correct, every later one wrong, with or without freeing the target during the capture; the same
// graph with fill kernels is correct).  That was the "GPU fault on the second replay" of round 2's inference graph: the per-segment
// counts were not zero on the second replay, route_finalize_kernel computed negative slots from them and wrote perm[] out of bounds.
static __global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
static inline hipError_t fill_u32_async(void* p, uint32_t v, size_t bytes, hipStream_t s) {   // bytes: a multiple of 4
  const long n = (long)(bytes / 4);
  if (n <= 0) return hipSuccess;
  int blocks = cdiv(n, 256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t*)p, v, n);
  return hipGetLastError();
}

// Ordered reduction of per-block partial sums: out[t] (+)= sum_b partial[b * n + t], the blocks in ascending order with a FIXED two-level
// association (16 runs of consecutive blocks, then the 16 run sums) - the same bits on every run, unlike one atomic per block.  The n
// values go to up to four destination arrays laid end to end (d.n[i] values each).  256 threads = 16 values x 16 runs.
struct OrdDst { float* p[4]; int n[4]; };
static __global__ __launch_bounds__(256) void ordered_reduce_kernel(const float* __restrict__ partial, int n_blocks, int n, OrdDst d,
                                                                    int accumulate) {
  __shared__ float red[16][17];
  const int pl = threadIdx.x & 15, c = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + pl;
  const int bpc = (n_blocks + 15) / 16;
  const int b0 = c * bpc, b1 = n_blocks < b0 + bpc ? n_blocks : b0 + bpc;
  float s = 0.f;
  if (t < n) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + u) * n + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += partial[(size_t)b * n + t];
  }
  red[c][pl] = s;
  __syncthreads();
  if (c == 0 && t < n) {
    float a = red[0][pl];
#pragma unroll
    for (int k = 1; k < 16; ++k) a += red[k][pl];
    int q = t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (q < d.n[i]) { float* o = d.p[i] + q; *o = accumulate ? *o + a : a; break; }
      q -= d.n[i];
    }
  }
}
static inline void ordered_reduce_async(const float* partial, int n_blocks, int n, const OrdDst& d, bool accumulate, hipStream_t s) {
  hipLaunchKernelGGL(ordered_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, s, partial, n_blocks, n, d, accumulate ? 1 : 0);
}

}  // namespace swn

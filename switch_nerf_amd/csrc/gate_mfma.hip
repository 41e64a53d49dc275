// Router forward on the matrix pipe (bf16 gate input, gate_dim 256, up to 8 experts): the LayerNorm + fp32 router + softmax + top-1 of
// NeRFMoE.forward / TopKGate (/root/reference/switch_nerf/models/nerf_moe.py:370-372, modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:
// 105-126) - the same contract as gate_fwd_kernel in elementwise.hip, which stays the kernel of the fp32 (parity) mode, of the fp16
// build and of the other shapes.
//
// Why: gate_fwd_kernel spends ~225 VALU instructions per token on 2048 fp32 multiply-adds and their 16-lane reductions (0.68 ms per
// 2M tokens against 0.25 ms for reading the rows).  Here the contraction runs as bf16 MFMAs WITHOUT giving up the fp32 router:
//   * the gate input g is bf16 (the front chain's output), so x is exact in bf16;
//   * the LayerNorm is folded into the weights:  logit[e] = rstd * (sum_k x_k W'[e][k] - mean * c1[e]) + c0[e],  W' = ln_w (.) wg[e],
//     c1[e] = sum_k W'[e][k], c0[e] = sum_k ln_b[k] wg[e][k];
//   * the fp32 W' is split into three bf16 terms hi + mid + lo (24 mantissa bits: exact to the last fp32 bit or two); products of two
//     bf16 numbers are exact in fp32 and the MFMA accumulates in fp32 - the three partial sums add up to the fp32 dot product;
//   * one 32x32x16 MFMA per 16 columns computes, for 32 tokens, the 8 x 3 partial logits AND sum_k x_k (a row of ones): 16 MFMAs per
//     32 tokens.  sum_k x_k^2 comes from the fragments the lane holds anyway (4 packed FMAs per step); variance = E[x^2] - mean^2.
// A wave owns a 32-token tile: rows are copied global -> LDS (1 KiB per `global_load_lds`, swizzled like the chain kernels' tiles), the
// weight fragments live in registers for the whole kernel.  No workgroup barrier in the loop; two 4-wave workgroups per CU.
#include "common.hpp"

#ifndef SWN_HALF_F16
namespace swn {

typedef __attribute__((ext_vector_type(8))) short gm_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float gm_f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t gm_u32x4_t;
typedef __attribute__((ext_vector_type(2))) float gm_f32x2_t;
#define GM_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define GM_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int GM_TILE_B = 32 * 512;          // 32 tokens x 256 bf16
constexpr int GM_CONST0 = 4 * GM_TILE_B;     // per-wave partial sums of c1[8], c0[8] (floats [4][16]) behind the four wave tiles
constexpr int GM_LDS_BYTES = GM_CONST0 + 256;

// byte address of element (row m, column k) of a [32][256] bf16 tile whose 16-byte chunks are XOR-swizzled with row & 15
__device__ __forceinline__ int gm_elem(int m, int k) { return m * 512 + ((((k >> 3) ^ (m & 15))) << 4) + (k & 7) * 2; }

template <bool LN>
__global__ __launch_bounds__(256, 2) void gate_fwd_mfma_kernel(const bf16_t* __restrict__ g, const float* __restrict__ ln_w,
                                                               const float* __restrict__ ln_b, const float* __restrict__ wg, int P, int E,
                                                               float* __restrict__ gates, int32_t* __restrict__ idx, float* __restrict__ gmax,
                                                               float* __restrict__ stats, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5, r15 = lane & 15;
  float* cst = (float*)(smem + GM_CONST0);

  // ---- prologue: the split weight table [32 rows][256] in the first tile: rows 0-7 hi, 8-15 mid, 16-23 lo, 24 ones, rest zero ----
  for (int i = tid; i < GM_TILE_B / 16; i += 256) ((gm_u32x4_t*)smem)[i] = gm_u32x4_t{0u, 0u, 0u, 0u};
  __syncthreads();
  {
    const int k = tid;
    const float lw = LN ? ln_w[k] : 1.f, lb = LN ? ln_b[k] : 0.f;
    for (int e = 0; e < E; ++e) {
      const float wv = wg[(long)e * 256 + k];
      const float wp = wv * lw;
      const bf16_t hi = f32_to_bf16(wp);
      const float r1 = wp - bf16_to_f32(hi);
      const bf16_t mid = f32_to_bf16(r1);
      const bf16_t lo = f32_to_bf16(r1 - bf16_to_f32(mid));
      *(bf16_t*)(smem + gm_elem(e, k)) = hi;
      *(bf16_t*)(smem + gm_elem(8 + e, k)) = mid;
      *(bf16_t*)(smem + gm_elem(16 + e, k)) = lo;
      if (LN) {
        // c1 must be the sum of what the MFMA multiplies: hi + mid + lo (= wp to the last bit or two)
        const float wq = bf16_to_f32(hi) + (bf16_to_f32(mid) + bf16_to_f32(lo));
        float a = wq, b = lb * wv;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (lane == 0) { cst[w * 16 + e] = a; cst[w * 16 + 8 + e] = b; }      // (summed below in a fixed order: every workgroup gets the same bits)
      }
    }
    *(bf16_t*)(smem + gm_elem(24, k)) = (bf16_t)0x3F80;      // 1.0
  }
  __syncthreads();
  gm_u32x4_t wfr[16];
  const uint32_t a_base = (uint32_t)(l31 * 512 + ((lhi ^ r15) << 4));      // this lane's fragment row incl. swizzle seed; ^ (ks << 5)
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) wfr[ks] = *(const gm_u32x4_t*)(smem + (a_base ^ (uint32_t)(ks << 5)));
  float c1[4], c0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c1[j] = c0[j] = 0.f;
    if (LN && 4 * lhi + j < E) {
      c1[j] = (cst[4 * lhi + j] + cst[16 + 4 * lhi + j]) + (cst[32 + 4 * lhi + j] + cst[48 + 4 * lhi + j]);
      c0[j] = (cst[8 + 4 * lhi + j] + cst[24 + 4 * lhi + j]) + (cst[40 + 4 * lhi + j] + cst[56 + 4 * lhi + j]);
    }
  }
  __syncthreads();

  char* tile = smem + w * GM_TILE_B;
  const uint32_t t_base = (uint32_t)(w * GM_TILE_B) + a_base;
  for (int t = blockIdx.x * 4 + w; t < n_tiles; t += gridDim.x * 4) {
    const long tok0 = (long)t * 32;
    // ---- the 32 rows -> the swizzled tile (rows past the end repeat the last token: computed, never written) ----
#pragma unroll 4
    for (int c = 0; c < 16; ++c) {
      const int r = 2 * c + lhi;
      long tok = tok0 + r;
      tok = tok < P ? tok : (long)P - 1;
      const int q = l31 ^ (r & 15);
      __builtin_amdgcn_global_load_lds(GM_GLB((const char*)g + tok * 512 + q * 16), GM_LDS(tile + c * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gm_f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    gm_f32x2_t q2a = {0.f, 0.f}, q2b = {0.f, 0.f};
    gm_u32x4_t xf[2];
    xf[0] = *(const gm_u32x4_t*)(smem + (t_base ^ 0u));
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 1 < 16) xf[(ks + 1) & 1] = *(const gm_u32x4_t*)(smem + (t_base ^ (uint32_t)((ks + 1) << 5)));
      const gm_u32x4_t x = xf[ks & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gm_bf16x8_t, wfr[ks]), __builtin_bit_cast(gm_bf16x8_t, x), acc, 0, 0, 0);
      if (LN) {
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const gm_f32x2_t u0 = {__uint_as_float(x[i] << 16), __uint_as_float(x[i] & 0xFFFF0000u)};
          const gm_f32x2_t u1 = {__uint_as_float(x[i + 1] << 16), __uint_as_float(x[i + 1] & 0xFFFF0000u)};
          q2a += u0 * u0;
          q2b += u1 * u1;
        }
      }
    }
    // ---- this lane: token l31, experts 4 lhi + j (accumulator rows 8 g4 + 4 lhi + j: g4 = 0 hi, 1 mid, 2 lo, 3: row 24 = sum x) ----
    float logit[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) logit[j] = (acc[j] + acc[4 + j]) + acc[8 + j];
    float mean = 0.f, rstd = 1.f;
    if (LN) {
      const float sx = __shfl(acc[12], l31);                     // row 24 lives in the lower half-wave
      float q2 = (q2a[0] + q2a[1]) + (q2b[0] + q2b[1]);
      q2 += __shfl_xor(q2, 32);
      mean = sx * (1.f / 256.f);
      const float var = fmaxf(q2 * (1.f / 256.f) - mean * mean, 0.f);
      rstd = 1.f / sqrtf(var + 1e-5f);                           // torch.nn.LayerNorm, eps 1e-5 (models/nerf_moe.py:301-302)
#pragma unroll
      for (int j = 0; j < 4; ++j) logit[j] = rstd * (logit[j] - mean * c1[j]) + c0[j];
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * lhi + j < E) mx = fmaxf(mx, logit[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float pr[4], den = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pr[j] = (4 * lhi + j < E) ? expf(logit[j] - mx) : 0.f;
      den += pr[j];
    }
    // (the 8 terms are summed in expert order like gate_fwd_kernel: lower half first)
    const float den_lo = __shfl(den, l31), den_hi = __shfl(den, l31 + 32);
    den = den_lo + den_hi;
    int best = 0;
    float bv = -1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pr[j] = pr[j] / den;
      if (4 * lhi + j < E && pr[j] > bv) { bv = pr[j]; best = 4 * lhi + j; }      // first maximum
    }
    const float bv_hi = __shfl(bv, l31 + 32);
    const int best_hi = __shfl(best, l31 + 32);
    if (bv_hi > bv) { bv = bv_hi; best = best_hi; }              // (lower half only: strictly greater = first maximum over all 8)
    const long tok = tok0 + l31;
    if (tok < P) {
      if (E == 8) {
        *(float4*)(gates + tok * 8 + 4 * lhi) = make_float4(pr[0], pr[1], pr[2], pr[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * lhi + j < E) gates[tok * E + 4 * lhi + j] = pr[j];
      }
      if (lhi == 0) {
        idx[tok] = best;
        gmax[tok] = bv;
        if (LN) *(float2*)(stats + tok * 2) = make_float2(mean, rstd);
      }
    }
  }
}

int gate_fwd_mfma_launch(const void* g, const float* ln_w, const float* ln_b, const float* wg, int n_tokens, int n_experts, float* gates,
                         int32_t* idx, float* gmax, float* stats, void* stream) {
  const int n_tiles = cdiv(n_tokens, 32);
  int blocks = cdiv(n_tiles, 4);
  if (blocks > 512) blocks = 512;
  const void* fn = ln_w ? (const void*)gate_fwd_mfma_kernel<true> : (const void*)gate_fwd_mfma_kernel<false>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const bf16_t* gp = (const bf16_t*)g;
  void* kargs[] = {(void*)&gp, (void*)&ln_w, (void*)&ln_b, (void*)&wg, (void*)&n_tokens, (void*)&n_experts, (void*)&gates, (void*)&idx,
                   (void*)&gmax, (void*)&stats, (void*)&n_tiles};
  e = hipLaunchKernel(fn, dim3(blocks), dim3(256), kargs, GM_LDS_BYTES, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_gate_fwd (mfma) launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace swn
#endif

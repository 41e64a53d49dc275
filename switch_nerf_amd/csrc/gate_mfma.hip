// Router forward on the matrix pipe (16-bit gate input, gate_dim 256, up to 8 experts): the LayerNorm + fp32 router + softmax + top-1 of
// NeRFMoE.forward / TopKGate (/root/reference/switch_nerf/models/nerf_moe.py:370-372, modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:
// 105-126) - the same contract as gate_fwd_kernel in elementwise.hip, which stays the kernel of the fp32 (parity) mode and of the
// other shapes.
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

namespace swn {

typedef __attribute__((ext_vector_type(8))) short gm_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float gm_f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t gm_u32x4_t;
typedef __attribute__((ext_vector_type(2))) float gm_f32x2_t;
// The 16-bit type of the build (bf16, or IEEE half in the -DSWN_HALF_F16 build): element access through common.hpp's bf16_to_f32 /
// f32_to_bf16, which are the fp16 conversions there.  fp16 has 5 exponent bits: the split terms of a weight (w ~ 0.05: remainders
// ~ 2e-5, ~ 1e-8) and small dlogits would land in its subnormals, so the fp16 build scales them by exact powers of two before the
// split and scales the MFMA results back (GM_SW, GM_SD; 1 for bf16).
#ifdef SWN_HALF_F16
#define GM_MFMA16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(swn_mfma16_t, a), __builtin_bit_cast(swn_mfma16_t, b), c, 0, 0, 0)
constexpr float GM_SW = 4096.f, GM_SD = 1024.f;
#else
#define GM_MFMA16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(swn_mfma16_t, a), __builtin_bit_cast(swn_mfma16_t, b), c, 0, 0, 0)
constexpr float GM_SW = 1.f, GM_SD = 1.f;
#endif
__device__ __forceinline__ float gm_lo(uint32_t v) { return bf16_to_f32((bf16_t)(v & 0xFFFFu)); }
__device__ __forceinline__ float gm_hi(uint32_t v) { return bf16_to_f32((bf16_t)(v >> 16)); }
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
      const float wp = wv * lw * GM_SW;
      const bf16_t hi = f32_to_bf16(wp);
      const float r1 = wp - bf16_to_f32(hi);
      const bf16_t mid = f32_to_bf16(r1);
      const bf16_t lo = f32_to_bf16(r1 - bf16_to_f32(mid));
      *(bf16_t*)(smem + gm_elem(e, k)) = hi;
      *(bf16_t*)(smem + gm_elem(8 + e, k)) = mid;
      *(bf16_t*)(smem + gm_elem(16 + e, k)) = lo;
      if (LN) {
        // c1 must be the sum of what the MFMA multiplies: hi + mid + lo (= wp to the last bit or two)
        const float wq = (bf16_to_f32(hi) + (bf16_to_f32(mid) + bf16_to_f32(lo))) * (1.f / GM_SW);
        float a = wq, b = lb * wv;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (lane == 0) { cst[w * 16 + e] = a; cst[w * 16 + 8 + e] = b; }      // (summed below in a fixed order: every workgroup gets the same bits)
      }
    }
    *(bf16_t*)(smem + gm_elem(24, k)) = (bf16_t)(SWN_HALF_ONE_X2 & 0xFFFFu);      // 1.0
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
      acc = SWN_MFMA_32x32x16(wfr[ks], x, acc);
      if (LN) {
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const gm_f32x2_t u0 = {gm_lo(x[i]), gm_hi(x[i])};
          const gm_f32x2_t u1 = {gm_lo(x[i + 1]), gm_hi(x[i + 1])};
          q2a += u0 * u0;
          q2b += u1 * u1;
        }
      }
    }
    // ---- this lane: token l31, experts 4 lhi + j (accumulator rows 8 g4 + 4 lhi + j: g4 = 0 hi, 1 mid, 2 lo, 3: row 24 = sum x) ----
    float logit[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) logit[j] = ((acc[j] + acc[4 + j]) + acc[8 + j]) * (1.f / GM_SW);
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Router backward, data path: dlogits (softmax + l_aux, as gate_bwd_kernel) -> d(xhat w) = dlogits @ W' -> LayerNorm backward -> dg.
// Same tiling as the forward kernel.  The contraction over the 8 experts is ONE K step of a 32x32x16 MFMA per 32 output columns: the
// K slots 0-7 carry the bf16 head of dlogits, 8-15 its bf16 remainder, against the bf16 head of W' (and a second MFMA against the
// remainder of W'): relative error 2^-17, far below the bf16 rounding of dg.  The row sums of the LayerNorm backward need every
// column of the row before any column can be finished: two passes over the 8 column tiles (the MFMAs are recomputed - 32 per tile -
// instead of holding 128 values per lane); the result replaces x in the LDS tile and leaves it in coalesced 1 KiB pieces.
// The parameter gradients are not formed here (gate_dwg_kernel / gate_dwg_finalize_kernel from the dlogits written out).
template <bool LN>
__global__ __launch_bounds__(256, 2) void gate_bwd_mfma_kernel(const bf16_t* __restrict__ g, const float* __restrict__ ln_w,
                                                               const float* __restrict__ wg, const float* __restrict__ gates,
                                                               const int32_t* __restrict__ idx, const float* __restrict__ d_gmax,
                                                               const float* __restrict__ stats, const int32_t* __restrict__ counts,
                                                               const float* __restrict__ laux_coef, int seg_tokens, int P, int E,
                                                               bf16_t* __restrict__ dg, float* __restrict__ dlogits, int n_tiles,
                                                               float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5, r15 = lane & 15;

  // ---- prologue: W' = ln_w (.) wg split into bf16 head / remainder, column-major: T[column][8 experts] (16 bytes per column) ----
  char* wtab = smem + GM_CONST0 + 256 + 4096;        // behind the tiles and the dlr^T staging buffers: 2 x 4 KiB
  {
    const int k = tid;
    const float lw = LN ? ln_w[k] : 1.f;
    uint32_t hi[4] = {0u, 0u, 0u, 0u}, mid[4] = {0u, 0u, 0u, 0u};
    for (int e = 0; e < E; ++e) {
      const float wp = wg[(long)e * 256 + k] * lw * GM_SW;
      const bf16_t h = f32_to_bf16(wp);
      const bf16_t m = f32_to_bf16(wp - bf16_to_f32(h));
      hi[e >> 1] |= (uint32_t)h << (16 * (e & 1));
      mid[e >> 1] |= (uint32_t)m << (16 * (e & 1));
    }
    *(gm_u32x4_t*)(wtab + k * 16) = gm_u32x4_t{hi[0], hi[1], hi[2], hi[3]};
    *(gm_u32x4_t*)(wtab + 4096 + k * 16) = gm_u32x4_t{mid[0], mid[1], mid[2], mid[3]};
  }
  __syncthreads();
  // A fragments of column tile nt: row = column 32 nt + l31 of W', K slots = the 8 experts (both half-waves alike) - read from the table
  // per use (64 registers otherwise)
  const char* wrow = wtab + l31 * 16;

  char* tile = smem + w * GM_TILE_B;
  // this lane's 8-byte element group (row l31, columns 32 nt + 8 g4 + 4 lhi .. + 3): chunk 4 nt + g4, swizzled with the row
  const uint32_t e_base = (uint32_t)(w * GM_TILE_B + l31 * 512 + (r15 << 4) + 8 * lhi);
  // ---- the parameter-gradient side (see gate_dwg_kernel): M[e][k] = sum_tok dlogits[tok][e] xhat[tok][k] = sum_tok dlr[tok][e] x[tok][k]
  //      - C[e], dlr = dlogits * rstd, C[e] = sum_tok dlr[tok][e] mean[tok]: a GEMM over the TOKENS with the exact bf16 rows as one operand.
  //      v_mfma_f32_16x16x32_bf16 per 16 columns: A = dlr^T (rows: 8 experts x {bf16 head, remainder}; K = the tile's 32 tokens, staged
  //      through a 1 KiB LDS buffer), B = x^T read straight from the row-major tile with the transposing LDS read (ds_read_b64_tr_b16:
  //      the 16 lanes of a group address a [4 tokens][16 columns] block, 8 bytes each, and receive one column's 4 tokens).
  typedef __attribute__((ext_vector_type(4))) float f32x4_t_;
  f32x4_t_ macc[16];
#pragma unroll
  for (int nt = 0; nt < 16; ++nt) macc[nt] = f32x4_t_{0.f, 0.f, 0.f, 0.f};
  float cacc[4] = {0.f, 0.f, 0.f, 0.f}, dlacc[4] = {0.f, 0.f, 0.f, 0.f};
  char* dstage = smem + GM_CONST0 + 256 + w * 1024;            // [16 rows][32 tokens] bf16
  const int kg = lane >> 4, n16 = lane & 15;
  // transposing read of (tokens 8 kg + 4 h + (n16 >> 2), columns 16 nt + 4 (n16 & 3) .. + 3): row byte address + swizzled chunk
  uint32_t tr_row[2], tr_sw[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 8 * kg + 4 * h + (n16 >> 2);
    tr_row[h] = (uint32_t)(w * GM_TILE_B + r * 512 + 8 * (n16 & 1));
    tr_sw[h] = (uint32_t)(r & 15);
  }
  for (int t = blockIdx.x * 4 + w; t < n_tiles; t += gridDim.x * 4) {
    const long tok0 = (long)t * 32;
#pragma unroll 4
    for (int c = 0; c < 16; ++c) {
      const int r = 2 * c + lhi;
      long tk = tok0 + r;
      tk = tk < P ? tk : (long)P - 1;
      const int q = l31 ^ (r & 15);
      __builtin_amdgcn_global_load_lds(GM_GLB((const char*)g + tk * 512 + q * 16), GM_LDS(tile + c * 1024), 16, 0, 0);
    }
    // ---- per token (lane: token l31, experts 4 lhi + j): dlogits ----
    long tok = tok0 + l31;
    const bool live = tok < P;
    tok = live ? tok : (long)P - 1;
    const int seg = (int)(tok / seg_tokens);
    const int my = idx[tok];
    const float coef = laux_coef ? laux_coef[seg] : 0.f;
    const float dgm = d_gmax ? d_gmax[tok] : 0.f;
    float mean = 0.f, rstd = 1.f;
    if (LN) { const float2 st = *(const float2*)(stats + tok * 2); mean = st.x; rstd = st.y; }
    float pr[4], dp[4], dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = 4 * lhi + j;
      pr[j] = e < E ? gates[tok * E + e] : 0.f;
      dp[j] = e < E ? coef * (float)counts[seg * E + e] + ((e == my) ? dgm : 0.f) : 0.f;
      dot += pr[j] * dp[j];
    }
    {
      const float d_lo = __shfl(dot, l31), d_hi = __shfl(dot, l31 + 32);       // experts 0-3 first, like gate_bwd_kernel's loop
      dot = d_lo + d_hi;
    }
    float dl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dl[j] = pr[j] * (dp[j] - dot);     // softmax backward
      if (live && 4 * lhi + j < E) dlogits[tok * E + 4 * lhi + j] = dl[j];
    }
    // the B fragment: all 8 dlogits of the token; the lower half-wave carries their bf16 heads, the upper one the remainders
    float da[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float o = __shfl_xor(dl[j], 32);
      da[j] = lhi ? o : dl[j];
      da[4 + j] = lhi ? dl[j] : o;
    }
    gm_u32x4_t bfr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t v = 0u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float x = da[2 * i + h] * GM_SD;
        const bf16_t hd = f32_to_bf16(x);
        const bf16_t val = lhi ? f32_to_bf16(x - bf16_to_f32(hd)) : hd;
        v |= (uint32_t)val << (16 * h);
      }
      bfr[i] = v;
    }
    if (partial) {          // dlr^T for the token GEMM: rows e (head) / 8 + e (remainder), column = this lane's token
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = live ? da[e] * rstd * GM_SD : 0.f;
        const bf16_t hd = f32_to_bf16(x);
        const bf16_t val = lhi ? f32_to_bf16(x - bf16_to_f32(hd)) : hd;
        *(bf16_t*)(dstage + (8 * lhi + e) * 64 + l31 * 2) = val;
      }
      if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { cacc[j] += dl[j] * rstd * mean; dlacc[j] += dl[j]; }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tile (and every load above) has landed
    if (partial) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const gm_u32x4_t afr = *(const gm_u32x4_t*)(dstage + n16 * 64 + kg * 16);      // row n16, tokens 8 kg .. + 7
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) {
        // columns 16 nt + 4 (n16 & 3) ..: chunk 2 nt + ((n16 & 3) >> 1)
        uint2 b0, b1;
        const uint32_t ch = (uint32_t)(2 * nt + ((n16 & 3) >> 1));
        const uint32_t a0 = tr_row[0] + ((ch ^ tr_sw[0]) << 4), a1 = tr_row[1] + ((ch ^ tr_sw[1]) << 4);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0) : "v"(a0) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1) : "v"(a1) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const gm_u32x4_t bx = {b0.x, b0.y, b1.x, b1.y};
        macc[nt] = GM_MFMA16x16x32(afr, bx, macc[nt]);
      }
    }

    auto dxh_tile = [&](int nt) -> gm_f32x16_t {
      gm_f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const gm_u32x4_t wh = *(const gm_u32x4_t*)(wrow + nt * 512), wm = *(const gm_u32x4_t*)(wrow + 4096 + nt * 512);
      acc = SWN_MFMA_32x32x16(wh, bfr, acc);
      acc = SWN_MFMA_32x32x16(wm, bfr, acc);
#ifdef SWN_HALF_F16
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= 1.f / (GM_SW * GM_SD);
#endif
      return acc;
    };
    const float mr = mean * rstd;
    float s1 = 0.f, s2 = 0.f;
    if (LN) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const gm_f32x16_t acc = dxh_tile(nt);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const uint2 xv = *(const uint2*)(smem + (e_base ^ (uint32_t)((4 * nt + g4) << 4)));
          const float x0 = gm_lo(xv.x), x1 = gm_hi(xv.x);
          const float x2 = gm_lo(xv.y), x3 = gm_hi(xv.y);
          const float d0 = acc[4 * g4], d1 = acc[4 * g4 + 1], d2 = acc[4 * g4 + 2], d3 = acc[4 * g4 + 3];
          s1 += (d0 + d1) + (d2 + d3);
          s2 += d0 * (x0 * rstd - mr) + d1 * (x1 * rstd - mr) + d2 * (x2 * rstd - mr) + d3 * (x3 * rstd - mr);
        }
      }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      s1 *= (1.f / 256.f);
      s2 *= (1.f / 256.f);
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const gm_f32x16_t acc = dxh_tile(nt);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const uint32_t a = e_base ^ (uint32_t)((4 * nt + g4) << 4);
        float o[4];
        if (LN) {
          const uint2 xv = *(const uint2*)(smem + a);
          const float xh0 = gm_lo(xv.x) * rstd - mr, xh1 = gm_hi(xv.x) * rstd - mr;
          const float xh2 = gm_lo(xv.y) * rstd - mr, xh3 = gm_hi(xv.y) * rstd - mr;
          o[0] = rstd * (acc[4 * g4] - s1 - xh0 * s2);
          o[1] = rstd * (acc[4 * g4 + 1] - s1 - xh1 * s2);
          o[2] = rstd * (acc[4 * g4 + 2] - s1 - xh2 * s2);
          o[3] = rstd * (acc[4 * g4 + 3] - s1 - xh3 * s2);
        } else {
          o[0] = acc[4 * g4]; o[1] = acc[4 * g4 + 1]; o[2] = acc[4 * g4 + 2]; o[3] = acc[4 * g4 + 3];
        }
        *(uint2*)(smem + a) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
      }
    }
    // ---- dg rows out of the tile: 1 KiB pieces ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
    for (int c = 0; c < 16; ++c) {
      const int r = 2 * c + lhi;
      const long tk = tok0 + r;
      const int q = l31 ^ (r & 15);
      const gm_u32x4_t v = *(const gm_u32x4_t*)(tile + c * 1024 + lane * 16);
      if (tk < P) *(gm_u32x4_t*)((char*)dg + tk * 512 + q * 16) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the tile is free for the next copy
  }
  if (partial) {
    // ---- block partial in gate_dwg_kernel's format: part[e * 256 + k] = M (this block's tokens), part[E * 256 + e] = DL ----
    __syncthreads();
    float* red = (float*)smem;                        // [16][256] + c[8] + dl[8]
    for (int i = tid; i < 16 * 256 + 16; i += 256) red[i] = 0.f;
    __syncthreads();
    float cs[4], dls[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = cacc[j], b = dlacc[j];
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      cs[j] = a; dls[j] = b;
    }
    // the four waves add their sums one after the other: a fixed order, the same bits on every run (a wave's lanes own distinct words)
    for (int wq = 0; wq < 4; ++wq) {
      if (w == wq) {
#pragma unroll
        for (int nt = 0; nt < 16; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(4 * kg + r) * 256 + 16 * nt + n16] += macc[nt][r];
        if (l31 == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { red[16 * 256 + 4 * lhi + j] += cs[j]; red[16 * 256 + 8 + 4 * lhi + j] += dls[j]; }
        }
      }
      __syncthreads();
    }
    float* part = partial + (size_t)blockIdx.x * (E * 256 + E);
    for (int i = tid; i < E * 256; i += 256) {
      const int e = i >> 8, k = i & 255;
      part[i] = (red[e * 256 + k] + red[(8 + e) * 256 + k]) * (1.f / GM_SD) - (LN ? red[16 * 256 + e] : 0.f);
    }
    if (tid < E) part[E * 256 + tid] = red[16 * 256 + 8 + tid];
  }
}

int gate_bwd_mfma_blocks(int n_tokens) {
  int blocks = cdiv(cdiv(n_tokens, 32), 4);
  return blocks > 512 ? 512 : blocks;
}

// partial != NULL: the kernel also leaves its block's share of the parameter-gradient sums there ([blocks][E * 256 + E], the format
// of gate_dwg_kernel): no separate pass over g for them
int gate_bwd_mfma_launch(const void* g, const float* ln_w, const float* wg, const float* gates, const int32_t* idx, const float* d_gmax,
                         const float* stats, const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int n_experts,
                         void* dg, float* dlogits, float* partial, void* stream) {
  const int n_tiles = cdiv(n_tokens, 32);
  const int blocks = gate_bwd_mfma_blocks(n_tokens);
  const void* fn = ln_w ? (const void*)gate_bwd_mfma_kernel<true> : (const void*)gate_bwd_mfma_kernel<false>;
  constexpr int LDS_B = GM_LDS_BYTES + 4096 + 8192;        // + dlr^T staging (4 x 1 KiB) + the W' table
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const bf16_t* gp = (const bf16_t*)g;
  bf16_t* dgp = (bf16_t*)dg;
  void* kargs[] = {(void*)&gp, (void*)&ln_w, (void*)&wg, (void*)&gates, (void*)&idx, (void*)&d_gmax, (void*)&stats, (void*)&counts,
                   (void*)&laux_coef, (void*)&seg_tokens, (void*)&n_tokens, (void*)&n_experts, (void*)&dgp, (void*)&dlogits, (void*)&n_tiles,
                   (void*)&partial};
  e = hipLaunchKernel(fn, dim3(blocks), dim3(256), kargs, LDS_B, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_gate_bwd (mfma) launch: %s", hipGetErrorString(e));
  return 0;
}


// =================================================================================================================================
// The 512-feature router (up to 16 experts: the Mission Bay recipe, configs/switch_nerf/mission_bay.yaml) on the matrix pipe.
//
// gate_fwd_kernel / gate_bwd_kernel<., 512, 16> spend 16 x (32 FMAs + a 16-lane reduction) per row on the VALU with their weights coming
// from LDS: 0.8 ms per 852 k rows each, a fifth of the rate the row reads allow.  Same arithmetic as the 256-feature kernels above (the
// LayerNorm folded into W' = ln_w (.) wg, W' split into three 16-bit terms, fp32 accumulation), different data movement: 49 table rows
// x 512 columns do not fit a wave's registers and four 32 KiB row tiles do not fit the LDS beside them, so
//   * the split weight table lives in LDS in FRAGMENT-MAJOR order (64 KiB: two 32-row A tiles x 32 K steps x 64 lanes x 16 bytes -
//     tile 0: experts 0-7 hi / mid / lo + the row of ones, tile 1: experts 8-15), two ds_read_b128 per K step;
//   * the rows never pass through LDS: a lane owns HALF a row (token l31, columns [256 lhi, +256)) and reads it as 32 consecutive
//     16-byte pieces straight into the B operand - the contraction order is free, so K step ks multiplies columns 8 ks .. + 7 of both
//     halves (the table is laid out to match); 8 K steps (8 KiB per wave) are in flight ahead of the MFMAs, across tile boundaries;
//   * forward: 2 MFMAs per K step, 64 per 32 tokens.  Backward: dxh = dlogits @ W' is ONE K step per 32 columns (K slots = the 16
//     experts; dlogits split into head + remainder against W' hi, head against W' mid: 3 MFMAs); the table rows are permuted so that a
//     lane's 16 accumulator values are 16 CONSECUTIVE columns of its token - x comes from and dg goes to global memory in 32-byte runs,
//     the whole tile's x (32 KiB per wave) stays in registers between the two passes of the LayerNorm backward.
// The parameter gradients stay with gate_dwg_kernel (from the dlogits written out).
// =================================================================================================================================
constexpr int GW_G = 512;                        // features
constexpr int GW_KS = GW_G / 16;                 // K steps
constexpr int GW_TAB_B = 2 * GW_KS * 64 * 16;    // 64 KiB
constexpr int GW_FWD_LDS = GW_TAB_B + 256;       // + c1[16], c0[16]
constexpr int GW_BWD_LDS = 2 * 16 * 64 * 16;     // W' hi / mid: [16 column tiles][64 lanes] x 16 bytes each

__device__ __forceinline__ void gw_split3(float wp, bf16_t& hi, bf16_t& mid, bf16_t& lo) {
  hi = f32_to_bf16(wp);
  const float r1 = wp - bf16_to_f32(hi);
  mid = f32_to_bf16(r1);
  lo = f32_to_bf16(r1 - bf16_to_f32(mid));
}

template <bool LN>
__global__ __launch_bounds__(256, 2) void gate_fwd_wide_kernel(const bf16_t* __restrict__ g, const float* __restrict__ ln_w,
                                                               const float* __restrict__ ln_b, const float* __restrict__ wg, int P, int E,
                                                               float* __restrict__ gates, int32_t* __restrict__ idx, float* __restrict__ gmax,
                                                               float* __restrict__ stats, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  float* cst = (float*)(smem + GW_TAB_B);

  // ---- prologue: the fragment-major split table.  Fragment (tile tl, K step ks, lane ln): table row m = ln & 31 (rows 0-7 hi, 8-15 mid,
  //      16-23 lo of experts 8 tl + (m & 7); row 24 of tile 0 = ones), columns 256 (ln >> 5) + 8 ks .. + 7 ----
  for (int f = tid; f < 2 * GW_KS * 64; f += 256) {
    const int ln = f & 63, ks = (f >> 6) & (GW_KS - 1), tl = f >> 11;
    const int m = ln & 31, h = ln >> 5, term = m >> 3, e = (m & 7) + 8 * tl;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (term < 3 && e < E) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = 256 * h + 8 * ks + i;
        bf16_t hi, mid, lo;
        gw_split3(wg[(long)e * GW_G + c] * (LN ? ln_w[c] : 1.f) * GM_SW, hi, mid, lo);
        const bf16_t val = term == 0 ? hi : (term == 1 ? mid : lo);
        v[i >> 1] |= (uint32_t)val << (16 * (i & 1));
      }
    } else if (m == 24 && tl == 0) {
      v[0] = v[1] = v[2] = v[3] = SWN_HALF_ONE_X2;
    }
    *(gm_u32x4_t*)(smem + (size_t)f * 16) = gm_u32x4_t{v[0], v[1], v[2], v[3]};
  }
  if (LN) {      // c1[e] = sum_k (what the MFMA multiplies), c0[e] = sum_k ln_b[k] wg[e][k]: 16 lanes per expert, a fixed order
    const int e = tid >> 4, jj = tid & 15;
    float a = 0.f, b = 0.f;
    if (e < E) {
      for (int c = jj; c < GW_G; c += 16) {
        const float wv = wg[(long)e * GW_G + c];
        bf16_t hi, mid, lo;
        gw_split3(wv * ln_w[c] * GM_SW, hi, mid, lo);
        a += (bf16_to_f32(hi) + (bf16_to_f32(mid) + bf16_to_f32(lo))) * (1.f / GM_SW);
        b += ln_b[c] * wv;
      }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (jj == 0) { cst[e] = a; cst[16 + e] = b; }
  }
  __syncthreads();
  float c1[2][4], c0[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = 8 * t + 4 * lhi + j;
      c1[t][j] = (LN && e < E) ? cst[e] : 0.f;
      c0[t][j] = (LN && e < E) ? cst[16 + e] : 0.f;
    }

  const uint32_t a_off = (uint32_t)lane * 16;
  const int stride = gridDim.x * 4;
  auto row_ptr = [&](int t) -> const char* {
    long tok = (long)t * 32 + l31;
    tok = tok < P ? tok : (long)P - 1;                 // (rows past the end repeat the last token: computed, never written)
    return (const char*)g + tok * (GW_G * 2) + lhi * GW_G;
  };
  gm_u32x4_t xb[2][8];
  int t = blockIdx.x * 4 + w;
  if (t < n_tiles) {
    const char* p0 = row_ptr(t);
#pragma unroll
    for (int s = 0; s < 8; ++s) xb[0][s] = *(const gm_u32x4_t*)(p0 + 16 * s);
  }
  for (; t < n_tiles; t += stride) {
    const char* p = row_ptr(t);
    const bool more = t + stride < n_tiles;
    const char* pn = row_ptr(more ? t + stride : t);
    gm_f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    gm_f32x2_t q2a = {0.f, 0.f}, q2b = {0.f, 0.f};
    // (one K step per scheduling region: left alone the scheduler hoists the 64 table reads of the unrolled loop - 256 registers, 238 spilled;
    //  the table fragments of step ks + 1 are requested in front of the MFMAs of step ks)
    gm_u32x4_t af[2][2];
    af[0][0] = *(const gm_u32x4_t*)(smem + a_off);
    af[0][1] = *(const gm_u32x4_t*)(smem + a_off + (uint32_t)GW_KS * 1024u);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < 3) {
#pragma unroll
        for (int s = 0; s < 8; ++s) xb[(c + 1) & 1][s] = *(const gm_u32x4_t*)(p + 16 * (8 * (c + 1) + s));
      } else if (more) {
#pragma unroll
        for (int s = 0; s < 8; ++s) xb[0][s] = *(const gm_u32x4_t*)(pn + 16 * s);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int ks = 8 * c + s;
        const gm_u32x4_t x = xb[c & 1][s];
        if (ks + 1 < GW_KS) {
          af[(ks + 1) & 1][0] = *(const gm_u32x4_t*)(smem + a_off + (uint32_t)(ks + 1) * 1024u);
          af[(ks + 1) & 1][1] = *(const gm_u32x4_t*)(smem + a_off + (uint32_t)(GW_KS + ks + 1) * 1024u);
        }
        acc0 = SWN_MFMA_32x32x16(af[ks & 1][0], x, acc0);
        acc1 = SWN_MFMA_32x32x16(af[ks & 1][1], x, acc1);
        if (LN) {
#pragma unroll
          for (int i = 0; i < 4; i += 2) {
            const gm_f32x2_t u0 = {gm_lo(x[i]), gm_hi(x[i])};
            const gm_f32x2_t u1 = {gm_lo(x[i + 1]), gm_hi(x[i + 1])};
            q2a += u0 * u0;
            q2b += u1 * u1;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- this lane: token l31, experts 4 lhi + j (tile 0) and 8 + 4 lhi + j (tile 1); accumulator rows 8 g4 + 4 lhi + j ----
    float logit[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      logit[0][j] = ((acc0[j] + acc0[4 + j]) + acc0[8 + j]) * (1.f / GM_SW);
      logit[1][j] = ((acc1[j] + acc1[4 + j]) + acc1[8 + j]) * (1.f / GM_SW);
    }
    float mean = 0.f, rstd = 1.f;
    if (LN) {
      const float sx = __shfl(acc0[12], l31);                    // table row 24 (the ones) lives in the lower half-wave
      float q2 = (q2a[0] + q2a[1]) + (q2b[0] + q2b[1]);
      q2 += __shfl_xor(q2, 32);
      mean = sx * (1.f / GW_G);
      const float var = fmaxf(q2 * (1.f / GW_G) - mean * mean, 0.f);
      rstd = 1.f / sqrtf(var + 1e-5f);                           // torch.nn.LayerNorm, eps 1e-5 (models/nerf_moe.py:301-302)
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int j = 0; j < 4; ++j) logit[tl][j] = rstd * (logit[tl][j] - mean * c1[tl][j]) + c0[tl][j];
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (8 * tl + 4 * lhi + j < E) mx = fmaxf(mx, logit[tl][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float pr[2][4], dsum[2] = {0.f, 0.f};
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pr[tl][j] = (8 * tl + 4 * lhi + j < E) ? expf(logit[tl][j] - mx) : 0.f;
        dsum[tl] += pr[tl][j];
      }
    // (summed in expert order, four at a time: 0-3, 4-7, 8-11, 12-15)
    const float den = ((__shfl(dsum[0], l31) + __shfl(dsum[0], l31 + 32)) + __shfl(dsum[1], l31)) + __shfl(dsum[1], l31 + 32);
    float bv[2] = {-1.f, -1.f};
    int be[2] = {0, 0};
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pr[tl][j] = pr[tl][j] / den;
        if (8 * tl + 4 * lhi + j < E && pr[tl][j] > bv[tl]) { bv[tl] = pr[tl][j]; be[tl] = 8 * tl + 4 * lhi + j; }      // first maximum
      }
    // first maximum over all 16 in expert order: (lower half, tile 0), (upper, 0), (lower, 1), (upper, 1)
    float best_v = __shfl(bv[0], l31);
    int best_e = __shfl(be[0], l31);
    {
      const float v1 = __shfl(bv[0], l31 + 32); const int e1 = __shfl(be[0], l31 + 32);
      if (v1 > best_v) { best_v = v1; best_e = e1; }
      const float v2 = __shfl(bv[1], l31); const int e2 = __shfl(be[1], l31);
      if (v2 > best_v) { best_v = v2; best_e = e2; }
      const float v3 = __shfl(bv[1], l31 + 32); const int e3 = __shfl(be[1], l31 + 32);
      if (v3 > best_v) { best_v = v3; best_e = e3; }
    }
    const long tok = (long)t * 32 + l31;
    if (tok < P) {
      if (E == 16) {
        *(float4*)(gates + tok * 16 + 4 * lhi) = make_float4(pr[0][0], pr[0][1], pr[0][2], pr[0][3]);
        *(float4*)(gates + tok * 16 + 8 + 4 * lhi) = make_float4(pr[1][0], pr[1][1], pr[1][2], pr[1][3]);
      } else {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (8 * tl + 4 * lhi + j < E) gates[tok * E + 8 * tl + 4 * lhi + j] = pr[tl][j];
      }
      if (lhi == 0) {
        idx[tok] = best_e;
        gmax[tok] = best_v;
        if (LN) *(float2*)(stats + tok * 2) = make_float2(mean, rstd);
      }
    }
  }
}

int gate_fwd_wide_launch(const void* g, const float* ln_w, const float* ln_b, const float* wg, int n_tokens, int n_experts, float* gates,
                         int32_t* idx, float* gmax, float* stats, void* stream) {
  const int n_tiles = cdiv(n_tokens, 32);
  int blocks = cdiv(n_tiles, 4);
  if (blocks > 512) blocks = 512;
  const void* fn = ln_w ? (const void*)gate_fwd_wide_kernel<true> : (const void*)gate_fwd_wide_kernel<false>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GW_FWD_LDS);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const bf16_t* gp = (const bf16_t*)g;
  void* kargs[] = {(void*)&gp, (void*)&ln_w, (void*)&ln_b, (void*)&wg, (void*)&n_tokens, (void*)&n_experts, (void*)&gates, (void*)&idx,
                   (void*)&gmax, (void*)&stats, (void*)&n_tiles};
  e = hipLaunchKernel(fn, dim3(blocks), dim3(256), kargs, GW_FWD_LDS, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_gate_fwd (wide mfma) launch: %s", hipGetErrorString(e));
  return 0;
}

// Router backward, data path, 512 features (see the header of this section): dlogits -> dxh = dlogits @ W' -> LayerNorm backward -> dg.
// Lane (l31, lhi): token l31; as the B operand it carries the dlogits of experts 8 lhi .. + 7, as the owner of accumulator values it
// holds columns 32 nt + 16 lhi .. + 15 of column tile nt (= the 32 bytes of its row it loads and stores).
// The row sums of the LayerNorm backward without a pass over the columns:
//   s1 = mean_k dxh_k        = (1 / G) sum_e dl_e c1_e,                       c1_e = sum_k W'[e][k]
//   s2 = mean_k dxh_k xhat_k = (1 / G) sum_e dl_e (rstd u_e - mean rstd c1_e),  u_e = sum_k W'[e][k] x_k
// (dxh_k = sum_e dl_e W'[e][k], xhat = x rstd - mean rstd) - u is the forward contraction again (W' as hi + mid, 32 MFMAs per 32 tokens,
// no VALU work), and its B fragments ARE the registers the output pass reads x from: K step 2 nt + hh multiplies columns
// 32 nt + 16 lhi + 8 hh .. + 7.
constexpr int GW_BWD_U0 = 2 * 16 * 64 * 16;              // the u table behind the two dxh tables: [32 K steps][64 lanes] x 16 bytes
constexpr int GW_BWD_C0 = GW_BWD_U0 + GW_KS * 64 * 16;   // c1[16]
constexpr int GW_BWD_LDS_ALL = GW_BWD_C0 + 64;

template <bool LN>
__global__ __launch_bounds__(256, 2) void gate_bwd_wide_kernel(const bf16_t* __restrict__ g, const float* __restrict__ ln_w,
                                                               const float* __restrict__ wg, const float* __restrict__ gates,
                                                               const int32_t* __restrict__ idx, const float* __restrict__ d_gmax,
                                                               const float* __restrict__ stats, const int32_t* __restrict__ counts,
                                                               const float* __restrict__ laux_coef, int seg_tokens, int P, int E,
                                                               bf16_t* __restrict__ dg, float* __restrict__ dlogits, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  float* cst = (float*)(smem + GW_BWD_C0);
  // ---- prologue 1: A fragments of the dxh contraction, column tile nt, lane ln: table row m = ln & 31 <-> column
  //      32 nt + 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3) (the accumulator of output lane (token, lhi) then holds columns 32 nt + 16 lhi + r,
  //      r = 0 .. 15), K slots = experts 8 (ln >> 5) + i ----
  for (int f = tid; f < 16 * 64; f += 256) {
    const int ln = f & 63, nt = f >> 6, m = ln & 31, h = ln >> 5;
    const int col = 32 * nt + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
    const float lw = LN ? ln_w[col] : 1.f;
    uint32_t hi[4] = {0u, 0u, 0u, 0u}, mid[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = 8 * h + i;
      if (e < E) {
        const float wp = wg[(long)e * GW_G + col] * lw * GM_SW;
        const bf16_t hh = f32_to_bf16(wp);
        const bf16_t mm = f32_to_bf16(wp - bf16_to_f32(hh));
        hi[i >> 1] |= (uint32_t)hh << (16 * (i & 1));
        mid[i >> 1] |= (uint32_t)mm << (16 * (i & 1));
      }
    }
    *(gm_u32x4_t*)(smem + (size_t)f * 16) = gm_u32x4_t{hi[0], hi[1], hi[2], hi[3]};
    *(gm_u32x4_t*)(smem + 16384 + (size_t)f * 16) = gm_u32x4_t{mid[0], mid[1], mid[2], mid[3]};
  }
  if (LN) {
    // ---- prologue 2: A fragments of the u contraction, K step ks = 2 nt + hh, lane ln: table row m = 8 g4 + 4 hm + j holds term g4 >> 1
    //      (hi / mid) of expert 8 hm + 4 (g4 & 1) + j - output lane (token, lhi) then holds u of experts 8 lhi + i, i = 4 (g4 & 1) + j, as
    //      acc[4 g4 + j] + acc[4 (g4 + 2) + j]; K slots = columns 32 nt + 16 (ln >> 5) + 8 hh + i ----
    for (int f = tid; f < GW_KS * 64; f += 256) {
      const int ln = f & 63, ks = f >> 6, m = ln & 31, h = ln >> 5;
      const int g4 = m >> 3, e = 8 * ((m >> 2) & 1) + 4 * (g4 & 1) + (m & 3), term = g4 >> 1;
      uint32_t v[4] = {0u, 0u, 0u, 0u};
      if (e < E) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = 32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + i;
          const float wp = wg[(long)e * GW_G + c] * ln_w[c] * GM_SW;
          const bf16_t hh = f32_to_bf16(wp);
          const bf16_t val = term == 0 ? hh : f32_to_bf16(wp - bf16_to_f32(hh));
          v[i >> 1] |= (uint32_t)val << (16 * (i & 1));
        }
      }
      *(gm_u32x4_t*)(smem + GW_BWD_U0 + (size_t)f * 16) = gm_u32x4_t{v[0], v[1], v[2], v[3]};
    }
    {      // c1[e] = sum_k (hi + mid)(e, k): 16 lanes per expert, a fixed order
      const int e = tid >> 4, jj = tid & 15;
      float a = 0.f;
      if (e < E) {
        for (int c = jj; c < GW_G; c += 16) {
          const float wp = wg[(long)e * GW_G + c] * ln_w[c] * GM_SW;
          const bf16_t hh = f32_to_bf16(wp);
          a += (bf16_to_f32(hh) + bf16_to_f32(f32_to_bf16(wp - bf16_to_f32(hh)))) * (1.f / GM_SW);
        }
      }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) a += __shfl_xor(a, o);
      if (jj == 0) cst[e] = a;
    }
  }
  __syncthreads();
  float c1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c1[i] = (LN && 8 * lhi + i < E) ? cst[8 * lhi + i] : 0.f;
  const char* wrow = smem + lane * 16;
  const int stride = gridDim.x * 4;
  for (int t = blockIdx.x * 4 + w; t < n_tiles; t += stride) {
    long tok = (long)t * 32 + l31;
    const bool live = tok < P;
    tok = live ? tok : (long)P - 1;
    // ---- the whole tile's rows into registers: this lane's 16 columns of every column tile (LayerNorm only: the plain router needs no x) ----
    const char* xp = (const char*)g + tok * (GW_G * 2) + lhi * 32;
    gm_u32x4_t xr[16][2];
    if (LN) {
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) {
        xr[nt][0] = *(const gm_u32x4_t*)(xp + nt * 64);
        xr[nt][1] = *(const gm_u32x4_t*)(xp + nt * 64 + 16);
      }
    }
    // ---- dlogits of experts 8 lhi .. + 7 ----
    const int seg = (int)(tok / seg_tokens);
    const int my = idx[tok];
    const float coef = laux_coef ? laux_coef[seg] : 0.f;
    const float dgm = d_gmax ? d_gmax[tok] : 0.f;
    float mean = 0.f, rstd = 1.f;
    if (LN) { const float2 st = *(const float2*)(stats + tok * 2); mean = st.x; rstd = st.y; }
    // (unconditional, vectorised loads: a guarded scalar load per expert is a branch and a round trip of its own - 20 of them in a row
    //  made the first version of this kernel no faster than the VALU one)
    float pr[8], dp[8], dot = 0.f;
    int cnt[8];
    if (E == 16) {
      const float4 g0 = *(const float4*)(gates + tok * 16 + 8 * lhi), g1 = *(const float4*)(gates + tok * 16 + 8 * lhi + 4);
      const int4 k0 = *(const int4*)(counts + (long)seg * 16 + 8 * lhi), k1 = *(const int4*)(counts + (long)seg * 16 + 8 * lhi + 4);
      pr[0] = g0.x; pr[1] = g0.y; pr[2] = g0.z; pr[3] = g0.w; pr[4] = g1.x; pr[5] = g1.y; pr[6] = g1.z; pr[7] = g1.w;
      cnt[0] = k0.x; cnt[1] = k0.y; cnt[2] = k0.z; cnt[3] = k0.w; cnt[4] = k1.x; cnt[5] = k1.y; cnt[6] = k1.z; cnt[7] = k1.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = 8 * lhi + i, ec = e < E ? e : E - 1;
        const float pv = gates[tok * E + ec];
        cnt[i] = counts[(long)seg * E + ec];
        pr[i] = e < E ? pv : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = 8 * lhi + i;
      dp[i] = e < E ? coef * (float)cnt[i] + ((e == my) ? dgm : 0.f) : 0.f;
      dot += pr[i] * dp[i];
    }
    dot = __shfl(dot, l31) + __shfl(dot, l31 + 32);           // experts 0-7 first
    float dl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dl[i] = pr[i] * (dp[i] - dot);      // softmax backward
    if (live) {
      if (E == 16) {
        *(float4*)(dlogits + tok * 16 + 8 * lhi) = make_float4(dl[0], dl[1], dl[2], dl[3]);
        *(float4*)(dlogits + tok * 16 + 8 * lhi + 4) = make_float4(dl[4], dl[5], dl[6], dl[7]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (8 * lhi + i < E) dlogits[tok * E + 8 * lhi + i] = dl[i];
      }
    }
    gm_u32x4_t bh, bl;                                         // B fragments: bf16 heads / remainders of the 8 dlogits
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t vh = 0u, vl = 0u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float x = dl[2 * i + h] * GM_SD;
        const bf16_t hd = f32_to_bf16(x);
        vh |= (uint32_t)hd << (16 * h);
        vl |= (uint32_t)f32_to_bf16(x - bf16_to_f32(hd)) << (16 * h);
      }
      bh[i] = vh;
      bl[i] = vl;
    }
    const float mr = mean * rstd;
    float s1 = 0.f, s2 = 0.f;
    if (LN) {
      // ---- u_e = sum_k W'[e][k] x_k: 32 K steps on two accumulators, the fragments of step ks + 1 requested in front of the MFMA of step ks ----
      gm_f32x16_t ua, ub;
#pragma unroll
      for (int r = 0; r < 16; ++r) ua[r] = ub[r] = 0.f;
      gm_u32x4_t uf[2];
      uf[0] = *(const gm_u32x4_t*)(wrow + GW_BWD_U0);
#pragma unroll
      for (int ks = 0; ks < GW_KS; ++ks) {
        if (ks + 1 < GW_KS) uf[(ks + 1) & 1] = *(const gm_u32x4_t*)(wrow + GW_BWD_U0 + (ks + 1) * 1024);
        if (ks & 1) ub = SWN_MFMA_32x32x16(uf[ks & 1], xr[ks >> 1][ks & 1], ub);
        else ua = SWN_MFMA_32x32x16(uf[ks & 1], xr[ks >> 1][ks & 1], ua);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int g4 = i >> 2, j = i & 3;
        const float u = ((ua[4 * g4 + j] + ub[4 * g4 + j]) + (ua[4 * (g4 + 2) + j] + ub[4 * (g4 + 2) + j])) * (1.f / GM_SW);
        s1 += dl[i] * c1[i];
        s2 += dl[i] * (rstd * u - mr * c1[i]);
      }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      s1 *= (1.f / GW_G);
      s2 *= (1.f / GW_G);
    }
    auto dxh_tile = [&](int nt) -> gm_f32x16_t {
      gm_f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const gm_u32x4_t wh = *(const gm_u32x4_t*)(wrow + nt * 1024), wm = *(const gm_u32x4_t*)(wrow + 16384 + nt * 1024);
      acc = SWN_MFMA_32x32x16(wh, bh, acc);
      acc = SWN_MFMA_32x32x16(wh, bl, acc);
      acc = SWN_MFMA_32x32x16(wm, bh, acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= 1.f / (GM_SW * GM_SD);
      return acc;
    };
    char* op = (char*)dg + tok * (GW_G * 2) + lhi * 32;
    // (a column tile per scheduling region, the next tile's MFMAs issued in front of this tile's VALU work)
    gm_f32x16_t acc_n = dxh_tile(0);
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
      const gm_f32x16_t acc = acc_n;
      if (nt + 1 < 16) acc_n = dxh_tile(nt + 1);
      uint32_t o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float o0 = acc[2 * q], o1 = acc[2 * q + 1];
        if (LN) {
          const uint32_t xv = xr[nt][q >> 2][q & 3];
          const float xh0 = gm_lo(xv) * rstd - mr, xh1 = gm_hi(xv) * rstd - mr;
          o0 = rstd * (o0 - s1 - xh0 * s2);
          o1 = rstd * (o1 - s1 - xh1 * s2);
        }
        o[q] = pack_bf16x2(o0, o1);
      }
      if (live) {
        *(gm_u32x4_t*)(op + nt * 64) = gm_u32x4_t{o[0], o[1], o[2], o[3]};
        *(gm_u32x4_t*)(op + nt * 64 + 16) = gm_u32x4_t{o[4], o[5], o[6], o[7]};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

int gate_bwd_wide_launch(const void* g, const float* ln_w, const float* wg, const float* gates, const int32_t* idx, const float* d_gmax,
                         const float* stats, const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int n_experts,
                         void* dg, float* dlogits, void* stream) {
  const int n_tiles = cdiv(n_tokens, 32);
  int blocks = cdiv(n_tiles, 4);
  if (blocks > 512) blocks = 512;
  const void* fn = ln_w ? (const void*)gate_bwd_wide_kernel<true> : (const void*)gate_bwd_wide_kernel<false>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GW_BWD_LDS_ALL);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const bf16_t* gp = (const bf16_t*)g;
  bf16_t* dgp = (bf16_t*)dg;
  void* kargs[] = {(void*)&gp, (void*)&ln_w, (void*)&wg, (void*)&gates, (void*)&idx, (void*)&d_gmax, (void*)&stats, (void*)&counts,
                   (void*)&laux_coef, (void*)&seg_tokens, (void*)&n_tokens, (void*)&n_experts, (void*)&dgp, (void*)&dlogits, (void*)&n_tiles};
  e = hipLaunchKernel(fn, dim3(blocks), dim3(256), kargs, GW_BWD_LDS_ALL, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_gate_bwd (wide mfma) launch: %s", hipGetErrorString(e));
  return 0;
}


// Router parameter gradients, 512 features: M[e][k] = sum_tok dlogits[tok][e] xhat[tok][k] and DL[e] = sum_tok dlogits[tok][e] per block, in
// gate_dwg_kernel's partial format (finished by ordered_reduce + gate_dwg_finalize_kernel).  gate_dwg_kernel<., 16, 512> walks its tokens
// with 128 accumulators per lane and 4 KiB per wave in flight: 0.41 ms per 852 k rows, latency-bound at a third of the rate the row reads
// allow.  Here the sum over the tokens is a GEMM on the matrix pipe, the 256-feature kernel's scheme with the tile shared by the block:
//   * a 32-token x 1 KiB tile comes in by LDS-DMA (double-buffered: the next tile travels while this one is multiplied), rows swizzled
//     like every tile of this file; wave w owns columns [128 w, +128);
//   * v_mfma_f32_16x16x32 per 16 columns: A = dlr^T (16 experts x 32 tokens; dlr = dlogits * rstd split into a 16-bit head and remainder:
//     two MFMAs), B = x^T read straight from the row-major tile with the transposing LDS read (ds_read_b64_tr_b16);
//   * xhat = (x - mean) rstd: M = dlr^T x - C, C[e] = sum_tok dlr[tok][e] mean[tok] (fp32, wave 0's lanes).
constexpr int GD_TILE_B = 32 * 1024;
constexpr int GD_DST0 = 2 * GD_TILE_B;            // dlr^T staging: head [16][32] + remainder [16][32] (1 KiB each)
constexpr int GD_CST0 = GD_DST0 + 2048;           // C[16], DL[16]
constexpr int GD_LDS = GD_CST0 + 128;

// 16 bytes per lane from the lane's own global address into LDS at lds_dst (wave-uniform) + lane * 16; inline asm: the compiler does not
// see the copy, the waits are written by hand (wgrad.hip's dma16)
__device__ __forceinline__ void gd_dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <bool LN>
__global__ __launch_bounds__(256, 2) void gate_dwg_wide_kernel(const bf16_t* __restrict__ g, const float* __restrict__ stats,
                                                               const float* __restrict__ dlogits, int P, int E, int n_tiles,
                                                               float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5, kg = lane >> 4, n16 = lane & 15;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  float* cst = (float*)(smem + GD_CST0);
  typedef __attribute__((ext_vector_type(4))) float f32x4_t_;
  f32x4_t_ macc[8][2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { macc[j][0] = f32x4_t_{0.f, 0.f, 0.f, 0.f}; macc[j][1] = f32x4_t_{0.f, 0.f, 0.f, 0.f}; }
  float cacc[16], dlacc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) cacc[e] = dlacc[e] = 0.f;
  // transposing read of (tokens 8 kg + 4 h + (n16 >> 2), columns 16 ct + 4 (n16 & 3) .. + 3): row byte address + swizzle seed
  uint32_t tr_row[2], tr_sw[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 8 * kg + 4 * h + (n16 >> 2);
    tr_row[h] = (uint32_t)(r * 1024 + 8 * (n16 & 1));
    tr_sw[h] = (uint32_t)(r & 15);
  }
  auto dma_tile = [&](int t, int buf) {          // wave w copies rows 8 w .. + 7: chunk (lane ^ (row & 15)) of the row -> position `lane`
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int r = 8 * w + c;
      long tok = (long)t * 32 + r;
      tok = tok < P ? tok : (long)P - 1;
      gd_dma16((const char*)g + tok * 1024 + ((lane ^ (r & 15)) << 4),
               __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(buf * GD_TILE_B + r * 1024)));
    }
  };
  float dlv[16], mu = 0.f, rs = 1.f;
  bool live = false;
  auto load_dl = [&](int t) {                    // (wave 0) this lane's token: its dlogits row and statistics
    long tok = (long)t * 32 + l31;
    live = tok < P;
    tok = live ? tok : (long)P - 1;
    if (E == 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *(const float4*)(dlogits + tok * 16 + 4 * q);
        dlv[4 * q] = v.x; dlv[4 * q + 1] = v.y; dlv[4 * q + 2] = v.z; dlv[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = dlogits[tok * E + (e < E ? e : E - 1)];
        dlv[e] = e < E ? v : 0.f;
      }
    }
    if (LN) { const float2 st = *(const float2*)(stats + tok * 2); mu = st.x; rs = st.y; }
  };
  const int stride = gridDim.x;
  int t = blockIdx.x;
  if (t < n_tiles) {
    dma_tile(t, 0);
    if (w == 0) load_dl(t);
  }
  for (int it = 0; t < n_tiles; t += stride, ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this tile's copy (issued one iteration ago) and the dlogits registers
    if (w == 0) {      // dlr^T: row e of the head (lower half-wave) / remainder (upper) table, column = this lane's token
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float d = live ? dlv[e] : 0.f;
        const float x = d * rs * GM_SD;
        const bf16_t hd = f32_to_bf16(x);
        const bf16_t val = lhi ? f32_to_bf16(x - bf16_to_f32(hd)) : hd;
        *(bf16_t*)(smem + GD_DST0 + lhi * 1024 + e * 64 + l31 * 2) = val;
        cacc[e] += d * rs * mu;
        dlacc[e] += d;
      }
    }
    __syncthreads();
    const int tn = t + stride;
    if (tn < n_tiles) {
      dma_tile(tn, buf ^ 1);
      if (w == 0) load_dl(tn);
    }
    const gm_u32x4_t afr_h = *(const gm_u32x4_t*)(smem + GD_DST0 + n16 * 64 + kg * 16);            // row n16, tokens 8 kg .. + 7
    const gm_u32x4_t afr_r = *(const gm_u32x4_t*)(smem + GD_DST0 + 1024 + n16 * 64 + kg * 16);
    const uint32_t tb = (uint32_t)(buf * GD_TILE_B);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t ch = (uint32_t)(2 * (8 * w + j) + ((n16 & 3) >> 1));
      const uint32_t a0 = tb + tr_row[0] + ((ch ^ tr_sw[0]) << 4), a1 = tb + tr_row[1] + ((ch ^ tr_sw[1]) << 4);
      uint2 b0, b1;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0) : "v"(a0) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1) : "v"(a1) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const gm_u32x4_t bx = {b0.x, b0.y, b1.x, b1.y};
      macc[j][0] = GM_MFMA16x16x32(afr_h, bx, macc[j][0]);
      macc[j][1] = GM_MFMA16x16x32(afr_r, bx, macc[j][1]);
    }
    __syncthreads();                                        // the tile and the staging rows are free for the next round
  }
  // ---- the block's partial: part[e * 512 + k] = M, part[E * 512 + e] = DL ----
  if (w == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float a = cacc[e], b = dlacc[e];
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }      // over the 32 tokens (both half-waves hold them)
      if (lane == 0) { cst[e] = a; cst[16 + e] = b; }
    }
  }
  __syncthreads();
  float* part = partial + (size_t)blockIdx.x * ((size_t)E * GW_G + E);
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = 4 * kg + r, col = 16 * (8 * w + j) + n16;
      if (e < E) part[(size_t)e * GW_G + col] = (macc[j][0][r] + macc[j][1][r]) * (1.f / GM_SD) - (LN ? cst[e] : 0.f);
    }
  if (tid < E) part[(size_t)E * GW_G + tid] = cst[16 + tid];
}

int gate_dwg_wide_blocks(int n_tokens) {
  const int n_tiles = cdiv(n_tokens, 32);
  return n_tiles > 512 ? 512 : n_tiles;
}

int gate_dwg_wide_launch(const void* g, bool layer_norm, const float* stats, const float* dlogits, int n_tokens, int n_experts, float* partial,
                         void* stream) {
  const int n_tiles = cdiv(n_tokens, 32);
  const int blocks = gate_dwg_wide_blocks(n_tokens);
  const void* fn = layer_norm ? (const void*)gate_dwg_wide_kernel<true> : (const void*)gate_dwg_wide_kernel<false>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GD_LDS);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const bf16_t* gp = (const bf16_t*)g;
  void* kargs[] = {(void*)&gp, (void*)&stats, (void*)&dlogits, (void*)&n_tokens, (void*)&n_experts, (void*)&n_tiles, (void*)&partial};
  e = hipLaunchKernel(fn, dim3(blocks), dim3(256), kargs, GD_LDS, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_gate_bwd (wide dwg) launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace swn

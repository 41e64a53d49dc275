// HBM-bound kernels of the Switch-NeRF hot path: ray sampling + positional encoding, gate (LayerNorm + router +
// softmax + top-1), dispatch / combine (Tutel sparse-kernel ABI), sigma/colour heads, volumetric compositing,
// Adam.  One 64-lane wave per token/ray row, 16-byte accesses, wave shuffles for the reductions.
#include <stdarg.h>
#include "common.hpp"
#include "pe_store.hpp"

namespace swn {

thread_local char g_err[512] = {0};
int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

// ------------------------------------------------------------------------------------------------ MFMA probe
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// For C/D: run D = A*B with A[i][k] = (i+1) for k == 0 else 0 and B[k][j] = (j+1)*1000 ... instead of inferring from
// values we test the documented maps directly: out[0..1023]   : D of bf16 mfma with A = one-hot rows, see host test.
__global__ void probe_kernel(int32_t* out) {
  const int lane = threadIdx.x;
  // Test 1 (bf16 32x32x16): A[i][k] = i*16+k (exact in bf16 for < 256? no -> use small ints), B = identity-like.
  // We encode A[i][k] = (i % 8) + 8 * (k % 8)?  Keep it simple: A[i][k] = 1 if k == (i % 16) else 0; B[k][j] = k + 16 * (j % 4).
  // Then D[i][j] = B[i % 16][j] = (i % 16) + 16 * (j % 4): every lane reports its 16 D values.
  bf16x8_t a, b;
  const int i = lane & 31, kb = (lane >> 5) * 8;
  for (int e = 0; e < 8; ++e) {
    const int k = kb + e;
    const float av = (k == (i % 16)) ? 1.f : 0.f;
    const float bv = (float)(k + 16 * (i % 4));  // here lane&31 plays the role of j for the B operand
    a[e] = (short)f32_to_bf16(av);
    b[e] = (short)f32_to_bf16(bv);
  }
  f32x16_t c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = SWN_MFMA_32x32x16(a, b, c);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = (int)c[r];
  // Test 2 (f32 32x32x2): A[i][k] = (k == i % 2), B[k][j] = k + 2 * (j % 8)  ->  D[i][j] = (i % 2) + 2 * (j % 8)
  const int k2 = lane >> 5;
  const float a2 = (k2 == (i % 2)) ? 1.f : 0.f;
  const float b2 = (float)(k2 + 2 * (i % 8));
  f32x16_t c2;
  for (int r = 0; r < 16; ++r) c2[r] = 0.f;
  c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, c2, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[1024 + lane * 16 + r] = (int)c2[r];
}

// ------------------------------------------------------------------------------------------------ sample + PE
__device__ __forceinline__ float z_of(float near, float far, float t) {
  // rendering.py:86  near * (1 - t) + far * t, each torch op rounded separately: fma contraction must stay off
  // (ROCm's __fmul_rn/__fadd_rn are plain operators and do not prevent it).
#pragma clang fp contract(off)
  const float a = near * (1.f - t);
  const float b = far * t;
  return a + b;
}

template <typename T, int LMAX>
__global__ __launch_bounds__(128) void sample_pe_kernel(const float* __restrict__ rays, const float* __restrict__ tsteps,
                                                        const float* __restrict__ prand, float perturb, int n_rays,
                                                        int S, int L, float* __restrict__ z_out, T* __restrict__ pe,
                                                        int pe_stride, const float* __restrict__ z_in) {
#pragma clang fp contract(off)
  const long p_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p_raw < (long)n_rays * S;
  const long p = live ? p_raw : (long)n_rays * S - 1;            // surplus threads recompute the last point, store nothing
  const int ray = (int)(p / S), s = (int)(p - (long)ray * S);
  const float* r = rays + (long)ray * 8;
  const float near = r[6], far = r[7];
  float z = z_in ? z_in[p] : z_of(near, far, tsteps[s]);      // z_in: depths supplied by the caller (fine pass)
  if (!z_in && perturb > 0.f && prand) {  // rendering.py:573-584
    const float zp = s > 0 ? z_of(near, far, tsteps[s - 1]) : z;
    const float zn = s < S - 1 ? z_of(near, far, tsteps[s + 1]) : z;
    const float lower = s > 0 ? 0.5f * (zp + z) : z;
    const float upper = s < S - 1 ? 0.5f * (z + zn) : z;
    const float pr = perturb * prand[p];
    const float span = (upper - lower) * pr;
    z = lower + span;
  }
  if (z_out && live) z_out[p] = z;
  float v[8 + 6 * LMAX + 8];
  float x[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dz = r[3 + c] * z;  // rendering.py:90 (mul, then add: two roundings)
    x[c] = r[c] + dz;
    v[c] = x[c];
  }
  if constexpr (sizeof(T) == 4) {   // fp32 (parity mode): every octave from its own accurately reduced argument
    float f = 1.f;
#pragma unroll
    for (int k = 0; k < LMAX; ++k) {
      if (k < L) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float sn, cs;
          sincosf(f * x[c], &sn, &cs);  // models/nerf.py:24
          v[3 + 6 * k + c] = sn;
          v[3 + 6 * k + 3 + c] = cs;
        }
      }
      f *= 2.f;
    }
  } else {   // bf16 output: one sincos per coordinate, higher octaves by angle doubling (error doubles per octave:
             // <= 2^11 * 1e-7 = 2e-4 at octave 11, far below the bf16 rounding of 4e-3) - 12x fewer trig evaluations
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float sn, cs;
      sincosf(x[c], &sn, &cs);
#pragma unroll
      for (int k = 0; k < LMAX; ++k) {
        if (k < L) {
          v[3 + 6 * k + c] = sn;
          v[3 + 6 * k + 3 + c] = cs;
        }
        const float s2 = 2.f * sn * cs, c2 = 1.f - 2.f * sn * sn;
        sn = s2;
        cs = c2;
      }
    }
  }
  pe_store_rows<T>(v, 3 + 6 * L, pe, pe_stride, p, live, (long)n_rays * S);
}

template <typename T, int LMAX>
__global__ void dir_pe_kernel(const float* __restrict__ rays, int n_rays, int L, T* __restrict__ pe, int stride) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays) return;
  const float* r = rays + (long)ray * 8 + 3;
  T* dst = pe + (long)ray * stride;
  for (int c = 0; c < stride; ++c) ElemIO<T>::st(dst + c, 0.f);
  for (int c = 0; c < 3; ++c) ElemIO<T>::st(dst + c, r[c]);
  float f = 1.f;
  for (int k = 0; k < L; ++k) {
    for (int c = 0; c < 3; ++c) {
      float sn, cs;
      sincosf(f * r[c], &sn, &cs);
      ElemIO<T>::st(dst + 3 + 6 * k + c, sn);
      ElemIO<T>::st(dst + 3 + 6 * k + 3 + c, cs);
    }
    f *= 2.f;
  }
}

// ------------------------------------------------------------------------------------------------ row kernels
// One token row per 16-lane group (4 rows per wave): a lane owns COLS/16 features in 16-byte chunks
// (chunk c = j + 16 q  ->  a 16-lane group reads 256 contiguous bytes per instruction), and row reductions are four
// DPP adds inside the 16-lane row (quad_perm, quad_perm, row_half_mirror, row_mirror) - no LDS traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum16(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

template <typename T, int COLS> struct Row16 {
  static constexpr int VPL = COLS / 16;                          // values per lane
  static constexpr int EPC = 16 / (int)sizeof(T);                // elements per 16-byte chunk
  static constexpr int NCH = VPL / EPC;                          // chunks per lane
  static_assert(NCH >= 1, "row too narrow for the 16-lane layout");
  static __device__ __forceinline__ int col(int j, int v) { return (j + 16 * (v / EPC)) * EPC + (v % EPC); }
  static __device__ __forceinline__ void load(const T* row, int j, float* x) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const uint4 u = *(const uint4*)(row + (j + 16 * q) * EPC);
      if constexpr (sizeof(T) == 2) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[q * 8 + 2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xFFFF));
          x[q * 8 + 2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16));
        }
      } else {
        x[q * 4 + 0] = __uint_as_float(u.x); x[q * 4 + 1] = __uint_as_float(u.y);
        x[q * 4 + 2] = __uint_as_float(u.z); x[q * 4 + 3] = __uint_as_float(u.w);
      }
    }
  }
  // the same in two halves, so that the next row's 16-byte loads can be in flight while this row is processed
  static __device__ __forceinline__ void load_raw(const T* row, int j, uint4* raw) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) raw[q] = *(const uint4*)(row + (j + 16 * q) * EPC);
  }
  static __device__ __forceinline__ void unpack(const uint4* raw, float* x) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const uint4 u = raw[q];
      if constexpr (sizeof(T) == 2) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[q * 8 + 2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xFFFF));
          x[q * 8 + 2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16));
        }
      } else {
        x[q * 4 + 0] = __uint_as_float(u.x); x[q * 4 + 1] = __uint_as_float(u.y);
        x[q * 4 + 2] = __uint_as_float(u.z); x[q * 4 + 3] = __uint_as_float(u.w);
      }
    }
  }
  static __device__ __forceinline__ void store(T* row, int j, const float* x) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      uint4 u;
      if constexpr (sizeof(T) == 2) {
        u.x = pack_bf16x2(x[q * 8 + 0], x[q * 8 + 1]); u.y = pack_bf16x2(x[q * 8 + 2], x[q * 8 + 3]);
        u.z = pack_bf16x2(x[q * 8 + 4], x[q * 8 + 5]); u.w = pack_bf16x2(x[q * 8 + 6], x[q * 8 + 7]);
      } else {
        u.x = __float_as_uint(x[q * 4 + 0]); u.y = __float_as_uint(x[q * 4 + 1]);
        u.z = __float_as_uint(x[q * 4 + 2]); u.w = __float_as_uint(x[q * 4 + 3]);
      }
      *(uint4*)(row + (j + 16 * q) * EPC) = u;
    }
  }
  // fp32 parameter vector laid out like the row
  static __device__ __forceinline__ void loadf(const float* vec, int j, float* x) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) x[v] = vec[col(j, v)];
  }
};

// ------------------------------------------------------------------------------------------------ gate
template <typename T, int G, int EMAX, int TB = 1>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ g, const float* __restrict__ ln_w,
                                                       const float* __restrict__ ln_b, const float* __restrict__ wg,
                                                       int P, int E, float* __restrict__ gates, int32_t* __restrict__ idx,
                                                       float* __restrict__ gmax, float* __restrict__ stats,
                                                       const float* __restrict__ noise, float noise_scale) {
  // noise != NULL: logits += noise_scale * noise[token][expert] before the softmax - the gate-noise branch of a training forward
  // (--gate_noise > 0: tutel_moe_layer_nobatch.py:119-122, noise_scale = gate_noise / E; swn_gate_fwd_noise).  NULL: nothing is added.
  // TB tokens per 16-lane group and pass (consecutive rows): every router weight read from LDS serves TB rows; the arithmetic of a row
  // is unchanged, value for value (TB = 1, the default everywhere: the kernel of rounds 1-3).  What made the 512-feature x 16-expert
  // instantiation slow (2.0 ms per 852 k rows, a tenth of the rate its row reads allow) was not the LDS traffic but the scheduler
  // hoisting all 128 weight reads of the unrolled expert loop: 512 registers + 168 spilled - see the sched_barrier below (0.74 ms).
  using R = Row16<T, G>;
  constexpr int VPL = R::VPL;
  constexpr int LS = VPL + 4;       // lane segment stride (floats): +16 B keeps the 16 lanes of a b128 read on distinct banks
  __shared__ float swg[EMAX * 16 * LS];   // router weights re-laid so that a lane's VPL values are contiguous: [e][j][v]
  const int j = threadIdx.x & 15;
  for (int t = threadIdx.x; t < EMAX * G; t += 256) {
    const int e = t / G, r = t % G, jj = r / VPL, v = r % VPL;
    swg[(e * 16 + jj) * LS + v] = e < E ? wg[(long)e * G + R::col(jj, v)] : 0.f;
  }
  float w[VPL], b[VPL];
  if (ln_w) { R::loadf(ln_w, j, w); R::loadf(ln_b, j, b); }
  __syncthreads();
  const long gid = (((long)blockIdx.x * 256 + threadIdx.x) >> 4) * TB;
  const long ngr = (((long)gridDim.x * 256) >> 4) * TB;
  uint4 nxt[TB][R::NCH];
#pragma unroll
  for (int t = 0; t < TB; ++t)
    if (gid + t < P) R::load_raw(g + (gid + t) * G, j, nxt[t]);
  for (long tok0 = gid; tok0 < P; tok0 += ngr) {
    float x[TB][VPL];
    float mean[TB], rstd[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      R::unpack(nxt[t], x[t]);
      if (tok0 + ngr + t < P) R::load_raw(g + (tok0 + ngr + t) * G, j, nxt[t]);     // next rows in flight during these rows' arithmetic
      mean[t] = 0.f; rstd[t] = 1.f;
      if (ln_w) {  // torch.nn.LayerNorm, eps 1e-5 (models/nerf_moe.py:301-302)
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) s += x[t][v];
        mean[t] = sum16(s) * (1.f / G);
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) q += (x[t][v] - mean[t]) * (x[t][v] - mean[t]);
        rstd[t] = 1.f / sqrtf(sum16(q) * (1.f / G) + 1e-5f);
#pragma unroll
        for (int v = 0; v < VPL; ++v) x[t][v] = (x[t][v] - mean[t]) * rstd[t] * w[v] + b[v];
      }
    }
    float logit[TB][EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      float d[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) d[t] = 0.f;
      const float4* wp = (const float4*)(swg + (e * 16 + j) * LS);
#pragma unroll
      for (int v4 = 0; v4 < VPL / 4; ++v4) {
        const float4 ww = wp[v4];
#pragma unroll
        for (int t = 0; t < TB; ++t)
          d[t] += x[t][4 * v4] * ww.x + x[t][4 * v4 + 1] * ww.y + x[t][4 * v4 + 2] * ww.z + x[t][4 * v4 + 3] * ww.w;
      }
#pragma unroll
      for (int t = 0; t < TB; ++t) logit[t][e] = sum16(d[t]);
      // (wide rows: one expert's weights at a time - left alone the scheduler hoists the LDS reads of all 16 experts to the top of the
      //  unrolled loop: 512 registers and 170-480 spilled ones at 512 features)
      if constexpr (G >= 512) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const long tok = tok0 + t;
      if (tok >= P) break;
      if (noise) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
          if (e < E) logit[t][e] = logit[t][e] + noise_scale * noise[tok * E + e];
      }
      float mx = logit[t][0];
#pragma unroll
      for (int e = 1; e < EMAX; ++e)
        if (e < E) mx = fmaxf(mx, logit[t][e]);
      float den = 0.f, pr[EMAX];
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        pr[e] = (e < E) ? expf(logit[t][e] - mx) : 0.f;
        den += pr[e];
      }
      int best = 0;
      float bv = -1.f;
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        pr[e] = pr[e] / den;
        if (e < E && pr[e] > bv) { bv = pr[e]; best = e; }  // first maximum
      }
#pragma unroll
      for (int e = 0; e < EMAX; ++e)
        if (j == (e & 15) && e < E) gates[tok * E + e] = pr[e];
      if (j == 0) {
        idx[tok] = best;
        gmax[tok] = bv;
        if (stats) { stats[tok * 2] = mean[t]; stats[tok * 2 + 1] = rstd[t]; }
      }
    }
  }
}

// backward: dlogits (softmax + l_aux), d(xn) = dlogits @ wg, LayerNorm backward -> dg; dlogits [P, E] written out for the parameter
// gradients: d_wg, d_ln_w and d_ln_b all follow from M[e][k] = sum_tok dlogits[tok][e] * xhat[tok][k] and DL[e] = sum_tok dlogits[tok][e]
// (gate_dwg_kernel + gate_dwg_finalize_kernel), so this kernel keeps no per-column accumulators.
template <typename T, int G, int EMAX, int TB = 1>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const T* __restrict__ g, const float* __restrict__ ln_w,
                                                       const float* __restrict__ ln_b, const float* __restrict__ wg,
                                                       const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                       const float* __restrict__ d_gmax, const float* __restrict__ stats,
                                                       const int32_t* __restrict__ counts, const float* __restrict__ laux_coef,
                                                       int seg_tokens, int P, int E, T* __restrict__ dg,
                                                       float* __restrict__ dlogits, float* __restrict__ d_ln_w,
                                                       float* __restrict__ d_ln_b, const float* __restrict__ d_probs,
                                                       const float* __restrict__ d_logits_add) {
  // d_probs [P, E] (may be NULL): a dense gradient w.r.t. the probabilities, added to the top-1 / l_aux terms (top-k gates);
  // d_logits_add [P, E] (may be NULL): a gradient w.r.t. the logits themselves, added behind the softmax backward (load / importance loss)
  // (TB rows per 16-lane group and pass: see gate_fwd_kernel)
  using R = Row16<T, G>;
  constexpr int VPL = R::VPL;
  constexpr int LS = VPL + 4;
  __shared__ float swg[EMAX * 16 * LS];
  const int j = threadIdx.x & 15;
  for (int t = threadIdx.x; t < EMAX * G; t += 256) {
    const int e = t / G, r = t % G, jj = r / VPL, v = r % VPL;
    swg[(e * 16 + jj) * LS + v] = e < E ? wg[(long)e * G + R::col(jj, v)] : 0.f;
  }
  float w[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) w[v] = 1.f;
  if (ln_w) R::loadf(ln_w, j, w);
  __syncthreads();
  const long gid = (((long)blockIdx.x * 256 + threadIdx.x) >> 4) * TB;
  const long ngr = (((long)gridDim.x * 256) >> 4) * TB;
  for (long tok0 = gid; tok0 < P; tok0 += ngr) {
    float xh[TB][VPL], dl[TB][EMAX], rstd[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const long tok = tok0 + t < P ? tok0 + t : (long)P - 1;      // (rows past the end repeat the last row: computed, never stored)
      R::load(g + tok * G, j, xh[t]);
      const float mean = ln_w ? stats[tok * 2] : 0.f;
      rstd[t] = ln_w ? stats[tok * 2 + 1] : 1.f;
      if (ln_w) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) xh[t][v] = (xh[t][v] - mean) * rstd[t];
      }
      const int seg = (int)(tok / seg_tokens);
      const int my = idx[tok];
      const float coef = laux_coef ? laux_coef[seg] : 0.f;
      const float dgm = d_gmax ? d_gmax[tok] : 0.f;
      float pr[EMAX], dp[EMAX], dot = 0.f;
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        pr[e] = (e < E) ? gates[tok * E + e] : 0.f;
        dp[e] = (e < E) ? coef * (float)counts[seg * E + e] + ((e == my) ? dgm : 0.f) + (d_probs ? d_probs[tok * E + e] : 0.f) : 0.f;
        dot += pr[e] * dp[e];
      }
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        dl[t][e] = pr[e] * (dp[e] - dot);  // softmax backward
        if (d_logits_add && e < E) dl[t][e] += d_logits_add[tok * E + e];
        if (e < E && j == (e & 15) && tok0 + t < P) dlogits[tok * E + e] = dl[t][e];
      }
    }
    float dxn[TB][VPL];
#pragma unroll
    for (int t = 0; t < TB; ++t)
#pragma unroll
      for (int v = 0; v < VPL; ++v) dxn[t][v] = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      const float4* wp = (const float4*)(swg + (e * 16 + j) * LS);
#pragma unroll
      for (int v4 = 0; v4 < VPL / 4; ++v4) {
        const float4 ww = wp[v4];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          dxn[t][4 * v4] += dl[t][e] * ww.x; dxn[t][4 * v4 + 1] += dl[t][e] * ww.y;
          dxn[t][4 * v4 + 2] += dl[t][e] * ww.z; dxn[t][4 * v4 + 3] += dl[t][e] * ww.w;
        }
      }
      if constexpr (G >= 512) __builtin_amdgcn_sched_barrier(0);      // (one expert's weights at a time: see gate_fwd_kernel)
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      float dx[VPL];
      if (ln_w) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const float dxh = dxn[t][v] * w[v];
          s1 += dxh;
          s2 += dxh * xh[t][v];
        }
        s1 = sum16(s1) * (1.f / G);
        s2 = sum16(s2) * (1.f / G);
#pragma unroll
        for (int v = 0; v < VPL; ++v) dx[v] = rstd[t] * (dxn[t][v] * w[v] - s1 - xh[t][v] * s2);
      } else {
#pragma unroll
        for (int v = 0; v < VPL; ++v) dx[v] = dxn[t][v];
      }
      if (tok0 + t < P) R::store(dg + (tok0 + t) * G, j, dx);
    }
  }
}

// M[e][c] = sum_tok dlogits[tok][e] * xhat[tok][c] (xhat = the normalised row before the LayerNorm's affine part, recomputed from g and
// the statistics; without LayerNorm xhat = x), DL[e] = sum_tok dlogits[tok][e]; per block, finished by gate_dwg_finalize_kernel:
//   d_wg[e][c] += ln_w[c] M[e][c] + ln_b[c] DL[e];   d_ln_w[c] += sum_e wg[e][c] M[e][c];   d_ln_b[c] += sum_e wg[e][c] DL[e]
// block = 256 threads = SG token sub-groups x LPT lanes; a lane owns one 16-byte chunk of the row (8 bf16 / 4 fp32 columns), so a
// sub-group reads a token's row as one contiguous run; each sub-group walks its tokens with UNR rows in flight and keeps
// E x VPT accumulators; the sub-groups are summed with LDS atomics and the block issues one global atomic per (expert, column).
template <typename T, int E, int G>
__global__ __launch_bounds__(256) void gate_dwg_kernel(const T* __restrict__ g, const float* __restrict__ ln_w,
                                                       const float* __restrict__ ln_b, const float* __restrict__ stats,
                                                       const float* __restrict__ dlogits, int P, int tok_per_block,
                                                       float* __restrict__ partial) {
  constexpr int VPT = 16 / (int)sizeof(T);     // columns per lane
  constexpr int LPT = G / VPT;                 // lanes per token
  constexpr int SG = 256 / LPT;                // token sub-groups per block
  constexpr int UNR = 4;
  static_assert(LPT <= 256 && 256 % LPT == 0, "gate width");
  __shared__ float red[E * G + E];
  const int lane = threadIdx.x % LPT, sg = threadIdx.x / LPT;
  const int c0 = lane * VPT;
  for (int t = threadIdx.x; t < E * G + E; t += 256) red[t] = 0.f;
  float acc[E][VPT], dls[E];
#pragma unroll
  for (int e = 0; e < E; ++e) dls[e] = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int v = 0; v < VPT; ++v) acc[e][v] = 0.f;
  const long t0 = (long)blockIdx.x * tok_per_block;
  const long t1 = min((long)P, t0 + tok_per_block);
  for (long tb = t0 + sg; tb < t1; tb += (long)SG * UNR) {
    uint4 raw[UNR];
    float dl[UNR][E], mu[UNR], rs[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {            // all loads of the UNR tokens first
      const long tok = min(tb + (long)u * SG, (long)P - 1);
      raw[u] = *(const uint4*)(g + tok * G + c0);
#pragma unroll
      for (int e4 = 0; e4 < E / 4; ++e4) {
        const float4 v = *(const float4*)(dlogits + tok * E + e4 * 4);
        dl[u][e4 * 4] = v.x; dl[u][e4 * 4 + 1] = v.y; dl[u][e4 * 4 + 2] = v.z; dl[u][e4 * 4 + 3] = v.w;
      }
      mu[u] = 0.f; rs[u] = 1.f;
      if (ln_w) { const float2 st = *(const float2*)(stats + tok * 2); mu[u] = st.x; rs[u] = st.y; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool live = tb + (long)u * SG < t1;
      float x[VPT];
      if constexpr (sizeof(T) == 2) {
        const uint32_t ww[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[2 * i] = bf16_to_f32((bf16_t)(ww[i] & 0xFFFF)); x[2 * i + 1] = bf16_to_f32((bf16_t)(ww[i] >> 16)); }
      } else {
        x[0] = __uint_as_float(raw[u].x); x[1] = __uint_as_float(raw[u].y); x[2] = __uint_as_float(raw[u].z); x[3] = __uint_as_float(raw[u].w);
      }
#pragma unroll
      for (int v = 0; v < VPT; ++v) {
        float xn = ln_w ? (x[v] - mu[u]) * rs[u] : x[v];
        xn = live ? xn : 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e][v] += dl[u][e] * xn;
      }
      if (lane == 0 && live) {
#pragma unroll
        for (int e = 0; e < E; ++e) dls[e] += dl[u][e];
      }
    }
  }
  __syncthreads();
  // the sub-groups add their sums one after the other (a fixed order: the result has the same bits on every run; LDS atomics would
  // add them in arrival order)
  for (int s = 0; s < SG; ++s) {
    if (sg == s) {
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int v = 0; v < VPT; ++v) red[e * G + c0 + v] += acc[e][v];
      if (lane == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) red[E * G + e] += dls[e];
      }
    }
    __syncthreads();
  }
  // the block's partial [E x G] goes to the workspace with plain stores; ordered_reduce_kernel sums the blocks (a thousand blocks
  // x 2048 device-scope atomics onto the same 2048 words took longer than reading the operands, and their order is not fixed)
  float* part = partial + (size_t)blockIdx.x * (E * G + E);
  for (int t = threadIdx.x; t < E * G + E; t += 256) part[t] = red[t];
}

// the three parameter gradients from M = msum[0 .. E G) and DL = msum[E G .. E G + E): one block, thread = column
__global__ void gate_dwg_finalize_kernel(const float* __restrict__ msum, int E, int G, const float* __restrict__ ln_w,
                                         const float* __restrict__ ln_b, const float* __restrict__ wg, float* __restrict__ d_wg,
                                         float* __restrict__ d_ln_w, float* __restrict__ d_ln_b) {
  const int k = threadIdx.x;
  if (k >= G) return;
  const float w = ln_w ? ln_w[k] : 1.f, b = ln_w ? ln_b[k] : 0.f;
  float lw = 0.f, lb = 0.f;
  for (int e = 0; e < E; ++e) {
    const float M = msum[e * G + k], DL = msum[E * G + e], wv = wg[(long)e * G + k];
    d_wg[(long)e * G + k] += w * M + b * DL;
    lw += wv * M;
    lb += wv * DL;
  }
  if (ln_w) { d_ln_w[k] += lw; d_ln_b[k] += lb; }
}

// ------------------------------------------------------------------------------------------------ dispatch / combine
// Tutel batched sparse kernels: row(i) = seg(i)*E*C + idx[i]*C + loc[i], dropped iff loc >= C or idx < 0.
template <typename T, int MODE>  // MODE 0: D[row] = g*x   1: out[i] = g*D[row] (+relu)   2: dgate[i] = <D[row], x[i]>   3: out[i] += g*D[row]
// begin != NULL: the no-batch layout (tutel_sparse_nobatch.py:24-133): row(i) = begin[seg(i) * E + idx[i]] + loc[i], rows packed
// contiguously per expert, NO capacity test (dropped iff idx < 0).
__global__ __launch_bounds__(256) void sparse_kernel(const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                     const int32_t* __restrict__ loc, T* __restrict__ tok_buf,
                                                     T* __restrict__ disp, float* __restrict__ dgate, int samples,
                                                     int hidden, int capacity, int seg_tokens, int n_experts, int relu,
                                                     const int32_t* __restrict__ begin) {
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  const int cpr = hidden * (int)sizeof(T) / 16;  // 16-byte chunks per row
  constexpr int EPC = 16 / (int)sizeof(T);
  for (long i = wid; i < samples; i += nw) {
    const int e = idx[i], l = loc[i];
    const bool keep = begin ? (e >= 0) : ((e >= 0) && (l < capacity) && (l >= 0));
    const long row = begin ? (long)begin[(i / seg_tokens) * (long)n_experts + max(e, 0)] + l
                           : (i / seg_tokens) * (long)n_experts * capacity + (long)e * capacity + l;
    const float gt = gates ? gates[i] : 1.f;
    if (MODE == 2) {
      float d = 0.f;
      if (keep)
        for (int ch = lane; ch < cpr; ch += 64) {
          const T* a = disp + row * hidden + ch * EPC;
          const T* b = tok_buf + i * hidden + ch * EPC;
#pragma unroll
          for (int j = 0; j < EPC; ++j) d += ElemIO<T>::ld(a + j) * ElemIO<T>::ld(b + j);
        }
      d = wave_sum(d);
      if (lane == 0) dgate[i] = d;
      continue;
    }
    for (int ch = lane; ch < cpr; ch += 64) {
      if (MODE == 0) {
        if (!keep) continue;
        const T* s = tok_buf + i * hidden + ch * EPC;
        T* dd = disp + row * hidden + ch * EPC;
        if (!gates) {
          *(uint4*)dd = *(const uint4*)s;
        } else {
#pragma unroll
          for (int j = 0; j < EPC; ++j) ElemIO<T>::st(dd + j, gt * ElemIO<T>::ld(s + j));
        }
      } else {
        T* o = tok_buf + i * hidden + ch * EPC;
        if (MODE == 3) {        // a further choice of a top-k routing: added to what the choices before it left (fp32 add, one rounding)
          if (!keep) continue;
          const T* s = disp + row * hidden + ch * EPC;
#pragma unroll
          for (int j = 0; j < EPC; ++j) ElemIO<T>::st(o + j, ElemIO<T>::ld(o + j) + gt * ElemIO<T>::ld(s + j));
        } else if (!keep) {
          *(uint4*)o = make_uint4(0, 0, 0, 0);
        } else {
          const T* s = disp + row * hidden + ch * EPC;
#pragma unroll
          for (int j = 0; j < EPC; ++j) {
            float v = gt * ElemIO<T>::ld(s + j);
            if (relu) v = fmaxf(v, 0.f);
            ElemIO<T>::st(o + j, v);
          }
        }
      }
    }
  }
}

// combine backward (fast path): dy = dy_in + dsig * wsig; dy *= (y > 0); dgate = <y, dy> / gate; dout = dy * gate
template <typename T, int H>
__global__ __launch_bounds__(256) void combine_bwd_kernel(const T* __restrict__ dy_in, const T* __restrict__ y,
                                                          const float* __restrict__ dsig, const float* __restrict__ wsig,
                                                          const float* __restrict__ gate, int P, T* __restrict__ dout,
                                                          float* __restrict__ dgate) {
  using R = Row16<T, H>;
  constexpr int VPL = R::VPL;
  const int j = threadIdx.x & 15;
  float ws[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) ws[v] = 0.f;
  if (wsig) R::loadf(wsig, j, ws);
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const long ngr = ((long)gridDim.x * 256) >> 4;
  for (long i = gid; i < P; i += ngr) {
    const float gt = gate[i];
    const float ds = dsig ? dsig[i] : 0.f;
    float yv[VPL], d[VPL], dot = 0.f;
    R::load(y + i * H, j, yv);
    R::load(dy_in + i * H, j, d);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float t = d[v] + ds * ws[v];
      t = yv[v] > 0.f ? t : 0.f;
      dot += yv[v] * t;
      d[v] = t * gt;
    }
    R::store(dout + i * H, j, d);
    dot = sum16(dot);
    if (j == 0) dgate[i] = dot / gt;
  }
}

// ------------------------------------------------------------------------------------------------ heads
// raw[i] = (sigmoid(h2 . Wc[c] + bc[c]) c<3, softplus(y . ws + bs + noise - 1))   models/nerf_moe.py:393-441
template <typename T, int M, int H2>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const T* __restrict__ y, const T* __restrict__ h2,
                                                        const float* __restrict__ ws, const float* __restrict__ bs,
                                                        const float* __restrict__ wc, const float* __restrict__ bc,
                                                        const float* __restrict__ noise, int P, float* __restrict__ raw) {
  using RY = Row16<T, M>;
  using RH = Row16<T, H2>;
  const int j = threadIdx.x & 15;
  float wsv[RY::VPL], wcv[3][RH::VPL];
  RY::loadf(ws, j, wsv);
#pragma unroll
  for (int c = 0; c < 3; ++c) RH::loadf(wc + c * H2, j, wcv[c]);
  const float b_s = bs[0], b0 = bc[0], b1 = bc[1], b2 = bc[2];
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const long ngr = ((long)gridDim.x * 256) >> 4;
  for (long i = gid; i < P; i += ngr) {
    float yv[RY::VPL], hv[RH::VPL];
    RY::load(y + i * M, j, yv);
    RH::load(h2 + i * H2, j, hv);
    float s = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int v = 0; v < RY::VPL; ++v) s += yv[v] * wsv[v];
#pragma unroll
    for (int v = 0; v < RH::VPL; ++v) { c0 += hv[v] * wcv[0][v]; c1 += hv[v] * wcv[1][v]; c2 += hv[v] * wcv[2][v]; }
    s = sum16(s); c0 = sum16(c0); c1 = sum16(c1); c2 = sum16(c2);
    if (j == 0) {
      const float u = s + b_s + (noise ? noise[i] : 0.f) - 1.f;  // ShiftedSoftplus, models/nerf.py:68-69
      float4 o;
      o.x = 1.f / (1.f + expf(-(c0 + b0)));
      o.y = 1.f / (1.f + expf(-(c1 + b1)));
      o.z = 1.f / (1.f + expf(-(c2 + b2)));
      o.w = u > 20.f ? u : log1pf(expf(u));
      *(float4*)(raw + i * 4) = o;
    }
  }
}

// d_raw -> dh2 (masked by h2 > 0), dsig_pre; the block's sums for d_ws, d_wc, d_bs, d_bc -> partial[block].
// rows_per_group > 0 (P a multiple of it; the samples of a ray): a block walks whole groups and also leaves rowsum[group][:] = the
// column sums of the group's dh2 rows AS STORED (rounded to T) - the per-ray bias gradient (swn_group_colsum) without reading dh2 back.
template <typename T, int M, int H2, bool HASY>      // HASY = false: no y, no sigma weight gradient (its 16+ registers are a fifth wave per SIMD)
__global__ __launch_bounds__(256) void heads_bwd_kernel(const T* __restrict__ y, const T* __restrict__ h2,
                                                        const float* __restrict__ wc, const float* __restrict__ raw,
                                                        const float* __restrict__ d_raw, int P, T* __restrict__ dh2,
                                                        float* __restrict__ dsig, float* __restrict__ partial, int rows_per_group,
                                                        float* __restrict__ rowsum) {
  using RY = Row16<T, M>;
  using RH = Row16<T, H2>;
  const int j = threadIdx.x & 15;
  float aws[RY::VPL], awc[3][RH::VPL], abs_ = 0.f, abc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int v = 0; v < RH::VPL; ++v) awc[c][v] = 0.f;
  }
  // the colour weights live in LDS and are read per row: in registers (24 per lane) they cost the kernel its fourth wave per SIMD -
  // one row per 16-lane group is 3 KB in flight per wave, and 12 waves per CU do not cover the memory latency (496 -> 444 us)
  __shared__ float wc_lds[3][16][RH::VPL];      // [colour][lane of the row's 16][value]: the lane's weights in its own value order
  if (threadIdx.x < 16) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t_[RH::VPL];
      RH::loadf(wc + c * H2, threadIdx.x, t_);
#pragma unroll
      for (int v = 0; v < RH::VPL; ++v) wc_lds[c][threadIdx.x][v] = t_[v];
    }
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < RY::VPL; ++v) aws[v] = 0.f;
  float o[RH::VPL];
  // a row's operands as loaded (all loads of a row issued before the first is unpacked; fetching the NEXT row ahead as well was
  // measured: 174-198 registers, two waves per SIMD, 457 us against 444)
  struct RowIn { float4 r, d; uint4 yr[RY::NCH], hr[RH::NCH]; };
  auto fetch = [&](long i, RowIn& in) {
    in.r = *(const float4*)(raw + i * 4);
    in.d = *(const float4*)(d_raw + i * 4);
    if constexpr (HASY) RY::load_raw(y + i * M, j, in.yr);      // (y == NULL: d_ws is formed elsewhere - swn_chain_desc.comb_dwsig - and stays untouched here)
    RH::load_raw(h2 + i * H2, j, in.hr);
  };
  auto row = [&](long i, const RowIn& in) {
    const float4 r = in.r;
    const float4 d = in.d;
    const float dc0 = d.x * r.x * (1.f - r.x), dc1 = d.y * r.y * (1.f - r.y), dc2 = d.z * r.z * (1.f - r.z);
    const float dsp = d.w * -expm1f(-r.w);  // softplus'(u) = sigmoid(u) = 1 - exp(-softplus(u)); expm1: no cancellation for near-empty samples
    if (j == 0) {
      dsig[i] = dsp;
      abs_ += dsp; abc[0] += dc0; abc[1] += dc1; abc[2] += dc2;
    }
    float hv[RH::VPL];
    asm volatile("" ::: "memory");        // (keeps the LDS reads of the colour weights inside the row loop: hoisted they are registers again)
    if constexpr (HASY) {
      float yv[RY::VPL];
      RY::unpack(in.yr, yv);
#pragma unroll
      for (int v = 0; v < RY::VPL; ++v) aws[v] += dsp * yv[v];
    }
    RH::unpack(in.hr, hv);
#pragma unroll
    for (int v = 0; v < RH::VPL; ++v) {
      awc[0][v] += dc0 * hv[v]; awc[1][v] += dc1 * hv[v]; awc[2][v] += dc2 * hv[v];
      const float gg = dc0 * wc_lds[0][j][v] + dc1 * wc_lds[1][j][v] + dc2 * wc_lds[2][j][v];
      o[v] = hv[v] > 0.f ? gg : 0.f;
    }
    RH::store(dh2 + i * H2, j, o);
  };
  // ONE instance of the row body for both walks (two inlined copies were contracted differently by the compiler: last-bit differences
  // in fp32): plain = one unit of P rows, 16-lane group q of block b takes rows 16 b + q, + 16 gridDim, ...; grouped = block b takes
  // the units b, b + gridDim, ..., its 16-lane groups the unit's rows q, q + 16, ...
  __shared__ float cs_lds[16][H2];
  const bool grouped = rows_per_group > 0;
  const int grp = threadIdx.x >> 4;
  const long rows = grouped ? rows_per_group : P, n_units = grouped ? P / rows_per_group : 1;
  const long r0 = grouped ? grp : (((long)blockIdx.x * 256 + threadIdx.x) >> 4), rstep = grouped ? 16 : (((long)gridDim.x * 256) >> 4);
  for (long u = grouped ? blockIdx.x : 0; u < n_units; u += grouped ? gridDim.x : 1) {
    float cs[RH::VPL];
#pragma unroll
    for (int v = 0; v < RH::VPL; ++v) cs[v] = 0.f;
    auto colsum = [&]() {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int v = 0; v < RH::VPL; v += 2) {           // what the store kept: the values rounded to T
          const uint32_t pk = pack_bf16x2(o[v], o[v + 1]);
          cs[v] += bf16_to_f32((bf16_t)(pk & 0xFFFF));
          cs[v + 1] += bf16_to_f32((bf16_t)(pk >> 16));
        }
      } else {
#pragma unroll
        for (int v = 0; v < RH::VPL; ++v) cs[v] += o[v];
      }
    };
    if constexpr (HASY) {
      for (long r = r0; r < rows; r += rstep) {
        RowIn cur_in;
        fetch(u * rows + r, cur_in);
        row(u * rows + r, cur_in);
        colsum();
      }
    } else {       // a row is 288 bytes here: two rows' loads in flight per 16-lane group (same rows, same order: the same bits;
                   // 307 -> 261 us at full size)
      for (long r = r0; r < rows; r += 2 * rstep) {
        RowIn in0, in1;
        const bool two = r + rstep < rows;
        fetch(u * rows + r, in0);
        if (two) fetch(u * rows + r + rstep, in1);
        row(u * rows + r, in0);
        colsum();
        if (two) {
          row(u * rows + r + rstep, in1);
          colsum();
        }
      }
    }
    if (grouped) {
#pragma unroll
      for (int v = 0; v < RH::VPL; ++v) cs_lds[grp][RH::col(j, v)] = cs[v];
      __syncthreads();
      for (int c = threadIdx.x; c < H2; c += 256) {
        float s_ = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s_ += cs_lds[q][c];   // the 16 row stripes in a fixed order
        rowsum[u * H2 + c] = s_;
      }
      __syncthreads();
    }
  }
  // block-level reduction in LDS: the 16 row groups add their sums one after the other (fixed order); the block's partial goes to the
  // workspace with plain stores and ordered_reduce_kernel adds the blocks in order - the same bits on every run (one atomic per
  // parameter per block, as in rounds 1-2, made the head gradients differ in the last bits from run to run)
  __shared__ float red[M + 3 * H2 + 4];
  for (int t = threadIdx.x; t < M + 3 * H2 + 4; t += 256) red[t] = 0.f;
  __syncthreads();
  for (int gq = 0; gq < 16; ++gq) {
    if ((int)(threadIdx.x >> 4) == gq) {
#pragma unroll
      for (int v = 0; v < RY::VPL; ++v) red[RY::col(j, v)] += aws[v];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int v = 0; v < RH::VPL; ++v) red[M + c * H2 + RH::col(j, v)] += awc[c][v];
      if (j == 0) {
        red[M + 3 * H2 + 0] += abs_;
        red[M + 3 * H2 + 1] += abc[0];
        red[M + 3 * H2 + 2] += abc[1];
        red[M + 3 * H2 + 3] += abc[2];
      }
    }
    __syncthreads();
  }
  float* part = partial + (size_t)blockIdx.x * (M + 3 * H2 + 4);     // [M | 3 H2 | b_sigma | b_color(3)]
  for (int t = threadIdx.x; t < M + 3 * H2 + 4; t += 256) part[t] = red[t];
}

// out[g][c] = sum over the group's rows of in[g*R + r][c]
// 256 threads per group: 16-byte pieces of a row across the lanes (coalesced), the block's 256 / (pieces per row) row
// stripes accumulate in registers and meet in LDS.  Needs C * sizeof(T) <= 4096 and a multiple of 16 (checked on the host).
template <typename T>
__global__ __launch_bounds__(256) void group_colsum_kernel(const T* __restrict__ in, int R, int C, float* __restrict__ out) {
  constexpr int EPC = 16 / (int)sizeof(T);
  __shared__ float red[256 * EPC];
  const int g = blockIdx.x;
  const int cpr = C / EPC;                       // pieces per row (<= 256)
  const int stripes = 256 / cpr;                 // row stripes handled concurrently
  const int piece = threadIdx.x % cpr, stripe = threadIdx.x / cpr;
  float acc[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
  if (stripe < stripes) {
    const T* base = in + (long)g * R * C + piece * EPC;
    for (int r = stripe; r < R; r += stripes) {
      const uint4 u = *(const uint4*)(base + (long)r * C);
      if constexpr (sizeof(T) == 2) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] += bf16_to_f32((bf16_t)(w[i] & 0xFFFF));
          acc[2 * i + 1] += bf16_to_f32((bf16_t)(w[i] >> 16));
        }
      } else {
        acc[0] += __uint_as_float(u.x); acc[1] += __uint_as_float(u.y);
        acc[2] += __uint_as_float(u.z); acc[3] += __uint_as_float(u.w);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < EPC; ++j) red[threadIdx.x * EPC + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s_ = 0.f;
    for (int st = 0; st < stripes; ++st) s_ += red[(st * cpr + c / EPC) * EPC + c % EPC];
    out[(long)g * C + c] = s_;
  }
}

// ------------------------------------------------------------------------------------------------ compositing
// one wave per ray; lane owns a contiguous run of ceil(S/64) samples.  rendering.py:435-494
template <int SPL>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                            float last_delta, float rgb_pad, int N, int S, float* __restrict__ rgb,
                                                            float* __restrict__ depth, float* __restrict__ dvar,
                                                            float* __restrict__ weights, const float* __restrict__ last_delta_ray,
                                                            float zsign, const float* __restrict__ depth_src,
                                                            float* __restrict__ bg_lambda) {
  const int lane = threadIdx.x & 63;
  const long ray = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= N) return;
  const float* zr = z + ray * S;
  const float4* rr = (const float4*)(raw + ray * S * 4);
  if (last_delta_ray) last_delta = last_delta_ray[ray];
  float al[SPL], zz[SPL];
  float4 cs[SPL];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    const int s = lane * SPL + j;
    al[j] = 0.f; zz[j] = 0.f; cs[j] = make_float4(0, 0, 0, 0);
    if (s < S) {
      zz[j] = zr[s];
      const float dl = (s + 1 < S) ? zsign * (zr[s + 1] - zz[j]) : last_delta;     // zsign = -1: descending depths (flip)
      cs[j] = rr[s];
      if (rgb_pad != 0.f) {   // rendering_mip.py:383-384: rgbs * (1 + 2 pad) - pad
        cs[j].x = cs[j].x * (1.f + 2.f * rgb_pad) - rgb_pad;
        cs[j].y = cs[j].y * (1.f + 2.f * rgb_pad) - rgb_pad;
        cs[j].z = cs[j].z * (1.f + 2.f * rgb_pad) - rgb_pad;
      }
      al[j] = 1.f - expf(-dl * cs[j].w);
      prod *= (1.f - al[j] + 1e-8f);
    }
  }
  // exclusive multiplicative scan over lanes
  float incl = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(incl, o, 64);
    if (lane >= o) incl *= t;
  }
  float T = __shfl_up(incl, 1, 64);
  if (lane == 0) T = 1.f;
  if (bg_lambda && lane == 63) bg_lambda[ray] = incl;          // transmittance behind the last sample (rendering.py:456-457)
  float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f;
  float w[SPL];
  const float* dsrc = depth_src ? depth_src + ray * S : nullptr;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    w[j] = al[j] * T;
    T *= (1.f - al[j] + 1e-8f);
    const int s = lane * SPL + j;
    const float dj = (dsrc && s < S) ? dsrc[s] : zz[j];        // depth map over the metric depths (:483-484)
    ar += w[j] * cs[j].x; ag += w[j] * cs[j].y; ab += w[j] * cs[j].z; ad += w[j] * dj;
    if (weights && s < S) weights[ray * S + s] = w[j];
  }
  ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); ad = wave_sum(ad);
  float av = 0.f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) av += w[j] * (zz[j] - ad) * (zz[j] - ad);
  av = wave_sum(av);
  if (lane == 0) {
    if (rgb) { rgb[ray * 3] = ar; rgb[ray * 3 + 1] = ag; rgb[ray * 3 + 2] = ab; }
    if (depth) depth[ray] = ad;
    if (dvar) dvar[ray] = av;
  }
}

template <int SPL>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                            float last_delta, float rgb_pad, const float* __restrict__ d_rgb, int N, int S,
                                                            float* __restrict__ d_raw, const float* __restrict__ last_delta_ray,
                                                            float zsign, const float* __restrict__ d_bg_lambda) {
  const int lane = threadIdx.x & 63;
  const long ray = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= N) return;
  const float* zr = z + ray * S;
  const float4* rr = (const float4*)(raw + ray * S * 4);
  if (last_delta_ray) last_delta = last_delta_ray[ray];
  const float g0 = d_rgb[ray * 3], g1 = d_rgb[ray * 3 + 1], g2 = d_rgb[ray * 3 + 2];
  float al[SPL], dl[SPL], cg[SPL];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    // (branch-free: with the per-sample values assigned inside `if (s < S)` the 16-sample instantiation - merged coarse + fine rays -
    //  carried its five arrays through every branch as whole vectors: 1900 accumulator-register moves, 842 spilled registers, 373 us
    //  per 8192 x 768 samples against 96 for the forward kernel)
    const int s = lane * SPL + j;
    const bool ok = s < S;
    const int sc = ok ? s : S - 1;
    const float zc = zr[sc];
    const float dlj = (sc + 1 < S) ? zsign * (zr[sc + 1] - zc) : last_delta;
    float4 c = rr[sc];
    if (rgb_pad != 0.f) {
      c.x = c.x * (1.f + 2.f * rgb_pad) - rgb_pad;
      c.y = c.y * (1.f + 2.f * rgb_pad) - rgb_pad;
      c.z = c.z * (1.f + 2.f * rgb_pad) - rgb_pad;
    }
    const float alj = 1.f - expf(-dlj * c.w);
    dl[j] = ok ? dlj : 0.f;
    al[j] = ok ? alj : 0.f;
    cg[j] = ok ? c.x * g0 + c.y * g1 + c.z * g2 : 0.f;
    prod *= ok ? (1.f - alj + 1e-8f) : 1.f;
  }
  float incl = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(incl, o, 64);
    if (lane >= o) incl *= t;
  }
  float T = __shfl_up(incl, 1, 64);
  if (lane == 0) T = 1.f;
  float Ts[SPL], u[SPL], usum = 0.f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    Ts[j] = T;
    u[j] = al[j] * T * cg[j];  // w_j * (c_j . g)
    usum += u[j];
    T *= (1.f - al[j] + 1e-8f);
  }
  // suffix sum over lanes: S_lane = sum of usum over lanes > lane
  float incl_s = usum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(incl_s, o, 64);
    if (lane + o < 64) incl_s += t;
  }
  float suffix = incl_s - usum;  // strictly after this lane
  // bg_lambda = prod_i (1 - alpha_i + 1e-8): d bg_lambda / d alpha_j = -bg_lambda / (1 - alpha_j + 1e-8), i.e. one more
  // term "behind the last sample" of the suffix sum
  if (d_bg_lambda) suffix += d_bg_lambda[ray] * __shfl(incl, 63, 64);
#pragma unroll
  for (int j = SPL - 1; j >= 0; --j) {
    const int s = lane * SPL + j;
    if (s < S) {
      // d/d alpha_j = T_j (c_j.g) - (sum_{i>j} u_i) / (1 - alpha_j + 1e-8)
      const float dalpha = Ts[j] * cg[j] - suffix / (1.f - al[j] + 1e-8f);
      const float dsigma = dalpha * dl[j] * (1.f - al[j]);  // d alpha / d sigma = delta * exp(-delta sigma)
      const float wj = al[j] * Ts[j] * (1.f + 2.f * rgb_pad);
      *(float4*)(d_raw + (ray * S + s) * 4) = make_float4(wj * g0, wj * g1, wj * g2, dsigma);
    }
    suffix += u[j];
    if constexpr (SPL >= 16) __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------ optimiser
template <typename T>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, T* __restrict__ shadow, long n, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2s, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;  // torch.optim.Adam: (sqrt(v) / sqrt(bias_correction2)) + eps
    const float pi = p[i] - (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (shadow) ElemIO<T>::st(shadow + i, pi);
  }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    ElemIO<T>::st(out + i, in[i]);
}

template <typename T>
__global__ void cast_transpose_kernel(const float* __restrict__ in, T* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const float* src = in + (long)blockIdx.z * rows * cols;
  T* dst = out + (long)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) ElemIO<T>::st(dst + (long)c * rows + r, tile[threadIdx.x][j]);
  }
}

// dst[r] = src[index[r]] (16-byte pieces), zero rows for index < 0: the send buffer of the expert-parallel exchange
__global__ void gather_rows_kernel(const char* __restrict__ src, const int32_t* __restrict__ index, long n_rows, int row_bytes,
                                   char* __restrict__ dst) {
  const int cpr = row_bytes >> 4;
  const long total = n_rows * cpr;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const long r = c / cpr;
    const int ch = (int)(c - r * cpr);
    const int s_ = index[r];
    uint4 v = make_uint4(0, 0, 0, 0);
    if (s_ >= 0) v = *(const uint4*)(src + (long)s_ * row_bytes + ch * 16);
    *(uint4*)(dst + c * 16) = v;
  }
}

}  // namespace swn

using namespace swn;

extern "C" const char* swn_last_error(void) { return g_err; }
extern "C" int swn_version(void) { return 2; }
extern "C" int swn_half_dtype(void) { return SWN_HALF; }      /* the 16-bit compute type of this build: SWN_BF16 or SWN_F16 */

extern "C" int swn_mfma_probe(int32_t* out, void* stream) {
  SWN_CHECK(out, "swn_mfma_probe: null");
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), out);
  SWN_LAUNCH_CHECK();
  return 0;
}

static inline int ew_blocks(long waves_needed) {
  long b = (waves_needed + 3) / 4;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int swn_sample_pe(const float* rays, const float* t_steps, const float* perturb_rand, float perturb,
                             int n_rays, int n_samples, int l_xyz, int l_dir, int dtype, float* z_out, void* pe_xyz,
                             int pe_stride, void* pe_dir, int dir_stride, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_sample_pe: bad dtype");
  SWN_CHECK(rays && t_steps && z_out && pe_xyz, "swn_sample_pe: null pointer");
  SWN_CHECK(l_xyz >= 0 && l_xyz <= 12 && l_dir >= 0 && l_dir <= 12, "swn_sample_pe: frequencies must be <= 12");
  const int epc = dtype == SWN_HALF ? 8 : 4;
  SWN_CHECK(pe_stride >= 3 + 6 * l_xyz && pe_stride % epc == 0, "swn_sample_pe: pe_stride %d too small / unaligned", pe_stride);
  const long P = (long)n_rays * n_samples;
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((sample_pe_kernel<bf16_t, 12>), dim3(cdiv(P, 128)), dim3(128), 0, as_stream(stream), rays, t_steps,
                       perturb_rand, perturb, n_rays, n_samples, l_xyz, z_out, (bf16_t*)pe_xyz, pe_stride, (const float*)nullptr);
  else
    hipLaunchKernelGGL((sample_pe_kernel<float, 12>), dim3(cdiv(P, 64)), dim3(64), 0, as_stream(stream), rays, t_steps,
                       perturb_rand, perturb, n_rays, n_samples, l_xyz, z_out, (float*)pe_xyz, pe_stride, (const float*)nullptr);
  SWN_LAUNCH_CHECK();
  if (pe_dir) {
    SWN_CHECK(dir_stride >= 3 + 6 * l_dir, "swn_sample_pe: dir_stride too small");
    if (dtype == SWN_HALF)
      hipLaunchKernelGGL((dir_pe_kernel<bf16_t, 12>), dim3(cdiv(n_rays, 256)), dim3(256), 0, as_stream(stream), rays,
                         n_rays, l_dir, (bf16_t*)pe_dir, dir_stride);
    else
      hipLaunchKernelGGL((dir_pe_kernel<float, 12>), dim3(cdiv(n_rays, 256)), dim3(256), 0, as_stream(stream), rays,
                         n_rays, l_dir, (float*)pe_dir, dir_stride);
    SWN_LAUNCH_CHECK();
  }
  return 0;
}

/* positional encoding of xyz = o + d * z for caller-supplied depths z[N,S] (the fine pass: rendering.py:246 xyz_fine_fn) */
extern "C" int swn_pe_from_z(const float* rays, const float* z, int n_rays, int n_samples, int l_xyz, int dtype, void* pe_xyz,
                             int pe_stride, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_pe_from_z: bad dtype");
  SWN_CHECK(rays && z && pe_xyz, "swn_pe_from_z: null pointer");
  SWN_CHECK(l_xyz >= 0 && l_xyz <= 12, "swn_pe_from_z: frequencies must be <= 12");
  const int epc = dtype == SWN_HALF ? 8 : 4;
  SWN_CHECK(pe_stride >= 3 + 6 * l_xyz && pe_stride % epc == 0, "swn_pe_from_z: pe_stride %d too small / unaligned", pe_stride);
  const long P = (long)n_rays * n_samples;
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((sample_pe_kernel<bf16_t, 12>), dim3(cdiv(P, 128)), dim3(128), 0, as_stream(stream), rays, (const float*)nullptr,
                       (const float*)nullptr, 0.f, n_rays, n_samples, l_xyz, (float*)nullptr, (bf16_t*)pe_xyz, pe_stride, z);
  else
    hipLaunchKernelGGL((sample_pe_kernel<float, 12>), dim3(cdiv(P, 64)), dim3(64), 0, as_stream(stream), rays, (const float*)nullptr,
                       (const float*)nullptr, 0.f, n_rays, n_samples, l_xyz, (float*)nullptr, (float*)pe_xyz, pe_stride, z);
  SWN_LAUNCH_CHECK();
  return 0;
}

#ifndef SWN_GATE_TB512
#define SWN_GATE_TB512 1          // rows per 16-lane group and pass of the 512-feature router FORWARD kernel (LDS weight reads per row / TB).
#endif                            // Measured on 852 k rows x 16 experts: 1 = 0.74 ms (163 registers, 3 waves per SIMD), 2 = 1.07, 4 = 1.02
#define GATE_DISPATCH_TB(T, KERNEL, TBV, ...)                                                                      \
  do {                                                                                                       \
    const bool e8 = n_experts <= 8;                                                                          \
    if (gate_dim == 256 && e8) hipLaunchKernelGGL((KERNEL<T, 256, 8>), __VA_ARGS__);                         \
    else if (gate_dim == 256) hipLaunchKernelGGL((KERNEL<T, 256, 16>), __VA_ARGS__);                         \
    else if (gate_dim == 128 && e8) hipLaunchKernelGGL((KERNEL<T, 128, 8>), __VA_ARGS__);                    \
    else if (gate_dim == 512) hipLaunchKernelGGL((KERNEL<T, 512, 16, TBV>), __VA_ARGS__);                    \
    else return swn::set_error("gate: unsupported gate_dim %d / experts %d", gate_dim, n_experts);          \
  } while (0)

static inline int row_blocks(long rows) {   // 16 rows per 256-thread block iteration
  long b = (rows + 15) / 16;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

static int gate_fwd_impl(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg, const float* noise, float noise_scale,
                         int n_tokens, int gate_dim, int n_experts, float* gates, int32_t* idx, float* gmax, float* stats, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_gate_fwd: bad dtype");
  SWN_CHECK(g && wg && gates && idx && gmax, "swn_gate_fwd: null pointer");
  SWN_CHECK(n_experts >= 1 && n_experts <= 16, "swn_gate_fwd: experts <= 16");
  SWN_CHECK((ln_w == nullptr) == (ln_b == nullptr), "swn_gate_fwd: ln_w / ln_b must both be given or both NULL");
  if (ln_w) SWN_CHECK(stats, "swn_gate_fwd: stats required with LayerNorm");
  const int blocks = row_blocks(n_tokens);
  // 16-bit rows of 256, up to 8 experts: the contraction on the matrix pipe (gate_mfma.hip; SWN_GATE_VALU=1 keeps the VALU kernel: A/B runs;
  // a forward with gate noise runs on the VALU kernel)
  static const bool valu_only = getenv("SWN_GATE_VALU") != nullptr;
  if (dtype == SWN_HALF && gate_dim == 256 && n_experts <= 8 && !valu_only && !noise)
    return swn::gate_fwd_mfma_launch(g, ln_w, ln_b, wg, n_tokens, n_experts, gates, idx, gmax, stats, stream);
  // ... and rows of 512 (Mission Bay's router: up to 16 experts)
  if (dtype == SWN_HALF && gate_dim == 512 && !valu_only && !noise)
    return swn::gate_fwd_wide_launch(g, ln_w, ln_b, wg, n_tokens, n_experts, gates, idx, gmax, stats, stream);
  if (dtype == SWN_HALF) {
    const bf16_t* gp = (const bf16_t*)g;
    GATE_DISPATCH_TB(bf16_t, gate_fwd_kernel, SWN_GATE_TB512, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, n_tokens, n_experts,
                  gates, idx, gmax, stats, noise, noise_scale);
  } else {
    const float* gp = (const float*)g;
    GATE_DISPATCH_TB(float, gate_fwd_kernel, 1, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, n_tokens, n_experts,
                  gates, idx, gmax, stats, noise, noise_scale);
  }
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_gate_fwd(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                            int n_tokens, int gate_dim, int n_experts, float* gates, int32_t* idx, float* gmax,
                            float* stats, void* stream) {
  return gate_fwd_impl(g, dtype, ln_w, ln_b, wg, nullptr, 0.f, n_tokens, gate_dim, n_experts, gates, idx, gmax, stats, stream);
}

extern "C" int swn_gate_fwd_noise(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg, const float* noise,
                                  float noise_scale, int n_tokens, int gate_dim, int n_experts, float* gates, int32_t* idx, float* gmax,
                                  float* stats, void* stream) {
  SWN_CHECK(noise, "swn_gate_fwd_noise: null noise");
  return gate_fwd_impl(g, dtype, ln_w, ln_b, wg, noise, noise_scale, n_tokens, gate_dim, n_experts, gates, idx, gmax, stats, stream);
}

#define DWG_LAUNCH(T, EV, GV, GP)                                                                                  \
  hipLaunchKernelGGL((gate_dwg_kernel<T, EV, GV>), dim3(dwg_blocks), dim3(256), 0, as_stream(stream), GP, ln_w, ln_b, stats, \
                     dlogits, n_tokens, tpb, dwg_partial)
#define DWG_DISPATCH(T, GP)                                                                                         \
  do {                                                                                                              \
    if (gate_dim == 256 && n_experts == 8) DWG_LAUNCH(T, 8, 256, GP);                                               \
    else if (gate_dim == 256 && n_experts == 16) DWG_LAUNCH(T, 16, 256, GP);                                        \
    else if (gate_dim == 256 && n_experts == 4) DWG_LAUNCH(T, 4, 256, GP);                                          \
    else if (gate_dim == 128 && n_experts == 8) DWG_LAUNCH(T, 8, 128, GP);                                          \
    else if (gate_dim == 128 && n_experts == 4) DWG_LAUNCH(T, 4, 128, GP);                                          \
    else if (gate_dim == 512 && n_experts == 16) DWG_LAUNCH(T, 16, 512, GP);                                        \
    else if (gate_dim == 512 && n_experts == 8) DWG_LAUNCH(T, 8, 512, GP);                                          \
    else return swn::set_error("gate dW: unsupported gate_dim %d / experts %d", gate_dim, n_experts);              \
  } while (0)

static int gate_dwg_tokens_per_block(int n_tokens) {
  int tpb = cdiv(n_tokens, 1024);
  return ((tpb < 256 ? 256 : (tpb > 4096 ? 4096 : tpb)) + 31) / 32 * 32;
}

extern "C" size_t swn_gate_bwd_scratch_floats(int n_tokens, int gate_dim, int n_experts) {   // dlogits | per-block partials | their sum
  const size_t ps = (size_t)n_experts * gate_dim + n_experts;
  size_t blocks = (size_t)cdiv(n_tokens, gate_dwg_tokens_per_block(n_tokens));
  if ((size_t)swn::gate_bwd_mfma_blocks(n_tokens) > blocks) blocks = (size_t)swn::gate_bwd_mfma_blocks(n_tokens);
  if (gate_dim == 512 && (size_t)swn::gate_dwg_wide_blocks(n_tokens) > blocks) blocks = (size_t)swn::gate_dwg_wide_blocks(n_tokens);
  return (size_t)n_tokens * n_experts + (blocks + 1) * ps;
}

static int gate_bwd_impl(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                         const float* gates, const int32_t* idx, const float* d_gmax, const float* stats,
                         const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int gate_dim,
                         int n_experts, void* dg, float* dlogits, float* d_wg, float* d_ln_w, float* d_ln_b, void* stream,
                         const float* d_probs, const float* d_logits_add = nullptr) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_gate_bwd: bad dtype");
  SWN_CHECK(g && wg && gates && idx && dg && d_wg && counts && dlogits, "swn_gate_bwd: null pointer");
  SWN_CHECK(n_experts >= 1 && n_experts <= 16 && seg_tokens > 0, "swn_gate_bwd: bad sizes");
  if (ln_w) SWN_CHECK(stats && d_ln_w && d_ln_b && ln_b, "swn_gate_bwd: LayerNorm buffers missing");
  int blocks = row_blocks(n_tokens);
  if (blocks > 2048) blocks = 2048;
  // router weight gradient: a block walks its tokens serially (32 per iteration), so its run time is set by tokens per block, not by
  // the batch - ~1024 blocks (4 per CU) whatever the batch size (4096 tokens per block took 0.5 ms for 2M and for 262144 tokens alike)
  const int tpb = gate_dwg_tokens_per_block(n_tokens);
  int dwg_blocks = cdiv(n_tokens, tpb);
  float* dwg_partial = dlogits + (size_t)n_tokens * n_experts;          // second part of the scratch: [dwg_blocks][E * G + E]
  const int ps = n_experts * gate_dim + n_experts;
  float* msum = dwg_partial + (size_t)dwg_blocks * ps;                  // third part: [E * G + E]
  static const bool valu_env = getenv("SWN_GATE_VALU") != nullptr;
  const bool valu_only = valu_env || d_probs != nullptr || d_logits_add != nullptr;       // (the dense operand exists in the VALU kernel only: top-k layers)
  const bool mfma_path = dtype == SWN_HALF && gate_dim == 256 && n_experts <= 8 && !valu_only;
  if (dtype == SWN_HALF) {
    const bf16_t* gp = (const bf16_t*)g;
    bf16_t* dgp = (bf16_t*)dg;
    if (mfma_path) {       // data path AND the per-block sums of the parameter gradients on the matrix pipe (gate_mfma.hip)
      dwg_blocks = swn::gate_bwd_mfma_blocks(n_tokens);
      msum = dwg_partial + (size_t)dwg_blocks * ps;
      const int rc = swn::gate_bwd_mfma_launch(g, ln_w, wg, gates, idx, d_gmax, stats, counts, laux_coef, seg_tokens, n_tokens, n_experts, dg,
                                               dlogits, dwg_partial, stream);
      if (rc) return rc;
    } else if (gate_dim == 512 && !valu_only) {      // 512-feature rows: data path and parameter-gradient sums on the matrix pipe
      int rc = swn::gate_bwd_wide_launch(g, ln_w, wg, gates, idx, d_gmax, stats, counts, laux_coef, seg_tokens, n_tokens, n_experts, dg,
                                         dlogits, stream);
      if (rc) return rc;
      dwg_blocks = swn::gate_dwg_wide_blocks(n_tokens);
      msum = dwg_partial + (size_t)dwg_blocks * ps;
      rc = swn::gate_dwg_wide_launch(g, ln_w != nullptr, stats, dlogits, n_tokens, n_experts, dwg_partial, stream);
      if (rc) return rc;
    } else
    GATE_DISPATCH_TB(bf16_t, gate_bwd_kernel, 1, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, gates, idx, d_gmax,
                  stats, counts, laux_coef, seg_tokens, n_tokens, n_experts, dgp, dlogits, d_ln_w, d_ln_b, d_probs, d_logits_add);
    if (!mfma_path && !(gate_dim == 512 && !valu_only)) {
      SWN_CHECK(n_experts == 4 || n_experts == 8 || n_experts == 16, "swn_gate_bwd: experts must be 4, 8 or 16");
      DWG_DISPATCH(bf16_t, gp);
    }
  } else {
    const float* gp = (const float*)g;
    float* dgp = (float*)dg;
    GATE_DISPATCH_TB(float, gate_bwd_kernel, 1, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, gates, idx, d_gmax,
                  stats, counts, laux_coef, seg_tokens, n_tokens, n_experts, dgp, dlogits, d_ln_w, d_ln_b, d_probs, d_logits_add);
    SWN_CHECK(n_experts == 4 || n_experts == 8 || n_experts == 16, "swn_gate_bwd: experts must be 4, 8 or 16");
    DWG_DISPATCH(float, gp);
  }
  {
    OrdDst od{{msum, nullptr, nullptr, nullptr}, {ps, 0, 0, 0}};
    ordered_reduce_async(dwg_partial, dwg_blocks, ps, od, false, as_stream(stream));
    hipLaunchKernelGGL(gate_dwg_finalize_kernel, dim3(1), dim3(gate_dim), 0, as_stream(stream), msum, n_experts, gate_dim, ln_w, ln_b, wg,
                       d_wg, d_ln_w, d_ln_b);
  }
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_gate_bwd(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                            const float* gates, const int32_t* idx, const float* d_gmax, const float* stats,
                            const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int gate_dim,
                            int n_experts, void* dg, float* dlogits, float* d_wg, float* d_ln_w, float* d_ln_b, void* stream) {
  return gate_bwd_impl(g, dtype, ln_w, ln_b, wg, gates, idx, d_gmax, stats, counts, laux_coef, seg_tokens, n_tokens, gate_dim, n_experts, dg,
                       dlogits, d_wg, d_ln_w, d_ln_b, stream, nullptr);
}
// ... with a dense gradient d_probs [n_tokens, n_experts] w.r.t. the softmax probabilities on top (the normalised gates of a top-k layer,
// swn_topk_gate_bwd); d_gmax may be NULL then
extern "C" int swn_gate_bwd_dense(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                                  const float* gates, const int32_t* idx, const float* d_gmax, const float* d_probs,
                                  const float* d_logits_add, const float* stats,
                                  const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int gate_dim,
                                  int n_experts, void* dg, float* dlogits, float* d_wg, float* d_ln_w, float* d_ln_b, void* stream) {
  SWN_CHECK(d_probs || d_logits_add, "swn_gate_bwd_dense: no dense operand (swn_gate_bwd is the entry point without one)");
  return gate_bwd_impl(g, dtype, ln_w, ln_b, wg, gates, idx, d_gmax, stats, counts, laux_coef, seg_tokens, n_tokens, gate_dim, n_experts, dg,
                       dlogits, d_wg, d_ln_w, d_ln_b, stream, d_probs, d_logits_add);
}

template <int MODE>
static int launch_sparse(const float* gates, const int32_t* idx, const int32_t* loc, void* tok, void* disp, float* dgate,
                         int dtype, int samples, int hidden, int capacity, int seg_tokens, int n_experts, int relu,
                         void* stream, const int32_t* begin = nullptr) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "sparse: bad dtype");
  SWN_CHECK(idx && loc && tok && disp, "sparse: null pointer");
  SWN_CHECK(hidden * (dtype == SWN_HALF ? 2 : 4) % 16 == 0, "sparse: hidden row must be a multiple of 16 bytes");
  SWN_CHECK(capacity > 0 && seg_tokens > 0 && n_experts > 0, "sparse: bad sizes");
  const int blocks = ew_blocks(samples);
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((sparse_kernel<bf16_t, MODE>), dim3(blocks), dim3(256), 0, as_stream(stream), gates, idx, loc,
                       (bf16_t*)tok, (bf16_t*)disp, dgate, samples, hidden, capacity, seg_tokens, n_experts, relu, begin);
  else
    hipLaunchKernelGGL((sparse_kernel<float, MODE>), dim3(blocks), dim3(256), 0, as_stream(stream), gates, idx, loc,
                       (float*)tok, (float*)disp, dgate, samples, hidden, capacity, seg_tokens, n_experts, relu, begin);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_dispatch_fwd(const float* gates, const int32_t* indices, const int32_t* locations,
                                const void* reshaped_input, void* dispatched, int dtype, int samples, int hidden,
                                int capacity, int n_experts, void* stream) {
  SWN_CHECK(dispatched, "swn_dispatch_fwd: null");
  const size_t bytes = (size_t)n_experts * capacity * hidden * (dtype == SWN_HALF ? 2 : 4);
  hipError_t e = fill_u32_async(dispatched, 0u, bytes, as_stream(stream));  // torch.zeros at tutel_fast_dispatch.py:25
  SWN_CHECK(e == hipSuccess, "swn_dispatch_fwd: memset failed: %s", hipGetErrorString(e));
  return launch_sparse<0>(gates, indices, locations, (void*)reshaped_input, dispatched, nullptr, dtype, samples, hidden,
                          capacity, samples, n_experts, 0, stream);
}
extern "C" int swn_dispatch_bwd_data(const float* gates, const int32_t* indices, const int32_t* locations,
                                     void* grad_reshaped_input, const void* dispatched, int dtype, int samples,
                                     int hidden, int capacity, void* stream) {
  return launch_sparse<1>(gates, indices, locations, grad_reshaped_input, (void*)dispatched, nullptr, dtype, samples,
                          hidden, capacity, samples, 1 << 20, 0, stream);
}
// top-k (k > 1): the second and later iterations of the reference's `for g, i, l in zip(gates, indices, locations)` loops
// (tutel_fast_dispatch.py:26-27 / :69-70: further rows of the SAME dispatched buffer - no zero fill; :34-37 / :59-62: `last_result + ...`)
extern "C" int swn_dispatch_fwd_more(const float* gates, const int32_t* indices, const int32_t* locations,
                                     const void* reshaped_input, void* dispatched, int dtype, int samples, int hidden,
                                     int capacity, int n_experts, void* stream) {
  SWN_CHECK(dispatched, "swn_dispatch_fwd_more: null");
  return launch_sparse<0>(gates, indices, locations, (void*)reshaped_input, dispatched, nullptr, dtype, samples, hidden,
                          capacity, samples, n_experts, 0, stream);
}
extern "C" int swn_dispatch_bwd_data_more(const float* gates, const int32_t* indices, const int32_t* locations,
                                          void* grad_reshaped_input, const void* dispatched, int dtype, int samples,
                                          int hidden, int capacity, void* stream) {
  return launch_sparse<3>(gates, indices, locations, grad_reshaped_input, (void*)dispatched, nullptr, dtype, samples,
                          hidden, capacity, samples, 1 << 20, 0, stream);
}
extern "C" int swn_dispatch_bwd_gate(float* grad_gates, const int32_t* indices, const int32_t* locations,
                                     const void* reshaped_input, const void* dispatched, int dtype, int samples,
                                     int hidden, int capacity, void* stream) {
  SWN_CHECK(grad_gates, "swn_dispatch_bwd_gate: null");
  return launch_sparse<2>(nullptr, indices, locations, (void*)reshaped_input, (void*)dispatched, grad_gates, dtype,
                          samples, hidden, capacity, samples, 1 << 20, 0, stream);
}

// ---- no-batch (evaluation) kernel ABI: the reference's argument order with expert_locations_begin as 4th argument
// (tutel_fast_dispatch_nobatch.py:36, :47, :53, :73, :87, :93).  n_experts: begin has n_experts entries per `samples` tokens.
extern "C" int swn_dispatch_nobatch_fwd(const float* gates, const int32_t* indices, const int32_t* locations,
                                        const int32_t* expert_locations_begin, const void* reshaped_input, void* dispatched, int dtype,
                                        int samples, int hidden, int capacity, int n_experts, long dispatched_rows, void* stream) {
  SWN_CHECK(dispatched && expert_locations_begin, "swn_dispatch_nobatch_fwd: null");
  const size_t bytes = (size_t)dispatched_rows * hidden * (dtype == SWN_HALF ? 2 : 4);
  hipError_t e = fill_u32_async(dispatched, 0u, bytes, as_stream(stream));  // torch.zeros at tutel_fast_dispatch_nobatch.py:34
  SWN_CHECK(e == hipSuccess, "swn_dispatch_nobatch_fwd: memset failed: %s", hipGetErrorString(e));
  return launch_sparse<0>(gates, indices, locations, (void*)reshaped_input, dispatched, nullptr, dtype, samples, hidden,
                          capacity > 0 ? capacity : 1, samples, n_experts, 0, stream, expert_locations_begin);
}
extern "C" int swn_dispatch_nobatch_bwd_data(const float* gates, const int32_t* indices, const int32_t* locations,
                                             const int32_t* expert_locations_begin, void* grad_reshaped_input, const void* dispatched,
                                             int dtype, int samples, int hidden, int capacity, int n_experts, void* stream) {
  SWN_CHECK(expert_locations_begin, "swn_dispatch_nobatch_bwd_data: null");
  return launch_sparse<1>(gates, indices, locations, grad_reshaped_input, (void*)dispatched, nullptr, dtype, samples,
                          hidden, capacity > 0 ? capacity : 1, samples, n_experts, 0, stream, expert_locations_begin);
}
extern "C" int swn_dispatch_nobatch_bwd_gate(float* grad_gates, const int32_t* indices, const int32_t* locations,
                                             const int32_t* expert_locations_begin, const void* reshaped_input, const void* dispatched,
                                             int dtype, int samples, int hidden, int capacity, int n_experts, void* stream) {
  SWN_CHECK(grad_gates && expert_locations_begin, "swn_dispatch_nobatch_bwd_gate: null");
  return launch_sparse<2>(nullptr, indices, locations, (void*)reshaped_input, (void*)dispatched, grad_gates, dtype,
                          samples, hidden, capacity > 0 ? capacity : 1, samples, n_experts, 0, stream, expert_locations_begin);
}

extern "C" int swn_combine_fwd(const float* gates, const int32_t* indices, const int32_t* locations, void* y,
                               const void* expert_out, int dtype, int samples, int hidden, int capacity, int seg_tokens,
                               int n_experts, int relu, void* stream) {
  return launch_sparse<1>(gates, indices, locations, y, (void*)expert_out, nullptr, dtype, samples, hidden, capacity,
                          seg_tokens, n_experts, relu, stream);
}

extern "C" int swn_combine_bwd(const void* dy_in, const void* y, const float* dsig, const float* wsig, const float* gate,
                               int dtype, int samples, int hidden, void* dout, float* dgate, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_combine_bwd: bad dtype");
  SWN_CHECK(dy_in && y && gate && dout && dgate, "swn_combine_bwd: null pointer");
  SWN_CHECK(hidden == 256 || hidden == 128 || hidden == 512, "swn_combine_bwd: hidden must be 128, 256 or 512");
  int blocks = row_blocks(samples);
#define SWN_CB(T, H) hipLaunchKernelGGL((combine_bwd_kernel<T, H>), dim3(blocks), dim3(256), 0, as_stream(stream), (const T*)dy_in, \
                                        (const T*)y, dsig, wsig, gate, samples, (T*)dout, dgate)
  if (dtype == SWN_HALF) { if (hidden == 256) SWN_CB(bf16_t, 256); else if (hidden == 128) SWN_CB(bf16_t, 128); else SWN_CB(bf16_t, 512); }
  else { if (hidden == 256) SWN_CB(float, 256); else if (hidden == 128) SWN_CB(float, 128); else SWN_CB(float, 512); }
#undef SWN_CB
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_heads_fwd(const void* y, const void* h2, int dtype, const float* w_sigma, const float* b_sigma,
                             const float* w_color, const float* b_color, const float* sigma_noise, int n_points,
                             int model_dim, int h2_dim, float* raw, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_heads_fwd: bad dtype");
  SWN_CHECK(y && h2 && w_sigma && b_sigma && w_color && b_color && raw, "swn_heads_fwd: null pointer");
  SWN_CHECK((model_dim == 256 || model_dim == 512) && (h2_dim == 128 || h2_dim == 256), "swn_heads_fwd: model_dim in {256,512}, h2_dim in {128,256}");
  const int blocks = row_blocks(n_points);
#define SWN_HF(T, M_, H_) hipLaunchKernelGGL((heads_fwd_kernel<T, M_, H_>), dim3(blocks), dim3(256), 0, as_stream(stream), (const T*)y, \
                                             (const T*)h2, w_sigma, b_sigma, w_color, b_color, sigma_noise, n_points, raw)
  if (dtype == SWN_HALF) {
    if (model_dim == 256 && h2_dim == 128) SWN_HF(bf16_t, 256, 128); else if (model_dim == 512 && h2_dim == 256) SWN_HF(bf16_t, 512, 256);
    else if (model_dim == 256) SWN_HF(bf16_t, 256, 256); else SWN_HF(bf16_t, 512, 128);
  } else {
    if (model_dim == 256 && h2_dim == 128) SWN_HF(float, 256, 128); else if (model_dim == 512 && h2_dim == 256) SWN_HF(float, 512, 256);
    else if (model_dim == 256) SWN_HF(float, 256, 256); else SWN_HF(float, 512, 128);
  }
#undef SWN_HF
  SWN_LAUNCH_CHECK();
  return 0;
}

static int heads_bwd_blocks(int n_points) {
  const int blocks = row_blocks(n_points);
  return blocks > 1024 ? 1024 : blocks;
}

extern "C" size_t swn_heads_bwd_workspace_bytes(int n_points, int model_dim, int h2_dim) {
  return (size_t)heads_bwd_blocks(n_points) * (size_t)(model_dim + 3 * h2_dim + 4) * sizeof(float);
}

extern "C" int swn_heads_bwd(const void* y, const void* h2, int dtype, const float* w_color, const float* raw,
                             const float* d_raw, int n_points, int model_dim, int h2_dim, void* dh2, float* dsig,
                             float* d_w_sigma, float* d_b_sigma, float* d_w_color, float* d_b_color, int rows_per_group,
                             float* group_colsum, void* workspace, size_t workspace_bytes, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_heads_bwd: bad dtype");
  SWN_CHECK(h2 && w_color && raw && d_raw && dh2 && dsig && d_w_sigma && d_b_sigma && d_w_color && d_b_color && workspace,
            "swn_heads_bwd: null pointer");      // (y may be NULL: d_w_sigma is left as it is - swn_chain_desc.comb_dwsig forms it)
  SWN_CHECK((model_dim == 256 || model_dim == 512) && (h2_dim == 128 || h2_dim == 256), "swn_heads_bwd: model_dim in {256,512}, h2_dim in {128,256}");
  SWN_CHECK(workspace_bytes >= swn_heads_bwd_workspace_bytes(n_points, model_dim, h2_dim), "swn_heads_bwd: workspace of %zu bytes, need %zu",
            workspace_bytes, swn_heads_bwd_workspace_bytes(n_points, model_dim, h2_dim));
  SWN_CHECK(rows_per_group >= 0 && (rows_per_group == 0 || (group_colsum && n_points % rows_per_group == 0)),
            "swn_heads_bwd: rows_per_group %d must divide n_points %d (and group_colsum be given)", rows_per_group, n_points);
  if (n_points <= 0) return 0;
  int blocks = heads_bwd_blocks(n_points);
  if (rows_per_group > 0 && n_points / rows_per_group < blocks) blocks = n_points / rows_per_group;     // (never more than the workspace was sized for)
  float* partial = (float*)workspace;
#define SWN_HB1(T, M_, H_, Y_) hipLaunchKernelGGL((heads_bwd_kernel<T, M_, H_, Y_>), dim3(blocks), dim3(256), 0, as_stream(stream), (const T*)y, \
                                             (const T*)h2, w_color, raw, d_raw, n_points, (T*)dh2, dsig, partial, rows_per_group, group_colsum)
#define SWN_HB(T, M_, H_) do { if (y) SWN_HB1(T, M_, H_, true); else SWN_HB1(T, M_, H_, false); } while (0)
  if (dtype == SWN_HALF) {
    if (model_dim == 256 && h2_dim == 128) SWN_HB(bf16_t, 256, 128); else if (model_dim == 512 && h2_dim == 256) SWN_HB(bf16_t, 512, 256);
    else if (model_dim == 256) SWN_HB(bf16_t, 256, 256); else SWN_HB(bf16_t, 512, 128);
  } else {
    if (model_dim == 256 && h2_dim == 128) SWN_HB(float, 256, 128); else if (model_dim == 512 && h2_dim == 256) SWN_HB(float, 512, 256);
    else if (model_dim == 256) SWN_HB(float, 256, 256); else SWN_HB(float, 512, 128);
  }
#undef SWN_HB
#undef SWN_HB1
  OrdDst od{{d_w_sigma, d_w_color, d_b_sigma, d_b_color}, {model_dim, 3 * h2_dim, 1, 3}};
  ordered_reduce_async(partial, blocks, model_dim + 3 * h2_dim + 4, od, true, as_stream(stream));
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_group_colsum(const void* in, int dtype, int n_groups, int rows_per_group, int cols, float* out,
                                void* stream) {
  SWN_CHECK(in && out, "swn_group_colsum: null pointer");
  const int esz = dtype == SWN_HALF ? 2 : 4;
  SWN_CHECK(cols > 0 && (cols * esz) % 16 == 0 && cols * esz <= 4096 && 256 % (cols * esz / 16) == 0,
            "swn_group_colsum: a row must be 16 * 2^k <= 4096 bytes (cols=%d)", cols);
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((group_colsum_kernel<bf16_t>), dim3(n_groups), dim3(256), 0, as_stream(stream), (const bf16_t*)in,
                       rows_per_group, cols, out);
  else
    hipLaunchKernelGGL((group_colsum_kernel<float>), dim3(n_groups), dim3(256), 0, as_stream(stream), (const float*)in,
                       rows_per_group, cols, out);
  SWN_LAUNCH_CHECK();
  return 0;
}

#define COMPOSITE_DISPATCH(KERNEL, ...)                                             \
  do {                                                                              \
    const int spl = cdiv(n_samples, 64);                                            \
    if (spl <= 1) hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__);                     \
    else if (spl <= 2) hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__);                \
    else if (spl <= 4) hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__);                \
    else if (spl <= 8) hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__);                \
    else if (spl <= 16) hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__);              \
    else return swn::set_error("composite: n_samples %d > 1024", n_samples);       \
  } while (0)

extern "C" int swn_composite_fwd(const float* raw, const float* z, float last_delta, float rgb_padding, int n_rays, int n_samples,
                                 float* rgb, float* depth, float* depth_var, float* weights, void* stream) {
  SWN_CHECK(raw && z, "swn_composite_fwd: null pointer");
  COMPOSITE_DISPATCH(composite_fwd_kernel, dim3(cdiv(n_rays, 4)), dim3(256), 0, as_stream(stream), raw, z, last_delta, rgb_padding,
                     n_rays, n_samples, rgb, depth, depth_var, weights, (const float*)nullptr, 1.f, (const float*)nullptr,
                     (float*)nullptr);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_composite_bwd(const float* raw, const float* z, float last_delta, float rgb_padding, const float* d_rgb,
                                 int n_rays, int n_samples, float* d_raw, void* stream) {
  SWN_CHECK(raw && z && d_rgb && d_raw, "swn_composite_bwd: null pointer");
  COMPOSITE_DISPATCH(composite_bwd_kernel, dim3(cdiv(n_rays, 4)), dim3(256), 0, as_stream(stream), raw, z, last_delta, rgb_padding,
                     d_rgb, n_rays, n_samples, d_raw, (const float*)nullptr, 1.f, (const float*)nullptr);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_composite_bounded_fwd(const float* raw, const float* z, const float* last_delta, int flip, const float* depth_real,
                                         int n_rays, int n_samples, float* rgb, float* depth, float* depth_var, float* weights,
                                         float* bg_lambda, void* stream) {
  SWN_CHECK(raw && z, "swn_composite_bounded_fwd: null pointer");
  if (n_rays <= 0) return 0;
  COMPOSITE_DISPATCH(composite_fwd_kernel, dim3(cdiv(n_rays, 4)), dim3(256), 0, as_stream(stream), raw, z, 1e10f, 0.f, n_rays,
                     n_samples, rgb, depth, depth_var, weights, last_delta, flip ? -1.f : 1.f, depth_real, bg_lambda);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_composite_bounded_bwd(const float* raw, const float* z, const float* last_delta, int flip, const float* d_rgb,
                                         const float* d_bg_lambda, int n_rays, int n_samples, float* d_raw, void* stream) {
  SWN_CHECK(raw && z && d_rgb && d_raw, "swn_composite_bounded_bwd: null pointer");
  if (n_rays <= 0) return 0;
  COMPOSITE_DISPATCH(composite_bwd_kernel, dim3(cdiv(n_rays, 4)), dim3(256), 0, as_stream(stream), raw, z, 1e10f, 0.f, d_rgb,
                     n_rays, n_samples, d_raw, last_delta, flip ? -1.f : 1.f, d_bg_lambda);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow, int dtype,
                             long n, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                             void* stream) {
  SWN_CHECK(param && grad && exp_avg && exp_avg_sq && step >= 1, "swn_adam_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  int blocks = cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  if (shadow && dtype == SWN_HALF)
    hipLaunchKernelGGL((adam_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, (bf16_t*)shadow, n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale);
  else
    hipLaunchKernelGGL((adam_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, (float*)shadow, n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_cast(const float* in, void* out, int dtype, long n, void* stream) {
  SWN_CHECK(in && out, "swn_cast: null pointer");
  int blocks = cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((cast_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), in, (bf16_t*)out, n);
  else
    hipLaunchKernelGGL((cast_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), in, (float*)out, n);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_cast_transpose(const float* in, void* out, int dtype, int batch, int rows, int cols, void* stream) {
  SWN_CHECK(in && out && batch >= 1 && rows >= 1 && cols >= 1, "swn_cast_transpose: bad arguments");
  dim3 grid(cdiv(cols, 32), cdiv(rows, 32), batch), block(32, 8);
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((cast_transpose_kernel<bf16_t>), grid, block, 0, as_stream(stream), in, (bf16_t*)out, rows, cols);
  else
    hipLaunchKernelGGL((cast_transpose_kernel<float>), grid, block, 0, as_stream(stream), in, (float*)out, rows, cols);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_gather_rows(const void* src, const int32_t* index, long n_rows, int row_bytes, void* dst, void* stream) {
  SWN_CHECK(src && index && dst, "swn_gather_rows: null pointer");
  SWN_CHECK(row_bytes > 0 && row_bytes % 16 == 0, "swn_gather_rows: row_bytes %d must be a multiple of 16", row_bytes);
  if (n_rows <= 0) return 0;
  long blocks = (n_rows * (row_bytes >> 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const char*)src, index, n_rows,
                     row_bytes, (char*)dst);
  SWN_LAUNCH_CHECK();
  return 0;
}

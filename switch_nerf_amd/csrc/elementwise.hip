// HBM-bound kernels of the Switch-NeRF hot path: ray sampling + positional encoding, gate (LayerNorm + router +
// softmax + top-1), dispatch / combine (Tutel sparse-kernel ABI), sigma/colour heads, volumetric compositing,
// Adam.  One 64-lane wave per token/ray row, 16-byte accesses, wave shuffles for the reductions.
#include <stdarg.h>
#include "common.hpp"

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
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
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

__device__ __forceinline__ float z_of(float near, float far, float t) {
  // rendering.py:86  near * (1 - t) + far * t, each torch op rounded separately: fma contraction must stay off
  // (ROCm's __fmul_rn/__fadd_rn are plain operators and do not prevent it).
#pragma clang fp contract(off)
  const float a = near * (1.f - t);
  const float b = far * t;
  return a + b;
}

template <typename T, int LMAX>
__global__ __launch_bounds__(256) void sample_pe_kernel(const float* __restrict__ rays, const float* __restrict__ tsteps,
                                                        const float* __restrict__ prand, float perturb, int n_rays,
                                                        int S, int L, float* __restrict__ z_out, T* __restrict__ pe,
                                                        int pe_stride) {
#pragma clang fp contract(off)
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long)n_rays * S) return;
  const int ray = (int)(p / S), s = (int)(p - (long)ray * S);
  const float* r = rays + (long)ray * 8;
  const float near = r[6], far = r[7];
  float z = z_of(near, far, tsteps[s]);
  if (perturb > 0.f && prand) {  // rendering.py:573-584
    const float zp = s > 0 ? z_of(near, far, tsteps[s - 1]) : z;
    const float zn = s < S - 1 ? z_of(near, far, tsteps[s + 1]) : z;
    const float lower = s > 0 ? 0.5f * (zp + z) : z;
    const float upper = s < S - 1 ? 0.5f * (z + zn) : z;
    const float pr = perturb * prand[p];
    const float span = (upper - lower) * pr;
    z = lower + span;
  }
  z_out[p] = z;
  float v[8 + 6 * LMAX + 8];
  float x[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dz = r[3 + c] * z;  // rendering.py:90 (mul, then add: two roundings)
    x[c] = r[c] + dz;
    v[c] = x[c];
  }
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
  const int used = 3 + 6 * L;
  T* dst = pe + p * pe_stride;
  const int step = 16 / (int)sizeof(T);
  // columns [used, pe_stride) are zero
  float tmp[8];
  for (int c0 = 0; c0 < pe_stride; c0 += step) {
#pragma unroll
    for (int j = 0; j < 8; ++j) tmp[j] = 0.f;
    for (int j = 0; j < step; ++j) {
      const int c = c0 + j;
      float val = 0.f;
      // v[] is indexed with a runtime index only here; keep it small
      if (c < used) val = v[c];
      tmp[j] = val;
    }
    store_vals<T>(dst + c0, tmp, step);
  }
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

// ------------------------------------------------------------------------------------------------ gate
// one wave per token; lane owns G/64 consecutive features (G in {64,128,...,1024})
template <typename T, int VPL>
__device__ __forceinline__ void load_row(const T* row, int lane, float* x) {
#pragma unroll
  for (int j = 0; j < VPL; ++j) x[j] = ElemIO<T>::ld(row + lane * VPL + j);
}

template <typename T, int VPL, int EMAX>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ g, const float* __restrict__ ln_w,
                                                       const float* __restrict__ ln_b, const float* __restrict__ wg,
                                                       int P, int E, float* __restrict__ gates, int32_t* __restrict__ idx,
                                                       float* __restrict__ gmax, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int G = VPL * 64;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  float w[VPL], b[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    w[j] = ln_w ? ln_w[lane * VPL + j] : 1.f;
    b[j] = ln_b ? ln_b[lane * VPL + j] : 0.f;
  }
  for (long tok = wid; tok < P; tok += nw) {
    float x[VPL];
    load_row<T, VPL>(g + tok * G, lane, x);
    float mean = 0.f, rstd = 1.f;
    if (ln_w) {  // torch.nn.LayerNorm, eps 1e-5 (models/nerf_moe.py:301-302)
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < VPL; ++j) s += x[j];
      mean = wave_sum(s) / G;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < VPL; ++j) q += (x[j] - mean) * (x[j] - mean);
      rstd = 1.f / sqrtf(wave_sum(q) / G + 1e-5f);
#pragma unroll
      for (int j = 0; j < VPL; ++j) x[j] = (x[j] - mean) * rstd * w[j] + b[j];
    }
    float logit[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      float d = 0.f;
      if (e < E) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) d += x[j] * wg[(long)e * G + lane * VPL + j];
        d = wave_sum(d);
      }
      logit[e] = d;
    }
    float mx = logit[0];
#pragma unroll
    for (int e = 1; e < EMAX; ++e)
      if (e < E) mx = fmaxf(mx, logit[e]);
    float den = 0.f, pr[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      pr[e] = (e < E) ? expf(logit[e] - mx) : 0.f;
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
      if (lane == e && e < E) gates[tok * E + e] = pr[e];
    if (lane == 0) {
      idx[tok] = best;
      gmax[tok] = bv;
      if (stats) { stats[tok * 2] = mean; stats[tok * 2 + 1] = rstd; }
    }
  }
}

template <typename T, int VPL, int EMAX>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const T* __restrict__ g, const float* __restrict__ ln_w,
                                                       const float* __restrict__ ln_b, const float* __restrict__ wg,
                                                       const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                       const float* __restrict__ d_gmax, const float* __restrict__ stats,
                                                       const int32_t* __restrict__ counts, const float* __restrict__ laux_coef,
                                                       int seg_tokens, int P, int E, T* __restrict__ dg,
                                                       float* __restrict__ d_wg, float* __restrict__ d_ln_w,
                                                       float* __restrict__ d_ln_b) {
  const int lane = threadIdx.x & 63;
  const int G = VPL * 64;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  float w[VPL], b[VPL], aw[VPL], ab[VPL];
  float awg[EMAX][VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    w[j] = ln_w ? ln_w[lane * VPL + j] : 1.f;
    b[j] = ln_b ? ln_b[lane * VPL + j] : 0.f;
    aw[j] = 0.f;
    ab[j] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < EMAX; ++e)
#pragma unroll
    for (int j = 0; j < VPL; ++j) awg[e][j] = 0.f;

  for (long tok = wid; tok < P; tok += nw) {
    float x[VPL], xh[VPL], xn[VPL];
    load_row<T, VPL>(g + tok * G, lane, x);
    const float mean = ln_w ? stats[tok * 2] : 0.f, rstd = ln_w ? stats[tok * 2 + 1] : 1.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      xh[j] = ln_w ? (x[j] - mean) * rstd : x[j];
      xn[j] = ln_w ? xh[j] * w[j] + b[j] : x[j];
    }
    const int seg = (int)(tok / seg_tokens);
    const int my = idx[tok];
    const float coef = laux_coef ? laux_coef[seg] : 0.f;
    const float dgm = d_gmax ? d_gmax[tok] : 0.f;
    float pr[EMAX], dp[EMAX], dot = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      pr[e] = (e < E) ? gates[tok * E + e] : 0.f;
      dp[e] = (e < E) ? coef * (float)counts[seg * E + e] + ((e == my) ? dgm : 0.f) : 0.f;
      dot += pr[e] * dp[e];
    }
    float dxn[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) dxn[j] = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      if (e < E) {
        const float dl = pr[e] * (dp[e] - dot);  // softmax backward
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          dxn[j] += dl * wg[(long)e * G + lane * VPL + j];
          awg[e][j] += dl * xn[j];
        }
      }
    }
    float dx[VPL];
    if (ln_w) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const float dxh = dxn[j] * w[j];
        s1 += dxh;
        s2 += dxh * xh[j];
        aw[j] += dxn[j] * xh[j];
        ab[j] += dxn[j];
      }
      s1 = wave_sum(s1) / G;
      s2 = wave_sum(s2) / G;
#pragma unroll
      for (int j = 0; j < VPL; ++j) dx[j] = rstd * (dxn[j] * w[j] - s1 - xh[j] * s2);
    } else {
#pragma unroll
      for (int j = 0; j < VPL; ++j) dx[j] = dxn[j];
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j) ElemIO<T>::st(dg + tok * G + lane * VPL + j, dx[j]);
  }
#pragma unroll
  for (int e = 0; e < EMAX; ++e)
    if (e < E)
#pragma unroll
      for (int j = 0; j < VPL; ++j) unsafeAtomicAdd(d_wg + (long)e * G + lane * VPL + j, awg[e][j]);
  if (ln_w) {
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      unsafeAtomicAdd(d_ln_w + lane * VPL + j, aw[j]);
      unsafeAtomicAdd(d_ln_b + lane * VPL + j, ab[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ dispatch / combine
// Tutel batched sparse kernels: row(i) = seg(i)*E*C + idx[i]*C + loc[i], dropped iff loc >= C or idx < 0.
template <typename T, int MODE>  // MODE 0: D[row] = g*x   1: out[i] = g*D[row] (+relu)   2: dgate[i] = <D[row], x[i]>
__global__ __launch_bounds__(256) void sparse_kernel(const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                     const int32_t* __restrict__ loc, T* __restrict__ tok_buf,
                                                     T* __restrict__ disp, float* __restrict__ dgate, int samples,
                                                     int hidden, int capacity, int seg_tokens, int n_experts, int relu) {
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  const int cpr = hidden * (int)sizeof(T) / 16;  // 16-byte chunks per row
  constexpr int EPC = 16 / (int)sizeof(T);
  for (long i = wid; i < samples; i += nw) {
    const int e = idx[i], l = loc[i];
    const bool keep = (e >= 0) && (l < capacity) && (l >= 0);
    const long row = (i / seg_tokens) * (long)n_experts * capacity + (long)e * capacity + l;
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
        if (!keep) {
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
template <typename T>
__global__ __launch_bounds__(256) void combine_bwd_kernel(const T* __restrict__ dy_in, const T* __restrict__ y,
                                                          const float* __restrict__ dsig, const float* __restrict__ wsig,
                                                          const float* __restrict__ gate, int P, int hidden,
                                                          T* __restrict__ dout, float* __restrict__ dgate) {
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cpr = hidden / EPC;
  for (long i = wid; i < P; i += nw) {
    const float gt = gate[i];
    const float ds = dsig ? dsig[i] : 0.f;
    float dot = 0.f;
    for (int ch = lane; ch < cpr; ch += 64) {
      const long off = i * hidden + ch * EPC;
      float o[EPC];
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        const float yv = ElemIO<T>::ld(y + off + j);
        float d = ElemIO<T>::ld(dy_in + off + j) + (wsig ? ds * wsig[ch * EPC + j] : 0.f);
        d = yv > 0.f ? d : 0.f;
        dot += yv * d;
        o[j] = d * gt;
      }
#pragma unroll
      for (int j = 0; j < EPC; ++j) ElemIO<T>::st(dout + off + j, o[j]);
    }
    dot = wave_sum(dot);
    if (lane == 0) dgate[i] = dot / gt;
  }
}

// ------------------------------------------------------------------------------------------------ heads
// raw[i] = (sigmoid(h2 . Wc[c] + bc[c]) c<3, softplus(y . ws + bs + noise - 1))   models/nerf_moe.py:393-441
template <typename T>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const T* __restrict__ y, const T* __restrict__ h2,
                                                        const float* __restrict__ ws, const float* __restrict__ bs,
                                                        const float* __restrict__ wc, const float* __restrict__ bc,
                                                        const float* __restrict__ noise, int P, int M, int H2,
                                                        float* __restrict__ raw) {
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long i = wid; i < P; i += nw) {
    float s = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int c = lane; c < M; c += 64) s += ElemIO<T>::ld(y + i * M + c) * ws[c];
    for (int c = lane; c < H2; c += 64) {
      const float h = ElemIO<T>::ld(h2 + i * H2 + c);
      c0 += h * wc[c];
      c1 += h * wc[H2 + c];
      c2 += h * wc[2 * H2 + c];
    }
    s = wave_sum(s); c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    if (lane == 0) {
      float u = s + bs[0] + (noise ? noise[i] : 0.f) - 1.f;  // ShiftedSoftplus, models/nerf.py:68-69
      const float sp = u > 20.f ? u : log1pf(expf(u));
      float4 o;
      o.x = 1.f / (1.f + expf(-(c0 + bc[0])));
      o.y = 1.f / (1.f + expf(-(c1 + bc[1])));
      o.z = 1.f / (1.f + expf(-(c2 + bc[2])));
      o.w = sp;
      *(float4*)(raw + i * 4) = o;
    }
  }
}

// d_raw -> dh2 (masked by h2 > 0), dsig_pre; accumulates d_wc, d_bc, d_ws, d_bs.
template <typename T>
__global__ __launch_bounds__(256) void heads_bwd_kernel(const T* __restrict__ y, const T* __restrict__ h2,
                                                        const float* __restrict__ wc, const float* __restrict__ raw,
                                                        const float* __restrict__ d_raw, const float* __restrict__ pre_sig,
                                                        int P, int M, int H2, T* __restrict__ dh2, float* __restrict__ dsig,
                                                        float* __restrict__ d_ws, float* __restrict__ d_bs,
                                                        float* __restrict__ d_wc, float* __restrict__ d_bc) {
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  // per-lane partial sums: d_ws for columns lane + 64*j (M <= 512), d_wc for columns lane + 64*j (H2 <= 256)
  float aws[8], awc[3][4], abs_ = 0.f, abc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) aws[j] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) awc[c][j] = 0.f;
  for (long i = wid; i < P; i += nw) {
    const float4 r = *(const float4*)(raw + i * 4);
    const float4 d = *(const float4*)(d_raw + i * 4);
    const float dc0 = d.x * r.x * (1.f - r.x), dc1 = d.y * r.y * (1.f - r.y), dc2 = d.z * r.z * (1.f - r.z);
    const float dsp = d.w * (1.f - expf(-r.w));  // softplus'(u) = sigmoid(u) = 1 - exp(-softplus(u))
    (void)pre_sig;
    if (lane == 0) dsig[i] = dsp;
    abs_ += dsp;
    abc[0] += dc0; abc[1] += dc1; abc[2] += dc2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + 64 * j;
      if (c < M) aws[j] += dsp * ElemIO<T>::ld(y + i * M + c);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      if (c < H2) {
        const float h = ElemIO<T>::ld(h2 + i * H2 + c);
        awc[0][j] += dc0 * h; awc[1][j] += dc1 * h; awc[2][j] += dc2 * h;
        const float g = dc0 * wc[c] + dc1 * wc[H2 + c] + dc2 * wc[2 * H2 + c];
        ElemIO<T>::st(dh2 + i * H2 + c, h > 0.f ? g : 0.f);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = lane + 64 * j;
    if (c < M) unsafeAtomicAdd(d_ws + c, aws[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    if (c < H2) {
      unsafeAtomicAdd(d_wc + c, awc[0][j]);
      unsafeAtomicAdd(d_wc + H2 + c, awc[1][j]);
      unsafeAtomicAdd(d_wc + 2 * H2 + c, awc[2][j]);
    }
  }
  if (lane == 0) {  // every lane accumulated the same per-token scalars; lane 0 publishes
    unsafeAtomicAdd(d_bs, abs_);
    unsafeAtomicAdd(d_bc + 0, abc[0]);
    unsafeAtomicAdd(d_bc + 1, abc[1]);
    unsafeAtomicAdd(d_bc + 2, abc[2]);
  }
}

// out[g][c] = sum over the group's rows of in[g*R + r][c]
template <typename T>
__global__ void group_colsum_kernel(const T* __restrict__ in, int R, int C, float* __restrict__ out) {
  const int g = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += ElemIO<T>::ld(in + ((long)g * R + r) * C + c);
    out[(long)g * C + c] = s;
  }
}

// ------------------------------------------------------------------------------------------------ compositing
// one wave per ray; lane owns a contiguous run of ceil(S/64) samples.  rendering.py:435-494
template <int SPL>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                            float last_delta, int N, int S, float* __restrict__ rgb,
                                                            float* __restrict__ depth, float* __restrict__ dvar,
                                                            float* __restrict__ weights) {
  const int lane = threadIdx.x & 63;
  const long ray = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= N) return;
  const float* zr = z + ray * S;
  const float4* rr = (const float4*)(raw + ray * S * 4);
  float al[SPL], zz[SPL];
  float4 cs[SPL];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    const int s = lane * SPL + j;
    al[j] = 0.f; zz[j] = 0.f; cs[j] = make_float4(0, 0, 0, 0);
    if (s < S) {
      zz[j] = zr[s];
      const float dl = (s + 1 < S) ? (zr[s + 1] - zz[j]) : last_delta;
      cs[j] = rr[s];
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
  float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f;
  float w[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    w[j] = al[j] * T;
    T *= (1.f - al[j] + 1e-8f);
    ar += w[j] * cs[j].x; ag += w[j] * cs[j].y; ab += w[j] * cs[j].z; ad += w[j] * zz[j];
    const int s = lane * SPL + j;
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
                                                            float last_delta, const float* __restrict__ d_rgb, int N, int S,
                                                            float* __restrict__ d_raw) {
  const int lane = threadIdx.x & 63;
  const long ray = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= N) return;
  const float* zr = z + ray * S;
  const float4* rr = (const float4*)(raw + ray * S * 4);
  const float g0 = d_rgb[ray * 3], g1 = d_rgb[ray * 3 + 1], g2 = d_rgb[ray * 3 + 2];
  float al[SPL], dl[SPL], cg[SPL];
  float prod = 1.f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    const int s = lane * SPL + j;
    al[j] = 0.f; dl[j] = 0.f; cg[j] = 0.f;
    if (s < S) {
      const float zc = zr[s];
      dl[j] = (s + 1 < S) ? (zr[s + 1] - zc) : last_delta;
      const float4 c = rr[s];
      al[j] = 1.f - expf(-dl[j] * c.w);
      cg[j] = c.x * g0 + c.y * g1 + c.z * g2;
      prod *= (1.f - al[j] + 1e-8f);
    }
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
#pragma unroll
  for (int j = SPL - 1; j >= 0; --j) {
    const int s = lane * SPL + j;
    if (s < S) {
      // d/d alpha_j = T_j (c_j.g) - (sum_{i>j} u_i) / (1 - alpha_j + 1e-8)
      const float dalpha = Ts[j] * cg[j] - suffix / (1.f - al[j] + 1e-8f);
      const float dsigma = dalpha * dl[j] * (1.f - al[j]);  // d alpha / d sigma = delta * exp(-delta sigma)
      const float wj = al[j] * Ts[j];
      *(float4*)(d_raw + (ray * S + s) * 4) = make_float4(wj * g0, wj * g1, wj * g2, dsigma);
    }
    suffix += u[j];
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

}  // namespace swn

using namespace swn;

extern "C" const char* swn_last_error(void) { return g_err; }
extern "C" int swn_version(void) { return 1; }

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
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_BF16, "swn_sample_pe: bad dtype");
  SWN_CHECK(rays && t_steps && z_out && pe_xyz, "swn_sample_pe: null pointer");
  SWN_CHECK(l_xyz >= 0 && l_xyz <= 12 && l_dir >= 0 && l_dir <= 12, "swn_sample_pe: frequencies must be <= 12");
  const int epc = dtype == SWN_BF16 ? 8 : 4;
  SWN_CHECK(pe_stride >= 3 + 6 * l_xyz && pe_stride % epc == 0, "swn_sample_pe: pe_stride %d too small / unaligned", pe_stride);
  const long P = (long)n_rays * n_samples;
  const int blocks = cdiv(P, 256);
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((sample_pe_kernel<bf16_t, 12>), dim3(blocks), dim3(256), 0, as_stream(stream), rays, t_steps,
                       perturb_rand, perturb, n_rays, n_samples, l_xyz, z_out, (bf16_t*)pe_xyz, pe_stride);
  else
    hipLaunchKernelGGL((sample_pe_kernel<float, 12>), dim3(blocks), dim3(256), 0, as_stream(stream), rays, t_steps,
                       perturb_rand, perturb, n_rays, n_samples, l_xyz, z_out, (float*)pe_xyz, pe_stride);
  SWN_LAUNCH_CHECK();
  if (pe_dir) {
    SWN_CHECK(dir_stride >= 3 + 6 * l_dir, "swn_sample_pe: dir_stride too small");
    if (dtype == SWN_BF16)
      hipLaunchKernelGGL((dir_pe_kernel<bf16_t, 12>), dim3(cdiv(n_rays, 256)), dim3(256), 0, as_stream(stream), rays,
                         n_rays, l_dir, (bf16_t*)pe_dir, dir_stride);
    else
      hipLaunchKernelGGL((dir_pe_kernel<float, 12>), dim3(cdiv(n_rays, 256)), dim3(256), 0, as_stream(stream), rays,
                         n_rays, l_dir, (float*)pe_dir, dir_stride);
    SWN_LAUNCH_CHECK();
  }
  return 0;
}

#define GATE_DISPATCH(T, KERNEL, ...)                                                                        \
  do {                                                                                                       \
    const int vpl = gate_dim / 64;                                                                           \
    const bool e8 = n_experts <= 8;                                                                          \
    if (vpl == 4 && e8) hipLaunchKernelGGL((KERNEL<T, 4, 8>), __VA_ARGS__);                                  \
    else if (vpl == 4) hipLaunchKernelGGL((KERNEL<T, 4, 16>), __VA_ARGS__);                                  \
    else if (vpl == 1 && e8) hipLaunchKernelGGL((KERNEL<T, 1, 8>), __VA_ARGS__);                             \
    else if (vpl == 8) hipLaunchKernelGGL((KERNEL<T, 8, 16>), __VA_ARGS__);                                  \
    else return swn::set_error("gate: unsupported gate_dim %d / experts %d", gate_dim, n_experts);          \
  } while (0)

extern "C" int swn_gate_fwd(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                            int n_tokens, int gate_dim, int n_experts, float* gates, int32_t* idx, float* gmax,
                            float* stats, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_BF16, "swn_gate_fwd: bad dtype");
  SWN_CHECK(g && wg && gates && idx && gmax, "swn_gate_fwd: null pointer");
  SWN_CHECK(gate_dim % 64 == 0 && n_experts >= 1 && n_experts <= 16, "swn_gate_fwd: gate_dim %% 64, experts <= 16");
  SWN_CHECK((ln_w == nullptr) == (ln_b == nullptr), "swn_gate_fwd: ln_w / ln_b must both be given or both NULL");
  if (ln_w) SWN_CHECK(stats, "swn_gate_fwd: stats required with LayerNorm");
  const int blocks = ew_blocks(n_tokens);
  if (dtype == SWN_BF16) {
    const bf16_t* gp = (const bf16_t*)g;
    GATE_DISPATCH(bf16_t, gate_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, n_tokens, n_experts,
                  gates, idx, gmax, stats);
  } else {
    const float* gp = (const float*)g;
    GATE_DISPATCH(float, gate_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, n_tokens, n_experts,
                  gates, idx, gmax, stats);
  }
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_gate_bwd(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                            const float* gates, const int32_t* idx, const float* d_gmax, const float* stats,
                            const int32_t* counts, const float* laux_coef, int seg_tokens, int n_tokens, int gate_dim,
                            int n_experts, void* dg, float* d_wg, float* d_ln_w, float* d_ln_b, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_BF16, "swn_gate_bwd: bad dtype");
  SWN_CHECK(g && wg && gates && idx && dg && d_wg && counts, "swn_gate_bwd: null pointer");
  SWN_CHECK(gate_dim % 64 == 0 && n_experts >= 1 && n_experts <= 16 && seg_tokens > 0, "swn_gate_bwd: bad sizes");
  if (ln_w) SWN_CHECK(stats && d_ln_w && d_ln_b && ln_b, "swn_gate_bwd: LayerNorm buffers missing");
  int blocks = ew_blocks(n_tokens);
  if (blocks > 1024) blocks = 1024;  // each wave publishes E*G atomics at the end
  if (dtype == SWN_BF16) {
    const bf16_t* gp = (const bf16_t*)g;
    bf16_t* dgp = (bf16_t*)dg;
    GATE_DISPATCH(bf16_t, gate_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, gates, idx, d_gmax,
                  stats, counts, laux_coef, seg_tokens, n_tokens, n_experts, dgp, d_wg, d_ln_w, d_ln_b);
  } else {
    const float* gp = (const float*)g;
    float* dgp = (float*)dg;
    GATE_DISPATCH(float, gate_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), gp, ln_w, ln_b, wg, gates, idx, d_gmax,
                  stats, counts, laux_coef, seg_tokens, n_tokens, n_experts, dgp, d_wg, d_ln_w, d_ln_b);
  }
  SWN_LAUNCH_CHECK();
  return 0;
}

template <int MODE>
static int launch_sparse(const float* gates, const int32_t* idx, const int32_t* loc, void* tok, void* disp, float* dgate,
                         int dtype, int samples, int hidden, int capacity, int seg_tokens, int n_experts, int relu,
                         void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_BF16, "sparse: bad dtype");
  SWN_CHECK(idx && loc && tok && disp, "sparse: null pointer");
  SWN_CHECK(hidden * (dtype == SWN_BF16 ? 2 : 4) % 16 == 0, "sparse: hidden row must be a multiple of 16 bytes");
  SWN_CHECK(capacity > 0 && seg_tokens > 0 && n_experts > 0, "sparse: bad sizes");
  const int blocks = ew_blocks(samples);
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((sparse_kernel<bf16_t, MODE>), dim3(blocks), dim3(256), 0, as_stream(stream), gates, idx, loc,
                       (bf16_t*)tok, (bf16_t*)disp, dgate, samples, hidden, capacity, seg_tokens, n_experts, relu);
  else
    hipLaunchKernelGGL((sparse_kernel<float, MODE>), dim3(blocks), dim3(256), 0, as_stream(stream), gates, idx, loc,
                       (float*)tok, (float*)disp, dgate, samples, hidden, capacity, seg_tokens, n_experts, relu);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_dispatch_fwd(const float* gates, const int32_t* indices, const int32_t* locations,
                                const void* reshaped_input, void* dispatched, int dtype, int samples, int hidden,
                                int capacity, int n_experts, void* stream) {
  SWN_CHECK(dispatched, "swn_dispatch_fwd: null");
  const size_t bytes = (size_t)n_experts * capacity * hidden * (dtype == SWN_BF16 ? 2 : 4);
  hipError_t e = hipMemsetAsync(dispatched, 0, bytes, as_stream(stream));  // torch.zeros at tutel_fast_dispatch.py:25
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
extern "C" int swn_dispatch_bwd_gate(float* grad_gates, const int32_t* indices, const int32_t* locations,
                                     const void* reshaped_input, const void* dispatched, int dtype, int samples,
                                     int hidden, int capacity, void* stream) {
  SWN_CHECK(grad_gates, "swn_dispatch_bwd_gate: null");
  return launch_sparse<2>(nullptr, indices, locations, (void*)reshaped_input, (void*)dispatched, grad_gates, dtype,
                          samples, hidden, capacity, samples, 1 << 20, 0, stream);
}

extern "C" int swn_combine_fwd(const float* gates, const int32_t* indices, const int32_t* locations, void* y,
                               const void* expert_out, int dtype, int samples, int hidden, int capacity, int seg_tokens,
                               int n_experts, int relu, void* stream) {
  return launch_sparse<1>(gates, indices, locations, y, (void*)expert_out, nullptr, dtype, samples, hidden, capacity,
                          seg_tokens, n_experts, relu, stream);
}

extern "C" int swn_combine_bwd(const void* dy_in, const void* y, const float* dsig, const float* wsig, const float* gate,
                               int dtype, int samples, int hidden, void* dout, float* dgate, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_BF16, "swn_combine_bwd: bad dtype");
  SWN_CHECK(dy_in && y && gate && dout && dgate, "swn_combine_bwd: null pointer");
  const int blocks = ew_blocks(samples);
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((combine_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)dy_in,
                       (const bf16_t*)y, dsig, wsig, gate, samples, hidden, (bf16_t*)dout, dgate);
  else
    hipLaunchKernelGGL((combine_bwd_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)dy_in,
                       (const float*)y, dsig, wsig, gate, samples, hidden, (float*)dout, dgate);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_heads_fwd(const void* y, const void* h2, int dtype, const float* w_sigma, const float* b_sigma,
                             const float* w_color, const float* b_color, const float* sigma_noise, int n_points,
                             int model_dim, int h2_dim, float* raw, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_BF16, "swn_heads_fwd: bad dtype");
  SWN_CHECK(y && h2 && w_sigma && b_sigma && w_color && b_color && raw, "swn_heads_fwd: null pointer");
  const int blocks = ew_blocks(n_points);
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((heads_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)y,
                       (const bf16_t*)h2, w_sigma, b_sigma, w_color, b_color, sigma_noise, n_points, model_dim, h2_dim, raw);
  else
    hipLaunchKernelGGL((heads_fwd_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)y,
                       (const float*)h2, w_sigma, b_sigma, w_color, b_color, sigma_noise, n_points, model_dim, h2_dim, raw);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_heads_bwd(const void* y, const void* h2, int dtype, const float* w_color, const float* raw,
                             const float* d_raw, int n_points, int model_dim, int h2_dim, void* dh2, float* dsig,
                             float* d_w_sigma, float* d_b_sigma, float* d_w_color, float* d_b_color, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_BF16, "swn_heads_bwd: bad dtype");
  SWN_CHECK(y && h2 && w_color && raw && d_raw && dh2 && dsig && d_w_sigma && d_b_sigma && d_w_color && d_b_color,
            "swn_heads_bwd: null pointer");
  SWN_CHECK(model_dim <= 512 && h2_dim <= 256, "swn_heads_bwd: model_dim <= 512, h2_dim <= 256");
  int blocks = ew_blocks(n_points);
  if (blocks > 1024) blocks = 1024;
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((heads_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)y,
                       (const bf16_t*)h2, w_color, raw, d_raw, nullptr, n_points, model_dim, h2_dim, (bf16_t*)dh2, dsig,
                       d_w_sigma, d_b_sigma, d_w_color, d_b_color);
  else
    hipLaunchKernelGGL((heads_bwd_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)y,
                       (const float*)h2, w_color, raw, d_raw, nullptr, n_points, model_dim, h2_dim, (float*)dh2, dsig,
                       d_w_sigma, d_b_sigma, d_w_color, d_b_color);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_group_colsum(const void* in, int dtype, int n_groups, int rows_per_group, int cols, float* out,
                                void* stream) {
  SWN_CHECK(in && out, "swn_group_colsum: null pointer");
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((group_colsum_kernel<bf16_t>), dim3(n_groups), dim3(128), 0, as_stream(stream), (const bf16_t*)in,
                       rows_per_group, cols, out);
  else
    hipLaunchKernelGGL((group_colsum_kernel<float>), dim3(n_groups), dim3(128), 0, as_stream(stream), (const float*)in,
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

extern "C" int swn_composite_fwd(const float* raw, const float* z, float last_delta, int n_rays, int n_samples,
                                 float* rgb, float* depth, float* depth_var, float* weights, void* stream) {
  SWN_CHECK(raw && z, "swn_composite_fwd: null pointer");
  COMPOSITE_DISPATCH(composite_fwd_kernel, dim3(cdiv(n_rays, 4)), dim3(256), 0, as_stream(stream), raw, z, last_delta,
                     n_rays, n_samples, rgb, depth, depth_var, weights);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_composite_bwd(const float* raw, const float* z, float last_delta, const float* d_rgb, int n_rays,
                                 int n_samples, float* d_raw, void* stream) {
  SWN_CHECK(raw && z && d_rgb && d_raw, "swn_composite_bwd: null pointer");
  COMPOSITE_DISPATCH(composite_bwd_kernel, dim3(cdiv(n_rays, 4)), dim3(256), 0, as_stream(stream), raw, z, last_delta,
                     d_rgb, n_rays, n_samples, d_raw);
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
  if (shadow && dtype == SWN_BF16)
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
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), in, (bf16_t*)out, n);
  else
    hipLaunchKernelGGL((cast_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), in, (float*)out, n);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_cast_transpose(const float* in, void* out, int dtype, int batch, int rows, int cols, void* stream) {
  SWN_CHECK(in && out && batch >= 1 && rows >= 1 && cols >= 1, "swn_cast_transpose: bad arguments");
  dim3 grid(cdiv(cols, 32), cdiv(rows, 32), batch), block(32, 8);
  if (dtype == SWN_BF16)
    hipLaunchKernelGGL((cast_transpose_kernel<bf16_t>), grid, block, 0, as_stream(stream), in, (bf16_t*)out, rows, cols);
  else
    hipLaunchKernelGGL((cast_transpose_kernel<float>), grid, block, 0, as_stream(stream), in, (float*)out, rows, cols);
  SWN_LAUNCH_CHECK();
  return 0;
}

// Multiresolution hash-grid input encoding (BASELINE.json configs[4]: "hash-encoded (NGP-style) input").
// The reference has no such encoder: the algorithm restated here is Mueller et al., "Instant
// Neural Graphics Primitives with a Multiresolution Hash Encoding" (SIGGRAPH 2022), section 3, with this file's own
// conventions (documented in include/swn.h, restated on the CPU in oracle/switchnerf_oracle.py hash_encode - parity unpinned):
//   x' = clamp((x - aabb_lo) / (aabb_hi - aabb_lo), 0, 1);  level l: scale_l = base_res * per_level_scale^l - 1,
//   R_l = ceil(scale_l) + 2 grid points per axis;  pos = x' * scale_l + 0.5, cell = floor(pos), w = pos - cell;
//   corner (cx, cy, cz) -> entry  cx + R_l (cy + R_l cz)            if R_l^3 <= T   (dense level)
//                                 (cx ^ cy * 2654435761 ^ cz * 805459861) mod T      otherwise (T = 2^log2_table)
//   feature_l = sum over the 8 corners of trilinear weight * table[l][entry][0..1]
// One thread per point: 16 levels x 8 corners of 8-byte gathers (the table, 64 MiB at L = 16, T = 2^19, lives in the
// 256 MiB Infinity Cache), output rows staged in LDS and written coalesced (pe_store.hpp).  Backward: one thread per point,
// fp32 atomics into the table gradient (collisions are the point of the hash: no sort/segment step would remove them).
#include <math.h>
#include "common.hpp"
#include "pe_store.hpp"

namespace swn {

constexpr int HASH_MAX_LEVELS = 16;

struct HashLevels {
  float scale[HASH_MAX_LEVELS];
  uint32_t res[HASH_MAX_LEVELS];     // grid points per axis
  uint32_t dense[HASH_MAX_LEVELS];
  float lo[3], inv_extent[3];
  int n_levels;
  uint32_t table_mask;               // T - 1
  long level_stride;                 // T * 2 floats
};

__device__ __forceinline__ uint32_t hash_entry(const HashLevels& h, int l, uint32_t cx, uint32_t cy, uint32_t cz) {
  if (h.dense[l]) return cx + h.res[l] * (cy + h.res[l] * cz);
  return (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & h.table_mask;
}

__device__ __forceinline__ void point_of(const float* __restrict__ rays, const float* __restrict__ z, long p, int S,
                                         const HashLevels& h, float x[3]) {
#pragma clang fp contract(off)
  const int ray = (int)(p / S);
  const float* r = rays + (long)ray * 8;
  const float zz = z[p];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dz = r[3 + c] * zz;                       // rendering.py:90: o + d * z (two roundings)
    const float w = ((r[c] + dz) - h.lo[c]) * h.inv_extent[c];
    x[c] = fminf(fmaxf(w, 0.f), 1.f);
  }
}

template <typename T>
__global__ __launch_bounds__(128) void hash_encode_fwd_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                              int n_rays, int S, HashLevels h, const float* __restrict__ table,
                                                              T* __restrict__ out, int out_stride) {
#pragma clang fp contract(off)
  const long total = (long)n_rays * S;
  const long p_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p_raw < total;
  const long p = live ? p_raw : total - 1;
  float x[3];
  point_of(rays, z, p, S, h, x);
  float v[2 * HASH_MAX_LEVELS + 8];
#pragma unroll
  for (int l = 0; l < HASH_MAX_LEVELS; ++l) {
    float f0 = 0.f, f1 = 0.f;
    if (l < h.n_levels) {
      uint32_t c0[3];
      float w[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float pos = x[c] * h.scale[l] + 0.5f;
        const float fl = floorf(pos);
        c0[c] = (uint32_t)fl;
        w[c] = pos - fl;
      }
      const float2* tl = (const float2*)(table + (long)l * h.level_stride);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
        const float wk = ((dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1])) * (dz ? w[2] : 1.f - w[2]);
        const float2 t = tl[hash_entry(h, l, c0[0] + dx, c0[1] + dy, c0[2] + dz)];
        f0 = f0 + wk * t.x;
        f1 = f1 + wk * t.y;
      }
    }
    v[2 * l] = f0;
    v[2 * l + 1] = f1;
  }
  pe_store_rows<T>(v, 2 * h.n_levels, out, out_stride, p, live, total);
}

// Backward: one thread per point, consecutive lanes = consecutive samples of a ray.  On the coarse levels dozens of
// consecutive samples sit in the same cell and would hammer the same 16 table floats with atomics (measured: 42 ms per
// 2M points).  Lanes of a wave that share a cell form contiguous runs; per level the runs are found once (ballot of
// "cell differs from the previous lane"), every corner/feature contribution is summed over its run with a segmented wave
// scan, and only the run's tail lane issues the atomic.
__device__ __forceinline__ float run_inclusive_scan(float v, int lane, int start) {     // sum over lanes [start, lane]
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane - o >= start) v += t;
  }
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void hash_encode_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                              int n_rays, int S, HashLevels h, const T* __restrict__ d_out,
                                                              int d_stride, float* __restrict__ d_table, long xcd_stride) {
#pragma clang fp contract(off)
  // xcd_stride != 0: d_table is one PRIVATE copy of the gradient table per XCD (copy x at d_table + x * xcd_stride, x = the XCC_ID the
  // workgroup runs on): every atomic of the launch meets its partners in ONE L2 (hash_reduce_kernel adds the copies afterwards)
  if (xcd_stride) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d_table += (long)(xcc & 7u) * xcd_stride;
  }
  const long total = (long)n_rays * S;
  const long p_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p_raw < total;                      // surplus lanes take part in the wave operations with zero gradient
  const long p = live ? p_raw : total - 1;
  const int lane = threadIdx.x & 63;
  float x[3];
  point_of(rays, z, p, S, h, x);
  const T* dr = d_out + p * d_stride;
  for (int l = 0; l < h.n_levels; ++l) {
    const float g0 = live ? ElemIO<T>::ld(dr + 2 * l) : 0.f, g1 = live ? ElemIO<T>::ld(dr + 2 * l + 1) : 0.f;
    uint32_t c0[3];
    float w[3];
    bool head = lane == 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pos = x[c] * h.scale[l] + 0.5f;
      const float fl = floorf(pos);
      c0[c] = (uint32_t)fl;
      w[c] = pos - fl;
      head = head || (__shfl_up(c0[c], 1, 64) != c0[c]);
    }
    const uint64_t heads = __ballot(head);
    // run of this lane: [start, end) with start = highest head at or below the lane, end = next head above it (or 64)
    const uint64_t below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int start = 63 - __clzll((long long)below);
    const uint64_t above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const bool tail = above == 0ull ? lane == 63 : (__ffsll((long long)above) == 1);
    float* tl = d_table + (long)l * h.level_stride;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      const float wk = ((dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1])) * (dz ? w[2] : 1.f - w[2]);
      const float t0 = run_inclusive_scan(wk * g0, lane, start), t1 = run_inclusive_scan(wk * g1, lane, start);
      if (tail) {
        float* e = tl + 2 * (long)hash_entry(h, l, c0[0] + dx, c0[1] + dy, c0[2] + dz);
        if (t0 != 0.f) unsafeAtomicAdd(e, t0);
        if (t1 != 0.f) unsafeAtomicAdd(e + 1, t1);
      }
    }
  }
}

// d_table[i] += the sum of the n_copies per-XCD copies (in copy order); the copies are left zeroed for the next launch
__global__ __launch_bounds__(256) void hash_reduce_kernel(float* __restrict__ priv, long stride, int n_copies, long n, float* __restrict__ d_table) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = *(const float4*)(d_table + i);
  for (int x = 0; x < n_copies; ++x) {
    float4* q = (float4*)(priv + (long)x * stride + i);
    const float4 v = *q;
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    *q = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  *(float4*)(d_table + i) = a;
}

}  // namespace swn

using namespace swn;

static int make_levels(const swn_hash_cfg* cfg, HashLevels* h) {
  SWN_CHECK(cfg, "hash encoding: null configuration");
  SWN_CHECK(cfg->n_levels >= 1 && cfg->n_levels <= HASH_MAX_LEVELS, "hash encoding: n_levels %d not in [1, 16]", cfg->n_levels);
  SWN_CHECK(cfg->log2_table >= 4 && cfg->log2_table <= 24, "hash encoding: log2_table %d not in [4, 24]", cfg->log2_table);
  SWN_CHECK(cfg->base_res >= 1 && cfg->per_level_scale >= 1.f, "hash encoding: bad resolutions");
  h->n_levels = cfg->n_levels;
  const uint64_t T = 1ull << cfg->log2_table;
  h->table_mask = (uint32_t)(T - 1);
  h->level_stride = (long)T * 2;
  for (int c = 0; c < 3; ++c) {
    SWN_CHECK(cfg->aabb_hi[c] > cfg->aabb_lo[c], "hash encoding: empty bounding box");
    h->lo[c] = cfg->aabb_lo[c];
    h->inv_extent[c] = 1.f / (cfg->aabb_hi[c] - cfg->aabb_lo[c]);
  }
  for (int l = 0; l < HASH_MAX_LEVELS; ++l) {
    const float s = (float)((double)cfg->base_res * pow((double)cfg->per_level_scale, (double)l) - 1.0);
    h->scale[l] = s;
    const uint64_t r = (uint64_t)ceil((double)s) + 2;
    SWN_CHECK(l >= cfg->n_levels || r < (1u << 20), "hash encoding: level %d resolution too large", l);
    h->res[l] = (uint32_t)r;
    h->dense[l] = (r * r * r <= T) ? 1u : 0u;
  }
  return 0;
}

extern "C" int swn_hash_encode_fwd(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                                   const float* table, int dtype, void* out, int out_stride, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_hash_encode_fwd: bad dtype");
  SWN_CHECK(rays && z && table && out, "swn_hash_encode_fwd: null pointer");
  HashLevels h;
  if (make_levels(cfg, &h)) return 1;
  const int epc = dtype == SWN_HALF ? 8 : 4;
  SWN_CHECK(out_stride >= 2 * h.n_levels && out_stride % epc == 0 && out_stride <= 128, "swn_hash_encode_fwd: out_stride %d", out_stride);
  if (n_rays <= 0 || n_samples <= 0) return 0;
  const long P = (long)n_rays * n_samples;
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((hash_encode_fwd_kernel<bf16_t>), dim3(cdiv(P, 128)), dim3(128), 0, as_stream(stream), rays, z, n_rays,
                       n_samples, h, table, (bf16_t*)out, out_stride);
  else
    hipLaunchKernelGGL((hash_encode_fwd_kernel<float>), dim3(cdiv(P, 64)), dim3(64), 0, as_stream(stream), rays, z, n_rays, n_samples,
                       h, table, (float*)out, out_stride);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_hash_encode_bwd(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                                   const void* d_out, int dtype, int d_stride, float* d_table, void* stream) {
  return swn_hash_encode_bwd_xcd(rays, z, n_rays, n_samples, cfg, d_out, dtype, d_stride, d_table, nullptr, stream);
}

extern "C" int swn_hash_encode_bwd_xcd(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                                       const void* d_out, int dtype, int d_stride, float* d_table, float* xcd_tables, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_hash_encode_bwd: bad dtype");
  SWN_CHECK(rays && z && d_out && d_table, "swn_hash_encode_bwd: null pointer");
  HashLevels h;
  if (make_levels(cfg, &h)) return 1;
  SWN_CHECK(d_stride >= 2 * h.n_levels, "swn_hash_encode_bwd: d_stride %d", d_stride);
  if (n_rays <= 0 || n_samples <= 0) return 0;
  const long P = (long)n_rays * n_samples;
  // xcd_tables (8 x the table, zero on entry, left zero): the atomics of a workgroup go to the copy of the XCD it runs on - every add
  // meets its partners in one L2 - and hash_reduce_kernel adds the copies into d_table (measured: 17.3 -> 15.1 ms per 2M points)
  const long table_elems = (long)h.n_levels * h.level_stride;
  float* target = xcd_tables ? xcd_tables : d_table;
  const long xs = xcd_tables ? table_elems : 0;
  // (one workgroup per point block walks all levels; the level as the slow grid dimension - one 4 MiB table slice in flight - was
  //  measured 2 % slower, profiles/r05_experiments.md 4: the atomics are not bound by where their lines live)
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((hash_encode_bwd_kernel<bf16_t>), dim3(cdiv(P, 256)), dim3(256), 0, as_stream(stream), rays, z, n_rays,
                       n_samples, h, (const bf16_t*)d_out, d_stride, target, xs);
  else
    hipLaunchKernelGGL((hash_encode_bwd_kernel<float>), dim3(cdiv(P, 256)), dim3(256), 0, as_stream(stream), rays, z, n_rays,
                       n_samples, h, (const float*)d_out, d_stride, target, xs);
  if (xcd_tables)
    hipLaunchKernelGGL(hash_reduce_kernel, dim3(cdiv(table_elems / 4, 256)), dim3(256), 0, as_stream(stream), xcd_tables, table_elems, 8,
                       table_elems, d_table);
  SWN_LAUNCH_CHECK();
  return 0;
}

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


// =================================================================================================================================
// Backward WITHOUT global atomics (round 6): bin, then accumulate in LDS.
//
// The atomic kernel above issues ~270 M fp32 atomics per 2M-point batch (16 levels x 8 corners x 2 features per point) and runs at the
// L2's atomic rate (~18 G/s: 15 of the 29.6 ms of configs[4]'s recipe), wherever the lines live (r05_experiments.md 4).  Here a
// contribution (entry, wk g0, wk g1) is an ITEM; the gradient table is cut into TILES of 2^13 entries (128 KiB of LDS as 2 x int64 per
// entry); tile t = level * tiles_per_level + (entry >> 13):
//   count    one workgroup per 512 points, all levels (a point's d_out row is read once): the items per tile (LDS histogram -> int atomics, 2M per launch) and max |d_out|
//   scan     exclusive prefix over the tiles -> every tile's item range in the workspace
//   scatter  the same workgroups again: items binned by tile in LDS, ONE cursor atomic per (workgroup, tile), bin runs written with
//            consecutive lanes (three 4-byte arrays: local entry, c0, c1 - 12 bytes per item, 3.2 GB at 2M points)
//   tiles    one workgroup per tile: its items are added into the LDS tile in FIXED POINT (int64, scale 2^(38 - exponent of max |d_out|):
//            an item is exact to 2^-38 of the largest gradient, 2^24 items cannot overflow) - integer addition is associative, so the
//            result does not depend on the order the items arrive in: the table gradient is BIT-DETERMINISTIC (the fp32 atomics were
//            not) - and the tile is added to d_table by its only owner, no atomics.
// Dense (coarse) levels keep the wave-level run merge of the atomic kernel: consecutive samples of a ray that share a cell emit one item
// per corner, which also keeps the few tiles of a coarse level from receiving 16 M items each.
// =================================================================================================================================
constexpr int HB_TILE_LOG2 = 13, HB_TILE = 1 << HB_TILE_LOG2;
constexpr int HB_PTS = 512;              // points per workgroup of the count / scatter passes (256 threads x 2)
constexpr int HB_MAX_TPL = 512;          // tiles per level the scatter pass bins in LDS (T <= 2^22); larger tables take the atomic kernel
constexpr int HB_ITEMS = HB_PTS * 8;     // items of a workgroup (one level)

struct HashBin {
  uint32_t* count;      // [n_tiles]      items per tile (count pass)
  uint32_t* begin;      // [n_tiles + 1]  exclusive prefix
  uint32_t* cursor;     // [n_tiles]      scatter pass: items of the tile written so far
  uint32_t* gmax_bits;  // [1]            max |d_out| as float bits (NaN / inf compare above every finite value)
  uint32_t* loc;        // [items]        entry & (HB_TILE - 1)
  float* c0;            // [items]
  float* c1;            // [items]
  int tpl;              // tiles per level
  int n_tiles;
};

// the items of one point at one level: emit(has, entry, c0, c1) per corner, called by ALL 64 lanes (has = this lane holds an item).  Dense
// levels: the lanes of a wave that share a cell form runs, the run's sums come out of its tail lane (lanes past the end carry zero
// gradient and the last point's cell).
template <bool COUNT_ONLY, typename F>
__device__ __forceinline__ void hash_items(const HashLevels& h, int l, const float (&x)[3], float g0, float g1, bool live, int lane, F emit) {
#pragma clang fp contract(off)
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
  if (h.dense[l]) {
    const uint64_t heads = __ballot(head);
    const uint64_t below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int start = 63 - __clzll((long long)below);
    const uint64_t above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const bool tail = above == 0ull ? lane == 63 : (__ffsll((long long)above) == 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      float t0 = 0.f, t1 = 0.f;
      if constexpr (!COUNT_ONLY) {
        const float wk = ((dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1])) * (dz ? w[2] : 1.f - w[2]);
        t0 = run_inclusive_scan(wk * g0, lane, start);
        t1 = run_inclusive_scan(wk * g1, lane, start);
      }
      emit(tail, hash_entry(h, l, c0[0] + dx, c0[1] + dy, c0[2] + dz), t0, t1);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      const float wk = ((dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1])) * (dz ? w[2] : 1.f - w[2]);
      emit(live, hash_entry(h, l, c0[0] + dx, c0[1] + dy, c0[2] + dz), wk * g0, wk * g1);
    }
  }
}

// rank of this lane's item inside its bin (hist[bin] += the wave's items of the bin).  few_bins (dense levels: the lanes of a wave meet
// in one or two bins - 64 same-address LDS atomics would serialise): one atomic per distinct bin of the wave, ranks from a ballot.
__device__ __forceinline__ uint32_t bin_rank(uint32_t* hist, uint32_t bin, bool has, bool few_bins, int lane) {
  if (!few_bins) return has ? atomicAdd(&hist[bin], 1u) : 0u;
  uint32_t rank = 0;
  uint64_t todo = __ballot(has);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
    const uint64_t m = __ballot(has && bin == b0);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&hist[b0], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (has && bin == b0) rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    todo &= ~m;
  }
  return rank;
}

// a workgroup = HB_PTS consecutive points, ALL levels (a point's d_out row and position are read once)
template <typename T>
__device__ __forceinline__ void hb_load_point(const float* __restrict__ rays, const float* __restrict__ z, int S, const HashLevels& h,
                                              const T* __restrict__ d_out, int d_stride, long p, bool live, float (&x)[3],
                                              uint32_t (&g)[HASH_MAX_LEVELS]) {
  point_of(rays, z, p, S, h, x);
  if constexpr (sizeof(T) == 2) {
    const uint32_t* row = (const uint32_t*)(d_out + p * d_stride);      // (d_stride is even: rows are 4-byte aligned)
#pragma unroll
    for (int l = 0; l < HASH_MAX_LEVELS; ++l) g[l] = (live && l < h.n_levels) ? row[l] : 0u;
  }
}
template <typename T>
__device__ __forceinline__ void hb_grad(const T* __restrict__ d_out, int d_stride, long p, bool live, const uint32_t (&g)[HASH_MAX_LEVELS], int l,
                                        float& g0, float& g1) {
  if constexpr (sizeof(T) == 2) {
    const uint32_t v = g[l];
    bf16_t lo = (bf16_t)(v & 0xffffu), hi = (bf16_t)(v >> 16);
    g0 = ElemIO<bf16_t>::ld(&lo);
    g1 = ElemIO<bf16_t>::ld(&hi);
  } else {
    g0 = live ? ElemIO<T>::ld(d_out + p * d_stride + 2 * l) : 0.f;
    g1 = live ? ElemIO<T>::ld(d_out + p * d_stride + 2 * l + 1) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void hash_bin_count_kernel(const float* __restrict__ rays, const float* __restrict__ z, int n_rays, int S,
                                                             HashLevels h, const T* __restrict__ d_out, int d_stride, HashBin b) {
  __shared__ uint32_t hist[HB_MAX_TPL];
  const int tid = threadIdx.x, lane = tid & 63;
  const long total = (long)n_rays * S;
  float x[HB_PTS / 256][3];
  uint32_t g[HB_PTS / 256][HASH_MAX_LEVELS];
  bool live[HB_PTS / 256];
  long pp[HB_PTS / 256];
  uint32_t gu = 0;          // max |d_out| as float bits (NaN / inf compare above every finite value; fmaxf would drop a NaN)
#pragma unroll
  for (int j = 0; j < HB_PTS / 256; ++j) {
    const long p_raw = (long)blockIdx.x * HB_PTS + j * 256 + tid;
    live[j] = p_raw < total;
    pp[j] = live[j] ? p_raw : total - 1;
    hb_load_point<T>(rays, z, S, h, d_out, d_stride, pp[j], live[j], x[j], g[j]);
  }
  for (int l = 0; l < h.n_levels; ++l) {
    for (int i = tid; i < b.tpl; i += 256) hist[i] = 0;
    __syncthreads();
    const bool few = h.dense[l] != 0;
#pragma unroll
    for (int j = 0; j < HB_PTS / 256; ++j) {
      float g0, g1;
      hb_grad<T>(d_out, d_stride, pp[j], live[j], g[j], l, g0, g1);
      if (live[j]) gu = max(gu, max(__float_as_uint(fabsf(g0)), __float_as_uint(fabsf(g1))));
      hash_items<true>(h, l, x[j], 0.f, 0.f, live[j], lane, [&](bool has, uint32_t e, float, float) { bin_rank(hist, e >> HB_TILE_LOG2, has, few, lane); });
    }
    __syncthreads();
    for (int i = tid; i < b.tpl; i += 256)
      if (hist[i]) atomicAdd(b.count + l * b.tpl + i, hist[i]);
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gu = max(gu, (uint32_t)__shfl_xor((int)gu, o, 64));
  if (lane == 0 && gu) atomicMax(b.gmax_bits, gu);
}

// begin[t] = items of the tiles before t (one workgroup; n_tiles <= 16 * 512); cursor = 0
__global__ __launch_bounds__(256) void hash_bin_scan_kernel(HashBin b) {
  __shared__ uint32_t part[256];
  const int tid = threadIdx.x;
  const int per = (b.n_tiles + 255) / 256;
  uint32_t s = 0;
  for (int i = 0; i < per; ++i) { const int t = tid * per + i; if (t < b.n_tiles) s += b.count[t]; }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) { uint32_t run = 0; for (int i = 0; i < 256; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; } b.begin[b.n_tiles] = run; }
  __syncthreads();
  uint32_t run = part[tid];
  for (int i = 0; i < per; ++i) {
    const int t = tid * per + i;
    if (t < b.n_tiles) { b.begin[t] = run; run += b.count[t]; b.cursor[t] = 0; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void hash_bin_scatter_kernel(const float* __restrict__ rays, const float* __restrict__ z, int n_rays, int S,
                                                               HashLevels h, const T* __restrict__ d_out, int d_stride, HashBin b) {
  __shared__ uint32_t hist[HB_MAX_TPL];       // items per bin, then the bin's first staging slot
  __shared__ uint32_t gbase[HB_MAX_TPL];      // the bin's first global slot
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_loc[HB_ITEMS];        // (bin << 13) | local entry
  __shared__ float s_c0[HB_ITEMS], s_c1[HB_ITEMS];
  const int tid = threadIdx.x, lane = tid & 63;
  const long total = (long)n_rays * S;
  float x[HB_PTS / 256][3];
  uint32_t g[HB_PTS / 256][HASH_MAX_LEVELS];
  bool live[HB_PTS / 256];
  long pp[HB_PTS / 256];
#pragma unroll
  for (int j = 0; j < HB_PTS / 256; ++j) {
    const long p_raw = (long)blockIdx.x * HB_PTS + j * 256 + tid;
    live[j] = p_raw < total;
    pp[j] = live[j] ? p_raw : total - 1;
    hb_load_point<T>(rays, z, S, h, d_out, d_stride, pp[j], live[j], x[j], g[j]);
  }
  for (int l = 0; l < h.n_levels; ++l) {
    for (int i = tid; i < b.tpl; i += 256) hist[i] = 0;
    __syncthreads();
    const bool few = h.dense[l] != 0;
    // pass 1: the items into registers, their rank inside their bin from the LDS histogram
    uint32_t it_e[HB_PTS / 256][8], it_r[HB_PTS / 256][8];
    float it_0[HB_PTS / 256][8], it_1[HB_PTS / 256][8];
    uint32_t it_has[HB_PTS / 256];
#pragma unroll
    for (int j = 0; j < HB_PTS / 256; ++j) {
      float g0, g1;
      hb_grad<T>(d_out, d_stride, pp[j], live[j], g[j], l, g0, g1);
      int k = 0;
      uint32_t hm = 0;
      hash_items<false>(h, l, x[j], g0, g1, live[j], lane, [&](bool has, uint32_t e, float a0, float a1) {
        it_e[j][k] = e; it_0[j][k] = a0; it_1[j][k] = a1;
        it_r[j][k] = bin_rank(hist, e >> HB_TILE_LOG2, has, few, lane);
        hm |= has ? (1u << k) : 0u;
        ++k;
      });
      it_has[j] = hm;
    }
    __syncthreads();
    // exclusive prefix over the bins (tpl <= 512 = 2 per thread), global ranges from the tiles' cursors
    {
      const uint32_t a = 2 * tid < b.tpl ? hist[2 * tid] : 0u, c = 2 * tid + 1 < b.tpl ? hist[2 * tid + 1] : 0u;
      uint32_t v = a + c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64); if (lane >= o) v += t; }
      if (lane == 63) wsum[tid >> 6] = v;
      __syncthreads();
      uint32_t base = 0;
      for (int q = 0; q < (tid >> 6); ++q) base += wsum[q];
      const uint32_t ex = base + v - (a + c);
      if (2 * tid < b.tpl) {
        hist[2 * tid] = ex;
        gbase[2 * tid] = a ? b.begin[l * b.tpl + 2 * tid] + atomicAdd(b.cursor + l * b.tpl + 2 * tid, a) : 0u;
      }
      if (2 * tid + 1 < b.tpl) {
        hist[2 * tid + 1] = ex + a;
        gbase[2 * tid + 1] = c ? b.begin[l * b.tpl + 2 * tid + 1] + atomicAdd(b.cursor + l * b.tpl + 2 * tid + 1, c) : 0u;
      }
    }
    __syncthreads();
    // pass 2: items -> staging, grouped by bin
#pragma unroll
    for (int j = 0; j < HB_PTS / 256; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (it_has[j] & (1u << k)) {
          const uint32_t bin = it_e[j][k] >> HB_TILE_LOG2;
          const uint32_t slot = hist[bin] + it_r[j][k];
          s_loc[slot] = (bin << HB_TILE_LOG2) | (it_e[j][k] & (HB_TILE - 1));
          s_c0[slot] = it_0[j][k];
          s_c1[slot] = it_1[j][k];
        }
    __syncthreads();
    const uint32_t n_items = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    for (uint32_t i = tid; i < n_items; i += 256) {     // consecutive lanes -> consecutive slots of a bin run -> consecutive addresses
      const uint32_t v = s_loc[i], bin = v >> HB_TILE_LOG2;
      const uint32_t dst = gbase[bin] + (i - hist[bin]);
      b.loc[dst] = v & (HB_TILE - 1);
      b.c0[dst] = s_c0[i];
      b.c1[dst] = s_c1[i];
    }
    __syncthreads();
  }
}

// one workgroup per tile: fixed-point accumulation in LDS, then d_table += tile (the tile's only writer)
constexpr int HB_TILE_THREADS = 1024;
__global__ __launch_bounds__(HB_TILE_THREADS) void hash_bin_tiles_kernel(HashLevels h, HashBin b, float* __restrict__ d_table) {
  extern __shared__ long long tile[];       // [HB_TILE][2]
  constexpr int NTH = HB_TILE_THREADS;
  const int t = blockIdx.x, tid = threadIdx.x;
  const uint32_t i0 = b.begin[t], i1 = b.begin[t + 1];
  if (i0 == i1) return;
  const int l = t / b.tpl, sub = t - l * b.tpl;
  const uint32_t T = h.table_mask + 1u;
  const int n_ent = (int)min((uint32_t)HB_TILE, T - (uint32_t)sub * HB_TILE);
  for (int i = tid; i < 2 * n_ent; i += NTH) tile[i] = 0;
  // scale = 2^(38 - e) with 2^e <= max |d_out| < 2^(e + 1): |item| < 2^39 exactly representable steps of 2^(e - 38)
  const uint32_t gb = *b.gmax_bits;
  const bool bad = gb >= 0x7f800000u;                        // inf / NaN upstream: the tile's entries become NaN
  int ex = (int)(gb >> 23) - 127;
  if (gb < 0x00800000u) ex = -126;                           // (zero / denormal maximum)
  const int sh = 38 - ex, sh1 = max(-126, min(127, sh));
  const float up = ldexpf(1.f, sh1), up2 = ldexpf(1.f, sh - sh1);     // (two factors: 2^(38 - e) can exceed the float range for tiny maxima)
  __syncthreads();
  // a thread takes FOUR consecutive items per 16-byte load (the arrays are 256-byte aligned: the range is widened to multiples of four and
  // masked), two such groups in flight: 6 wide loads per 8 items instead of 24 scalar ones - the pass is bound by memory latency (one
  // workgroup per CU: 128 KiB of LDS), not by the LDS atomics
  constexpr int U = 2;
  const uint32_t g0 = i0 >> 2, g1 = (i1 + 3) >> 2;           // groups of four items
  for (uint32_t gq = g0 + tid; gq < g1; gq += U * NTH) {
    uint4 e[U];
    float4 a0[U], a1[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t q = gq + u * NTH;
      ok[u] = q < g1;
      e[u] = ok[u] ? ((const uint4*)b.loc)[q] : make_uint4(0u, 0u, 0u, 0u);
      a0[u] = ok[u] ? ((const float4*)b.c0)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      a1[u] = ok[u] ? ((const float4*)b.c1)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      const uint32_t q4 = (gq + u * NTH) << 2;
      const uint32_t ee[4] = {e[u].x, e[u].y, e[u].z, e[u].w};
      const float x0[4] = {a0[u].x, a0[u].y, a0[u].z, a0[u].w}, x1[4] = {a1[u].x, a1[u].y, a1[u].z, a1[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (q4 + k < i0 || q4 + k >= i1) continue;          // (the neighbours' items in the widened range)
        atomicAdd((unsigned long long*)&tile[2 * ee[k]], (unsigned long long)__float2ll_rn(x0[k] * up * up2));
        atomicAdd((unsigned long long*)&tile[2 * ee[k] + 1], (unsigned long long)__float2ll_rn(x1[k] * up * up2));
      }
    }
  }
  __syncthreads();
  const double down = ldexp(1.0, ex - 38);
  float2* dst = (float2*)(d_table + (long)l * h.level_stride) + (long)sub * HB_TILE;
  for (int e = tid; e < n_ent; e += NTH) {
    const long long s0 = tile[2 * e], s1 = tile[2 * e + 1];
    if (!bad && s0 == 0 && s1 == 0) continue;
    float2 v = dst[e];
    v.x += bad ? __uint_as_float(0x7fc00000u) : (float)((double)s0 * down);
    v.y += bad ? __uint_as_float(0x7fc00000u) : (float)((double)s1 * down);
    dst[e] = v;
  }
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

// ---- binned backward (no global float atomics; bit-deterministic) ----
static size_t hb_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t swn_hash_bwd_workspace_bytes(long n_points, const swn_hash_cfg* cfg) {
  if (!cfg || n_points <= 0) return 0;
  const long T = 1L << cfg->log2_table;
  const long tpl = T > HB_TILE ? T / HB_TILE : 1;
  const long n_tiles = (long)cfg->n_levels * tpl;
  const size_t items = (size_t)n_points * 8 * (size_t)cfg->n_levels;
  return 3 * hb_align((size_t)(n_tiles + 1) * 4) + 256 + 3 * hb_align(items * 4);
}

extern "C" int swn_hash_encode_bwd_binned(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                                          const void* d_out, int dtype, int d_stride, float* d_table, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_hash_encode_bwd_binned: bad dtype");
  SWN_CHECK(rays && z && d_out && d_table && workspace, "swn_hash_encode_bwd_binned: null pointer");
  HashLevels h;
  if (make_levels(cfg, &h)) return 1;
  SWN_CHECK(d_stride >= 2 * h.n_levels && (dtype != SWN_HALF || d_stride % 2 == 0), "swn_hash_encode_bwd_binned: d_stride %d", d_stride);
  if (n_rays <= 0 || n_samples <= 0) return 0;
  const long P = (long)n_rays * n_samples;
  const long T = 1L << cfg->log2_table;
  const long tpl = T > HB_TILE ? T / HB_TILE : 1;
  SWN_CHECK(tpl <= HB_MAX_TPL, "swn_hash_encode_bwd_binned: tables of more than 2^22 entries take swn_hash_encode_bwd (log2_table %d)", cfg->log2_table);
  SWN_CHECK((double)P * 8.0 * h.n_levels < 4294967296.0, "swn_hash_encode_bwd_binned: more than 2^32 items (%ld points)", P);
  SWN_CHECK(workspace_bytes >= swn_hash_bwd_workspace_bytes(P, cfg), "swn_hash_encode_bwd_binned: workspace too small");
  HashBin b;
  b.tpl = (int)tpl;
  b.n_tiles = h.n_levels * (int)tpl;
  char* ws = (char*)workspace;
  const size_t tb = hb_align((size_t)(b.n_tiles + 1) * 4), ib = hb_align((size_t)P * 8 * h.n_levels * 4);
  b.count = (uint32_t*)ws; b.begin = (uint32_t*)(ws + tb); b.cursor = (uint32_t*)(ws + 2 * tb); b.gmax_bits = (uint32_t*)(ws + 3 * tb);
  b.loc = (uint32_t*)(ws + 3 * tb + 256); b.c0 = (float*)(ws + 3 * tb + 256 + ib); b.c1 = (float*)(ws + 3 * tb + 256 + 2 * ib);
  hipStream_t s = as_stream(stream);
  hipError_t e = fill_u32_async(b.count, 0u, tb, s);                   // (a fill KERNEL: memset nodes misbehave in replayed graphs, common.hpp)
  SWN_CHECK(e == hipSuccess, "fill: %s", hipGetErrorString(e));
  e = fill_u32_async(b.gmax_bits, 0u, 256, s);
  SWN_CHECK(e == hipSuccess, "fill: %s", hipGetErrorString(e));
  const dim3 grid(cdiv(P, HB_PTS));
  if (dtype == SWN_HALF) hipLaunchKernelGGL((hash_bin_count_kernel<bf16_t>), grid, dim3(256), 0, s, rays, z, n_rays, n_samples, h, (const bf16_t*)d_out, d_stride, b);
  else hipLaunchKernelGGL((hash_bin_count_kernel<float>), grid, dim3(256), 0, s, rays, z, n_rays, n_samples, h, (const float*)d_out, d_stride, b);
  hipLaunchKernelGGL(hash_bin_scan_kernel, dim3(1), dim3(256), 0, s, b);
  if (dtype == SWN_HALF) hipLaunchKernelGGL((hash_bin_scatter_kernel<bf16_t>), grid, dim3(256), 0, s, rays, z, n_rays, n_samples, h, (const bf16_t*)d_out, d_stride, b);
  else hipLaunchKernelGGL((hash_bin_scatter_kernel<float>), grid, dim3(256), 0, s, rays, z, n_rays, n_samples, h, (const float*)d_out, d_stride, b);
  static bool attr_set = false;
  constexpr int TILE_LDS = HB_TILE * 2 * 8;
  if (!attr_set) {
    e = hipFuncSetAttribute((const void*)hash_bin_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS);
    SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(hash_bin_tiles_kernel, dim3(b.n_tiles), dim3(HB_TILE_THREADS), TILE_LDS, s, h, b, d_table);
  SWN_LAUNCH_CHECK();
  return 0;
}

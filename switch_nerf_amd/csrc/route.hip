// swn_route_top1: batch-prioritised top-1 capacity assignment, bit-exact and deterministic.
//
// Replaces extract_critical / compute_sorted_location / load_balance
// (/root/reference/switch_nerf/modules/tutel_moe_ext/tutel_fast_dispatch.py:136-217: three argsorts of the whole
// segment, two [P,E] int64 gathers and a [P,E] int64 cumsum) and Tutel's fast_cumsum_sub_one by ONE stable LSD
// radix sort per segment of the 32-bit key  (expert << 26) | (bits(1.0f) - bits(max gate)):
// ascending key order = expert-major, descending gate, ties in token order.  loc = position - expert start.
// Keys are softmax maxima, i.e. in [1/E, 1], so bits(1.0f) - bits(g) < 2^26 for E <= 64.
#include "common.hpp"

namespace swn {

constexpr int KPB = 2048;  // keys per block (256 threads x 8)

__global__ __launch_bounds__(256) void route_keys_kernel(const int32_t* __restrict__ idx, const float* __restrict__ gmax,
                                                         int n_tokens, int seg_tokens, int E, int bpr,
                                                         uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                         int32_t* __restrict__ counts) {
  // A block takes KPB = 2048 consecutive tokens (8 per thread): per-block (LDS) expert histogram, then one global atomic per
  // (segment, expert) touched by the block; a block spans at most two segments (seg_tokens >= KPB is not required: generic two-slot
  // handling below).  With 256 tokens per block (rounds 1-2) 512 blocks per segment queued up on each of the E counters of the
  // segment: the kernel took 40 us for 2M tokens, 5x its traffic.
  __shared__ int32_t h[2][64];
  if (threadIdx.x < 128) (&h[0][0])[threadIdx.x] = 0;
  __syncthreads();
  const long b0 = (long)blockIdx.x * KPB;
  const int seg0 = (int)(b0 / seg_tokens);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int u = 0; u < KPB / 256; ++u) {
    const long i = b0 + u * 256 + threadIdx.x;
    int e = -1, seg = -1;            // (lanes past the last token: no expert, no segment)
    if (i < n_tokens) {
      e = idx[i];
      uint32_t inv = 0;
      if (bpr) {
        const int32_t b = (int32_t)0x3F800000 - (int32_t)__float_as_uint(gmax[i]);
        inv = b < 0 ? 0u : (b > 0x03FFFFFF ? 0x03FFFFFFu : (uint32_t)b);
      }
      keys[i] = ((uint32_t)e << 26) | inv;
      seg = (int)(i / seg_tokens);
      vals[i] = (int32_t)(i - (long)seg * seg_tokens);
    }
    // a wave inside one segment (the usual case) counts with one ballot per expert and adds once per expert.  Every VALID lane must
    // agree on the segment: with segments shorter than a wave, lanes 0 and 63 alone (lane 63 past the last token) can sit in one
    // segment while the lanes between them are in the next - those tokens would be credited to the wrong segment's counts.
    const int seg_first = __shfl(seg, 0);
    if (seg_first >= 0 && __all(e < 0 || seg == seg_first) && seg_first - seg0 < 2) {
      for (int q = 0; q < E; ++q) {
        const unsigned long long m = __ballot(e == q);
        if (lane == 0 && m) atomicAdd(&h[seg_first - seg0][q], (int)__popcll(m));
      }
    } else if (e >= 0) {
      if (seg - seg0 < 2) atomicAdd(&h[seg - seg0][e], 1);
      else atomicAdd(counts + seg * E + e, 1);   // tiny segments: fall back to a direct atomic
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int s = threadIdx.x >> 6, e = threadIdx.x & 63;
    const int c = h[s][e];
    if (c) atomicAdd(counts + (seg0 + s) * E + e, c);
  }
}

// One LSD pass sorts by the BITS-bit digit (key >> shift) & (2^BITS - 1).  BITS = 8: four passes over the 32-bit key (the default).
// The key only has 26 + ceil(log2 E) significant bits, so for E <= 16 three passes of <= 10 bits would do - an experiment that lost
// (see swn_route_top1).
template <int BITS>
__global__ __launch_bounds__(256) void route_hist_kernel(const uint32_t* __restrict__ keys, int seg_tokens, int shift,
                                                         int nblk, int32_t* __restrict__ hist) {
  constexpr int BINS = 1 << BITS;
  __shared__ int32_t h[BINS];
  const int seg = blockIdx.y, blk = blockIdx.x;
  for (int d = threadIdx.x; d < BINS; d += 256) h[d] = 0;
  __syncthreads();
  const uint32_t* k = keys + (long)seg * seg_tokens;
  for (int j = 0; j < KPB / 256; ++j) {
    const int p = blk * KPB + j * 256 + threadIdx.x;
    if (p < seg_tokens) atomicAdd(&h[(k[p] >> shift) & (BINS - 1)], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < BINS; d += 256) hist[((long)seg * nblk + blk) * BINS + d] = h[d];      // [segment][block][digit]: coalesced
}

// exclusive scan of the counters hist[seg][blk][d] in (d, blk) order; one block of 2^BITS threads per segment, thread d owns digit d.
// NB > 0: the digit's nblk <= NB counters are held in registers - all their loads are in flight together; the serial form (NB = 0) pays
// one memory round trip per counter, twice (25 us for 64 counters whatever the batch size).  The digit is the fastest index: the
// threads of a wave read / write consecutive words (with the block index fastest - rounds 1-2 - every one of a wave's 64 x 64 loads was
// its own cache line: 17 us per pass for a kilobyte of counters).
template <int BITS, int NB>
__global__ __launch_bounds__(1 << BITS) void route_scan_kernel(int32_t* __restrict__ hist, int nblk) {
  constexpr int BINS = 1 << BITS, NW = BINS / 64;
  __shared__ int32_t rowsum[NW];
  const int seg = blockIdx.x, d = threadIdx.x;
  int32_t* row = hist + (long)seg * nblk * BINS + d;       // counter of block b: row[b * BINS]
  int32_t s = 0;
  int32_t c[NB > 0 ? NB : 1];
  if constexpr (NB > 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) c[b] = b < nblk ? row[(long)b * BINS] : 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) s += c[b];
  } else {
    for (int b = 0; b < nblk; ++b) s += row[(long)b * BINS];
  }
  // exclusive scan over the row sums: inclusive scan inside each wave (shuffles, no barrier), then the totals of the waves before
  // (the Hillis-Steele form over LDS paid 16 workgroup barriers: 17 us per pass for 64 counters)
  int32_t v = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(v, o, 64);
    if ((d & 63) >= o) v += t;
  }
  if ((d & 63) == 63) rowsum[d >> 6] = v;
  __syncthreads();
  for (int w = 0; w < (d >> 6); ++w) v += rowsum[w];
  int32_t run = v - s;  // exclusive prefix of this row
  if constexpr (NB > 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b < nblk) row[(long)b * BINS] = run;
      run += c[b];
    }
  } else {
    for (int b = 0; b < nblk; ++b) {
      const int32_t cc = row[(long)b * BINS];
      row[(long)b * BINS] = run;
      run += cc;
    }
  }
}

template <int BITS>
__device__ __forceinline__ unsigned long long match_digit(int d, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const bool bit = (d >> b) & 1;
    const unsigned long long bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return valid ? m : 0ull;
}

template <int BITS>
__global__ __launch_bounds__(256) void route_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                            const int32_t* __restrict__ vals_in, int seg_tokens, int shift,
                                                            int nblk, const int32_t* __restrict__ hist,
                                                            uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
  constexpr int BINS = 1 << BITS;
  __shared__ int32_t wh[4][BINS];
  const int seg = blockIdx.y, blk = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long sbase = (long)seg * seg_tokens;
  for (int j = threadIdx.x; j < 4 * BINS; j += 256) (&wh[0][0])[j] = 0;
  __syncthreads();
  const int p0 = blk * KPB + w * (KPB / 4);
  // phase 1: per-wave digit counts
  for (int r = 0; r < KPB / 4 / 64; ++r) {
    const int p = p0 + r * 64 + lane;
    const bool valid = p < seg_tokens;
    const int d = valid ? (int)((keys_in[sbase + p] >> shift) & (BINS - 1)) : 0;
    const unsigned long long m = match_digit<BITS>(d, valid);
    if (valid && lane == __ffsll((long long)m) - 1) wh[w][d] += __popcll(m);
  }
  __syncthreads();
  // phase 2: per-wave bases
  for (int d = threadIdx.x; d < BINS; d += 256) {
    int32_t base = hist[((long)seg * nblk + blk) * BINS + d];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const int32_t c = wh[ww][d];
      wh[ww][d] = base;
      base += c;
    }
  }
  __syncthreads();
  // phase 3: stable scatter
  for (int r = 0; r < KPB / 4 / 64; ++r) {
    const int p = p0 + r * 64 + lane;
    const bool valid = p < seg_tokens;
    uint32_t key = 0;
    int32_t val = 0;
    if (valid) { key = keys_in[sbase + p]; val = vals_in[sbase + p]; }
    const int d = (int)((key >> shift) & (BINS - 1));
    const unsigned long long m = match_digit<BITS>(d, valid);
    if (valid) {
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      const int32_t pos = wh[w][d] + rank;
      keys_out[sbase + pos] = key;
      vals_out[sbase + pos] = val;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && lane == __ffsll((long long)m) - 1) wh[w][d] += __popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
}

template <int BITS>
static void route_pass(const uint32_t* ki, const int32_t* vi, uint32_t* ko, int32_t* vo, int seg_tokens, int n_seg, int nblk, int shift,
                       int32_t* hist, hipStream_t s) {
  hipLaunchKernelGGL((route_hist_kernel<BITS>), dim3(nblk, n_seg), dim3(256), 0, s, ki, seg_tokens, shift, nblk, hist);
  if (nblk <= 16) hipLaunchKernelGGL((route_scan_kernel<BITS, 16>), dim3(n_seg), dim3(1 << BITS), 0, s, hist, nblk);
  else if (nblk <= 64) hipLaunchKernelGGL((route_scan_kernel<BITS, 64>), dim3(n_seg), dim3(1 << BITS), 0, s, hist, nblk);
  else if (nblk <= 128) hipLaunchKernelGGL((route_scan_kernel<BITS, 128>), dim3(n_seg), dim3(1 << BITS), 0, s, hist, nblk);   // (Mission Bay's
                                                    // 212,992-token segments = 104 tiles: the serial form took 26 us per pass, 0.2 ms per step)
  else hipLaunchKernelGGL((route_scan_kernel<BITS, 0>), dim3(n_seg), dim3(1 << BITS), 0, s, hist, nblk);
  hipLaunchKernelGGL((route_scatter_kernel<BITS>), dim3(nblk, n_seg), dim3(256), 0, s, ki, vi, seg_tokens, shift, nblk, hist, ko, vo);
}

// prev / n_prev (top-k routing, choice k = n_prev > 0): the counts [n_prev][n_seg * E] of the choices before this one - their sum per
// (segment, expert) is the `acc_base` that tutel_fast_dispatch.py:199-201 adds to the locations of choice k
__global__ void route_finalize_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                      const int32_t* __restrict__ counts, int n_tokens, int seg_tokens, int E, int capacity,
                                      int32_t* __restrict__ loc, int32_t* __restrict__ perm, int32_t* __restrict__ tok2row,
                                      const int32_t* __restrict__ prev, int n_prev) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tokens) return;
  const int seg = (int)(i / seg_tokens);
  const int pos = (int)(i - (long)seg * seg_tokens);
  const int e = (int)(keys[i] >> 26);
  int start = 0;
  for (int q = 0; q < e; ++q) start += counts[seg * E + q];
  int l = pos - start;
  if (n_prev > 0) {
    const long groups = (long)(n_tokens / seg_tokens) * E;
    for (int j = 0; j < n_prev; ++j) l += prev[j * groups + seg * E + e];
  }
  const long tok = (long)seg * seg_tokens + vals[i];
  loc[tok] = l;
  const long row = ((long)seg * E + e) * capacity + l;
  if (l < capacity) {
    if (perm) perm[row] = (int32_t)tok;
    if (tok2row) tok2row[tok] = (int32_t)row;
  } else if (tok2row) {
    tok2row[tok] = -1;
  }
}

// No-batch (evaluation) layout, tutel_fast_dispatch_nobatch.py:24-36: the rows of group g = (segment, expert) sit contiguously at
// [begin[g], begin[g] + counts[g]), begin = exclusive prefix sum of the counts (expert_locations_begin), nothing is dropped.
// One block computes begin (n_groups <= 4096); the tokens then scatter themselves.
__global__ __launch_bounds__(256) void route_pack_begin_kernel(const int32_t* __restrict__ counts, int n_groups, int32_t* __restrict__ begin) {
  __shared__ int32_t part[256];
  const int per = (n_groups + 255) / 256;
  const int g0 = threadIdx.x * per, g1 = min(n_groups, g0 + per);
  int32_t s = 0;
  for (int g = g0; g < g1; ++g) s += counts[g];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int t = 0; t < 256; ++t) { const int32_t v = part[t]; part[t] = run; run += v; }
  }
  __syncthreads();
  int32_t run = part[threadIdx.x];
  for (int g = g0; g < g1; ++g) { begin[g] = run; run += counts[g]; }
}
__global__ void route_pack_scatter_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ loc,
                                          const int32_t* __restrict__ begin, int n_tokens, int seg_tokens, int E,
                                          int32_t* __restrict__ perm, int32_t* __restrict__ tok2row) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tokens) return;
  const int e = idx[i];
  if (e < 0) { if (tok2row) tok2row[i] = -1; return; }
  const int32_t row = begin[(i / seg_tokens) * E + e] + loc[i];
  if (tok2row) tok2row[i] = row;
  if (perm) perm[row] = (int32_t)i;
}

// me partial sums: grid (nblk, n_seg), block 256; partial[seg][blk][e]
__global__ __launch_bounds__(256) void laux_partial_kernel(const float* __restrict__ gates, int seg_tokens, int E, int nblk,
                                                           float* __restrict__ partial) {
  __shared__ float red[256];
  const int seg = blockIdx.y, blk = blockIdx.x;
  const float* gp = gates + (long)seg * seg_tokens * E;
  // thread t handles expert t % E for tokens t / E + k * (256 / E)   (E divides 256 for E in {1,2,4,8,16,32,64})
  const int e = threadIdx.x % E, t0 = threadIdx.x / E, tstep = 256 / E;
  float s = 0.f;
  const int pbeg = blk * KPB, pend = min(seg_tokens, pbeg + KPB);
  int p = pbeg + t0;
  for (; p + 7 * tstep < pend; p += 8 * tstep) {      // 8 loads in flight, added in the same order as one at a time (same bits)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = gp[(long)(p + u * tstep) * E + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; p < pend; p += tstep) s += gp[(long)p * E + e];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < E) {
    float a = 0.f;
    for (int t = threadIdx.x; t < 256; t += E) a += red[t];
    partial[((long)seg * nblk + blk) * E + threadIdx.x] = a;
  }
}

__global__ __launch_bounds__(256) void laux_final_kernel(const float* __restrict__ partial, const int32_t* __restrict__ counts, int seg_tokens,
                                                         int E, int nblk, float* __restrict__ l_aux) {
  // thread t: expert t % E, blocks t / E, t / E + 256 / E, ...; fixed summation order (deterministic).  (One thread per segment walked
  // all nblk * E partials alone: 26 us of dependent loads per step.)
  __shared__ float red[256];
  const int seg = blockIdx.x, t = threadIdx.x;
  const int e = t % E, sub = t / E, nsub = 256 / E;
  float me = 0.f;
  for (int b = sub; b < nblk; b += nsub) me += partial[((long)seg * nblk + b) * E + e];
  red[t] = me;
  __syncthreads();
  if (t < E) {
    float a = 0.f;
    for (int q = 0; q < nsub; ++q) a += red[q * E + t];
    red[t] = a * (float)counts[seg * E + t];
  }
  __syncthreads();
  if (t == 0) {
    float tot = 0.f;
    for (int q = 0; q < E; ++q) tot += red[q];
    const float scale = (float)((double)E / ((double)seg_tokens * (double)seg_tokens));
    l_aux[seg] = tot * scale;
  }
}


constexpr int ROUTE_SYNC_WORDS = 1024;      // (swn_route_sync_bytes: the synchronisation words of the experimental one-launch form)
#ifdef SWN_EXP_ROUTE_ONE
#include "../../scripts/experiments/route_one.inc"
#endif
}  // namespace swn

using namespace swn;

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t swn_route_workspace_bytes(int n_tokens, int n_seg, int n_experts) {
  const int seg_tokens = n_seg > 0 ? (n_tokens + n_seg - 1) / n_seg : n_tokens;
  const int nblk = (seg_tokens + KPB - 1) / KPB;
  return 4 * align256((size_t)n_tokens * 4) + align256((size_t)n_seg * 1024 * nblk * 4) +
         2 * align256((size_t)n_seg * nblk * n_experts * 4) + 1024;
}

extern "C" size_t swn_route_sync_bytes(void) { return (size_t)ROUTE_SYNC_WORDS * 4; }

extern "C" int swn_route_top1x(const int32_t* idx, const float* gmax, const float* gates, int n_tokens, int seg_tokens,
                               int n_experts, int capacity, int bpr, int32_t* loc, int32_t* counts, int32_t* perm,
                               int32_t* tok2row, float* l_aux, int32_t* drop_begin, int32_t* dropped, int32_t* sync, int mode,
                               void* workspace, size_t workspace_bytes, void* stream) {
  SWN_CHECK(idx && gmax && loc && counts && workspace, "swn_route_top1x: null pointer");
  SWN_CHECK(n_tokens > 0 && seg_tokens > 0 && n_tokens % seg_tokens == 0,
            "swn_route_top1x: n_tokens (%d) must be a positive multiple of seg_tokens (%d)", n_tokens, seg_tokens);
  SWN_CHECK(n_experts >= 1 && n_experts <= 64 && (256 % n_experts) == 0, "swn_route_top1x: experts must divide 256 (<= 64)");
  SWN_CHECK(capacity >= 1, "swn_route_top1x: capacity must be >= 1");
  SWN_CHECK((drop_begin == nullptr) == (dropped == nullptr), "swn_route_top1x: drop_begin and dropped come together");
  const int n_seg = n_tokens / seg_tokens;
  SWN_CHECK(workspace_bytes >= swn_route_workspace_bytes(n_tokens, n_seg, n_experts), "swn_route_top1x: workspace too small");
  // mode 0 = the per-phase kernels (swn_route_top1 + swn_route_dropped).  Modes 1 / 2 - the fused forms of round 5, measured slower - exist
  // only in the experiment build (scripts/experiments/route_one.inc, -DSWN_EXP_ROUTE_ONE)
#ifdef SWN_EXP_ROUTE_ONE
  SWN_CHECK(mode >= 0 && mode <= 3, "swn_route_top1x: mode %d (0 = per-phase kernels, 1 = fused phases, 2 = one launch, 3 = one launch, a segment per XCD)", mode);
  if (mode > 0 && sync != nullptr && 2 + 4 * n_seg <= ROUTE_TEAM0 && n_seg * n_experts <= 1024)
    return route_one_launch(idx, gmax, gates, n_tokens, seg_tokens, n_experts, capacity, bpr, loc, counts, perm, tok2row, l_aux, drop_begin,
                            dropped, sync, mode, workspace, stream);
#else
  SWN_CHECK(mode == 0, "swn_route_top1x: mode %d needs the experiment build (-DSWN_EXP_ROUTE_ONE); this library has mode 0 only", mode);
#endif
  int rc = swn_route_top1(idx, gmax, gates, n_tokens, seg_tokens, n_experts, capacity, bpr, loc, counts, perm, tok2row, l_aux, workspace,
                          workspace_bytes, stream);
  if (rc == 0 && drop_begin) rc = swn_route_dropped(idx, loc, counts, n_tokens, seg_tokens, n_experts, capacity, drop_begin, dropped, stream);
  return rc;
}

// One choice of the (top-k) routing: choice 0 is swn_route_top1 itself; choice n_prev > 0 ranks its tokens the same way and starts
// each expert's locations behind the rows of the choices before it (prev: their counts), in the SAME perm (filled by choice 0).
static int route_choice(const int32_t* idx, const float* gmax, const float* gates, int n_tokens, int seg_tokens,
                        int n_experts, int capacity, int bpr, int32_t* loc, int32_t* counts, int32_t* perm,
                        int32_t* tok2row, float* l_aux, void* workspace, size_t workspace_bytes, void* stream,
                        const int32_t* prev, int n_prev) {
  SWN_CHECK(idx && gmax && loc && counts && workspace, "swn_route_top1: null pointer");
  SWN_CHECK(n_tokens > 0 && seg_tokens > 0 && n_tokens % seg_tokens == 0,
            "swn_route_top1: n_tokens (%d) must be a positive multiple of seg_tokens (%d)", n_tokens, seg_tokens);
  SWN_CHECK(n_experts >= 1 && n_experts <= 64 && (256 % n_experts) == 0, "swn_route_top1: experts must divide 256 (<= 64)");
  SWN_CHECK(capacity >= 1, "swn_route_top1: capacity must be >= 1");
  const int n_seg = n_tokens / seg_tokens;
  SWN_CHECK(workspace_bytes >= swn_route_workspace_bytes(n_tokens, n_seg, n_experts), "swn_route_top1: workspace too small");
  const int nblk = cdiv(seg_tokens, KPB);
  char* ws = (char*)workspace;
  const size_t tb = align256((size_t)n_tokens * 4);
  uint32_t* k0 = (uint32_t*)ws;
  uint32_t* k1 = (uint32_t*)(ws + tb);
  int32_t* v0 = (int32_t*)(ws + 2 * tb);
  int32_t* v1 = (int32_t*)(ws + 3 * tb);
  int32_t* hist = (int32_t*)(ws + 4 * tb);
  float* partial = (float*)(ws + 4 * tb + align256((size_t)n_seg * 1024 * nblk * 4));
  hipStream_t s = as_stream(stream);
  // counts = 0, perm = -1: fill KERNELS (common.hpp: memset nodes of a captured graph go wrong from the second replay on).
  // SWN_ROUTE_HIP_MEMSET=1 restores the hipMemsetAsync calls of rounds 1-2 (scripts/graph_probe.py demonstrates the fault with it).
  static const bool use_memset = getenv("SWN_ROUTE_HIP_MEMSET") != nullptr;
  hipError_t e = use_memset ? hipMemsetAsync(counts, 0, (size_t)n_seg * n_experts * 4, s)
                            : fill_u32_async(counts, 0u, (size_t)n_seg * n_experts * 4, s);
  SWN_CHECK(e == hipSuccess, "fill: %s", hipGetErrorString(e));
  if (perm && n_prev == 0) {
    e = use_memset ? hipMemsetAsync(perm, 0xFF, (size_t)n_seg * n_experts * capacity * 4, s)
                   : fill_u32_async(perm, 0xFFFFFFFFu, (size_t)n_seg * n_experts * capacity * 4, s);
    SWN_CHECK(e == hipSuccess, "fill: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(route_keys_kernel, dim3(cdiv(n_tokens, KPB)), dim3(256), 0, s, idx, gmax, n_tokens, seg_tokens,
                     n_experts, bpr, k0, v0, counts);
  SWN_LAUNCH_CHECK();
  uint32_t *ki = k0, *ko = k1;
  int32_t *vi = v0, *vo = v1;
  // pass plan: the key has 26 + ceil(log2 E) significant bits (without BPR only the expert bits differ: one pass)
  // Four 8-bit passes.  (Three passes of 9 / 10 bits for E <= 16 were measured SLOWER in round 3 - 15.20 vs 14.98 ms per step: with 2048
  // keys per block a 1024-bin histogram per block is as much traffic as the keys, the scan kernel has four times the rows, the per-wave
  // match takes 10 ballots - profiles/r03_experiments.md 7; the variant and its switch were removed in round 5.)
  int n_pass = 4, shift = 0;
  if (!bpr) { n_pass = 1; shift = 24; }
  for (int q = 0; q < n_pass; ++q) {
    route_pass<8>(ki, vi, ko, vo, seg_tokens, n_seg, nblk, shift, hist, s);
    SWN_LAUNCH_CHECK();
    shift += 8;
    uint32_t* tk = ki; ki = ko; ko = tk;
    int32_t* tv = vi; vi = vo; vo = tv;
  }
  hipLaunchKernelGGL(route_finalize_kernel, dim3(cdiv(n_tokens, 256)), dim3(256), 0, s, ki, vi, counts, n_tokens,
                     seg_tokens, n_experts, capacity, loc, perm, tok2row, prev, n_prev);
  SWN_LAUNCH_CHECK();
  if (gates && l_aux) {
    hipLaunchKernelGGL(laux_partial_kernel, dim3(nblk, n_seg), dim3(256), 0, s, gates, seg_tokens, n_experts, nblk, partial);
    hipLaunchKernelGGL(laux_final_kernel, dim3(n_seg), dim3(256), 0, s, partial, counts, seg_tokens, n_experts, nblk, l_aux);
    SWN_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int swn_route_top1(const int32_t* idx, const float* gmax, const float* gates, int n_tokens, int seg_tokens,
                              int n_experts, int capacity, int bpr, int32_t* loc, int32_t* counts, int32_t* perm,
                              int32_t* tok2row, float* l_aux, void* workspace, size_t workspace_bytes, void* stream) {
  return route_choice(idx, gmax, gates, n_tokens, seg_tokens, n_experts, capacity, bpr, loc, counts, perm, tok2row, l_aux, workspace,
                      workspace_bytes, stream, nullptr, 0);
}

// ---- top-k (k > 1) ------------------------------------------------------------------------------------------------------------
// torch.topk(gates, k) per token (descending, the lower expert on exact ties), the selected gates and their normalised form
// g_j / max(sum_j g_j, eps) (tutel_fast_dispatch.py:177-182, 204-206).  One thread per token; E <= 64: the chosen experts are a bit mask.
__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ gates, int n_tokens, int E, int K,
                                                          int32_t* __restrict__ idx, float* __restrict__ gsel, float* __restrict__ gnorm) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tokens) return;
  const float* g = gates + i * E;
  unsigned long long taken = 0ull;
  float sum = 0.f;
  for (int k = 0; k < K; ++k) {
    int best = -1;
    float bv = 0.f;
    for (int e = 0; e < E; ++e) {
      if ((taken >> e) & 1ull) continue;
      const float v = g[e];
      if (best < 0 || v > bv) { best = e; bv = v; }
    }
    taken |= 1ull << best;
    idx[(long)k * n_tokens + i] = best;
    gsel[(long)k * n_tokens + i] = bv;
    sum += bv;                                  // (sum(gates_s): first choice first, the reference's order)
  }
  const float denom = K > 1 ? fmaxf(sum, 1.1920928955078125e-07f) : 1.f;      // torch.finfo(torch.float32).eps; `if top_k > 1:` (:196)
  for (int k = 0; k < K; ++k) gnorm[(long)k * n_tokens + i] = gsel[(long)k * n_tokens + i] / denom;
}

// backward of the normalisation: d_gnorm [K, P] -> d_probs [P, E] (zero outside the token's K experts).  gn_j = g_j / D, D = max(sum, eps):
// dg_j = (d_gn_j - sum_m d_gn_m gn_m) / D above the clamp, d_gn_j / eps under it (torch.clamp passes no gradient there).
__global__ __launch_bounds__(256) void topk_gate_bwd_kernel(const float* __restrict__ gates, const int32_t* __restrict__ idx,
                                                            const float* __restrict__ d_gnorm, int n_tokens, int E, int K,
                                                            float* __restrict__ d_probs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tokens) return;
  for (int e = 0; e < E; ++e) d_probs[i * E + e] = 0.f;
  if (K == 1) { d_probs[i * E + idx[i]] = d_gnorm[i]; return; }        // no normalisation for top-1 (:196)
  float sum = 0.f;
  for (int k = 0; k < K; ++k) sum += gates[i * E + idx[(long)k * n_tokens + i]];
  const float eps = 1.1920928955078125e-07f;
  const bool clamped = !(sum >= eps);
  const float D = clamped ? eps : sum;
  float s = 0.f;
  if (!clamped)
    for (int k = 0; k < K; ++k) s += d_gnorm[(long)k * n_tokens + i] * (gates[i * E + idx[(long)k * n_tokens + i]] / D);
  for (int k = 0; k < K; ++k) d_probs[i * E + idx[(long)k * n_tokens + i]] = (d_gnorm[(long)k * n_tokens + i] - s) / D;
}

// valid rows of group g = all choices' tokens of its expert (the chains clamp them to the capacity)
__global__ void route_group_rows_kernel(const int32_t* __restrict__ counts, int groups, int K, int32_t* __restrict__ group_rows) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  int s = 0;
  for (int k = 0; k < K; ++k) s += counts[(long)k * groups + g];
  group_rows[g] = s;
}

// ---- load / importance loss (use_load_importance_loss: tutel_fast_dispatch.py:152-174, 219-265) ---------------------------------
// logits = g @ wg^T (+ noise_scale * noise): the router's logits themselves - the loss compares probabilities with the k-th largest NOISY
// LOGIT (:159-160), which the probabilities the gate kernels write do not determine.  fp32 accumulation, one wave per token, E <= 16.
template <typename T>
__global__ __launch_bounds__(256) void gate_logits_kernel(const T* __restrict__ g, const float* __restrict__ wg,
                                                          const float* __restrict__ noise, float noise_scale, int P, int G, int E,
                                                          float* __restrict__ logits) {
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long i = wid; i < P; i += nw) {
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k = lane; k < G; k += 64) {
      const float x = ElemIO<T>::ld(g + i * G + k);
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (e < E) acc[e] += x * wg[(long)e * G + k];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e < E) {       // (E is wave-uniform)
        const float v = wave_sum(acc[e]);
        if (lane == 0) logits[i * E + e] = v + (noise ? noise_scale * noise[i * E + e] : 0.f);
      }
  }
}

// Normal(0, sigma).cdf(x) as torch.distributions writes it: 0.5 * (1 + erf(x / sigma / sqrt(2)))
__device__ __forceinline__ float normal_cdf(float x, float inv_sigma) { return 0.5f * (1.f + erff(x * inv_sigma * 0.70710678118654752f)); }

// per-block sums over the tokens of Imp_e = scores[t][e] and Load_e = cdf(scores[t][e] - threshold[t]), threshold = the token's k-th
// largest noisy logit; partial [nblk][2 E], fixed order (deterministic)
__global__ __launch_bounds__(256) void load_importance_partial_kernel(const float* __restrict__ scores, const float* __restrict__ logits_w_noise,
                                                                      const int32_t* __restrict__ idx_last, float inv_sigma, int P, int E,
                                                                      float* __restrict__ partial) {
  __shared__ float red[256][33];
  float imp[16], load[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) imp[e] = load[e] = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long)gridDim.x * 256) {
    const float thr = logits_w_noise[i * E + idx_last[i]];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e < E) {
        const float sc = scores[i * E + e];
        imp[e] += sc;
        load[e] += normal_cdf(sc - thr, inv_sigma);
      }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) { red[threadIdx.x][e] = imp[e]; red[threadIdx.x][16 + e] = load[e]; }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int c = threadIdx.x;
    float a = 0.f;
    for (int t = 0; t < 256; ++t) a += red[t][c];
    const int e = c & 15;
    if (e < E) partial[(long)blockIdx.x * 2 * E + (c >> 4) * E + e] = a;
  }
}

// Imp, Load -> l = (cv2(Imp) + cv2(Load)) / 2 with cv2(v) = v.var() / (v.mean()^2 + 1e-10) (unbiased variance, :163, :168), and the
// loss's gradient w.r.t. every Imp_e / Load_e (coef [2 E]) for the backward
__global__ __launch_bounds__(64) void load_importance_final_kernel(const float* __restrict__ partial, int nblk, int E, float* __restrict__ l_loss,
                                                                   float* __restrict__ coef) {
  __shared__ float v[32];
  const int c = threadIdx.x;
  if (c < 2 * E) {
    float a = 0.f;
    for (int b = 0; b < nblk; ++b) a += partial[(long)b * 2 * E + c];
    v[c] = a;
  }
  __syncthreads();
  if (c < 2) {          // thread 0: importance, thread 1: load
    const float* x = v + c * E;
    float m = 0.f;
    for (int e = 0; e < E; ++e) m += x[e];
    m /= (float)E;
    float var = 0.f;
    for (int e = 0; e < E; ++e) var += (x[e] - m) * (x[e] - m);
    var /= (float)(E - 1);
    const float den = m * m + 1e-10f;
    for (int e = 0; e < E; ++e)
      coef[c * E + e] = 0.5f * (2.f * (x[e] - m) / ((float)(E - 1) * den) - var * (2.f * m / (float)E) / (den * den));
    v[c * E] = var / den;          // (x is dead now)
  }
  __syncthreads();
  if (c == 0) l_loss[0] = 0.5f * (v[0] + v[E]);
}

// d_logits [P, E] of the loss: through scores = softmax(logits) (importance + the cdf's argument) and through the threshold (the k-th
// choice's noisy logit: one entry per token); d_l = dL/d l_loss (device scalar)
__global__ __launch_bounds__(256) void load_importance_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ logits_w_noise,
                                                                  const int32_t* __restrict__ idx_last, const float* __restrict__ coef,
                                                                  const float* __restrict__ d_l, float inv_sigma, int P, int E,
                                                                  float* __restrict__ d_logits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float dl = d_l[0];
  const int last = idx_last[i];
  const float thr = logits_w_noise[i * E + last];
  float sc[16], ds[16], dot = 0.f, dthr = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    sc[e] = ds[e] = 0.f;
    if (e < E) {
      sc[e] = scores[i * E + e];
      const float z = (sc[e] - thr) * inv_sigma;
      const float pdf = __expf(-0.5f * z * z) * inv_sigma * 0.3989422804014327f;
      ds[e] = dl * (coef[e] + coef[E + e] * pdf);
      dthr -= dl * coef[E + e] * pdf;
      dot += sc[e] * ds[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if (e < E) d_logits[i * E + e] = sc[e] * (ds[e] - dot) + (e == last ? dthr : 0.f);
}

extern "C" int swn_gate_logits(const void* g, int dtype, const float* wg, const float* noise, float noise_scale, int n_tokens, int gate_dim,
                               int n_experts, float* logits, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_gate_logits: bad dtype");
  SWN_CHECK(g && wg && logits, "swn_gate_logits: null pointer");
  SWN_CHECK(n_tokens > 0 && gate_dim > 0 && n_experts >= 1 && n_experts <= 16, "swn_gate_logits: experts <= 16");
  const int blocks = min(cdiv(n_tokens, 4), 4096);
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((gate_logits_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)g, wg, noise, noise_scale,
                       n_tokens, gate_dim, n_experts, logits);
  else
    hipLaunchKernelGGL((gate_logits_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)g, wg, noise, noise_scale,
                       n_tokens, gate_dim, n_experts, logits);
  SWN_LAUNCH_CHECK();
  return 0;
}

static int load_importance_blocks(int n_tokens) { return min(cdiv(n_tokens, 256), 512); }
extern "C" size_t swn_load_importance_workspace_floats(int n_tokens, int n_experts) {
  return (size_t)load_importance_blocks(n_tokens) * 2 * n_experts;
}

extern "C" int swn_load_importance_fwd(const float* scores_wo_noise, const float* logits_w_noise, const int32_t* idx_last, float sigma,
                                       int n_tokens, int n_experts, float* l_loss, float* coef, float* workspace, void* stream) {
  SWN_CHECK(scores_wo_noise && logits_w_noise && idx_last && l_loss && coef && workspace, "swn_load_importance_fwd: null pointer");
  SWN_CHECK(sigma > 0.f, "swn_load_importance_fwd: `gate_noise` must be > 0 for normalization in load_importance_loss()");
  SWN_CHECK(n_tokens > 0 && n_experts >= 2 && n_experts <= 16, "swn_load_importance_fwd: 2 <= experts <= 16");
  const int nblk = load_importance_blocks(n_tokens);
  hipLaunchKernelGGL(load_importance_partial_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), scores_wo_noise, logits_w_noise, idx_last,
                     1.f / sigma, n_tokens, n_experts, workspace);
  hipLaunchKernelGGL(load_importance_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), workspace, nblk, n_experts, l_loss, coef);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_load_importance_bwd(const float* scores_wo_noise, const float* logits_w_noise, const int32_t* idx_last, const float* coef,
                                       const float* d_l_loss, float sigma, int n_tokens, int n_experts, float* d_logits, void* stream) {
  SWN_CHECK(scores_wo_noise && logits_w_noise && idx_last && coef && d_l_loss && d_logits, "swn_load_importance_bwd: null pointer");
  SWN_CHECK(sigma > 0.f && n_tokens > 0 && n_experts >= 2 && n_experts <= 16, "swn_load_importance_bwd: bad sizes");
  hipLaunchKernelGGL(load_importance_bwd_kernel, dim3(cdiv(n_tokens, 256)), dim3(256), 0, as_stream(stream), scores_wo_noise, logits_w_noise,
                     idx_last, coef, d_l_loss, 1.f / sigma, n_tokens, n_experts, d_logits);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_topk_select(const float* gates, int n_tokens, int n_experts, int top_k, int32_t* idx, float* gsel, float* gnorm,
                               void* stream) {
  SWN_CHECK(gates && idx && gsel && gnorm, "swn_topk_select: null pointer");
  SWN_CHECK(n_tokens > 0 && n_experts >= 1 && n_experts <= 64 && top_k >= 1 && top_k <= n_experts, "swn_topk_select: need 1 <= k <= E <= 64");
  hipLaunchKernelGGL(topk_select_kernel, dim3(cdiv(n_tokens, 256)), dim3(256), 0, as_stream(stream), gates, n_tokens, n_experts, top_k, idx,
                     gsel, gnorm);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_topk_gate_bwd(const float* gates, const int32_t* idx, const float* d_gnorm, int n_tokens, int n_experts, int top_k,
                                 float* d_probs, void* stream) {
  SWN_CHECK(gates && idx && d_gnorm && d_probs, "swn_topk_gate_bwd: null pointer");
  SWN_CHECK(n_tokens > 0 && n_experts >= 1 && n_experts <= 64 && top_k >= 1 && top_k <= n_experts, "swn_topk_gate_bwd: need 1 <= k <= E <= 64");
  hipLaunchKernelGGL(topk_gate_bwd_kernel, dim3(cdiv(n_tokens, 256)), dim3(256), 0, as_stream(stream), gates, idx, d_gnorm, n_tokens,
                     n_experts, top_k, d_probs);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_route_topk(const int32_t* idx, const float* gmax, const float* gates, int n_tokens, int seg_tokens, int n_experts,
                              int capacity, int bpr, int top_k, int32_t* loc, int32_t* counts, int32_t* perm, int32_t* tok2row,
                              int32_t* group_rows, float* l_aux, void* workspace, size_t workspace_bytes, void* stream) {
  SWN_CHECK(top_k >= 1 && top_k <= n_experts, "swn_route_topk: need 1 <= k <= E");
  SWN_CHECK(idx && loc && counts && group_rows, "swn_route_topk: null pointer");
  SWN_CHECK(n_tokens > 0 && seg_tokens > 0 && n_tokens % seg_tokens == 0, "swn_route_topk: n_tokens must be a multiple of seg_tokens");
  const long groups = (long)(n_tokens / seg_tokens) * n_experts;
  for (int k = 0; k < top_k; ++k) {
    const int rc = route_choice(idx + (long)k * n_tokens, gmax, k == 0 ? gates : nullptr, n_tokens, seg_tokens, n_experts, capacity, bpr,
                                loc + (long)k * n_tokens, counts + k * groups, perm, tok2row ? tok2row + (long)k * n_tokens : nullptr,
                                k == 0 ? l_aux : nullptr, workspace, workspace_bytes, stream, counts, k);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(route_group_rows_kernel, dim3(cdiv(groups, 256)), dim3(256), 0, as_stream(stream), counts, (int)groups, top_k,
                     group_rows);
  SWN_LAUNCH_CHECK();
  return 0;
}

// ---- the tokens no expert kept (the fused tail of swn_mlp_chain runs them as zero rows) -------------------------------------
// drop_begin[g] = dropped tokens of the groups before g (group = (segment, expert)); one workgroup: the counts go through LDS, the
// prefix is a serial walk of one thread over at most a few thousand words
__global__ __launch_bounds__(256) void route_drop_begin_kernel(const int32_t* __restrict__ counts, int n_groups, int capacity,
                                                               int32_t* __restrict__ drop_begin) {
  __shared__ int32_t part[256];
  const int per = (n_groups + 255) / 256;
  const int g0 = threadIdx.x * per;
  int run = 0;
  for (int q = 0; q < per; ++q) {
    const int g = g0 + q;
    if (g < n_groups) run += max(counts[g] - capacity, 0);
  }
  part[threadIdx.x] = run;
  __syncthreads();
  int base = 0;
  for (int t = 0; t < (int)threadIdx.x; ++t) base += part[t];
  for (int q = 0; q < per; ++q) {
    const int g = g0 + q;
    if (g < n_groups) {
      drop_begin[g] = base;
      base += max(counts[g] - capacity, 0);
    }
  }
  if (threadIdx.x == 255) drop_begin[n_groups] = base;
}

__global__ __launch_bounds__(256) void route_drop_list_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ loc,
                                                              const int32_t* __restrict__ drop_begin, int n_tokens, int seg_tokens, int E,
                                                              int capacity, int32_t* __restrict__ dropped) {
  const long tok = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tok >= n_tokens) return;
  const int l = loc[tok];
  if (l < capacity) return;
  const int g = (int)(tok / seg_tokens) * E + idx[tok];
  dropped[drop_begin[g] + (l - capacity)] = (int32_t)tok;
}

extern "C" int swn_route_dropped(const int32_t* idx, const int32_t* loc, const int32_t* counts, int n_tokens, int seg_tokens, int n_experts,
                                 int capacity, int32_t* drop_begin, int32_t* dropped, void* stream) {
  SWN_CHECK(idx && loc && counts && drop_begin && dropped, "swn_route_dropped: null pointer");
  SWN_CHECK(n_tokens > 0 && seg_tokens > 0 && n_tokens % seg_tokens == 0, "swn_route_dropped: n_tokens must be a multiple of seg_tokens");
  const int n_groups = (n_tokens / seg_tokens) * n_experts;
  SWN_CHECK(n_groups <= (1 << 20), "swn_route_dropped: too many groups");
  hipLaunchKernelGGL(route_drop_begin_kernel, dim3(1), dim3(256), 0, as_stream(stream), counts, n_groups, capacity, drop_begin);
  hipLaunchKernelGGL(route_drop_list_kernel, dim3(cdiv(n_tokens, 256)), dim3(256), 0, as_stream(stream), idx, loc, drop_begin, n_tokens,
                     seg_tokens, n_experts, capacity, dropped);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_route_pack(const int32_t* idx, const int32_t* loc, const int32_t* counts, int n_tokens, int seg_tokens,
                              int n_experts, int32_t* begin, int32_t* perm, int32_t* tok2row, void* stream) {
  SWN_CHECK(idx && loc && counts && begin, "swn_route_pack: null pointer");
  SWN_CHECK(n_tokens > 0 && seg_tokens > 0 && n_tokens % seg_tokens == 0, "swn_route_pack: n_tokens must be a multiple of seg_tokens");
  const int n_groups = (n_tokens / seg_tokens) * n_experts;
  SWN_CHECK(n_groups <= (1 << 20), "swn_route_pack: too many groups");
  hipLaunchKernelGGL(route_pack_begin_kernel, dim3(1), dim3(256), 0, as_stream(stream), counts, n_groups, begin);
  hipLaunchKernelGGL(route_pack_scatter_kernel, dim3(cdiv(n_tokens, 256)), dim3(256), 0, as_stream(stream), idx, loc, begin, n_tokens,
                     seg_tokens, n_experts, perm, tok2row);
  SWN_LAUNCH_CHECK();
  return 0;
}

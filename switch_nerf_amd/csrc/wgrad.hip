// swn_wgrad: grouped weight-gradient GEMM  dW[g % n_wsets] += A_g^T @ B_g,  db += colsum(B_g)   (fp32 atomics).
//
// Replaces autograd's backward of torch.baddbmm w.r.t. the expert weights (ExpertMLP,
// /root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:908) and of F.linear (Mlp,
// models/nerf_moe.py:34).  The reduction runs over ROWS (tokens), which is the slow dimension of both row-major
// operands, while the bf16 MFMA wants 8 consecutive k per lane.  Operand slabs [32 rows][<=256 cols] are staged
// in LDS row-major; a lane reads an 8-byte (A) / 4-byte (B) piece of 8 rows and interleaves row pairs in registers,
// which yields 4 (A) / 2 (B) fragments whose MFMA row/col labels are a fixed permutation of the real columns.
// The bias gradient is one extra MFMA per step with an all-ones A fragment (no extra LDS traffic).
// The roofline of this kernel is HBM: each operand row is read exactly once ((m+n) elements per 2*m*n flops).
//
// Workgroup = 512 threads = 8 waves as 2 (m) x 4 (n); wave tile 128 x 64; full 256 x 256 dW tile per workgroup;
// grid = (groups, row splits).
// Staging: the slabs come in by `global_load ... lds` (no staging registers, per-lane source address = row gather for free) into a
// ring of NS = 4 slots, three slabs (96 KiB per CU) in flight: with one 128-accumulator workgroup per CU the previous register
// double buffer held one slab (32 KiB) in flight, i.e. ~4 TB/s chip-wide at ~2 us of HBM latency - the kernel ran at that rate.
// Rows past the end of the group read a zero page.
#include "common.hpp"

namespace swn {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int WG_NT = 512;
constexpr int WG_NS = 4;          // LDS ring slots (A slab + B slab each)
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[256];   // 1 KiB of zeros: source of the rows past a group's end
// 16 bytes per lane from each lane's own global address into LDS at lds_dst (wave-uniform byte address) + lane * 16.  Inline asm on
// purpose: with the builtin, hipcc treats every later ds_read of the same array as possibly aliasing the pending copy and drains the
// queue (s_waitcnt vmcnt(0)) in front of it; the waits here are counted by hand.  M0 is saved / restored (cdna_hip_programming.md 5.7).
// Cache policy of the copy: non-temporal where every operand row is read exactly once (2.555 -> 2.49-2.50 ms for the expert launch, 2.04 -> 2.01 ms
// for the dense ones, round 4, scripts/wgrad_check.py); the default policy for an operand that ANOTHER job of the launch reads as well - the
// 256-column blocks of a wider weight gradient (swn_wgrad_blocks: dW [512 x 512] = 4 jobs, every operand block in two of them; the jobs that
// share a block run on the same XCD at the same time, and a non-temporal line is the first to leave the L2 the second reader hopes to hit:
// Mission Bay recipe 33.8 -> 33.3 ms, FETCH_SIZE of the weight-gradient launches -11 %, round 6).  -DSWN_WG_NT=0: default policy everywhere.
template <bool NT>
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
#if !defined(SWN_WG_NT) || SWN_WG_NT
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
#endif
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <typename T> struct WCfg;
template <> struct WCfg<bf16_t> { static constexpr int BKR = 32; };
template <> struct WCfg<float> { static constexpr int BKR = 16; };

struct WgradArgs {
  swn_wgrad_item it[8];      // one GEMM per blockIdx.z (same shapes and grouping); a_gather / b_gather: row (in the grouped row
                             // space) -> source row of a / b, or NULL: the operand is read through the routing permutation
                             // instead of from a gathered copy (saves writing that copy)
  size_t partial_stride;     // floats between the partial-tile areas of consecutive items
  int lda, ldb, ldw;         // row strides (elements) of a, b and dw: the items may be column blocks of wider matrices
  size_t dw_set_stride, db_set_stride;   // elements between the weight sets of dw / db
  int m_dim, n_dim, n_groups, n_wsets, group_stride, clamp, rows_per_split;
  const int32_t* group_rows;
  float* partial;   // [n_groups * n_splits][m_dim * n_dim + n_dim] fp32 partial tiles (plain stores), or NULL = atomics
  int n_splits;
};

__device__ __forceinline__ bf16x8_t as_frag(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3) {
  u32x4_t v = {r0, r1, r2, r3};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <typename T, int TAG>
__global__ __launch_bounds__(WG_NT) void wgrad_kernel(const WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKR = WCfg<T>::BKR;
  constexpr int RS = 256 * (int)sizeof(T);  // LDS row stride in bytes
  constexpr int SLAB = BKR * RS;            // bytes of one operand slab
  auto sa = [&](int b_) -> char* { return smem + b_ * 2 * SLAB; };
  auto sb = [&](int b_) -> char* { return smem + b_ * 2 * SLAB + SLAB; };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;
  // (blocks go to the XCDs round-robin: rotate the group index per 8 groups so that an XCD does not own one weight set - one expert -
  //  of every segment: with unbalanced routing the XCD of the most popular expert would be the long pole of the launch)
  int g = blockIdx.x;
  if ((p.n_groups & 7) == 0) g = (((g & 7) + (g >> 3)) & 7) + (g & ~7);
  const int split = blockIdx.y;
  const swn_wgrad_item& it = p.it[blockIdx.z];
  int rows_valid = p.group_stride;
  if (p.group_rows) rows_valid = min(p.group_rows[g], p.clamp);
  const int r_begin = split * p.rows_per_split;
  const int r_end = min(rows_valid, r_begin + p.rows_per_split);
  if (r_begin >= r_end) return;
  const long grow0 = (long)g * p.group_stride;
  const int wset = g % p.n_wsets;
  const int m_dim = p.m_dim, n_dim = p.n_dim;

  f32x16_t acc[4][2];
  f32x16_t accb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[j][r] = 0.f;

  // ---- staging by LDS DMA.  A slab = BKR rows x RS bytes = 16 pieces of 1 KiB (RPP rows each); wave w copies pieces 2 w, 2 w + 1 of
  // the A slab and of the B slab: 4 copies per wave and slab.  Lane -> (row in piece, 16-byte column): columns past the operand's
  // width re-read column 0 (those tile columns are never used), rows past r_end read the zero page.
  constexpr int RPP = 1024 / RS;                       // rows per piece: 2 (bf16) / 1 (fp32)
  constexpr int LPR = RS / 16;                         // lanes per row: 32 / 64
  const int prow = lane / LPR, pcol = (lane % LPR) * 16;
  const int a_colb = pcol < m_dim * (int)sizeof(T) ? pcol : 0, b_colb = pcol < n_dim * (int)sizeof(T) ? pcol : 0;
  const char* zero = (const char*)g_zero_page + (lane & 31) * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;     // LDS byte address of the ring
  // The gather indices are fetched with SCALAR loads (wave-uniform addresses: a piece holds RPP consecutive rows): a vector load
  // here would sit behind the copies in flight in the in-order vmcnt queue, and waiting for it would drain them.
  auto dma_slab = [&](int slot, int r0) {              // rows r0 .. r0 + BKR of the group -> ring slot
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = 2 * wave + i;
      const int rf = r0 + piece * RPP;                 // first row of the piece (wave-uniform)
      long asr[RPP], bsr[RPP];
#pragma unroll
      for (int q = 0; q < RPP; ++q) {
        const long row = grow0 + min(rf + q, r_end - 1);
        // (constant address space + uniform address = s_load_dword; the routing permutation is not written while this kernel runs)
        typedef const __attribute__((address_space(4))) int32_t* cidx_t;
        asr[q] = it.a_gather ? (long)max(((cidx_t)it.a_gather)[row], 0) : row;      // valid rows (< group_rows) always carry a source row
        bsr[q] = it.b_gather ? (long)max(((cidx_t)it.b_gather)[row], 0) : row;
      }
      const bool ok = rf + prow < r_end;
      const long as = RPP == 2 ? (prow ? asr[RPP - 1] : asr[0]) : asr[0];
      const long bs = RPP == 2 ? (prow ? bsr[RPP - 1] : bsr[0]) : bsr[0];
      const char* ap = ok ? (const char*)it.a + (as * (long)p.lda) * sizeof(T) + a_colb : zero;
      const char* bp = ok ? (const char*)it.b + (bs * (long)p.ldb) * sizeof(T) + b_colb : zero;
      dma16<true>(ap, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(slot * 2 * SLAB + piece * 1024)));
      dma16<true>(bp, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(slot * 2 * SLAB + SLAB + piece * 1024)));
    }
  };

  const bool active = (wm * 128 < m_dim) && (wn * 64 < n_dim);
  const bool do_bias = (it.db != nullptr) && wm == 0 && (wn * 64 < n_dim);

  // prologue: slabs 0, 1, 2 in flight (slabs past the end copy zeros: every wave issues 4 copies per slab, the counted waits below
  // rely on it)
#pragma unroll
  for (int s0 = 0; s0 < WG_NS - 1; ++s0) dma_slab(s0, r_begin + s0 * BKR);
  int slot = 0;
  for (int r0 = r_begin; r0 < r_end; r0 += BKR) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's copies of the current slab have landed (younger: two slabs x 4)
    __builtin_amdgcn_s_barrier();                      // ... everybody's; and every wave is done reading the previous slab's slot
    dma_slab((slot + WG_NS - 1) % WG_NS, r0 + (WG_NS - 1) * BKR);
    const int buf = slot;
    const char* A = sa(buf);
    const char* B = sb(buf);
    if (active || do_bias) {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < BKR / 16; ++kk) {
          const int rb0 = kk * 16 + lhi * 8;
          uint2 pa[8];
          uint32_t pb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            pa[j] = *(const uint2*)(A + (rb0 + j) * RS + (wm * 128 + 4 * l31) * 2);
            pb[j] = *(const uint32_t*)(B + (rb0 + j) * RS + (wn * 64 + 2 * l31) * 2);
          }
          bf16x8_t fa[4], fb[2];
          {
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              lo[t] = (pa[2 * t].x & 0xFFFFu) | (pa[2 * t + 1].x << 16);
              hi[t] = (pa[2 * t].x >> 16) | (pa[2 * t + 1].x & 0xFFFF0000u);
            }
            fa[0] = as_frag(lo[0], lo[1], lo[2], lo[3]);
            fa[1] = as_frag(hi[0], hi[1], hi[2], hi[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              lo[t] = (pa[2 * t].y & 0xFFFFu) | (pa[2 * t + 1].y << 16);
              hi[t] = (pa[2 * t].y >> 16) | (pa[2 * t + 1].y & 0xFFFF0000u);
            }
            fa[2] = as_frag(lo[0], lo[1], lo[2], lo[3]);
            fa[3] = as_frag(hi[0], hi[1], hi[2], hi[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              lo[t] = (pb[2 * t] & 0xFFFFu) | (pb[2 * t + 1] << 16);
              hi[t] = (pb[2 * t] >> 16) | (pb[2 * t + 1] & 0xFFFF0000u);
            }
            fb[0] = as_frag(lo[0], lo[1], lo[2], lo[3]);
            fb[1] = as_frag(hi[0], hi[1], hi[2], hi[3]);
          }
          if (active) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
                acc[q][qq] = SWN_MFMA_32x32x16(fa[q], fb[qq], acc[q][qq]);
          }
          if (do_bias) {
            const bf16x8_t ones = as_frag(SWN_HALF_ONE_X2, SWN_HALF_ONE_X2, SWN_HALF_ONE_X2, SWN_HALF_ONE_X2);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
              accb[qq] = SWN_MFMA_32x32x16(ones, fb[qq], accb[qq]);
          }
        }
      } else {
#pragma unroll 2
        for (int kk = 0; kk < BKR / 2; ++kk) {
          const int row = kk * 2 + lhi;
          float fa[4], fb[2];
#pragma unroll
          for (int q = 0; q < 4; ++q) fa[q] = *(const float*)(A + row * RS + (wm * 128 + q * 32 + l31) * 4);
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) fb[qq] = *(const float*)(B + row * RS + (wn * 64 + qq * 32 + l31) * 4);
          if (active) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
                acc[q][qq] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[qq], acc[q][qq], 0, 0, 0);
          }
          if (do_bias) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
              accb[qq] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, fb[qq], accb[qq], 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of the slab are done before the next barrier frees its slot
    slot = (slot + 1) % WG_NS;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS copy may be in flight when the workgroup retires

  // ---- epilogue: this workgroup's partial tile goes to the workspace with plain stores (a reduce kernel sums the
  // partials; device-scope fp32 atomics from hundreds of workgroups onto one 256 KiB tile are fabric-bound), or,
  // without a workspace, straight into dW with atomics.
  float* dw = it.dw + (size_t)wset * p.dw_set_stride;
  float* part = p.partial ? p.partial + blockIdx.z * p.partial_stride + ((size_t)g * p.n_splits + split) * ((size_t)m_dim * n_dim + n_dim)
                          : nullptr;
  if (active) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * lhi;
          int m, n;
          if constexpr (sizeof(T) == 2) {
            m = wm * 128 + 4 * i + q;
            n = wn * 64 + 2 * l31 + qq;
          } else {
            m = wm * 128 + q * 32 + i;
            n = wn * 64 + qq * 32 + l31;
          }
          if (m < m_dim && n < n_dim) {
            if (part) part[(size_t)m * n_dim + n] = acc[q][qq][r];
            else unsafeAtomicAdd(dw + (size_t)m * p.ldw + n, acc[q][qq][r]);
          }
        }
  }
  if (do_bias && lhi == 0) {
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      int n;
      if constexpr (sizeof(T) == 2) n = wn * 64 + 2 * l31 + qq; else n = wn * 64 + qq * 32 + l31;
      if (n < n_dim) {  // D row i = 0 (every row equal)
        if (part) part[(size_t)m_dim * n_dim + n] = accb[qq][0];
        else unsafeAtomicAdd(it.db + (size_t)wset * p.db_set_stride + n, accb[qq][0]);
      }
    }
  }
}

struct WgradReduceArgs {
  float* dw[8];
  float* db[8];
};
// dw[wset][:] += sum over the partial tiles of (group % n_wsets == wset, split) that were actually produced.  A thread owns 4
// consecutive elements (16-byte loads); the list of partial tiles of a weight set is cut into chunks of RED_CHUNK (grid y), each
// chunk ends in one fp32 atomic per element - enough workgroups to pull the partials at HBM speed (256 row splits x 256 KiB for
// a dense layer) without a second pass.
constexpr int RED_CHUNK = 16;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial_all, size_t partial_stride,
                                                           const int32_t* __restrict__ group_rows, int clamp, int group_stride,
                                                           int rows_per_split, int n_groups, int n_wsets, int n_splits, int tile_elems,
                                                           int mn, int n_dim, int ldw, size_t dw_set_stride, size_t db_set_stride,
                                                           int n_chunks, const WgradReduceArgs ra) {
  const float* partial = partial_all + blockIdx.z * partial_stride;
  float* dw = ra.dw[blockIdx.z];
  float* db = ra.db[blockIdx.z];
  const int wset = blockIdx.y / n_chunks, chunk = blockIdx.y % n_chunks;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= tile_elems) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  // partial tile t of this weight set = (group wset + (t / n_splits) * n_wsets, split t % n_splits)
  const int t0 = chunk * RED_CHUNK;
  const int t1 = min(t0 + RED_CHUNK, ((n_groups - wset + n_wsets - 1) / n_wsets) * n_splits);
  for (int t = t0; t < t1; ++t) {
    const int g = wset + (t / n_splits) * n_wsets, sp = t % n_splits;
    const int rows = group_rows ? min(group_rows[g], clamp) : group_stride;
    if (sp * rows_per_split >= rows) continue;
    const float4 v = *(const float4*)(partial + ((size_t)g * n_splits + sp) * tile_elems + e);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* dst = nullptr;
  if (e < mn) dst = dw + (size_t)wset * dw_set_stride + (size_t)(e / n_dim) * ldw + e % n_dim;
  else if (db) dst = db + (size_t)wset * db_set_stride + (e - mn);
  if (dst) {
    unsafeAtomicAdd(dst, s.x); unsafeAtomicAdd(dst + 1, s.y); unsafeAtomicAdd(dst + 2, s.z); unsafeAtomicAdd(dst + 3, s.w);
  }
}


// =================================================================================================================================
// Balanced stream kernel (round 3): the same slab pipeline, but the WORK - not the grid - follows the kept rows.
//
// The kernel above launches one workgroup per (group, row split, layer): with the router's own routing some (segment, expert) groups
// are full and others hold a third of their capacity, the splits are sized by the capacity, the full groups are the long pole of the
// launch (2.19 .. 6.21 ms per call in a training run, slower at 80 % kept rows than at 100 %), and every workgroup leaves a 257 KiB
// partial tile behind (1792 of them: 0.9 GB written and read again per step).  Here a launch is a list of up to 8 JOBS (one GEMM
// each: its own operands, widths and strides; one common row grouping).  All rows of all jobs form one line, measured in slabs of
// BKR rows weighted by the bytes a slab moves ((m_dim + n_dim) / 32); the line is ordered job-major, then weight set, then group,
// and cut into n_wg equal parts (n_wg = the CUs of the device: one 128 KiB workgroup per CU, every one resident from start to end).
// A workgroup walks its part; whenever the (job, weight set) PAIR changes it stores the accumulators as one partial tile - so a
// launch leaves n_wg + pairs partial tiles (312 for the 7 expert layers instead of 1792) and no workgroup runs longer than
// its share of the kept rows plus one slab.  The cut depends on the row counts only: results are deterministic, the reduce kernel
// recomputes the same cut and adds a pair's tiles in workgroup order (no atomics anywhere).
constexpr int WS_MAX_JOBS = 8;
constexpr int WS_MAX_GROUPS = 2048;
constexpr int WS_MAX_PIECES = 512;            // (job, weight set) pairs of one launch: n_jobs * n_wsets
constexpr int WS_LDS_INTS = 2 * WS_MAX_GROUPS + 1 + 8 + 1 + 4 * WS_MAX_PIECES;
constexpr int WS_TILE = 256 * 256 + 256;      // floats per partial tile slot (dW tile + db row), whatever the job's widths
constexpr int WS_HDR_INTS = 128;              // header in front of the partial tiles: [0] = slabs per job, [1 + e] = first slab of weight set e

struct WsJob {
  const void* a; const void* b; const int32_t* a_gather; const int32_t* b_gather; float* dw; float* db;
  size_t dw_set_stride, db_set_stride;
  int m_dim, n_dim, lda, ldb, ldw, weight;
};
struct WsArgs {
  WsJob job[WS_MAX_JOBS];
  const int32_t* group_rows;
  const int32_t* group_begin;    // first row of every group (packed layouts), or NULL: group g starts at row g * group_stride
  float* partial;            // [WS_HDR_INTS ints][n_wg + n_jobs * n_wsets][WS_TILE]
  int n_jobs, n_groups, n_wsets, group_stride, clamp, n_wg;
};

// first slab of a job (base = the job's offset on the weighted line, w = its weight, T = its slab count) whose start is >= x
__device__ __forceinline__ int ws_first_slab(long x, long base, int w, int T) {
  const long d = x - base;
  if (d <= 0) return 0;
  const long s = (d + w - 1) / w;
  return s > T ? T : (int)s;
}
__device__ __forceinline__ long ws_cut(int k, long total, int n_wg) { return ((long)k * total) / n_wg; }

template <typename T, int TAG>
__global__ __launch_bounds__(WG_NT) void wgrad_stream_kernel(const WsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKR = WCfg<T>::BKR;
  constexpr int RS = 256 * (int)sizeof(T);
  constexpr int SLAB = BKR * RS;
  constexpr int RING = WG_NS * 2 * SLAB;
  int32_t* pre = (int32_t*)(smem + RING);                 // [n_groups + 1] slabs before ordered group i (i = wset * segs + segment)
  int32_t* rws = pre + (WS_MAX_GROUPS + 1);               // [n_groups]     valid rows of ordered group i
  int32_t* wtot = rws + WS_MAX_GROUPS;                    // [8]            scan scratch

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n_groups = p.n_groups, n_wsets = p.n_wsets, segs = n_groups / n_wsets;

  // ---- slabs per ordered group and their exclusive prefix (every workgroup computes the same table) ----
  {
    const int ipt = (n_groups + WG_NT - 1) / WG_NT;       // ordered groups per thread (<= 4)
    const int i0 = tid * ipt;
    int loc[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q;
      int sl = 0;
      if (q < ipt && i < n_groups) {
        const int g = (i % segs) * n_wsets + i / segs;
        const int rows = p.group_rows ? min(p.group_rows[g], p.clamp) : p.group_stride;
        rws[i] = rows;
        sl = (rows + BKR - 1) / BKR;
      }
      loc[q] = run;
      run += sl;
    }
    int inc = run;                                         // inclusive scan of the per-thread sums over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int base = inc - run;
    for (int w2 = 0; w2 < wave; ++w2) base += wtot[w2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q;
      if (q < ipt && i < n_groups) pre[i] = base + loc[q];
    }
    if (tid == WG_NT - 1) pre[n_groups] = base + run;
    __syncthreads();
  }
  const int Tj = pre[n_groups];                           // slabs of one job
  const int k = blockIdx.x;
  if (k == 0 && tid <= n_wsets) {                         // header for the reduce kernel
    int32_t* hdr = (int32_t*)p.partial;
    if (tid == 0) hdr[0] = Tj;
    hdr[1 + tid] = pre[tid * segs];
  }
  // ---- this workgroup's pieces: (job, weight set, first slab, end slab), listed by one thread (the cut needs 64-bit divisions: kept
  //      out of the slab loop's register budget) ----
  int32_t* plist = wtot + 8;                              // [1 + 4 * WS_MAX_PIECES]
  if (tid == 0) {
    long total = 0;
    for (int j = 0; j < p.n_jobs; ++j) total += (long)p.job[j].weight * Tj;
    int np = 0;
    if (total > 0) {
      const long x0 = ws_cut(k, total, p.n_wg), x1 = ws_cut(k + 1, total, p.n_wg);
      long wbase = 0;                                     // the job's offset on the weighted line
      for (int j = 0; j < p.n_jobs; ++j) {
        const int w = p.job[j].weight;
        const int s_lo = ws_first_slab(x0, wbase, w, Tj), s_hi = ws_first_slab(x1, wbase, w, Tj);
        wbase += (long)w * Tj;
        if (s_lo >= s_hi) continue;
        for (int e = 0; e < n_wsets; ++e) {
          const int pa = max(s_lo, pre[e * segs]), pb = min(s_hi, pre[(e + 1) * segs]);
          if (pa >= pb) continue;
          plist[1 + 4 * np] = j; plist[2 + 4 * np] = e; plist[3 + 4 * np] = pa; plist[4 + 4 * np] = pb;
          ++np;
        }
      }
    }
    plist[0] = np;
  }
  __syncthreads();
  const int n_pieces = __builtin_amdgcn_readfirstlane(plist[0]);
  float* tiles = p.partial + WS_HDR_INTS;

  // staging geometry (as in wgrad_kernel)
  constexpr int RPP = 1024 / RS;
  constexpr int LPR = RS / 16;
  const int prow = lane / LPR, pcol = (lane % LPR) * 16;
  const char* zero = (const char*)g_zero_page + (lane & 31) * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  auto sa = [&](int b_) -> char* { return smem + b_ * 2 * SLAB; };
  auto sb = [&](int b_) -> char* { return smem + b_ * 2 * SLAB + SLAB; };

  for (int pi = 0; pi < n_pieces; ++pi) {
    {
      const int j = __builtin_amdgcn_readfirstlane(plist[1 + 4 * pi]), e = __builtin_amdgcn_readfirstlane(plist[2 + 4 * pi]);
      const int pa = __builtin_amdgcn_readfirstlane(plist[3 + 4 * pi]), pb = __builtin_amdgcn_readfirstlane(plist[4 + 4 * pi]);
      const WsJob& it = p.job[j];
      const int m_dim = it.m_dim, n_dim = it.n_dim;
      const int a_colb = pcol < m_dim * (int)sizeof(T) ? pcol : 0, b_colb = pcol < n_dim * (int)sizeof(T) ? pcol : 0;
      const bool active = (wm * 128 < m_dim) && (wn * 64 < n_dim);
      const bool do_bias = (it.db != nullptr) && wm == 0 && (wn * 64 < n_dim);
      // a column block of a wider operand pair: the block of A is read again by the job of the next column block of B, and the other way round
      const bool a_shared = it.ldb > n_dim, b_shared = it.lda > m_dim;
      // ---------------------------------------------------------------- one piece: slabs [pa, pb) of (job j, weight set e)
      f32x16_t acc[4][2];
      f32x16_t accb[2];
      float dbv[2] = {0.f, 0.f};                 // 16-bit operands: this lane's share of the bias gradient (its two columns, its 8 rows per step)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[jj][r] = 0.f;

      // producer cursor: ordered group ip (binary search for the group that holds slab pa), row offset rp, slabs left
      int ip;
      {
        int lo = e * segs, hi = (e + 1) * segs - 1;        // last i with pre[i] <= pa
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (pre[mid] <= pa) lo = mid; else hi = mid - 1;
        }
        ip = lo;
      }
      int rp = (pa - pre[ip]) * BKR;
      int rows_p = rws[ip];
      int left = pb - pa;                                  // slabs not yet issued
      ip = __builtin_amdgcn_readfirstlane(ip); rp = __builtin_amdgcn_readfirstlane(rp);
      rows_p = __builtin_amdgcn_readfirstlane(rows_p); left = __builtin_amdgcn_readfirstlane(left);

      // The producer in two halves: prepare() works out the four source addresses of the next slab (scalar row arithmetic, the gather
      // indices by scalar loads, one 64-bit multiply-add per address) and advances the cursor; issue() is the four LDS copies alone.
      // The loop prepares BEFORE it waits for the current slab and the barrier, so the copies go out right behind the barrier
      // (the address arithmetic used to sit between the barrier and the copies: ~300 instructions per slab on every wave).
      int gp = (ip % segs) * n_wsets + e;                  // group of the cursor (ordered group ip + 1 = group gp + n_wsets)
      const char* a_lane = (const char*)it.a + a_colb;
      const char* b_lane = (const char*)it.b + b_colb;
      const uint32_t a_rb = (uint32_t)it.lda * (uint32_t)sizeof(T), b_rb = (uint32_t)it.ldb * (uint32_t)sizeof(T);
      typedef const __attribute__((address_space(4))) int32_t* cidx_t;
      const cidx_t ag = (cidx_t)it.a_gather, bg = (cidx_t)it.b_gather;
      const char* src[4];                                  // [2 pieces] x (A, B)
      // Gather indices of the slab the cursor points to, fetched ONE slab ahead: fetch() only issues the scalar loads (behind the
      // cursor's advance, at the end of prepare()); their values are first looked at in the next prepare(), a whole slab of MFMA work
      // later.  With the loads at the top of prepare() every wave sat through two scalar-load round trips per slab in the two gathered
      // jobs of the expert launch: 2.93 ms against 2.49 without any gather (scripts/wgrad_check.py, PERM=none), whatever the order
      // of the gathered rows.
      // (The loads are unconditional straight-line code - without a gather they read a zero word - and the choice between the loaded
      //  index and the row itself is made where the value is used: a load inside `if (gather)` is waited for at the end of its branch.)
      const cidx_t agp = ag ? ag : (cidx_t)g_zero_page, bgp = bg ? bg : (cidx_t)g_zero_page;
      const bool gathered = ag != nullptr || bg != nullptr;
      long rr0[2], rr1[2];
      int ia0[2], ia1[2], ib0[2], ib1[2];
      auto group_row0 = [&]() -> long { return p.group_begin ? (long)((cidx_t)p.group_begin)[gp] : (long)gp * p.group_stride; };
      long grow0 = group_row0();                           // first row of the cursor's group: re-read only when the cursor changes group
      auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rf = rp + (2 * wave + i) * RPP;        // first row of this wave's piece (wave-uniform)
          rr0[i] = grow0 + max(min(rf, rows_p - 1), 0);
          rr1[i] = grow0 + max(min(rf + RPP - 1, rows_p - 1), 0);
        }
        if (gathered) {                                    // (jobs without a gathered operand - five of the expert launch's seven, all dense
          ia0[0] = agp[ag ? rr0[0] : 0]; ia1[0] = agp[ag ? rr1[0] : 0];      //  ones - skip the loads; ia / ib are then never looked at)
          ia0[1] = agp[ag ? rr0[1] : 0]; ia1[1] = agp[ag ? rr1[1] : 0];
          ib0[0] = bgp[bg ? rr0[0] : 0]; ib1[0] = bgp[bg ? rr1[0] : 0];
          ib0[1] = bgp[bg ? rr0[1] : 0]; ib1[1] = bgp[bg ? rr1[1] : 0];
        }
      };
      auto prepare = [&]() {
        const bool live = left > 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rf = rp + (2 * wave + i) * RPP;
          const long a0 = ag ? (long)max(ia0[i], 0) : rr0[i], a1 = ag ? (long)max(ia1[i], 0) : rr1[i];      // (-1 = an empty slot of the permutation)
          const long b0 = bg ? (long)max(ib0[i], 0) : rr0[i], b1 = bg ? (long)max(ib1[i], 0) : rr1[i];
          const bool ok = live && (rf + prow < rows_p);
          const uint32_t as = (uint32_t)((RPP == 2 && prow) ? a1 : a0), bs = (uint32_t)((RPP == 2 && prow) ? b1 : b0);
          src[2 * i] = ok ? a_lane + (uint64_t)as * a_rb : zero;
          src[2 * i + 1] = ok ? b_lane + (uint64_t)bs * b_rb : zero;
        }
        if (live) {                                        // advance (wave-uniform)
          --left;
          rp += BKR;
          if (rp >= rows_p && left > 0) {
            do { ++ip; gp += n_wsets; rows_p = __builtin_amdgcn_readfirstlane(rws[ip]); } while (rows_p == 0);
            rp = 0;
            grow0 = group_row0();
          }
        }
        fetch();
      };
      auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int piece = 2 * wave + i;
          const uint32_t da = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(slot * 2 * SLAB + piece * 1024));
          const uint32_t db_ = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(slot * 2 * SLAB + SLAB + piece * 1024));
          if (a_shared) dma16<false>(src[2 * i], da); else dma16<true>(src[2 * i], da);
          if (b_shared) dma16<false>(src[2 * i + 1], db_); else dma16<true>(src[2 * i + 1], db_);
        }
      };

      const int nsl = pb - pa;
      fetch();
#pragma unroll
      for (int s0 = 0; s0 < WG_NS - 1; ++s0) { prepare(); issue(s0); }
      int slot = 0;
      for (int n = 0; n < nsl; ++n) {
        prepare();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue((slot + WG_NS - 1) % WG_NS);
        const char* A = sa(slot);
        const char* B = sb(slot);
#ifdef SWN_WG_ABL_STREAM      // (experiment: the staging pipeline alone - no LDS reads, no MFMAs; profiles/r04_experiments.md)
        if (false) {
#else
        if (active || do_bias) {
#endif
          if constexpr (sizeof(T) == 2) {
            // One slab = two K steps of 16 rows.  The 32 LDS reads of BOTH steps go out first (one round trip per slab), an operand
            // fragment (8 consecutive rows of one column) is put together from the row-major dwords with one v_perm_b32 per dword, and
            // the permutes of step 1 sit BETWEEN the matrix instructions of step 0: the workgroup's eight waves leave the slab's barrier
            // together, and with reads -> permutes -> MFMAs in sequence per step they all read, then all permute, then all multiply
            // (LDS, VALU and the matrix pipe busy one after the other: 4.9 TB/s against the 6.2 the staging pipeline alone sustains,
            // profiles/r04_experiments.md).  Same products in the same order per accumulator: results unchanged bit for bit.
            static_assert(BKR == 32, "two K steps per slab");
            uint2 pa_[2][8];
            uint32_t pb_[2][8];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                pa_[kk][q] = *(const uint2*)(A + (kk * 16 + lhi * 8 + q) * RS + (wm * 128 + 4 * l31) * 2);
                pb_[kk][q] = *(const uint32_t*)(B + (kk * 16 + lhi * 8 + q) * RS + (wn * 64 + 2 * l31) * 2);
              }
            // dword t of a fragment = (row 2 t, row 2 t + 1) of one column: low halves 0x05040100, high halves 0x07060302
            auto plo = [](uint32_t r0, uint32_t r1) -> uint32_t { return __builtin_amdgcn_perm(r1, r0, 0x05040100u); };
            auto phi = [](uint32_t r0, uint32_t r1) -> uint32_t { return __builtin_amdgcn_perm(r1, r0, 0x07060302u); };
            bf16x8_t fa[2][4], fb[2][2];
            auto frag_a = [&](int kk, int q) {      // column 4 l31 + q of the A slab
              const bool y = q >= 2, h = q & 1;
              uint32_t d[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const uint32_t r0 = y ? pa_[kk][2 * t].y : pa_[kk][2 * t].x, r1 = y ? pa_[kk][2 * t + 1].y : pa_[kk][2 * t + 1].x;
                d[t] = h ? phi(r0, r1) : plo(r0, r1);
              }
              fa[kk][q] = as_frag(d[0], d[1], d[2], d[3]);
            };
            auto frag_b = [&](int kk, int qq) {     // column 2 l31 + qq of the B slab
              uint32_t d[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) d[t] = qq ? phi(pb_[kk][2 * t], pb_[kk][2 * t + 1]) : plo(pb_[kk][2 * t], pb_[kk][2 * t + 1]);
              fb[kk][qq] = as_frag(d[0], d[1], d[2], d[3]);
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) frag_a(0, q);
            frag_b(0, 0);
            frag_b(0, 1);
            // db = the column sums of the B slab, from the raw row dwords (low half: column 2 l31, high half: 2 l31 + 1): 32 VALU
            // instructions per K step instead of two more matrix instructions against a fragment of ones - the launch is bound by the
            // chip's power budget (profiles/r04_experiments.md 10) and an MFMA costs what ~500 such additions do
            auto dbsum = [&](int kk) {
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                dbv[0] += bf16_to_f32((bf16_t)(pb_[kk][q] & 0xFFFFu));
                dbv[1] += bf16_to_f32((bf16_t)(pb_[kk][q] >> 16));
              }
            };
            __builtin_amdgcn_sched_barrier(0);
#if defined(SWN_WG_ABL) && SWN_WG_ABL == 2      // (experiment: staging + the LDS reads, nothing else)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int q = 0; q < 8; ++q) asm volatile("" :: "v"(pa_[kk][q]), "v"(pb_[kk][q]));
            if (false) {
#elif defined(SWN_WG_ABL) && SWN_WG_ABL == 3    // (experiment: ... + the fragment permutes, no matrix instructions)
#pragma unroll
            for (int q = 0; q < 4; ++q) { frag_a(1, q); asm volatile("" :: "v"(fa[0][q]), "v"(fa[1][q])); }
#pragma unroll
            for (int q = 0; q < 2; ++q) { frag_b(1, q); asm volatile("" :: "v"(fb[0][q]), "v"(fb[1][q])); }
            if (false) {
#else
            if (active) {      // (a wave outside the job's widths only takes part in the staging; do_bias implies active)
#endif
#pragma unroll
              for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                  acc[q][qq] = SWN_MFMA_32x32x16(fa[0][q], fb[0][qq], acc[q][qq]);
                  __builtin_amdgcn_sched_barrier(0);
                  const int j = 2 * q + qq;      // six fragments of step 1 behind the first six matrix instructions of step 0
                  if (j < 4) { frag_a(1, j); asm volatile("" : "+v"(fa[1][j])); }      // (HERE: pure code is otherwise sunk to its use)
                  else if (j < 6) { frag_b(1, j - 4); asm volatile("" : "+v"(fb[1][j - 4])); }
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
              if (do_bias) dbsum(0);
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) acc[q][qq] = SWN_MFMA_32x32x16(fa[1][q], fb[1][qq], acc[q][qq]);
              if (do_bias) dbsum(1);
            }
          } else {
#pragma unroll 2
            for (int kk = 0; kk < BKR / 2; ++kk) {
              const int row = kk * 2 + lhi;
              float fa[4], fb[2];
#pragma unroll
              for (int q = 0; q < 4; ++q) fa[q] = *(const float*)(A + row * RS + (wm * 128 + q * 32 + l31) * 4);
#pragma unroll
              for (int qq = 0; qq < 2; ++qq) fb[qq] = *(const float*)(B + row * RS + (wn * 64 + qq * 32 + l31) * 4);
              if (active) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                  for (int qq = 0; qq < 2; ++qq)
                    acc[q][qq] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[qq], acc[q][qq], 0, 0, 0);
              }
              if (do_bias) {
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                  accb[qq] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, fb[qq], accb[qq], 0, 0, 0);
              }
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot = (slot + 1) % WG_NS;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero copies issued past the end of the piece

      // ---- the piece's partial tile: slot k + pair (unique: both the cut and the pair index grow along the line) ----
      float* part = tiles + (size_t)(k + j * n_wsets + e) * WS_TILE;
      if (active) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int i = (r & 3) + 8 * (r >> 2) + 4 * lhi;
              int m, n;
              if constexpr (sizeof(T) == 2) {
                m = wm * 128 + 4 * i + q;
                n = wn * 64 + 2 * l31 + qq;
              } else {
                m = wm * 128 + q * 32 + i;
                n = wn * 64 + qq * 32 + l31;
              }
              if (m < m_dim && n < n_dim) part[(size_t)m * n_dim + n] = acc[q][qq][r];
            }
      }
      if constexpr (sizeof(T) == 2) {          // the two half-waves hold the two halves of every K step's rows
        if (do_bias) {
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) dbv[qq] += __shfl_xor(dbv[qq], 32);
        }
      }
      if (do_bias && lhi == 0) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          int n;
          if constexpr (sizeof(T) == 2) n = wn * 64 + 2 * l31 + qq; else n = wn * 64 + qq * 32 + l31;
          if (n < n_dim) part[(size_t)m_dim * n_dim + n] = sizeof(T) == 2 ? dbv[qq] : accb[qq][0];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (stores and copies share vmcnt: the next piece counts from zero)
      __syncthreads();                                       // every wave has left the ring before the next piece refills it
    }
  }
}

// dw[pair] += the pair's partial tiles in workgroup order.  grid (tile blocks, pairs); a thread owns 4 consecutive elements and adds
// the tiles in ascending workgroup order (deterministic); the loads of 8 tiles are in flight together (one at a time the loop paid a
// memory round trip per tile: 130 us for a dense layer's 256 tiles).
__global__ __launch_bounds__(256) void wgrad_stream_reduce_kernel(const WsArgs p) {
  __shared__ int32_t flag[1024 + 8];
  __shared__ int32_t krange[2];
  const int pair = blockIdx.y, j = pair / p.n_wsets, e = pair % p.n_wsets;
  const WsJob& it = p.job[j];
  const int32_t* hdr = (const int32_t*)p.partial;
  const int Tj = hdr[0], P0 = hdr[1 + e], P1 = hdr[2 + e];
  long wpre = 0, total = 0;
  for (int q = 0; q < p.n_jobs; ++q) {
    if (q == j) wpre = total;
    total += (long)p.job[q].weight * Tj;
  }
  if (threadIdx.x == 0) { krange[0] = p.n_wg; krange[1] = -1; }
  for (int k = threadIdx.x; k < 1024 + 8; k += 256) flag[k] = 0;
  __syncthreads();
  if (total > 0 && P0 < P1) {
    for (int k = threadIdx.x; k < p.n_wg; k += 256) {
      const int s_lo = ws_first_slab(ws_cut(k, total, p.n_wg), wpre, it.weight, Tj);
      const int s_hi = ws_first_slab(ws_cut(k + 1, total, p.n_wg), wpre, it.weight, Tj);
      const int ok = max(s_lo, P0) < min(s_hi, P1);
      flag[k] = ok;
      if (ok) { atomicMin(&krange[0], k); atomicMax(&krange[1], k); }
    }
  }
  __syncthreads();
  const int k0 = krange[0], k1 = krange[1];
  if (k1 < k0) return;
  const int mn = it.m_dim * it.n_dim, tile_elems = mn + (it.db ? it.n_dim : 0);
  const int el = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (el >= tile_elems) return;
  const float* tiles = p.partial + WS_HDR_INTS + (size_t)pair * WS_TILE + el;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = k0; k <= k1; k += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {            // (flag[] is zero past n_wg: the tail of the last batch reads nothing)
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (flag[k + u]) v[u] = *(const float4*)(tiles + (size_t)(k + u) * WS_TILE);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (flag[k + u]) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  float* dst = el < mn ? it.dw + (size_t)e * it.dw_set_stride + (size_t)(el / it.n_dim) * it.ldw + el % it.n_dim
                       : it.db + (size_t)e * it.db_set_stride + (el - mn);
  float4 d = *(float4*)dst;
  d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
  *(float4*)dst = d;
}

}  // namespace swn

using namespace swn;

static int ws_n_wg() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const char* ov = getenv("SWN_WGRAD_WGS");       // experiments: workgroups of the balanced kernel (default: one per CU)
    if (ov && atoi(ov) > 0) cus = atoi(ov);
    n = cus > 1024 ? 1024 : cus;
  }
  return n;
}

extern "C" size_t swn_wgrad_multi_workspace_bytes(int n_jobs, int n_wsets) {
  return (size_t)WS_HDR_INTS * 4 + ((size_t)ws_n_wg() + (size_t)n_jobs * n_wsets) * WS_TILE * sizeof(float);
}

// the groupings the stream kernel's LDS tables hold (checked on EVERY swn_wgrad_multi call: an oversized grouping would overrun them)
static bool ws_geometry_ok(int n_groups, int n_wsets) {
  return n_groups % n_wsets == 0 && n_groups <= WS_MAX_GROUPS && n_wsets <= WS_MAX_PIECES / WS_MAX_JOBS && n_wsets + 2 <= WS_HDR_INTS;
}
// ... and whether swn_wgrad / swn_wgrad_blocks route to it (SWN_WGRAD_LEGACY keeps them on the row-split kernel: experiments)
static bool ws_eligible(int n_groups, int n_wsets) {
  static const bool legacy = getenv("SWN_WGRAD_LEGACY") != nullptr;
  return !legacy && ws_geometry_ok(n_groups, n_wsets);
}

extern "C" int swn_wgrad_multi(const swn_wgrad_job* jobs, int n_jobs, int dtype, int n_groups, int n_wsets, int group_stride,
                               const int32_t* group_rows, int group_rows_clamp, const int32_t* group_begin, int tag, void* workspace,
                               size_t workspace_bytes, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_wgrad_multi: bad dtype %d", dtype);
  SWN_CHECK(jobs && n_jobs >= 1 && n_jobs <= WS_MAX_JOBS, "swn_wgrad_multi: 1..%d jobs", WS_MAX_JOBS);
  SWN_CHECK(n_groups >= 1 && n_wsets >= 1 && group_stride >= 1, "swn_wgrad_multi: bad geometry");
  SWN_CHECK(ws_geometry_ok(n_groups, n_wsets), "swn_wgrad_multi: n_groups (%d) must be a multiple of n_wsets (%d) and <= %d", n_groups,
            n_wsets, WS_MAX_GROUPS);
  SWN_CHECK(tag == 0 || tag == 1, "swn_wgrad_multi: tag must be 0 or 1");
  WsArgs p;
  memset(&p, 0, sizeof(p));
  for (int i = 0; i < n_jobs; ++i) {
    const swn_wgrad_job& s = jobs[i];
    SWN_CHECK(s.a && s.b && s.dw, "swn_wgrad_multi: null pointer in job %d", i);
    SWN_CHECK(s.m_dim >= 32 && s.m_dim <= 256 && s.m_dim % 32 == 0 && s.n_dim >= 32 && s.n_dim <= 256 && s.n_dim % 32 == 0,
              "swn_wgrad_multi: job %d: m_dim=%d n_dim=%d must be multiples of 32 in [32,256]", i, s.m_dim, s.n_dim);
    SWN_CHECK(s.lda >= s.m_dim && s.ldb >= s.n_dim && s.ldw >= s.n_dim && s.ldw % 4 == 0, "swn_wgrad_multi: job %d: bad leading dimensions", i);
    SWN_CHECK(((uintptr_t)s.dw & 15) == 0 && (!s.db || ((uintptr_t)s.db & 15) == 0) && s.dw_set_stride % 4 == 0 && s.db_set_stride % 4 == 0,
              "swn_wgrad_multi: job %d: dw / db must be 16-byte aligned", i);
    WsJob& d = p.job[i];
    d.a = s.a; d.b = s.b; d.a_gather = s.a_gather; d.b_gather = s.b_gather; d.dw = s.dw; d.db = s.db;
    d.dw_set_stride = s.dw_set_stride; d.db_set_stride = s.db_set_stride;
    d.m_dim = s.m_dim; d.n_dim = s.n_dim; d.lda = s.lda; d.ldb = s.ldb; d.ldw = s.ldw; d.weight = (s.m_dim + s.n_dim) / 32;
  }
  p.group_rows = group_rows;
  p.group_begin = group_begin;
  p.n_jobs = n_jobs; p.n_groups = n_groups; p.n_wsets = n_wsets; p.group_stride = group_stride;
  p.clamp = group_rows ? group_rows_clamp : group_stride;
  p.n_wg = ws_n_wg();
  const size_t need = swn_wgrad_multi_workspace_bytes(n_jobs, n_wsets);
  SWN_CHECK(workspace && workspace_bytes >= need, "swn_wgrad_multi: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  p.partial = (float*)workspace;
  const int bkr = dtype == SWN_HALF ? 32 : 16;
  const int lds = WG_NS * 2 * bkr * 256 * (dtype == SWN_HALF ? 2 : 4) + WS_LDS_INTS * 4;
  const void* fn;
  if (dtype == SWN_HALF) fn = tag ? (const void*)wgrad_stream_kernel<bf16_t, 1> : (const void*)wgrad_stream_kernel<bf16_t, 0>;
  else fn = tag ? (const void*)wgrad_stream_kernel<float, 1> : (const void*)wgrad_stream_kernel<float, 0>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  void* kargs[] = {(void*)&p};
  e = hipLaunchKernel(fn, dim3(p.n_wg), dim3(WG_NT), kargs, lds, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_wgrad_multi launch: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(wgrad_stream_reduce_kernel, dim3(cdiv(WS_TILE, 1024), n_jobs * n_wsets), dim3(256), 0, as_stream(stream), p);
  SWN_LAUNCH_CHECK();
  return 0;
}

static int wgrad_launch(const swn_wgrad_item* items, int n_items, int dtype, int m_dim, int n_dim, int lda, int ldb, int ldw,
                        size_t dw_set_stride, size_t db_set_stride, int n_groups, int n_wsets, int group_stride,
                        const int32_t* group_rows, int group_rows_clamp, int n_splits, int tag, void* workspace,
                        size_t workspace_bytes, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_wgrad: bad dtype %d", dtype);
  SWN_CHECK(m_dim >= 32 && m_dim <= 256 && m_dim % 32 == 0 && n_dim >= 32 && n_dim <= 256 && n_dim % 32 == 0,
            "swn_wgrad: m_dim=%d n_dim=%d must be multiples of 32 in [32,256]", m_dim, n_dim);
  SWN_CHECK(n_groups >= 1 && n_wsets >= 1 && group_stride >= 1 && n_splits >= 1, "swn_wgrad: bad geometry");
  SWN_CHECK(items && n_items >= 1 && n_items <= 8, "swn_wgrad: 1..8 items");
  if (workspace && ws_eligible(n_groups, n_wsets) && workspace_bytes >= swn_wgrad_multi_workspace_bytes(n_items, n_wsets)) {
    // the balanced stream kernel (work follows the kept rows; n_splits is irrelevant there)
    swn_wgrad_job jobs[8];
    for (int i = 0; i < n_items; ++i) {
      jobs[i].a = items[i].a; jobs[i].b = items[i].b; jobs[i].a_gather = items[i].a_gather; jobs[i].b_gather = items[i].b_gather;
      jobs[i].dw = items[i].dw; jobs[i].db = items[i].db;
      jobs[i].m_dim = m_dim; jobs[i].n_dim = n_dim; jobs[i].lda = lda; jobs[i].ldb = ldb; jobs[i].ldw = ldw;
      jobs[i].dw_set_stride = dw_set_stride; jobs[i].db_set_stride = db_set_stride;
    }
    return swn_wgrad_multi(jobs, n_items, dtype, n_groups, n_wsets, group_stride, group_rows, group_rows_clamp, nullptr, tag, workspace,
                           workspace_bytes, stream);
  }
  WgradArgs p;
  WgradReduceArgs ra;
  for (int i = 0; i < 8; ++i) {
    p.it[i] = items[i < n_items ? i : 0];
    ra.dw[i] = p.it[i].dw;
    ra.db[i] = p.it[i].db;
    SWN_CHECK(p.it[i].a && p.it[i].b && p.it[i].dw, "swn_wgrad: null pointer in item %d", i);
  }
  const int bkr = dtype == SWN_HALF ? 32 : 16;
  const int max_rows = group_rows ? (group_rows_clamp < group_stride ? group_rows_clamp : group_stride) : group_stride;
  int rps = cdiv(max_rows, n_splits);
  rps = cdiv(rps, bkr) * bkr;
  const int splits = cdiv(max_rows, rps);
  SWN_CHECK(lda >= m_dim && ldb >= n_dim && ldw >= n_dim, "swn_wgrad: leading dimensions smaller than the block");
  p.lda = lda; p.ldb = ldb; p.ldw = ldw; p.dw_set_stride = dw_set_stride; p.db_set_stride = db_set_stride;
  p.m_dim = m_dim; p.n_dim = n_dim; p.n_groups = n_groups; p.n_wsets = n_wsets;
  p.group_stride = group_stride; p.clamp = group_rows ? group_rows_clamp : group_stride; p.rows_per_split = rps;
  p.group_rows = group_rows;
  p.n_splits = splits;
  const size_t tile_elems = (size_t)m_dim * n_dim + n_dim;
  p.partial_stride = (size_t)n_groups * splits * tile_elems;
  const size_t need = (size_t)n_items * p.partial_stride * sizeof(float);
  p.partial = (workspace && workspace_bytes >= need) ? (float*)workspace : nullptr;
  if (workspace && !p.partial) return swn::set_error("swn_wgrad: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
  const int lds = WG_NS * 2 * bkr * 256 * (dtype == SWN_HALF ? 2 : 4);      // ring of WG_NS (A slab + B slab) slots: 128 KiB
  SWN_CHECK(tag == 0 || tag == 1, "swn_wgrad: tag must be 0 or 1");
  const void* fn;
  if (dtype == SWN_HALF) fn = tag ? (const void*)wgrad_kernel<bf16_t, 1> : (const void*)wgrad_kernel<bf16_t, 0>;
  else fn = tag ? (const void*)wgrad_kernel<float, 1> : (const void*)wgrad_kernel<float, 0>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  void* kargs[] = {(void*)&p};
  e = hipLaunchKernel(fn, dim3(n_groups, splits, n_items), dim3(WG_NT), kargs, lds, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_wgrad launch: %s", hipGetErrorString(e));
  if (p.partial) {
    const int tiles_per_set = cdiv(n_groups, n_wsets) * splits;
    const int n_chunks = cdiv(tiles_per_set, RED_CHUNK);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((long)tile_elems, 1024), n_wsets * n_chunks, n_items), dim3(256), 0,
                       as_stream(stream), p.partial, p.partial_stride, group_rows, p.clamp, group_stride, rps, n_groups, n_wsets, splits,
                       (int)tile_elems, m_dim * n_dim, n_dim, ldw, dw_set_stride, db_set_stride, n_chunks, ra);
  }
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_wgrad(const void* a, const void* b, const int32_t* a_gather, const int32_t* b_gather, int dtype, int m_dim,
                         int n_dim, int n_groups, int n_wsets, int group_stride, const int32_t* group_rows, int group_rows_clamp,
                         float* dw, float* db, int n_splits, int tag, void* workspace, size_t workspace_bytes, void* stream) {
  swn_wgrad_item it = {a, b, a_gather, b_gather, dw, db};
  return wgrad_launch(&it, 1, dtype, m_dim, n_dim, m_dim, n_dim, n_dim, (size_t)m_dim * n_dim, (size_t)n_dim, n_groups, n_wsets,
                      group_stride, group_rows, group_rows_clamp, n_splits, tag, workspace, workspace_bytes, stream);
}

extern "C" int swn_wgrad_batched(const swn_wgrad_item* items, int n_items, int dtype, int m_dim, int n_dim, int n_groups,
                                 int n_wsets, int group_stride, const int32_t* group_rows, int group_rows_clamp, int n_splits,
                                 int tag, void* workspace, size_t workspace_bytes, void* stream) {
  return wgrad_launch(items, n_items, dtype, m_dim, n_dim, m_dim, n_dim, n_dim, (size_t)m_dim * n_dim, (size_t)n_dim, n_groups, n_wsets,
                      group_stride, group_rows, group_rows_clamp, n_splits, tag, workspace, workspace_bytes, stream);
}

extern "C" int swn_wgrad_blocks(const swn_wgrad_item* items, int n_items, int dtype, int m_dim, int n_dim, int lda, int ldb, int ldw,
                                size_t dw_set_stride, size_t db_set_stride, int n_groups, int n_wsets, int group_stride,
                                const int32_t* group_rows, int group_rows_clamp, int n_splits, int tag, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return wgrad_launch(items, n_items, dtype, m_dim, n_dim, lda, ldb, ldw, dw_set_stride, db_set_stride, n_groups, n_wsets, group_stride,
                      group_rows, group_rows_clamp, n_splits, tag, workspace, workspace_bytes, stream);
}

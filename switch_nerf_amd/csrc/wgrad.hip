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
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
#if defined(SWN_WG_NT) && SWN_WG_NT
#define SWN_WG_LOAD_POLICY " nt"      // experiment: non-temporal operand loads
#else
#define SWN_WG_LOAD_POLICY ""
#endif
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" SWN_WG_LOAD_POLICY "\n\ts_mov_b32 m0, %0"
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
      dma16(ap, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(slot * 2 * SLAB + piece * 1024)));
      dma16(bp, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(slot * 2 * SLAB + SLAB + piece * 1024)));
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

}  // namespace swn

using namespace swn;

static int wgrad_launch(const swn_wgrad_item* items, int n_items, int dtype, int m_dim, int n_dim, int lda, int ldb, int ldw,
                        size_t dw_set_stride, size_t db_set_stride, int n_groups, int n_wsets, int group_stride,
                        const int32_t* group_rows, int group_rows_clamp, int n_splits, int tag, void* workspace,
                        size_t workspace_bytes, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_wgrad: bad dtype %d", dtype);
  SWN_CHECK(m_dim >= 32 && m_dim <= 256 && m_dim % 32 == 0 && n_dim >= 32 && n_dim <= 256 && n_dim % 32 == 0,
            "swn_wgrad: m_dim=%d n_dim=%d must be multiples of 32 in [32,256]", m_dim, n_dim);
  SWN_CHECK(n_groups >= 1 && n_wsets >= 1 && group_stride >= 1 && n_splits >= 1, "swn_wgrad: bad geometry");
  SWN_CHECK(items && n_items >= 1 && n_items <= 8, "swn_wgrad: 1..8 items");
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

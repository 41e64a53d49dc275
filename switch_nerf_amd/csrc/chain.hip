// swn_mlp_chain: fused Linear(+bias,+ReLU,+skip) chains on CDNA4 MFMA with the activation tile resident in LDS.
//
// Replaces ExpertMLP.forward (/root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:887-924,
// a baddbmm per layer with activations round-tripping HBM) and Mlp.forward (models/nerf_moe.py:30-49), and - run
// with transposed weights and the stored ReLU masks - their backward-data passes.
//
// Geometry: one workgroup = 256 threads = 4 waves; wave w owns output features [64 w, 64 w + 64) of ALL rows of the
// tile (64 rows = 2 x 2 MFMA tiles of 32x32 per wave, 64 accumulator VGPRs; SWN_BF16_BM=128 selects 128-row tiles).
//   * activations: one tile in LDS (32 KiB bf16 / 64 KiB fp32, 16-byte chunks XOR-swizzled with row&15 ->
//     conflict-free fragment reads);
//   * weights: never touch LDS.  They are pre-packed (swn_pack_weights) in MFMA-fragment-major order, so that the
//     fragment of (32 features x 16 k) is one contiguous, fully coalesced 1 KiB wave load straight into registers
//     (buffer loads: SGPR descriptor + one lane-offset VGPR + scalar step offset, no address VGPRs); a 2-step register
//     ring keeps the next fragments in flight (also across the layer boundary), served by L1/L2 (an expert's 7 layers
//     = 0.9 MB stay in the XCD's L2: workgroup b uses expert b % 8 = its XCD);
//   * the K loop therefore has NO barrier and its instruction order is pinned (sched_barrier): read fragment i+1,
//     two MFMAs on fragment i, ...; a layer costs two workgroup barriers (before / after the epilogue rewrites the LDS
//     tile).  33 KiB of LDS and <= 128 VGPRs per workgroup -> four workgroups (16 waves) per CU overlap each other's
//     epilogues, write-outs and weight-load latencies.
// The MFMA is issued "transposed" (weight fragment = A operand, activation fragment = B operand): a lane ends up
// with 4 consecutive output FEATURES of one row, so the epilogue packs them and writes the next layer's input tile
// row-major with 8-byte LDS stores.
#include "common.hpp"

#ifndef SWN_WIDE
#define SWN_WIDE 0
#endif
#ifndef SWN_CONCAT
#define SWN_CONCAT 0     // 1: third build of this file with the concat-skip layer mode (skip = 2) enabled, namespace swn_cat
#endif
#if SWN_WIDE == 2
#define SWN_NS swn_wide2
#elif SWN_WIDE
#define SWN_NS swn_wide
#elif SWN_CONCAT
#define SWN_NS swn_cat
#else
#define SWN_NS swn
#endif
#define SWN_AUX (SWN_WIDE || SWN_CONCAT)      // an auxiliary build: kernels + launcher only, no C entry points
#ifndef SWN_STATIC_EPI
#define SWN_STATIC_EPI 1                      // compile-time variants of the epilogue for the common layers (0: the dynamic one only)
#endif
// -DSWN_EXP_TIMING: s_memtime phase timers of wave 0 of the first 4096 workgroups, written through d.y_add_gather (reinterpreted
// as int64 [4096][8]; scripts/chain_timing.py, scripts/experiments/chain_wide_timing.py).  Default and 8-wave 512-feature builds; off = no code.
#if defined(SWN_EXP_TIMING) && (!SWN_AUX || SWN_WIDE == 2)
#define SWN_TIMING_ON 1
#else
#define SWN_TIMING_ON 0
#endif

namespace SWN_NS {
using namespace swn;

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// This file is compiled twice (build.sh): as is - layers up to 256 features, the tuned geometry described above - and with
// -DSWN_WIDE=1 into a second set of kernels for layers up to 512 features (the Mission Bay recipe's model width): a wave then owns
// 128 output features (4 MFMA feature tiles, 128 accumulator VGPRs), the LDS tile is 64 rows x 1 KiB (bf16; two workgroups per CU)
// or 2 KiB (fp32; one), and the K loop is left to the compiler's scheduler.
// -DSWN_WIDE=2 (round 4): a third geometry for 512-feature layers in the 16-bit types - EIGHT waves (512 threads), one workgroup per CU
// on a 128-row x 1 KiB tile, a wave owns 64 output features of all 128 rows (4 x 2 MFMA tiles, 128 accumulators, the hand-scheduled
// MI = 4 K loop): a layer's 512 KiB of weights are streamed once per 128 rows instead of once per 64 - these chains are bound by the
// CU's L2 -> L1 path like every other one (profiles/r04_experiments.md 1, 15).  Chains with the fused heads stay on -DSWN_WIDE=1.
constexpr int NT = SWN_WIDE == 2 ? 512 : 256;         // threads per workgroup (4 waves; 8 in the SWN_WIDE = 2 build)
constexpr int NI = SWN_WIDE == 1 ? 4 : 2;   // 32-wide feature tiles per wave (waves * NI * 32 = max features)
constexpr int ROW_ELEMS = (NT / 64) * 32 * NI;        // features per LDS tile row: 256 / 512
#ifndef SWN_RING_W2
#define SWN_RING_W2 4      // ... of the 8-wave 512-feature build (one workgroup per CU, fewer waves to hide the L2 round trip behind: 4 measured best - 39.5 ms at 2, 40.5 at 3, 38.8 at 4 on the Mission Bay recipe; 6 spills 42 registers)
#endif
#ifndef SWN_RING
#define SWN_RING (SWN_WIDE == 2 ? SWN_RING_W2 : 2)
#endif
constexpr int RING = SWN_RING;  // weight-fragment steps in flight
#if SWN_WIDE == 1
typedef uint64_t mbits_t;
#else
typedef uint32_t mbits_t;
#endif

template <typename T> struct Cfg;
#ifndef SWN_BF16_BM
#define SWN_BF16_BM 64
#endif
#ifndef SWN_WIDE_BM
#define SWN_WIDE_BM 64      // rows per tile of the 512-feature build (128: one workgroup per CU, 256 accumulator registers per wave - experiment)
#endif
template <> struct Cfg<bf16_t> {
  static constexpr int BM = SWN_WIDE == 2 ? 128 : (SWN_WIDE ? SWN_WIDE_BM : SWN_BF16_BM), MI = BM / 32, KSTEP = 16;   // one ring step = K 16: one 16-byte fragment load per feature tile
  static constexpr int ROWB = ROW_ELEMS * 2;           // LDS tile row stride in bytes
  static constexpr int ACT = BM * ROWB;                // LDS tile bytes
#ifndef SWN_OCC
#define SWN_OCC 4
#endif
#ifndef SWN_CAT_OCC
#define SWN_CAT_OCC 3      // workgroups per CU of the concat-skip build (dense NeRF trunk)
#endif
  static constexpr int OCC = SWN_WIDE == 2 ? 1 : SWN_WIDE ? (SWN_WIDE_BM == 128 ? 1 : 2) : (BM == 128 ? 2 : (SWN_CONCAT ? SWN_CAT_OCC : SWN_OCC));   // workgroups per CU (= waves per SIMD) the register budget must allow
  typedef bf16x8_t wfrag_t;
};
template <> struct Cfg<float> {
  static constexpr int BM = 64, MI = 2, KSTEP = 8;     // one ring step = K 8: a float4 feeds four 32x32x2 MFMAs
  static constexpr int ROWB = ROW_ELEMS * 4;
  static constexpr int ACT = BM * ROWB;
  static constexpr int OCC = SWN_WIDE ? 1 : 2;
  typedef f32x4_t wfrag_t;
};

// ---- LDS addressing: activation tile, row-major, row stride 256 elements -------------------------------------
__device__ __forceinline__ int act_off(bf16_t*, int row, int col) {  // byte offset of element (row, col)
  return row * Cfg<bf16_t>::ROWB + ((((col >> 3) ^ (row & 15))) << 4) + ((col & 7) << 1);
}
__device__ __forceinline__ int act_off(float*, int row, int col) { return row * Cfg<float>::ROWB + ((col ^ (row & 31)) << 2); }

struct ChainArgs {
  swn_chain_desc d;
  int tiles_per_group;
};

template <typename T>
__device__ __forceinline__ void store_chunk_to_act(char* act, int row, int chunk, uint4 v) {
  if constexpr (sizeof(T) == 2) {
    *(uint4*)(act + act_off((bf16_t*)nullptr, row, chunk * 8)) = v;
  } else {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) *(uint32_t*)(act + act_off((float*)nullptr, row, chunk * 4 + j)) = w[j];
  }
}
template <typename T>
__device__ __forceinline__ uint4 load_chunk_from_act(const char* act, int row, int chunk) {
  if constexpr (sizeof(T) == 2) {
    return *(const uint4*)(act + act_off((bf16_t*)nullptr, row, chunk * 8));
  } else {
    uint4 v;
    v.x = *(const uint32_t*)(act + act_off((float*)nullptr, row, chunk * 4 + 0));
    v.y = *(const uint32_t*)(act + act_off((float*)nullptr, row, chunk * 4 + 1));
    v.z = *(const uint32_t*)(act + act_off((float*)nullptr, row, chunk * 4 + 2));
    v.w = *(const uint32_t*)(act + act_off((float*)nullptr, row, chunk * 4 + 3));
    return v;
  }
}

// a 16-byte chunk of T <-> fp32 values
template <typename T>
__device__ __forceinline__ void chunk_to_f32(uint4 a, float* x) {
  const uint32_t aa[4] = {a.x, a.y, a.z, a.w};
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[2 * j] = bf16_to_f32((bf16_t)(aa[j] & 0xFFFF)); x[2 * j + 1] = bf16_to_f32((bf16_t)(aa[j] >> 16)); }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = __uint_as_float(aa[j]);
  }
}
template <typename T>
__device__ __forceinline__ uint4 f32_to_chunk(const float* x) {
  if constexpr (sizeof(T) == 2) return make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7]));
  else return make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
}

// out = a + b elementwise on a 16-byte chunk of T
template <typename T>
__device__ __forceinline__ uint4 add_chunks(uint4 a, uint4 b) {
  uint4 r;
  if constexpr (sizeof(T) == 2) {
    const uint32_t aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
    uint32_t rr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = bf16_to_f32((bf16_t)(aa[j] & 0xFFFF)) + bf16_to_f32((bf16_t)(bb[j] & 0xFFFF));
      float hi = bf16_to_f32((bf16_t)(aa[j] >> 16)) + bf16_to_f32((bf16_t)(bb[j] >> 16));
      rr[j] = pack_bf16x2(lo, hi);
    }
    r = make_uint4(rr[0], rr[1], rr[2], rr[3]);
  } else {
    r.x = __float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x));
    r.y = __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y));
    r.z = __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z));
    r.w = __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w));
  }
  return r;
}

// Load the (gathered) input rows of this tile into an LDS tile with the activation layout.
// kfeat * sizeof(T) / 16 (chunks per row) is a power of two; all global loads of a batch are issued before any is
// consumed (no per-load wait), out-of-range work is clamped to a valid address and masked at the store.
template <typename T, int B = 4>
__device__ __forceinline__ void load_rows_to_lds(char* dst, const void* src, const int32_t* gather, void* save,
                                                 long grow0, int rows_valid_in_tile, int kfeat, int tid,
                                                 const float* scale = nullptr, int relu = 0) {
  constexpr int BM = Cfg<T>::BM;
  const int row_bytes = kfeat * (int)sizeof(T);
  const int cpr = row_bytes >> 4;  // 16-byte chunks per row (power of two)
  const int sh = 31 - __builtin_clz(cpr);
  const int total = BM * cpr;      // B = loads in flight per thread (1: minimal register footprint, for the concat re-stage)
  for (int c0 = tid; c0 < total; c0 += NT * B) {
    long srow[B];
    int row[B], ch[B];
    bool in[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int c = c0 + NT * i;
      in[i] = c < total;
      const int cc = in[i] ? c : 0;
      row[i] = cc >> sh;
      ch[i] = cc & (cpr - 1);
      const bool valid = in[i] && row[i] < rows_valid_in_tile;
      srow[i] = valid ? (gather ? (long)gather[grow0 + row[i]] : grow0 + row[i]) : -1;
    }
    uint4 v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const long sr = srow[i] >= 0 ? srow[i] : 0;
      v[i] = *(const uint4*)((const char*)src + sr * row_bytes + ch[i] * 16);
    }
    asm volatile("" ::: "memory");  // keep the whole batch of loads in flight: do not sink them next to their uses
#pragma unroll
    for (int i = 0; i < B; ++i) {
      if (srow[i] < 0) v[i] = make_uint4(0, 0, 0, 0);
      else if (scale) {   // fused combine: x = relu?(scale[row] * x)   (GatingDecoder + the MoE layer's ReLU)
        const float sc = scale[grow0 + row[i]];
        if constexpr (sizeof(T) == 2) {
          uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float lo = sc * bf16_to_f32((bf16_t)(w[q] & 0xFFFF)), hi = sc * bf16_to_f32((bf16_t)(w[q] >> 16));
            if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            w[q] = pack_bf16x2(lo, hi);
          }
          v[i] = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
          float f[4] = {__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w)};
#pragma unroll
          for (int q = 0; q < 4; ++q) { f[q] *= sc; if (relu) f[q] = fmaxf(f[q], 0.f); }
          v[i] = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
        }
      }
      if (in[i]) {
        if (save && row[i] < rows_valid_in_tile) *(uint4*)((char*)save + (grow0 + row[i]) * row_bytes + ch[i] * 16) = v[i];
        store_chunk_to_act<T>(dst, row[i], ch[i], v[i]);
      }
    }
  }
}

// ---- the K loop of one layer ----------------------------------------------------------------------------------
// ring[r][ni] holds the weight fragments of step r of this layer on entry (r < RING); on exit it holds the first RING
// steps of the next layer (prefetched while the last steps of this one run).
// wcur / wnxt: this wave's fragment streams: [tile ni][step][lane][16 B]; tile stride = steps * 1 KiB.
// The write-out of a layer's INPUT tile (= the previous layer's output: a saved activation), 8-wave 512-feature build.  Behind the second
// barrier, in front of the next K loop, this copy was 16 % of a layer's time with the matrix pipe idle.  It cannot ride the K loop: the
// loop's weight stream already fills the CU's vector memory path at the matrix pipe's pace (512 KiB per 128-row tile-layer = 31 B/clk),
// and stores retire through the same in-order counter as the fragment loads - measured: the loop got slower by exactly what the copy had
// cost (profiles/r06_experiments.md 4).  It rides the EPILOGUE instead, which is VALU-bound and issues no loads: wave w copies out the
// 64-feature column block it is about to overwrite (rows of 128 contiguous bytes: 8 lanes per row), one 32-row tile ahead of its own writes.
struct WoArgs {
  char* out;          // first row of the tile in the save tensor (row-major, row_bytes per row); nullptr: nothing to write
  int row_bytes, rows;
};
template <typename T, int Q0 = 0, int NQ = 4>
__device__ __forceinline__ void writeout_cols(const char* act, const WoArgs& wo, int wn, int lane, int mi) {
  const int cpr = wo.row_bytes >> 4;                    // 16-byte chunks per row
  const int ch = wn * (NI * 4) + (lane & 7);            // (a wave owns NI * 32 features = NI * 4 chunks of 16-bit elements)
  if (ch >= cpr) return;
  uint4 v[NQ];                                          // 8 rows per slice q in [Q0, Q0 + NQ): four slices cover the row tile
  int row[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    // (adjacent 8-lane groups read rows 8 apart: their swizzled chunks fall into different halves of the 256-byte bank span - with
    //  consecutive rows the 16-byte reads of a 16-lane pass were 2-way conflicted)
    row[q] = mi * 32 + 16 * ((Q0 + q) >> 1) + 4 * ((Q0 + q) & 1) + (lane >> 4) + 8 * ((lane >> 3) & 1);
    v[q] = load_chunk_from_act<T>(act, row[q] < Cfg<T>::BM ? row[q] : 0, ch);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (row[q] < wo.rows) *(uint4*)(wo.out + (long)row[q] * wo.row_bytes + ch * 16) = v[q];
}
template <typename T, int NSTEPS>
__device__ __forceinline__ void k_loop(f32x16_t (&acc)[Cfg<T>::MI][NI], typename Cfg<T>::wfrag_t (&ring)[RING][NI],
                                       const char* act, const int (&aoff)[16], __amdgpu_buffer_rsrc_t wcur,
                                       __amdgpu_buffer_rsrc_t wnxt, int nxt_steps, int lane16) {
  constexpr int MI = Cfg<T>::MI;
  typedef typename Cfg<T>::wfrag_t wfrag_t;
  const int ts_n = nxt_steps * 1024;
  constexpr int ts_c = NSTEPS * 1024;
  // refill ring slot r with step ks + RING of this layer, else step (ks + RING - NSTEPS) of the next layer.
  // Buffer loads: descriptor (SGPRs) + one per-lane offset VGPR + scalar step offset -> no address VGPRs.
  auto refill = [&](int ks, int r) {
    const int nx = ks + RING;
    if (nx < NSTEPS) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        ring[r][ni] = __builtin_bit_cast(wfrag_t, __builtin_amdgcn_raw_buffer_load_b128(wcur, lane16, ni * ts_c + nx * 1024, 0));
    } else {   // wnxt is a valid stream even at the end of the chain (re-read, discarded)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        ring[r][ni] = __builtin_bit_cast(wfrag_t, __builtin_amdgcn_raw_buffer_load_b128(wnxt, lane16, ni * ts_n + (nx - NSTEPS) * 1024, 0));
    }
  };
  if constexpr (sizeof(T) == 2) {
    // activation fragments are software-pipelined through two alternating registers: the read of fragment i+1 is
    // issued before the two MFMAs of fragment i; a ring slot is refilled right after its last use (no copies).
    auto aread = [&](int ks, int mi) -> bf16x8_t {
      return *(const bf16x8_t*)(act + aoff[ks & 7] + mi * (32 * Cfg<T>::ROWB) + (ks >> 3) * 256);
    };
#if SWN_WIDE == 1
#pragma unroll
    for (int ks = 0; ks < NSTEPS; ++ks) {
      const int r = ks % RING;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const bf16x8_t a = aread(ks, mi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = SWN_MFMA_32x32x16(ring[r][ni], a, acc[mi][ni]);
      }
      refill(ks, r);
    }
#else
    bf16x8_t a0 = aread(0, 0), a1;
#define SWN_PIN() __builtin_amdgcn_sched_barrier(0)
    if constexpr (MI == 4) {
#pragma unroll
      for (int ks = 0; ks < NSTEPS; ++ks) {
        const int r = ks % RING;
        a1 = aread(ks, 1);
        SWN_PIN();
        acc[0][0] = SWN_MFMA_32x32x16(ring[r][0], a0, acc[0][0]);
        acc[0][1] = SWN_MFMA_32x32x16(ring[r][1], a0, acc[0][1]);
        SWN_PIN();
        a0 = aread(ks, 2);
        SWN_PIN();
        acc[1][0] = SWN_MFMA_32x32x16(ring[r][0], a1, acc[1][0]);
        acc[1][1] = SWN_MFMA_32x32x16(ring[r][1], a1, acc[1][1]);
        SWN_PIN();
        a1 = aread(ks, 3);
        SWN_PIN();
        acc[2][0] = SWN_MFMA_32x32x16(ring[r][0], a0, acc[2][0]);
        acc[2][1] = SWN_MFMA_32x32x16(ring[r][1], a0, acc[2][1]);
        SWN_PIN();
        if (ks + 1 < NSTEPS) a0 = aread(ks + 1, 0);
        SWN_PIN();
        acc[3][0] = SWN_MFMA_32x32x16(ring[r][0], a1, acc[3][0]);
        acc[3][1] = SWN_MFMA_32x32x16(ring[r][1], a1, acc[3][1]);
        SWN_PIN();
        refill(ks, r);
        SWN_PIN();
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < NSTEPS; ++ks) {
        const int r = ks % RING;
        a1 = aread(ks, 1);
        SWN_PIN();
        acc[0][0] = SWN_MFMA_32x32x16(ring[r][0], a0, acc[0][0]);
        acc[0][1] = SWN_MFMA_32x32x16(ring[r][1], a0, acc[0][1]);
        SWN_PIN();
        if (ks + 1 < NSTEPS) a0 = aread(ks + 1, 0);
        SWN_PIN();
        acc[1][0] = SWN_MFMA_32x32x16(ring[r][0], a1, acc[1][0]);
        acc[1][1] = SWN_MFMA_32x32x16(ring[r][1], a1, acc[1][1]);
        SWN_PIN();
        refill(ks, r);
        SWN_PIN();
      }
    }
#undef SWN_PIN
#endif
  } else {
#pragma unroll
    for (int ks = 0; ks < NSTEPS; ++ks) {
      const int r = ks % RING;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float af[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          af[mi] = *(const float*)(act + aoff[(ks & 3) * 4 + j] + mi * (32 * Cfg<T>::ROWB) + (ks >> 2) * 128);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[r][ni][j], af[mi], acc[mi][ni], 0, 0, 0);
        }
      }
      refill(ks, r);
    }
  }
}

#if SWN_WIDE == 2
// ---- the packed 16-bit epilogue of the 8-wave 512-feature build (the idiom of chain_big.hip's epilogue) --------------------------------
// A non-packed VALU instruction costs a wave 4 clocks and this workgroup runs its epilogue with the matrix pipe idle (two waves per SIMD,
// lockstep): ~5.5 VALU per value (add, compare, two selects, shift-or, half a convert) were half as long as the K loop.  ReLU and its mask
// work on the PACKED results - bf16 / fp16 are sign-magnitude, as int16 a negative value or -0 is < 0:
//   forward:  p = cvt_pk(z0 + b0, z1 + b1);  p = pk_max_i16(p, 0);  m |= pk_min_u16(p, 1) << d        (mask bit = rounded output > 0)
//   backward: p = pk_mul_lo_u16(cvt_pk(g0, g1), pk_min_u16(m & (0x00010001 << d), 1))
// Mask layout of THIS build (forward and backward chains of a model run on the same one): one dword per (mi, lane); packed pair
// d = ni * 8 + g4 * 2 + i has its low half at bit d, its high half at bit d + 16.  Layers with a residual input or a per-row bias keep
// the generic epilogue below (their masks use the same layout: see `PK_MASK`).
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
__device__ __forceinline__ uint32_t pk_relu16(uint32_t p) {
  const i16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, p), z));
}
__device__ __forceinline__ uint32_t pk_nonzero16(uint32_t p) {      // 0 / 1 per half: min(half, 1) unsigned
  uint32_t q;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(q) : "v"(p), "s"(0x00010001u));
  return q;
}
__device__ __forceinline__ uint32_t pk_mul16(uint32_t p, uint32_t t) {
  uint32_t q;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(q) : "v"(p), "v"(t));
  return q;
}
#endif

#if SWN_WIDE == 2
template <typename T, int RELU, bool BIAS>
__device__ __forceinline__ void epilogue_packed(f32x16_t (&acc)[Cfg<T>::MI][NI], char* act, const char* bias_lds, uint32_t* mk, int wn, int l31,
                                                int lhi, const uint32_t* mpre, const WoArgs& wo, int lane_wo) {
  constexpr int MI = Cfg<T>::MI;
  const int cbase = wn * (NI * 4);               // first 16-byte chunk of this wave's columns
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mi * 32 + l31;
    // (the input tile's rows 32 mi .. + 31 of this wave's columns leave before this row tile is rewritten - all four 8-row slices
    //  in front of the first write, two in flight at a time: registers)
    if (wo.out) { writeout_cols<T, 0, 2>(act, wo, wn, lane_wo, mi); writeout_cols<T, 2, 2>(act, wo, wn, lane_wo, mi); }
    uint32_t mb = 0;
    if constexpr (RELU == 2) mb = mpre ? mpre[mi] : mk[mi * 64];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float4 bq[4];
      if constexpr (BIAS) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) bq[g4] = *(const float4*)(bias_lds + (wn * (32 * NI) + ni * 32 + g4 * 8 + lhi * 4) * 4);
      }
      uint32_t pk[4][2];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float z0 = acc[mi][ni][g4 * 4 + 2 * i], z1 = acc[mi][ni][g4 * 4 + 2 * i + 1];
          if constexpr (BIAS) { z0 += i ? bq[g4].z : bq[g4].x; z1 += i ? bq[g4].w : bq[g4].y; }
          uint32_t p = pack_bf16x2(z0, z1);
          const int d = ni * 8 + g4 * 2 + i;
          if constexpr (RELU == 1) {
            p = pk_relu16(p);
            mb |= pk_nonzero16(p) << d;
          } else if constexpr (RELU == 2) {
            p = pk_mul16(p, pk_nonzero16(mb & (0x00010001u << d)));
          }
          pk[g4][i] = p;
        }
      }
      // The half-waves exchange 8-byte pieces (v_permlane32_swap) so that a lane writes one whole 16-byte chunk: the lower half-wave
      // ends up with chunk g of its row, the upper one with chunk g + 1 (conflict-free ds_write_b128; the 8-byte form is 2-way conflicted)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
        const uint4 o = make_uint4((uint32_t)r0[0], (uint32_t)r1[0], (uint32_t)r0[1], (uint32_t)r1[1]);
        *(uint4*)(act + m * Cfg<T>::ROWB + (((cbase + ni * 4 + g + lhi) ^ (m & 15)) << 4)) = o;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (RELU == 1) { if (mk) mk[mi * 64] = mb; }
  }
}
#endif

// ---- epilogue of one layer: accumulators -> (+bias, +per-ray bias, +skip input) -> ReLU / stored mask -> LDS tile ------
// A lane owns row m = mi*32 + l31 and, per feature tile ni and group g4, features n0 .. n0+3 (n0 = wn*64+ni*32+g4*8+lhi*4).
// Compile-time flags keep the hot variants straight-line; `mk` points at this lane's first mask word (stride 64 per mi).
template <typename T, bool DYN, int RELU_, bool SKIP_, bool BIAS_, bool RB_>
__device__ __forceinline__ void epilogue_body(f32x16_t (&acc)[Cfg<T>::MI][NI], char* act, const char* bias_lds, const float* rbp,
                                              uint32_t* mk, int wn, int l31, int lhi, int nvalid, long grow0, int rows_per_bias,
                                              int n, int relu_d, bool skip_d, bool bias_d, int rows_in_tile,
                                              const uint32_t* mpre = nullptr, const WoArgs wo = WoArgs{nullptr, 0, 0}, int lane_wo = 0) {
  constexpr int MI = Cfg<T>::MI;
  const int relu = DYN ? relu_d : RELU_;
  const bool skip = DYN ? skip_d : SKIP_;
  const bool bias = DYN ? bias_d : BIAS_;
  const bool rowb = DYN ? (rbp != nullptr) : RB_;
#if SWN_WIDE == 2
  if constexpr (sizeof(T) == 2) {
    if (!skip && !rowb && nvalid >= 32 * NI) {       // the packed epilogue (see pk_relu16): every expert / front layer of the 512-feature models
      // (compile-time variants, dispatched ONCE: with `relu` / `bias` tested inside the unrolled loops every packed pair carried two
      //  scalar compare-and-branch pairs - the epilogue segment was 2400 scalar and 300 idle instructions for 3200 vector ones)
      if (relu == 1) { if (bias) epilogue_packed<T, 1, true>(acc, act, bias_lds, mk, wn, l31, lhi, mpre, wo, lane_wo); else epilogue_packed<T, 1, false>(acc, act, bias_lds, mk, wn, l31, lhi, mpre, wo, lane_wo); }
      else if (relu == 2) { if (bias) epilogue_packed<T, 2, true>(acc, act, bias_lds, mk, wn, l31, lhi, mpre, wo, lane_wo); else epilogue_packed<T, 2, false>(acc, act, bias_lds, mk, wn, l31, lhi, mpre, wo, lane_wo); }
      else { if (bias) epilogue_packed<T, 0, true>(acc, act, bias_lds, mk, wn, l31, lhi, mpre, wo, lane_wo); else epilogue_packed<T, 0, false>(acc, act, bias_lds, mk, wn, l31, lhi, mpre, wo, lane_wo); }
      return;
    }
  }
#endif
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mi * 32 + l31;
#if SWN_WIDE == 2
    if (wo.out) writeout_cols<T>(act, wo, wn, lane_wo, mi);      // the input tile's rows of this row tile, before they are overwritten
#endif
    mbits_t mbits = 0;       // 16 bits per feature tile: one word per pair of tiles ([half][mi][lane] per wave)
    if (relu == 2) {
#if SWN_WIDE == 2
      mbits = mpre ? mpre[mi] : mk[mi * 64];
#else
      mbits = mk[mi * 64];
#endif
      if constexpr (NI == 4) mbits |= (mbits_t)((uint64_t)mk[(MI + mi) * 64] << 32);
    }
    const float* rb = nullptr;
    if (rowb) rb = rbp + (((grow0 % rows_per_bias) + min(m, rows_in_tile - 1)) / rows_per_bias) * (size_t)n;   // clamp: rows past the tile end
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if (ni * 32 < nvalid) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = wn * (32 * NI) + ni * 32 + g4 * 8 + lhi * 4;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][g4 * 4 + j];
          if (bias) {
            const float4 b4 = *(const float4*)(bias_lds + n0 * 4);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          }
          if (rowb) {
            const float4 b4 = *(const float4*)(rb + n0);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
          }
          if (skip) {
            if constexpr (sizeof(T) == 2) {
              const uint2 xv = *(const uint2*)(act + act_off((bf16_t*)nullptr, m, n0));
              v[0] += bf16_to_f32((bf16_t)(xv.x & 0xFFFF)); v[1] += bf16_to_f32((bf16_t)(xv.x >> 16));
              v[2] += bf16_to_f32((bf16_t)(xv.y & 0xFFFF)); v[3] += bf16_to_f32((bf16_t)(xv.y >> 16));
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] += *(const float*)(act + act_off((float*)nullptr, m, n0 + j));
            }
          }
#if SWN_WIDE == 2
#define SWN_MASK_BIT(ni, g4, j) ((ni) * 8 + (g4) * 2 + ((j) >> 1) + 16 * ((j) & 1))      // the packed epilogue's layout (value j of a group: pair j / 2, half j % 2)
#else
#define SWN_MASK_BIT(ni, g4, j) ((ni) * 16 + (g4) * 4 + (j))
#endif
          if (relu == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool pos = v[j] > 0.f;
              mbits |= (mbits_t)(pos ? 1u : 0u) << SWN_MASK_BIT(ni, g4, j);
              v[j] = pos ? v[j] : 0.f;
            }
          } else if (relu == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((mbits >> SWN_MASK_BIT(ni, g4, j)) & (mbits_t)1) ? v[j] : 0.f;
          }
          if constexpr (sizeof(T) == 2) {
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)(act + act_off((bf16_t*)nullptr, m, n0)) = pk;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(float*)(act + act_off((float*)nullptr, m, n0 + j)) = v[j];
          }
          __builtin_amdgcn_sched_barrier(0);   // one group at a time: keeps the epilogue's register footprint small
        }
      }
    }
    if (relu == 1 && mk) {
      mk[mi * 64] = (uint32_t)mbits;
      if constexpr (NI == 4) mk[(MI + mi) * 64] = (uint32_t)((uint64_t)mbits >> 32);
    }
  }
}
// Fused heads (chain_kernel, TAG 4): acc[s] += <tile row `row`, chunks [chunk0, chunk0 + NC)> . w[s * wstride + ...] for NSETS weight
// rows.  The weights are wave-uniform: all their scalar loads are issued together, ahead of the LDS reads (one s_load per loop
// iteration, as the first version had it, is a ~300-clock round trip each: +0.26 ms on the tail forward chain).
template <typename T, int NC, int NSETS>
__device__ __forceinline__ void heads_dot(const char* act, int row, int chunk0, const float* __restrict__ w, int wstride, float (&acc)[NSETS]) {
  constexpr int EPC = 16 / (int)sizeof(T);
  float wv[NSETS][NC * EPC];
#pragma unroll
  for (int s_ = 0; s_ < NSETS; ++s_)
#pragma unroll
    for (int i = 0; i < NC * EPC; ++i) wv[s_][i] = w[s_ * wstride + i];
  uint4 raw[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) raw[c] = load_chunk_from_act<T>(act, row, chunk0 + c);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float v[EPC];
    chunk_to_f32<T>(raw[c], v);
#pragma unroll
    for (int e = 0; e < EPC; ++e)
#pragma unroll
      for (int s_ = 0; s_ < NSETS; ++s_) acc[s_] += v[e] * wv[s_][c * EPC + e];
  }
}

template <typename T, int TAG>
__global__ __launch_bounds__(NT, Cfg<T>::OCC) void chain_kernel(const ChainArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = Cfg<T>::BM, MI = Cfg<T>::MI, KSTEP = Cfg<T>::KSTEP;
  typedef typename Cfg<T>::wfrag_t wfrag_t;
  const swn_chain_desc& d = args.d;
  char* act = smem;
  char* bias_lds = smem + Cfg<T>::ACT;  // ROW_ELEMS floats

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = feature slab
  const int l31 = lane & 31, lhi = lane >> 5;

  // workgroup -> (group, tile).  Workgroups are dealt round-robin to the 8 XCDs, and group g uses weight set g % n_wsets:
  // with a multiple of 8 weight sets XCD x works on the experts e = x (mod 8) (its L2 holds their weights) either way; the "sequential" mapping
  // additionally lets the workgroups that are resident together on an XCD walk CONSECUTIVE tiles of one group, so that their
  // save stores form a few long sequential streams per buffer instead of hundreds of scattered 32 KiB bursts.
  int g = blockIdx.x % d.n_groups;
  int tile = blockIdx.x / d.n_groups;
#ifndef SWN_NO_SEQ_TILES
  if ((d.n_wsets & 7) == 0 && (d.n_groups & 7) == 0) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int s_ = q / args.tiles_per_group;
    tile = q - s_ * args.tiles_per_group;
    g = ((x + s_) & 7) + 8 * s_;     // rotated per 8 groups: every XCD meets every weight set - an expert that draws more rows than the
                                     // others (unbalanced routing) would otherwise make ITS XCD the long pole of the launch
  } else if (d.n_groups == 1 && args.tiles_per_group >= 64) {
    // one group (the dense chains): XCD x walks its own contiguous eighth of the rows - the workgroups resident together on an XCD
    // write a few MB of consecutive rows instead of every eighth 32 KiB piece of a 32 MB window (grid = 8 * ceil(tiles / 8))
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    tile = x * ((args.tiles_per_group + 7) >> 3) + q;
  }
#endif
  int rows_valid = d.group_stride;
  if (d.group_rows) rows_valid = d.group_rows[g];
  if (rows_valid > d.group_rows_clamp) rows_valid = d.group_rows_clamp;
  const int row0 = tile * BM;
  if (row0 >= rows_valid) return;
  const int rows_in_tile = min(BM, rows_valid - row0);
  const long grow0 = (d.group_begin ? (long)d.group_begin[g] : (long)g * d.group_stride) + row0;
  const int wset = g % d.n_wsets;

  // per-lane LDS byte offsets of the activation fragments (the swizzle makes them non-affine in k: 8 (bf16) / 16
  // (fp32) distinct values; everything else is an immediate offset)
  int aoff[16];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int q = 0; q < 8; ++q) aoff[q] = l31 * Cfg<T>::ROWB + (((2 * q + lhi) ^ (l31 & 15)) << 4);
#pragma unroll
    for (int q = 8; q < 16; ++q) aoff[q] = 0;
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) aoff[q] = l31 * Cfg<T>::ROWB + ((((q >> 2) * 8 + 2 * (q & 3) + lhi) ^ l31) << 2);
  }

  // this wave's weight-fragment stream of layer L: [feature tile (2 wn + ni)][step][lane][16 B]
  auto wstream = [&](int L_) -> const char* {
    const swn_chain_layer& q = d.layers[L_];
    const size_t steps = q.k / KSTEP;
    const int nt0 = min(NI * wn, q.n / 32 - NI);    // waves beyond the layer width read valid tiles and discard
    return (const char*)q.w + ((size_t)wset * (q.n / 32) + (nt0 < 0 ? 0 : nt0)) * steps * 1024;
  };
  auto stage_bias = [&](int L_) {
    const swn_chain_layer& q = d.layers[L_];
    if (q.b && tid < (q.n >> 2)) *(float4*)(bias_lds + tid * 16) = *(const float4*)(q.b + (size_t)wset * q.n + tid * 4);
  };

  auto wrsrc = [&](int L_) -> __amdgpu_buffer_rsrc_t {   // descriptor over this wave's two feature tiles of layer L_
    return __builtin_amdgcn_make_buffer_rsrc((void*)wstream(L_), 0, NI * (d.layers[L_].k / KSTEP) * 1024, 0x00020000);
  };
  const int lane16 = lane * 16;
  wfrag_t ring[RING][NI];
  {
    const __amdgpu_buffer_rsrc_t w0 = wrsrc(0);
    const int ts = (d.layers[0].k / KSTEP) * 1024;
#pragma unroll
    for (int r = 0; r < RING; ++r) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        ring[r][ni] = __builtin_bit_cast(wfrag_t, __builtin_amdgcn_raw_buffer_load_b128(w0, lane16, ni * ts + r * 1024, 0));
    }
  }
  stage_bias(0);
  // ---- stage the chain input ----
  if constexpr (TAG == 4)      // the tail forward chain gathers its rows through tok2row: 8 chunks in flight per thread (one round trip per tile)
    load_rows_to_lds<T, 8>(act, d.x, d.x_gather, d.x_save, grow0, rows_in_tile, d.layers[0].k, tid, d.x_scale, d.x_relu);
  else
    load_rows_to_lds<T>(act, d.x, d.x_gather, d.x_save, grow0, rows_in_tile, d.layers[0].k, tid, d.x_scale, d.x_relu);
  __syncthreads();

  // ---- fused sigma / colour heads (tail forward chain only: include/swn.h, heads_raw) ----
  // The sigma head reads the chain INPUT row y (the decoded, gate-scaled, ReLU'd expert output): its dot product is taken from the
  // staged tile now and parked in LDS until the colour head's turn after the last layer.  Neither y nor h2 is read back from memory
  // (swn_heads_fwd: 768 B per point at 256 / 128 features).  Lane = row (BM = 64), wave = a quarter of the columns: the weights are
  // wave-uniform (scalar loads), no cross-lane reduction - the four quarter sums of a row meet in LDS.  (A first version with a
  // shuffle reduction per row cost the chain as much as the separate heads launch had: these 64-row chains are latency chains.)
  float* heads_part = (float*)(smem + Cfg<T>::ACT + ROW_ELEMS * 4);       // [16][64]: sigma quarters 0..3, colour c quarters 4 + 4 c ..
  if constexpr (TAG == 4 && BM == 64) {
    if (d.heads_raw) {
      constexpr int EPC = 16 / (int)sizeof(T);
      const int cq = d.layers[0].k / EPC / 4;                // chunks per quarter row
      const float* wq = d.heads_ws + wn * cq * EPC;
      int ln = lane;
      asm volatile("" : "+v"(ln));                           // (own copy of the lane index: nothing of this block stays live across the layers)
      float sg[1] = {0.f};
      constexpr int NCS = 64 / EPC;                          // chunks per call: 64 weights in scalar registers
      for (int c = 0; c < cq; c += NCS) heads_dot<T, NCS, 1>(act, ln, wn * cq + c, wq + c * EPC, 0, sg);
      heads_part[wn * 64 + ln] = sg[0];
    }
  }

  f32x16_t acc[MI][NI];
#if SWN_TIMING_ON
  long long tk = 0, tb1 = 0, tep = 0, tb2 = 0, two = 0, tstart = __builtin_amdgcn_s_memtime();
#define TICK() __builtin_amdgcn_s_memtime()
#endif

#if SWN_CONCAT
  // K loop of entry Lx: no barrier.  (Waves beyond the layer width run it too on a clamped stream - keeps the ring logic
  // uniform - and discard the result.  Only the 128-wide layer "2" has such waves.)
  auto run_k = [&](int Lx) {
    const bool nx = (Lx + 1) < d.n_layers;
    const int steps = d.layers[Lx].k / KSTEP;
    const __amdgpu_buffer_rsrc_t wcur = wrsrc(Lx);
    const __amdgpu_buffer_rsrc_t wnxt = nx ? wrsrc(Lx + 1) : wcur;
    const int nsteps_next = nx ? d.layers[Lx + 1].k / KSTEP : steps;
#if SWN_WIDE
    if (steps == 512 / KSTEP) k_loop<T, 512 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
    else
#endif
    if (steps == 256 / KSTEP) k_loop<T, 256 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
    else if (steps == 128 / KSTEP) k_loop<T, 128 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
    else k_loop<T, 64 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
  };

  for (int L0 = 0; L0 < d.n_layers; ++L0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    run_k(L0);
    __syncthreads();   // every wave has finished reading the activation tile of this layer

    int L = L0;
#if SWN_CONCAT
    if (d.layers[L0].skip == 2) {
      // concat-skip (models/nerf.py:155-156, Linear(cat([enc, h])) = h W_h + enc W_enc): entry L0 was the h half.  No
      // epilogue yet: re-stage the chain input (enc) as the tile and let the next entry (the enc half, K = the chain input
      // width) accumulate on top; bias / ReLU / mask / save belong to that entry.  Both K loops live in this iteration so
      // that the accumulators are never carried across the layer loop's back edge.
      load_rows_to_lds<T, 1>(act, d.x, d.x_gather, nullptr, grow0, rows_in_tile, d.layers[0].k, tid, d.x_scale, d.x_relu);
      stage_bias(L0 + 1);
      __syncthreads();
      L = L0 + 1;
      run_k(L);
      __syncthreads();
      L0 = L;
    }
#endif
    const swn_chain_layer& ly = d.layers[L];
    const int n = ly.n;
    const bool wave_active = (wn * 32 * NI) < n;
    const bool has_next = (L + 1) < d.n_layers;

#else
  for (int L = 0; L < d.n_layers; ++L) {
    const swn_chain_layer& ly = d.layers[L];
    const int n = ly.n, k = ly.k;
    const bool wave_active = (wn * 32 * NI) < n;
    const bool has_next = (L + 1) < d.n_layers;
    const int steps = k / KSTEP;
    const __amdgpu_buffer_rsrc_t wcur = wrsrc(L);
    const __amdgpu_buffer_rsrc_t wnxt = has_next ? wrsrc(L + 1) : wcur;
    const int nsteps_next = has_next ? d.layers[L + 1].k / KSTEP : steps;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // K loop: no barrier.  (Waves beyond the layer width run it too on a clamped stream - keeps the ring logic
    // uniform - and discard the result.  Only the 128-wide layer "2" has such waves.)
#if SWN_WIDE == 2
    uint32_t mpre[MI];        // this layer's stored ReLU masks (backward chains), requested here: the K loop hides their round trip
    const bool mpre_on = ly.relu == 2 && ly.mask != nullptr && wave_active;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      mpre[mi] = mpre_on ? ly.mask[(size_t)((g * args.tiles_per_group + tile) * (NT / 64) + wn) * MI * 64 * (NI / 2) + lane + mi * 64] : 0u;
#endif
#if SWN_TIMING_ON
    long long q0 = TICK();
#endif
#if SWN_WIDE
    if (steps == 512 / KSTEP) k_loop<T, 512 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
    else
#endif
    // (a wave beyond the width of the LAST layer has nothing to compute and nothing to prefetch for: it skips the K loop - for the
    //  128-feature last layer of the tail forward chain that is half the waves, a quarter of the launch's weight stream through the
    //  CU's L2 -> L1 path, which bounds these kernels)
    if (!wave_active && !has_next) {}
    else
    if (steps == 256 / KSTEP) k_loop<T, 256 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
    else if (steps == 128 / KSTEP) k_loop<T, 128 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
    else k_loop<T, 64 / KSTEP>(acc, ring, act, aoff, wcur, wnxt, nsteps_next, lane16);
#if SWN_TIMING_ON
    long long q1 = TICK();
#endif

    __syncthreads();   // every wave has finished reading the activation tile of this layer
#if SWN_TIMING_ON
    long long q2 = TICK();
#endif

#endif
#if SWN_WIDE == 2 && !SWN_CONCAT
    // the previous layer's saved activation = this layer's input tile, still in LDS: written out by the epilogue below (WoArgs)
    WoArgs wo{nullptr, 0, 0};
    if (TAG != 4 && L > 0 && d.layers[L - 1].save) {      // (tag 4 - the tail chain without fused heads - keeps the copy behind the barrier: registers)
      const int rb = d.layers[L - 1].n * (int)sizeof(T);
      wo = WoArgs{(char*)d.layers[L - 1].save + grow0 * rb, rb, rows_in_tile};
    }
    if (wo.out && (ly.skip || !wave_active)) {      // (a skip layer overwrites the tile with the chain input first; a wave beyond the layer's
      int ln = lane;                                //  width runs no epilogue: these copy their column block out right here)
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) writeout_cols<T>(act, wo, wn, ln, mi);
      wo.out = nullptr;
      if (ly.skip) __syncthreads();
    }
#endif
    if (ly.skip) {  // the input tile is dead: bring the chain input x back into the SAME LDS tile; each lane then reads
                    // its x values and writes h over them in place
      load_rows_to_lds<T>(act, d.x, d.x_gather, nullptr, grow0, rows_in_tile, d.layers[0].k, tid, d.x_scale, d.x_relu);
      __syncthreads();
    }

    // ---- epilogue: bias / row-bias / skip / ReLU (or stored mask) -> next layer's input tile ----
    if (wave_active) {
      // launder the lane coordinates: stops the compiler from hoisting every (mi, ni, g4) LDS offset of the epilogue
      // out of the layer loop (dozens of long-lived VGPRs -> spills in the MFMA loop)
      int l31e = l31, lhie = lhi;
      asm volatile("" : "+v"(l31e), "+v"(lhie));
      const float* rbp = ly.rowbias ? ly.rowbias + (grow0 / ly.rows_per_bias) * (size_t)n : nullptr;   // tile-aligned per-ray bias
      uint32_t* mk = ly.mask ? ly.mask + (size_t)((g * args.tiles_per_group + tile) * (NT / 64) + wn) * MI * 64 * (NI / 2) + lane : nullptr;
      const int nvalid = n - wn * 32 * NI;   // feature tiles of this wave that exist: nvalid >= 32 NI -> all
#if SWN_WIDE == 2 && !SWN_CONCAT
      epilogue_body<T, true, 0, false, false, false>(acc, act, bias_lds, rbp, mk, wn, l31e, lhie, nvalid, grow0, ly.rows_per_bias, n,
                                                     ly.relu, ly.skip == 1, ly.b != nullptr, rows_in_tile, mpre_on ? mpre : nullptr, wo,
                                                     l31e + 32 * lhie);
#else
      // the common layers (no residual input, no per-row bias) as compile-time variants, dispatched once - with the flags tested inside the
      // unrolled loops every group of four values carried scalar compare-and-branch pairs (profiles/r06_experiments.md 4); the rest: dynamic
#define SWN_EPI(R_, B_) epilogue_body<T, false, R_, false, B_, false>(acc, act, bias_lds, rbp, mk, wn, l31e, lhie, nvalid, grow0, ly.rows_per_bias, \
                                                                      n, R_, false, B_, rows_in_tile)
      // (not in the default build's 16-bit 64-row kernels: two workgroups per CU = 128 registers, the variants spill 28 of them)
      if (SWN_STATIC_EPI && (SWN_AUX || sizeof(T) == 4) && ly.skip != 1 && rbp == nullptr) {
        if (ly.relu == 1) { if (ly.b) SWN_EPI(1, true); else SWN_EPI(1, false); }
        else if (ly.relu == 2) { if (ly.b) SWN_EPI(2, true); else SWN_EPI(2, false); }
        else { if (ly.b) SWN_EPI(0, true); else SWN_EPI(0, false); }
      } else if (SWN_STATIC_EPI && (SWN_AUX || sizeof(T) == 4) && ly.skip != 1 && rbp != nullptr && !ly.b && ly.relu == 1) {
        // (layer "2" of the tail chain: per-ray bias, ReLU, no layer bias)
        epilogue_body<T, false, 1, false, false, true>(acc, act, bias_lds, rbp, mk, wn, l31e, lhie, nvalid, grow0, ly.rows_per_bias, n, 1, false,
                                                       false, rows_in_tile);
      } else
#undef SWN_EPI
      epilogue_body<T, true, 0, false, false, false>(acc, act, bias_lds, rbp, mk, wn, l31e, lhie, nvalid, grow0, ly.rows_per_bias, n,
                                                     ly.relu, ly.skip == 1, ly.b != nullptr, rows_in_tile);
#endif
    }
#if SWN_TIMING_ON
    long long q3 = TICK();
#endif
    __syncthreads();
#if SWN_TIMING_ON
    long long q4 = TICK();
    tk += q1 - q0; tb1 += q2 - q1; tep += q3 - q2; tb2 += q4 - q3;
#endif
    if (has_next) stage_bias(L + 1);   // read after the next layer's post-K-loop barrier

    // ---- write-out (row-major, coalesced) ----
    int tidw = tid;
    asm volatile("" : "+v"(tidw));     // same reason: keep the write-out index math inside the loop
    const bool last = (L == d.n_layers - 1);
    void* outp = last ? d.y : ly.save;
    if (TAG == 5 && last && d.comb_y) outp = nullptr;       // the fused combine backward writes the last layer out behind the loop
#if SWN_WIDE == 2 && !SWN_CONCAT
    if (TAG != 4 && !last) outp = nullptr;      // rides the next layer's epilogue (WoArgs)
#endif
    if (outp) {
      const int row_bytes = n * (int)sizeof(T);
      const int cpr = row_bytes >> 4;
      const int sh = 31 - __builtin_clz(cpr);
      const int total = rows_in_tile * cpr;
      if ((TAG == 6 || TAG == 2) && last && d.y_add) {
        // The backward chains' output + the rows of another tensor (the expert input gradients through tok2row / the skip gradient):
        // the row indices of UB chunks, then their 16-byte pieces, are fetched TOGETHER - one chunk at a time the loop paid two dependent
        // global-load latencies per 16 bytes stored.  (Only these two instantiations carry the code: registers.)
        constexpr int UB = 8;
        const int ch = tidw & (cpr - 1);
        for (int c0 = tidw; c0 < total; c0 += UB * NT) {
          long ar[UB];
          uint4 a[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int c = c0 + u * NT;
            const long gr = grow0 + ((c < total ? c : c0) >> sh);
            ar[u] = d.y_add_gather ? (long)d.y_add_gather[gr] : gr;
          }
#pragma unroll
          for (int u = 0; u < UB; ++u)
            a[u] = *(const uint4*)((const char*)d.y_add + (ar[u] >= 0 ? ar[u] : 0) * row_bytes + ch * 16);
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int c = c0 + u * NT;
            if (c < total) {
              const int row = c >> sh;
              uint4 v = load_chunk_from_act<T>(act, row, ch);
              if (ar[u] >= 0) v = add_chunks<T>(v, a[u]);
              *(uint4*)((char*)outp + (grow0 + row) * row_bytes + ch * 16) = v;
            }
          }
        }
      } else
      for (int c = tidw; c < total; c += NT) {
        const int row = c >> sh, ch = c & (cpr - 1);
        uint4 v = load_chunk_from_act<T>(act, row, ch);
        if (last && d.y_add) {
          long arow = grow0 + row;
          if (d.y_add_gather) arow = d.y_add_gather[grow0 + row];
          if (arow >= 0) {
            const uint4 a = *(const uint4*)((const char*)d.y_add + arow * row_bytes + ch * 16);
            v = add_chunks<T>(v, a);
          }
        }
        // (plain stores: these tensors are read back by the next kernels while still on-die; non-temporal stores measured 5 % slower
        //  per step here - unlike the expert chains' seven write-only activation streams, chain_big.hip)
        *(uint4*)((char*)outp + (grow0 + row) * row_bytes + ch * 16) = v;
      }
    }
    // no barrier needed here: the next layer only reads `act` until its own post-K-loop barrier.
#if SWN_TIMING_ON
    two += TICK() - q4;
#endif
  }
  // ---- combine backward fused into the write-out of the LAST layer (tail backward chain; behind the layer loop for the same reason
  //      as the fused heads: inside the loop body its operand loads were hoisted across the K loops) ----
  if constexpr (TAG == 5) {
    if (d.comb_y) {
      int tidw = tid;
      asm volatile("" : "+v"(tidw));
      const int n = d.layers[d.n_layers - 1].n;
      void* outp = d.y;
      const int row_bytes = n * (int)sizeof(T);
      const int cpr = row_bytes >> 4;
      const int sh = 31 - __builtin_clz(cpr);
      const int total = rows_in_tile * cpr;
    // Combine backward on the way out (include/swn.h).  A row's cpr chunks sit in cpr consecutive lanes (NT is a multiple of cpr:
    // a thread keeps its chunk column).  The global operands of UB chunks are fetched together, ahead of the arithmetic.
    constexpr int EPC = 16 / (int)sizeof(T), UB = 8;
    const int ch = tidw & (cpr - 1);
    float wv[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) wv[e] = d.comb_wsig ? d.comb_wsig[ch * EPC + e] : 0.f;
    for (int c0 = tidw; c0 < total; c0 += UB * NT) {
      uint4 yc[UB];
      float gt[UB], ds[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int c = c0 + u * NT;
        const long gr = grow0 + ((c < total ? c : c0) >> sh);
        yc[u] = *(const uint4*)((const char*)d.comb_y + gr * row_bytes + ch * 16);
        gt[u] = d.comb_gate[gr];
        ds[u] = d.comb_dsig ? d.comb_dsig[gr] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int c = c0 + u * NT;
        if (c < total) {
          const int row = c >> sh;
          float z[EPC], yv[EPC], dot = 0.f;
          chunk_to_f32<T>(load_chunk_from_act<T>(act, row, ch), z);
          chunk_to_f32<T>(yc[u], yv);
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            float t = z[e] + ds[u] * wv[e];          // (one fma, like combine_bwd_kernel)
            t = yv[e] > 0.f ? t : 0.f;
            dot += yv[e] * t;
            z[e] = t * gt[u];
          }
          for (int o = cpr >> 1; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
          if (ch == 0) d.comb_dgate[grow0 + row] = dot / gt[u];
          *(uint4*)((char*)outp + (grow0 + row) * row_bytes + ch * 16) = f32_to_chunk<T>(z);
        }
      }
    }
    }
  }
  // ---- fused heads, second half (after the layer loop: the tile holds the last layer's output h2; kept out of the loop body, where its
  //      kernel-argument loads were hoisted and cost 9 more spilled registers - 0.4 GB of scratch traffic per launch; the barrier
  //      behind the last epilogue already ordered the tile) ----
  if constexpr (TAG == 4 && BM == 64) {
    if (d.heads_raw) {      // colour head from the h2 tile, sigma from the parked quarter sums -> raw[row] = (rgb, sigma)
      constexpr int EPC = 16 / (int)sizeof(T);
      const int n = d.layers[d.n_layers - 1].n;
      const int cq = n / EPC / 4;
      const float* w0 = d.heads_wc + wn * cq * EPC;
      int lane_ = tid & 63;
      asm volatile("" : "+v"(lane_));                      // (own copy of the lane index)
      const int lane = lane_;
      float cc[3] = {0.f, 0.f, 0.f};
      constexpr int NCC = 16 / EPC;                        // 3 x 16 weights in scalar registers per call
      for (int c = 0; c < cq; c += NCC) heads_dot<T, NCC, 3>(act, lane, wn * cq + c, w0 + c * EPC, n, cc);
      heads_part[(4 + wn) * 64 + lane] = cc[0];
      heads_part[(8 + wn) * 64 + lane] = cc[1];
      heads_part[(12 + wn) * 64 + lane] = cc[2];
      __syncthreads();
      if (wn == 0 && lane < rows_in_tile) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          v[q] = (heads_part[(4 * q) * 64 + lane] + heads_part[(4 * q + 1) * 64 + lane]) +
                 (heads_part[(4 * q + 2) * 64 + lane] + heads_part[(4 * q + 3) * 64 + lane]);
        const long gr = grow0 + lane;
        const float u = v[0] + d.heads_bs[0] + (d.heads_noise ? d.heads_noise[gr] : 0.f) - 1.f;   // ShiftedSoftplus, models/nerf.py:68-69
        float4 o;
        o.x = 1.f / (1.f + expf(-(v[1] + d.heads_bc[0])));
        o.y = 1.f / (1.f + expf(-(v[2] + d.heads_bc[1])));
        o.z = 1.f / (1.f + expf(-(v[3] + d.heads_bc[2])));
        o.w = u > 20.f ? u : log1pf(expf(u));
        *(float4*)(d.heads_raw + gr * 4) = o;
      }
    }
  }
#if SWN_TIMING_ON
  if (d.y_add_gather && tid == 0 && blockIdx.x < 4096) {
    long long* dbg = (long long*)d.y_add_gather + (long)blockIdx.x * 8;
    dbg[0] = tk; dbg[1] = tb1; dbg[2] = tep; dbg[3] = tb2; dbg[4] = two; dbg[5] = TICK() - tstart; dbg[6] = tstart;
  }
#endif
}

#if !SWN_AUX
// Pack a master weight [wsets][in][out] (fp32) into the fragment-major compute layout of swn_mlp_chain:
//   transpose = 1 (forward):       W[n = out][k = in]  = master[k][n]
//   transpose = 0 (backward-data): W[n = in][k = out]  = master[n][k]
// bf16: out[wset][n/32][k/16][lane][8]: lane l, j -> W[nt*32 + (l&31)][ks*16 + (l>>5)*8 + j]
// fp32: out[wset][n/32][k/8 ][lane][4]: lane l, j -> W[nt*32 + (l&31)][kq*8 + 2*j + (l>>5)]
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ master, T* __restrict__ out, int in_dim, int out_dim,
                                    int transpose, long total_chunks) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int KSTEP = Cfg<T>::KSTEP;
  const int N = transpose ? out_dim : in_dim, K = transpose ? in_dim : out_dim;
  const long per_set = (long)(N / 32) * (K / KSTEP) * 64;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total_chunks; c += (long)gridDim.x * blockDim.x) {
    const long ws = c / per_set;
    long r = c - ws * per_set;
    const int lane = (int)(r & 63);
    r >>= 6;
    const int ks = (int)(r % (K / KSTEP)), nt = (int)(r / (K / KSTEP));
    const int n = nt * 32 + (lane & 31);
    const float* m = master + ws * (long)in_dim * out_dim;
    T* o = out + c * EPC;
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      int kk;
      if constexpr (sizeof(T) == 2) kk = ks * 16 + (lane >> 5) * 8 + j; else kk = ks * 8 + 2 * j + (lane >> 5);
      const float v = transpose ? m[(long)kk * out_dim + n] : m[(long)n * out_dim + kk];
      ElemIO<T>::st(o + j, v);
    }
  }
}

// the same for a table of weights in ONE launch (blockIdx.y = entry): the per-step refresh of every compute copy after Adam
struct PackTable {
  swn_pack_item it[SWN_MAX_PACK_ITEMS];
};
template <typename T>
__global__ void pack_weights_batched_kernel(const PackTable tab) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int KSTEP = Cfg<T>::KSTEP;
  const swn_pack_item& q = tab.it[blockIdx.y];
  const int in_dim = q.in_dim, out_dim = q.out_dim, transpose = q.transpose;
  const int in_rows = q.in_rows > 0 ? q.in_rows : in_dim;          // rows / columns of the master that exist (the rest packs as zeros)
  const int out_cols = q.out_cols > 0 ? q.out_cols : out_dim;
  const int N = transpose ? out_dim : in_dim, K = transpose ? in_dim : out_dim;
  const long per_set = (long)(N / 32) * (K / KSTEP) * 64;
  const long total_chunks = per_set * q.n_wsets;
  const float* master = q.master;
  T* out = (T*)q.out;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total_chunks; c += (long)gridDim.x * blockDim.x) {
    const long ws = c / per_set;
    long r = c - ws * per_set;
    const int lane = (int)(r & 63);
    r >>= 6;
    const int ks = (int)(r % (K / KSTEP)), nt = (int)(r / (K / KSTEP));
    const int n = nt * 32 + (lane & 31);
    const float* m = master + ws * (long)in_rows * out_cols;
    T vals[EPC];        // (one 16-byte store per lane: the eight 2-byte stores of the first version were a fabric write each)
    bool done = false;
    if constexpr (sizeof(T) == 2) {
      // backward-data layout: a lane's 8 values are 32 consecutive bytes of master row n - two 16-byte loads instead of eight scalar ones
      // (each lane on its own row: the scalar form took 0.2 ms per step for Mission Bay's 30 M parameters, a quarter of the copy rate)
      const int k0 = ks * 16 + (lane >> 5) * 8;
      if (!transpose && n < in_rows && k0 + 8 <= out_cols && (out_cols & 3) == 0 && (((uintptr_t)m) & 15) == 0) {
        const float4 a = *(const float4*)(m + (long)n * out_cols + k0), b = *(const float4*)(m + (long)n * out_cols + k0 + 4);
        ElemIO<T>::st(vals + 0, a.x); ElemIO<T>::st(vals + 1, a.y); ElemIO<T>::st(vals + 2, a.z); ElemIO<T>::st(vals + 3, a.w);
        ElemIO<T>::st(vals + 4, b.x); ElemIO<T>::st(vals + 5, b.y); ElemIO<T>::st(vals + 6, b.z); ElemIO<T>::st(vals + 7, b.w);
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        int kk;
        if constexpr (sizeof(T) == 2) kk = ks * 16 + (lane >> 5) * 8 + j; else kk = ks * 8 + 2 * j + (lane >> 5);
        const int row = transpose ? kk : n, col = transpose ? n : kk;          // master[row][col]
        const float v = (row < in_rows && col < out_cols) ? m[(long)row * out_cols + col] : 0.f;
        ElemIO<T>::st(vals + j, v);
      }
    }
    uint4 pk;
    __builtin_memcpy(&pk, vals, 16);
    *(uint4*)(out + c * EPC) = pk;
  }
}

#endif   // !SWN_AUX

// host side of one launch (shared by both builds; the narrow build's swn_mlp_chain forwards wide descriptors to the wide one)
static int chain_launch(const swn_chain_desc& d, void* stream) {
  ChainArgs a;
  a.d = d;
  const int bm = d.dtype == SWN_HALF ? Cfg<bf16_t>::BM : Cfg<float>::BM;
  a.tiles_per_group = cdiv(d.group_rows ? (d.group_rows_clamp < d.group_stride ? d.group_rows_clamp : d.group_stride) : d.group_stride, bm);
  if (!d.group_rows) a.d.group_rows_clamp = d.group_stride;
  long grid = (long)a.tiles_per_group * d.n_groups;
#ifndef SWN_NO_SEQ_TILES
  if (d.n_groups == 1 && a.tiles_per_group >= 64) grid = 8L * ((a.tiles_per_group + 7) >> 3);      // (see the tile mapping in the kernel)
#endif
  SWN_CHECK(grid > 0 && grid < (1L << 31), "swn_mlp_chain: grid %ld out of range", grid);
  int lds = (d.dtype == SWN_HALF ? Cfg<bf16_t>::ACT : Cfg<float>::ACT) + ROW_ELEMS * 4 + 4096;     // tile, bias row, the fused heads' quarter sums
#ifdef SWN_EXP_LDSPAD
  lds += SWN_EXP_LDSPAD;      // experiment: fewer resident workgroups per CU (scripts/chain_timing.py)
#endif
  const void* fn = nullptr;
#define SWN_PICK(TAGV)                                                                         \
  case TAGV:                                                                                   \
    fn = d.dtype == SWN_HALF ? (const void*)chain_kernel<bf16_t, TAGV> : (const void*)chain_kernel<float, TAGV>; \
    break;
  switch (d.tag) {
    SWN_PICK(0) SWN_PICK(1) SWN_PICK(2) SWN_PICK(3) SWN_PICK(4) SWN_PICK(5) SWN_PICK(6)
  }
#undef SWN_PICK
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(NT), kargs, lds, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_mlp_chain launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace SWN_NS

#if SWN_WIDE == 2
namespace swn {
int chain_wide2_launch(const swn_chain_desc& d, void* stream) { return swn_wide2::chain_launch(d, stream); }
int chain_wide2_tile_rows() { return swn_wide2::Cfg<bf16_t>::BM; }
}  // namespace swn
#elif SWN_WIDE
namespace swn {
int chain_wide_launch(const swn_chain_desc& d, void* stream) { return swn_wide::chain_launch(d, stream); }
int chain_wide_tile_rows(int dtype) { return dtype == SWN_HALF ? swn_wide::Cfg<bf16_t>::BM : swn_wide::Cfg<float>::BM; }
}  // namespace swn
#elif SWN_CONCAT
namespace swn {
int chain_concat_launch(const swn_chain_desc& d, void* stream) { return swn_cat::chain_launch(d, stream); }
}  // namespace swn
#else

using namespace swn;

extern "C" int swn_pack_weights(const float* master, void* out, int dtype, int n_wsets, int in_dim, int out_dim,
                                int transpose, void* stream) {
  SWN_CHECK(master && out, "swn_pack_weights: null pointer");
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_pack_weights: bad dtype");
  const int N = transpose ? out_dim : in_dim, K = transpose ? in_dim : out_dim;
  const int kstep = dtype == SWN_HALF ? 16 : 8;
  SWN_CHECK(N % 32 == 0 && K % kstep == 0 && n_wsets >= 1, "swn_pack_weights: N=%d must be a multiple of 32, K=%d of %d", N, K, kstep);
  const long chunks = (long)n_wsets * (N / 32) * (K / kstep) * 64;
  int blocks = cdiv(chunks, 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), master, (bf16_t*)out, in_dim,
                       out_dim, transpose, chunks);
  else
    hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(blocks), dim3(256), 0, as_stream(stream), master, (float*)out, in_dim,
                       out_dim, transpose, chunks);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_pack_weights_batched(const swn_pack_item* items, int n_items, int dtype, void* stream) {
  SWN_CHECK(items && n_items >= 1 && n_items <= SWN_MAX_PACK_ITEMS, "swn_pack_weights_batched: 1..%d items", SWN_MAX_PACK_ITEMS);
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_pack_weights_batched: bad dtype");
  PackTable tab;
  const int kstep = dtype == SWN_HALF ? 16 : 8;
  long max_chunks = 0;
  for (int i = 0; i < n_items; ++i) {
    const swn_pack_item& q = items[i];
    SWN_CHECK(q.master && q.out && q.n_wsets >= 1, "swn_pack_weights_batched: item %d: null pointer / no weight sets", i);
    SWN_CHECK(q.in_rows >= 0 && q.in_rows <= q.in_dim && q.out_cols >= 0 && q.out_cols <= q.out_dim,
              "swn_pack_weights_batched: item %d: in_rows %d / out_cols %d not in [0, in_dim = %d] / [0, out_dim = %d]", i, q.in_rows, q.out_cols, q.in_dim, q.out_dim);
    const int N = q.transpose ? q.out_dim : q.in_dim, K = q.transpose ? q.in_dim : q.out_dim;
    SWN_CHECK(N % 32 == 0 && K % kstep == 0, "swn_pack_weights_batched: item %d: N=%d must be a multiple of 32, K=%d of %d", i, N, K, kstep);
    const long chunks = (long)q.n_wsets * (N / 32) * (K / kstep) * 64;
    if (chunks > max_chunks) max_chunks = chunks;
    tab.it[i] = q;
  }
  int blocks = cdiv(max_chunks, 256);
  if (blocks > 512) blocks = 512;
  if (dtype == SWN_HALF) hipLaunchKernelGGL((pack_weights_batched_kernel<bf16_t>), dim3(blocks, n_items), dim3(256), 0, as_stream(stream), tab);
  else hipLaunchKernelGGL((pack_weights_batched_kernel<float>), dim3(blocks, n_items), dim3(256), 0, as_stream(stream), tab);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_chain_tile_rows(int dtype) { return dtype == SWN_HALF ? Cfg<bf16_t>::BM : Cfg<float>::BM; }

/* uint32 words of one ReLU mask buffer for a chain over n_groups x group_stride rows whose widest layer has max_width features */
extern "C" long swn_chain_mask_words(int dtype, int n_groups, int group_stride, int max_width) {
  const bool wide = max_width > 256;
  const int bm = wide ? chain_wide_tile_rows(dtype) : swn_chain_tile_rows(dtype);
  long words = (long)cdiv(group_stride, bm) * n_groups * bm * (wide ? 16 : 8);
  if (wide && dtype != SWN_F32) {              // ... or the 8-wave 128-row geometry of the 512-feature chains (same words per row)
    const int bm2 = chain_wide2_tile_rows();
    const long w2 = (long)cdiv(group_stride, bm2) * n_groups * bm2 * 16;
    if (w2 > words) words = w2;
  }
  if (!wide && dtype != SWN_F32) {             // ... or any of the chain_big.hip geometries (one buffer size fits all)
    for (int geo = 2; geo <= 3; ++geo) {
      const long w = (long)cdiv(group_stride, chain_big_tile_rows(geo)) * n_groups * chain_big_mask_words_per_tile(geo);
      if (w > words) words = w;
    }
  }
  return words;
}

extern "C" int swn_mlp_chain(const swn_chain_desc* desc, void* stream) {
  SWN_CHECK(desc != nullptr, "swn_mlp_chain: null descriptor");
  const swn_chain_desc& d = *desc;
  SWN_CHECK(d.dtype == SWN_F32 || d.dtype == SWN_HALF, "swn_mlp_chain: bad dtype %d (this build of the library computes in fp32 and %s)", d.dtype, SWN_HALF == SWN_F16 ? "fp16" : "bf16");
  SWN_CHECK(d.n_layers >= 1 && d.n_layers <= SWN_MAX_CHAIN_LAYERS, "swn_mlp_chain: n_layers %d not in [1,%d]", d.n_layers, SWN_MAX_CHAIN_LAYERS);
  SWN_CHECK(d.n_groups >= 1 && d.n_wsets >= 1 && d.group_stride >= 1, "swn_mlp_chain: bad group geometry");
  bool wide = false, concat = false;
  for (int l = 0; l < d.n_layers; ++l) {
    const swn_chain_layer& ly = d.layers[l];
    SWN_CHECK(ly.n >= 64 && ly.n <= 512 && ly.n % 64 == 0, "swn_mlp_chain: layer %d n=%d must be a multiple of 64 in [64, 512]", l, ly.n);
    SWN_CHECK(ly.k == 64 || ly.k == 128 || ly.k == 256 || ly.k == 512, "swn_mlp_chain: layer %d k=%d must be 64, 128, 256 or 512", l, ly.k);
    wide = wide || ly.n > 256 || ly.k > 256;
    concat = concat || ly.skip == 2;
    if (l > 0 && d.layers[l - 1].skip == 2)
      SWN_CHECK(ly.k == d.layers[0].k && ly.n == d.layers[l - 1].n, "swn_mlp_chain: layer %d after a concat half needs k = chain input width "
                "and the same n", l);
    else if (l > 0) SWN_CHECK(ly.k == d.layers[l - 1].n, "swn_mlp_chain: layer %d k=%d != previous n=%d", l, ly.k, d.layers[l - 1].n);
    SWN_CHECK(ly.w != nullptr, "swn_mlp_chain: layer %d has no weights", l);
    SWN_CHECK(ly.skip >= 0 && ly.skip <= 2, "swn_mlp_chain: skip mode %d", ly.skip);
    if (ly.skip == 1) SWN_CHECK(ly.n == d.layers[0].k, "swn_mlp_chain: skip layer %d needs n == chain input width", l);
    if (ly.skip == 2) SWN_CHECK(l + 1 < d.n_layers && !ly.save && !ly.mask && !ly.b && !ly.rowbias && ly.relu == 0,
                                "swn_mlp_chain: a concat half (skip = 2) carries no bias / activation / save and is followed by its other half");
    if (ly.rowbias) SWN_CHECK(ly.rows_per_bias > 0 || (d.tail_first > 0 && d.tail_bias_row), "swn_mlp_chain: rows_per_bias must be > 0 (or tail_bias_row in tail mode)");
    SWN_CHECK(ly.relu >= 0 && ly.relu <= 2, "swn_mlp_chain: relu mode %d", ly.relu);
    if (ly.relu == 2) SWN_CHECK(ly.mask != nullptr, "swn_mlp_chain: relu=2 (apply stored mask) needs a mask");
  }
  SWN_CHECK(d.x != nullptr && (d.y != nullptr || d.heads_raw != nullptr), "swn_mlp_chain: x / y must not be null");
  if (d.heads_raw) {
    const int nl = d.layers[d.n_layers - 1].n, k0 = d.layers[0].k;
    SWN_CHECK(d.heads_ws && d.heads_bs && d.heads_wc && d.heads_bc, "swn_mlp_chain: fused heads need heads_ws / heads_bs / heads_wc / heads_bc");
    SWN_CHECK((d.geometry < 2 && d.tag == 4) || (d.tail_first > 0 && d.geometry == 7 && d.tag == 7),
              "swn_mlp_chain: the fused heads run on the 64-row kernels (geometry 0 / 1, tag 4) or inside a fused tail (geometry 7, tag 7)");
    SWN_CHECK((nl == 128 || nl == 256) && (k0 == 256 || k0 == 512) && k0 * (d.dtype == SWN_F32 ? 4 : 2) <= 1024,
              "swn_mlp_chain: fused heads: chain input of 256 / 512 features in rows of at most 1 KiB, last layer of 128 / 256 (k0=%d, n=%d)", k0, nl);
    for (int l = 0; l < d.n_layers; ++l) SWN_CHECK(d.layers[l].skip != 2, "swn_mlp_chain: fused heads: no concat-skip layers");
  }
  SWN_CHECK(d.tag >= 0 && d.tag <= 8, "swn_mlp_chain: tag %d not in [0,8]", d.tag);
  SWN_CHECK(d.head_layers == 0 || (d.geometry == 7 && d.tag == 8), "swn_mlp_chain: head layers (head_layers > 0) run on geometry 7 with tag 8");
  SWN_CHECK(!d.comb_dwsig == !d.comb_dwsig_ws && (!d.comb_dwsig || (d.head_layers > 0 && d.comb_y && d.comb_dsig)),
            "swn_mlp_chain: comb_dwsig and comb_dwsig_ws come together, with the combine backward of a fused backward chain (head_layers > 0, comb_y, comb_dsig)");
  SWN_CHECK(d.tag != 8 || d.head_layers > 0, "swn_mlp_chain: tag 8 is the expert backward chain behind the tail's backward layers (head_layers > 0)");
  SWN_CHECK(d.tail_first == 0 || (d.geometry == 7 && d.tag == 7), "swn_mlp_chain: a fused tail (tail_first > 0) runs on geometry 7 with tag 7");
  SWN_CHECK(d.tag != 7 || d.tail_first > 0, "swn_mlp_chain: tag 7 is the fused-tail forward chain (tail_first > 0)");
  SWN_CHECK(d.x_features == 0 || d.x_features == d.layers[0].k || (d.x_features == 128 && d.layers[0].k == 256 && d.geometry >= 6),
            "swn_mlp_chain: x_features = %d: only 128-feature rows under a zero-padded k = 256 first layer on geometry 6 / 7", d.x_features);
  SWN_CHECK(!(wide && concat), "swn_mlp_chain: concat-skip layers are built for the 256-feature kernels only");
  SWN_CHECK(d.geometry >= 0 && d.geometry <= 7, "swn_mlp_chain: geometry %d not in [0,7]", d.geometry);
  if (d.comb_y && d.head_layers == 0) {      // (head_layers > 0: the combine backward sits behind the head layers - chain_persistent_eligible)
    const int nl = d.layers[d.n_layers - 1].n;
    SWN_CHECK(d.comb_gate && d.comb_dgate, "swn_mlp_chain: combine backward needs comb_gate and comb_dgate");
    SWN_CHECK((nl == 128 || nl == 256 || nl == 512) && nl * (d.dtype == SWN_F32 ? 4 : 2) <= 1024,
              "swn_mlp_chain: combine backward: last layer of 128 / 256 / 512 features, at most 1 KiB per row");
    SWN_CHECK((d.geometry < 2 || d.geometry >= 6) && (d.tag & 0xFF) == 5,
              "swn_mlp_chain: combine backward runs on the 64-row kernels (geometry 0 / 1) or the persistent ones (6 / 7), tag 5");
    SWN_CHECK(d.geometry < 2 || nl == 256, "swn_mlp_chain: combine backward on geometry 6 / 7: last layer of 256 features");
  }
  {   // 256-row geometry (chain_big.hip).  Never chosen silently for bf16: the ReLU mask layout differs between the geometries,
      // and a backward chain must run on the geometry of the forward chain that recorded its masks - the caller pairs them.
    const bool can = d.geometry >= 6 ? chain_persistent_eligible(d) : chain_big_eligible(d);
    if (d.tail_first > 0 || d.head_layers > 0)
      SWN_CHECK(can, "swn_mlp_chain: a fused tail (tail_first / head_layers) needs bf16 / fp16 layers of 256 x 256 (the last tail layer zero-padded "
                "from y_features = 128; x_features = 128 under the head layers), x_gather, the gate values / combine operands, the dropped-token "
                "list, tail_tokens * 512 bytes below 4 GiB, no ReLU on the gate layer and ReLU on the last tail layer (include/swn.h)");
    SWN_CHECK(d.geometry < 2 || can, "swn_mlp_chain: geometries 2 - 7 need bf16 / fp16 chains of 256 x 256 layers without rowbias / x_scale / x_save / y_add_gather");
    if (d.geometry >= 2) return chain_big_launch(d, stream);
  }
  if (wide) {      // 512-feature geometries: this file compiled with -DSWN_WIDE=2 (16-bit types, no fused heads: 128-row tiles, 8 waves) / =1
    static const bool no_w2 = getenv("SWN_NO_WIDE2") != nullptr;
    if (d.dtype == SWN_HALF && !d.heads_raw && !no_w2) return chain_wide2_launch(d, stream);
    return chain_wide_launch(d, stream);
  }
  if (concat) return chain_concat_launch(d, stream);    // concat-skip layers (this file compiled with -DSWN_CONCAT=1)
  return chain_launch(d, stream);
}
#endif   // !SWN_AUX

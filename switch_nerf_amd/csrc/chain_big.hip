// swn_mlp_chain, 256-row geometry ("big tile"): chains of 256 x 256 layers in bf16 / fp16 with one workgroup per CU.
//
// Same contract as chain.hip (ExpertMLP.forward, /root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:887-924,
// and its backward-data pass); selected by swn_mlp_chain for the expert chains of the 256-feature recipes.  What is different and why
// (profiles/r01_chain_store_experiments.md: the 64-row chain re-reads an expert's 0.9 MB of weights from L2 for every 64 rows - 29 GB per
// 2M-row pass - and that stream plus the activation saves saturate the CU's vector-memory path):
//   * one workgroup = 512 threads = 8 waves owns a 256-row tile (128 KiB of LDS, 16-byte chunks XOR-swizzled with row & 15); wave
//     (rg, fg) computes rows [128 rg, +128) x features [64 fg, +64): 4 x 2 MFMA tiles of 32x32, 128 accumulator registers;
//   * the weights of a K step (16 k x 256 features = 8 KiB, fragment-major as packed by swn_pack_weights) are brought in ONCE per
//     workgroup by `buffer_load ... lds` (each wave copies one 1 KiB fragment) into a 3-slot LDS ring, two K steps ahead of their
//     use, and read by all eight waves: a quarter of the L2 -> CU weight traffic per row of the 64-row geometry;
//   * one workgroup barrier per K step: [fragments of step k in registers, copy of step k+1 landed (counted vmcnt)] -> barrier ->
//     issue the fragment reads of step k+1, the copy of step k+3 into the slot of step k, one 1 KiB piece of the write-out -> 8 MFMAs;
//   * the saved activation of layer l (the weight-gradient GEMM's operand) is written out DURING the K loop of layer l+1, whose
//     input tile it is: 16 steps x 8 waves x 1 KiB, row-major and fully coalesced, through a buffer descriptor clipped to the
//     valid rows - the store stream is spread evenly over the MFMA work instead of following it;
//   * the chain input rows are gathered straight into the swizzled tile with `global_load ... lds` (per-lane source address).
// The MFMA is issued transposed like chain.hip (weights = A operand): a lane owns 4 consecutive features of one row.
#include "common.hpp"

namespace swn_big {
using namespace swn;

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
#define SWN_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define SWN_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int NT = 512;                 // 8 waves
constexpr int BM = 256;                 // rows per tile
constexpr int ROWB = 512;               // tile row stride in bytes (256 two-byte features)
constexpr int TILE_B = BM * ROWB;       // 128 KiB
constexpr int SLOT_B = 8192;            // weights of one K step: 8 feature tiles x 1 KiB
constexpr int NSLOT = 3;
constexpr int RING0 = TILE_B;
constexpr int IDX0 = RING0 + NSLOT * SLOT_B;   // int32 [256]: source row of every tile row
constexpr int BIAS0 = IDX0 + 1024;             // f32 [256]
constexpr int LDS_BYTES = BIAS0 + 1024;        // 157,696 B of the CU's 163,840
constexpr int KSTEPS = 16;                     // 256 / 16

struct Args {
  swn_chain_desc d;
  int tiles_per_group;
};

// element type: packs / unpacks two features per dword and picks the MFMA
struct Bf16 {
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
  static __device__ __forceinline__ float lo(uint32_t v) { return __uint_as_float(v << 16); }
  static __device__ __forceinline__ float hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
struct Fp16 {
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};     // round to nearest even (v_cvt_f16_f32)
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ float lo(uint32_t v) { return (float)__builtin_bit_cast(f16x2_t, v)[0]; }
  static __device__ __forceinline__ float hi(uint32_t v) { return (float)__builtin_bit_cast(f16x2_t, v)[1]; }
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// buffer descriptor over [p, p + bytes) with every word provably wave-uniform (a descriptor the compiler believes to be divergent costs a
// waterfall loop around each access - cdna_hip_programming.md T20)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, int bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

#define SWN_PIN() __builtin_amdgcn_sched_barrier(0)
#define SWN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SWN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// per-wave / per-lane constants of a workgroup
struct Ctx {
  char* smem;
  int w, lane, l31, lhi;
  uint32_t a_base;       // LDS byte address of this lane's activation-fragment row (mi = 0) incl. the swizzle seed; ^ (ks << 5) per step
  uint32_t e_base;       // ... of this lane's epilogue row incl. swizzle seed, half-wave and wave feature offset; ^ ((4 ni + g4) << 4)
  uint32_t wf_base;      // RING0 + this wave's first feature tile + lane * 16 (add the slot offset)
  uint32_t wo_base;      // write-out: LDS byte address of this lane's 16 bytes of chunk (ks = 0); + ks * 8192
  int slot_off[3];       // byte offset of the ring slot of K step (16 L + j), j mod 3 - refreshed per layer
};

// ---- the K loop of one layer ------------------------------------------------------------------------------------------------
// SAVE: the input tile of this layer is written out (1 KiB per wave and K step) through rs_save.
template <typename E, bool SAVE>
__device__ __forceinline__ void k_loop(f32x16_t (&acc)[4][2], const Ctx& cx, __amdgpu_buffer_rsrc_t rs_cur, __amdgpu_buffer_rsrc_t rs_nxt,
                                       __amdgpu_buffer_rsrc_t rs_save) {
  char* smem = cx.smem;
  const int lane16 = cx.lane * 16;
  u32x4_t fa[2][4], fw[2][2];
  u32x4_t wo = {0u, 0u, 0u, 0u};
  auto read_frags = [&](int ks, int set) {
    const uint32_t ab = cx.a_base ^ (uint32_t)(ks << 5);
    const uint32_t wb = cx.wf_base + cx.slot_off[ks % 3];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) fw[set][ni] = *(const u32x4_t*)(smem + wb + ni * 1024);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) fa[set][mi] = *(const u32x4_t*)(smem + ab + mi * (32 * ROWB));
  };
  read_frags(0, 0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int cur = ks & 1;
    SWN_WAIT_LGKM0();                       // the fragments of step ks (and the write-out piece read in step ks - 1) are in registers
    if constexpr (SAVE) SWN_WAIT_VM(2); else SWN_WAIT_VM(1);   // this wave's copy of step ks + 1 has landed (in-order counter: every
                                                               // step issues [store,] copy - see the header)
    __builtin_amdgcn_s_barrier();           // ... and everybody else's; every wave is done reading the slot of step ks
    SWN_PIN();
    if constexpr (SAVE) {                   // (first: the piece was read a step ago, no LDS read of this step is pending yet)
      if (ks >= 1) __builtin_amdgcn_raw_buffer_store_b128(wo, rs_save, lane16, ((ks - 1) * 8 + cx.w) * 1024, 0);
    }
    SWN_PIN();
    if (ks + 1 < KSTEPS) read_frags(ks + 1, cur ^ 1);
    {   // copy of K step (ks + 3) of the stream into the slot of step ks: this wave's feature tile, 1 KiB
      const int nx = ks + 3;
      if (nx < KSTEPS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cur, SWN_LDS(smem + RING0 + cx.slot_off[nx % 3] + cx.w * 1024), 16, lane16, nx * 1024, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_nxt, SWN_LDS(smem + RING0 + cx.slot_off[nx % 3] + cx.w * 1024), 16, lane16, (nx - KSTEPS) * 1024, 0, 0);
    }
    if constexpr (SAVE) wo = *(const u32x4_t*)(smem + cx.wo_base + ks * 8192);
    SWN_PIN();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = E::mfma(fw[cur][ni], fa[cur][mi], acc[mi][ni]);
    }
    SWN_PIN();
  }
  if constexpr (SAVE) {
    SWN_WAIT_LGKM0();
    __builtin_amdgcn_raw_buffer_store_b128(wo, rs_save, lane16, ((KSTEPS - 1) * 8 + cx.w) * 1024, 0);
  }
}

// ---- epilogue of one layer: accumulators (+bias, +skip input) -> ReLU (recording the mask) / stored mask -> the tile, in place ----
// A lane owns row 128 rg + 32 mi + l31 and, per (ni, g4), features 64 fg + 32 ni + 8 g4 + 4 lhi .. + 3.
// Mask layout: 128 bits per lane and layer = one dword per mi; value e = ni * 16 + g4 * 4 + j sits at bit 31 - e.
template <typename E, int RELU, bool BIAS, bool SKIP>
__device__ __forceinline__ void epilogue(f32x16_t (&acc)[4][2], const Ctx& cx, u32x4_t& mk) {
  char* smem = cx.smem;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    uint32_t mbits = (RELU == 2) ? mk[mi] : 0u;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const uint32_t addr = (cx.e_base ^ (uint32_t)((4 * ni + g4) << 4)) + mi * (32 * ROWB);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][g4 * 4 + j];
        if constexpr (BIAS) {
          // bias of features 64 fg + 32 ni + 8 g4 + 4 lhi ..: e_base's feature part is not needed, rebuild the index
          const f32x4_t b4 = *(const f32x4_t*)(smem + BIAS0 + (((cx.w & 3) * 64 + 32 * ni + 8 * g4 + 4 * cx.lhi) << 2));
          v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
        }
        if constexpr (SKIP) {
          const u32x2_t xv = *(const u32x2_t*)(smem + addr);
          v[0] += E::lo(xv[0]); v[1] += E::hi(xv[0]); v[2] += E::lo(xv[1]); v[3] += E::hi(xv[1]);
        }
        if constexpr (RELU == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool pos = v[j] > 0.f;
            mbits = (mbits << 1) | (pos ? 1u : 0u);
            v[j] = pos ? v[j] : 0.f;
          }
        } else if constexpr (RELU == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = ni * 16 + g4 * 4 + j;
            v[j] = ((mbits >> (31 - e)) & 1u) ? v[j] : 0.f;
          }
        }
        u32x2_t pk;
        pk[0] = E::pack2(v[0], v[1]);
        pk[1] = E::pack2(v[2], v[3]);
        *(u32x2_t*)(smem + addr) = pk;
        SWN_PIN();                            // one group at a time: small register footprint
      }
    }
    if constexpr (RELU == 1) mk[mi] = mbits;
  }
}

template <typename E>
__device__ __forceinline__ void epilogue_dispatch(f32x16_t (&acc)[4][2], const Ctx& cx, u32x4_t& mk, int relu, bool bias, bool skip) {
  if (relu == 1) {
    if (skip) { if (bias) epilogue<E, 1, true, true>(acc, cx, mk); else epilogue<E, 1, false, true>(acc, cx, mk); }
    else { if (bias) epilogue<E, 1, true, false>(acc, cx, mk); else epilogue<E, 1, false, false>(acc, cx, mk); }
  } else if (relu == 2) {
    if (skip) epilogue<E, 2, false, true>(acc, cx, mk); else epilogue<E, 2, false, false>(acc, cx, mk);
  } else {
    if (skip) { if (bias) epilogue<E, 0, true, true>(acc, cx, mk); else epilogue<E, 0, false, true>(acc, cx, mk); }
    else { if (bias) epilogue<E, 0, true, false>(acc, cx, mk); else epilogue<E, 0, false, false>(acc, cx, mk); }
  }
}

// the (gathered) chain input rows -> the swizzled tile: 128 pieces of 1 KiB (2 rows), 16 per wave, global_load ... lds
__device__ __forceinline__ void stage_input(const Ctx& cx, const char* x) {
  const int* idx = (const int*)(cx.smem + IDX0);
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int c = j * 8 + cx.w;
    const int r = 2 * c + cx.lhi;
    const long src = idx[r];
    const int q = cx.l31 ^ (r & 15);          // LDS position l31 of row r holds chunk q
    __builtin_amdgcn_global_load_lds(SWN_GLB(x + src * ROWB + q * 16), SWN_LDS(cx.smem + c * 1024), 16, 0, 0);
  }
}

template <typename E, int TAG>
__global__ __launch_bounds__(NT, 2) void chainb_kernel(const Args args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const swn_chain_desc& d = args.d;
  Ctx cx;
  cx.smem = smem;
  const int tid = threadIdx.x;
  cx.lane = tid & 63;
  cx.w = __builtin_amdgcn_readfirstlane(tid >> 6);
  cx.l31 = cx.lane & 31;
  cx.lhi = cx.lane >> 5;
  const int rg = cx.w >> 2, fg = cx.w & 3;
  const int r15 = cx.lane & 15;

  // workgroup -> (group, tile): same mapping as chain.hip (XCD x works on the weight sets = x mod 8, consecutive tiles of a group)
  int g = blockIdx.x % d.n_groups;
  int tile = blockIdx.x / d.n_groups;
  if ((d.n_wsets & 7) == 0 && (d.n_groups & 7) == 0) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int s_ = q / args.tiles_per_group;
    tile = q - s_ * args.tiles_per_group;
    g = x + 8 * s_;
  }
  int rows_valid = d.group_stride;
  if (d.group_rows) rows_valid = d.group_rows[g];
  if (rows_valid > d.group_rows_clamp) rows_valid = d.group_rows_clamp;
  const int row0 = tile * BM;
  if (row0 >= rows_valid) return;
  const int rows_in_tile = min(BM, rows_valid - row0);
  const long grow0 = (long)g * d.group_stride + row0;
  const int wset = g % d.n_wsets;
  const int n_layers = d.n_layers;

  cx.a_base = (uint32_t)((128 * rg + cx.l31) * ROWB + ((cx.lhi ^ r15) << 4));
  cx.e_base = (uint32_t)((128 * rg + cx.l31) * ROWB + (r15 << 4) + 8 * cx.lhi) ^ (uint32_t)(fg << 7);
  cx.wf_base = (uint32_t)(RING0 + (2 * fg) * 1024 + cx.lane * 16);
  cx.wo_base = (uint32_t)((2 * cx.w + cx.lhi) * ROWB + ((cx.l31 ^ ((2 * cx.w + cx.lhi) & 15)) << 4));
  const int lane16 = cx.lane * 16;

  // this wave's weight-fragment stream of layer L: feature tile w, 16 K steps of 1 KiB
  auto wrs = [&](int L) -> __amdgpu_buffer_rsrc_t {
    const char* p = (const char*)d.layers[L].w + ((size_t)wset * 8 + cx.w) * (KSTEPS * 1024);
    return uniform_rsrc(p, KSTEPS * 1024);
  };
  auto out_rs = [&](void* base) -> __amdgpu_buffer_rsrc_t {     // rows of this tile in a row-major [*, 256] tensor, clipped to the valid rows
    return uniform_rsrc((char*)base + grow0 * ROWB, rows_in_tile * ROWB);
  };
  auto stage_bias = [&](int L) {        // 1 KiB, all waves issue (waves 4..7 repeat 0..3): keeps the per-wave vmcnt bookkeeping uniform
    const float* b = d.layers[L].b;
    if (b) {
      const __amdgpu_buffer_rsrc_t rb = uniform_rsrc(b + (size_t)wset * 256, 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, SWN_LDS(smem + BIAS0 + (cx.w & 3) * 256), 4, cx.lane * 4, (cx.w & 3) * 256, 0, 0);
    }
  };

  // ---- prologue: weight ring (steps 0..2 of layer 0), source rows, input tile, bias ----
  cx.slot_off[0] = 0; cx.slot_off[1] = SLOT_B; cx.slot_off[2] = 2 * SLOT_B;
  {
    const __amdgpu_buffer_rsrc_t r0 = wrs(0);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, SWN_LDS(smem + RING0 + s * SLOT_B + cx.w * 1024), 16, lane16, s * 1024, 0, 0);
  }
  if (tid < BM) {
    const long gr = grow0 + (tid < rows_in_tile ? tid : 0);      // rows past the end repeat the first row (computed, never stored)
    long src = d.x_gather ? (long)d.x_gather[gr] : gr;
    if (src < 0) src = 0;
    ((int*)(smem + IDX0))[tid] = (int)src;
  }
  SWN_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();
  stage_input(cx, (const char*)d.x);
  stage_bias(0);
  SWN_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();

  f32x16_t acc[4][2];
  for (int L = 0; L < n_layers; ++L) {
    const swn_chain_layer& ly = d.layers[L];
    const bool has_next = (L + 1) < n_layers;
    const __amdgpu_buffer_rsrc_t rs_cur = wrs(L);
    const __amdgpu_buffer_rsrc_t rs_nxt = has_next ? wrs(L + 1) : rs_cur;     // (end of chain: a valid stream, copied and never read)
    void* save_in = L > 0 ? d.layers[L - 1].save : nullptr;
    u32x4_t mk = {0u, 0u, 0u, 0u};
    uint32_t* mkp = ly.mask ? ly.mask + ((size_t)(blockIdx.x * 8 + cx.w) * 64 + cx.lane) * 4 : nullptr;
    if (ly.relu == 2) mk = *(const u32x4_t*)mkp;
    {   // ring slots of this layer's K steps: step (16 L + j) -> slot (L + j) mod 3
      const int s0 = L % 3;
      cx.slot_off[0] = s0 * SLOT_B;
      cx.slot_off[1] = ((s0 + 1) % 3) * SLOT_B;
      cx.slot_off[2] = ((s0 + 2) % 3) * SLOT_B;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (save_in) k_loop<E, true>(acc, cx, rs_cur, rs_nxt, out_rs(save_in));
    else k_loop<E, false>(acc, cx, rs_cur, rs_nxt, rs_cur);

    SWN_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();       // every wave has finished reading the tile (fragments and write-out)
    if (ly.skip) {                      // the residual input: bring the chain input back into the (dead) tile; the epilogue reads it in place
      stage_input(cx, (const char*)d.x);
      SWN_WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
    }
    epilogue_dispatch<E>(acc, cx, mk, ly.relu, ly.b != nullptr, ly.skip != 0);
    if (ly.relu == 1 && mkp) *(u32x4_t*)mkp = mk;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // copies of the next steps landed long ago; the tile is rewritten
    __builtin_amdgcn_s_barrier();
    if (has_next) stage_bias(L + 1);    // (the bias slot is free: every wave is past its epilogue)
  }

  // ---- the chain output: row-major, coalesced, + y_add rows ----
  {
    const __amdgpu_buffer_rsrc_t ry = out_rs(d.y);
    const __amdgpu_buffer_rsrc_t ra = d.y_add ? out_rs((void*)d.y_add) : ry;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      u32x4_t v = *(const u32x4_t*)(smem + cx.wo_base + j * 8192);
      const int soff = (j * 8 + cx.w) * 1024;
      if (d.y_add) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(ra, lane16, soff, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = E::pack2(E::lo(v[q]) + E::lo(a[q]), E::hi(v[q]) + E::hi(a[q]));
      }
      __builtin_amdgcn_raw_buffer_store_b128(v, ry, lane16, soff, 0);
    }
  }
  SWN_WAIT_VM(0);      // no LDS copy may be in flight when the workgroup retires
}

}  // namespace swn_big

namespace swn {

bool chain_big_eligible(const swn_chain_desc& d) {
  if (d.dtype != SWN_BF16 && d.dtype != SWN_F16) return false;
  if (d.x_save || d.x_scale || d.y_add_gather) return false;
  for (int l = 0; l < d.n_layers; ++l) {
    const swn_chain_layer& ly = d.layers[l];
    if (ly.n != 256 || ly.k != 256 || ly.rowbias || ly.skip > 1) return false;
  }
  return true;
}

int chain_big_launch(const swn_chain_desc& d, void* stream) {
  using namespace swn_big;
  Args a;
  a.d = d;
  a.tiles_per_group = cdiv(d.group_rows ? (d.group_rows_clamp < d.group_stride ? d.group_rows_clamp : d.group_stride) : d.group_stride, BM);
  if (!d.group_rows) a.d.group_rows_clamp = d.group_stride;
  const long grid = (long)a.tiles_per_group * d.n_groups;
  SWN_CHECK(grid > 0 && grid < (1L << 31), "swn_mlp_chain: grid %ld out of range", grid);
  const void* fn = nullptr;
#define SWN_PICKB(TAGV)                                                                                                     \
  case TAGV:                                                                                                                \
    fn = d.dtype == SWN_BF16 ? (const void*)chainb_kernel<Bf16, TAGV> : (const void*)chainb_kernel<Fp16, TAGV>;             \
    break;
  switch (d.tag) {
    SWN_PICKB(1) SWN_PICKB(2)
    default: fn = d.dtype == SWN_BF16 ? (const void*)chainb_kernel<Bf16, 0> : (const void*)chainb_kernel<Fp16, 0>;
  }
#undef SWN_PICKB
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(NT), kargs, LDS_BYTES, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_mlp_chain (256-row geometry) launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace swn

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

// Two workgroup geometries share the code (template parameters NW = waves, MI = 32-row tiles per wave):
//   G256 (NW 8, MI 4): ONE 512-thread workgroup per CU on a 256-row tile; wave (rg, fg) owns rows [128 rg, +128) x features [64 fg, +64);
//                      LDS 128 KiB tile + 24 KiB ring + index / bias scratch.
//   G96  (NW 4, MI 3): TWO independent 256-thread workgroups per CU on 96-row tiles (48 KiB tile + 24 KiB ring each); wave fg owns all
//                      96 rows x features [64 fg, +64).  A workgroup's epilogue / prologue / write-out (VALU, LDS, latency) overlaps the
//                      other one's K loop on the same SIMDs - the lockstep G256 workgroup leaves the matrix pipe idle there - at the
//                      price of 2.7x the weight copies per row of G256 (still 1.5x fewer than the 64-row kernels).
constexpr int ROWB = 512;               // tile row stride in bytes (256 two-byte features)
constexpr int SLOT_B = 8192;            // weights of one K step: 8 feature tiles x 1 KiB
constexpr int NSLOT = 3;
constexpr int KSTEPS = 16;              // 256 / 16

template <int NW_, int MI_> struct Geo {
  static constexpr int NW = NW_, MI = MI_;
  static constexpr int RG = NW / 4;                  // row groups (waves with the same feature slab)
  static constexpr int BM = RG * MI * 32;            // rows per tile
  static constexpr int NT = NW * 64;
  static constexpr int TILE_B = BM * ROWB;
  static constexpr int RING0 = TILE_B;
  static constexpr int IDX0 = RING0 + NSLOT * SLOT_B;   // int32 [BM]: source row of every tile row
  static constexpr int BIAS0 = IDX0 + 1024;             // f32 [256]
  static constexpr int LDS_BYTES = BIAS0 + 1024;
  static constexpr int PIECES = BM / 2;              // 1 KiB pieces (2 rows) of a tile
  static constexpr int NCOPY = NW / 2;               // copy waves (w < NCOPY), each copies FPC feature tiles per K step
  static constexpr int FPC = 8 / NCOPY;
  static constexpr int OCC = 2;                      // waves per SIMD (launch bound): 1 x 8 or 2 x 4
};
typedef Geo<8, 4> G256;
typedef Geo<4, 3> G96;

struct Args {
  swn_chain_desc d;
  int tiles_per_group;
};

// -DSWN_BIG_TIMING: s_memtime phase timers of wave 0 of the first 4096 workgroups, written through d.y_add_gather (reinterpreted as
// int64 [4096][8]: K-loop waits, K-loop barriers, K loops, epilogues, post-K barrier + restage, prologue, write-out, total;
// scripts/chain_big_check.py timing).  Off = no code.
#ifdef SWN_BIG_TIMING
#define TICK() __builtin_amdgcn_s_memtime()
struct Timers { long long wait = 0, bar = 0, kloop = 0, epi = 0, mid = 0, pro = 0, wout = 0; };
#define SWN_TM(...) __VA_ARGS__
#else
#define SWN_TM(...)
struct Timers {};
#endif

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

#ifndef SWN_BIG_STORE_AUX
#define SWN_BIG_STORE_AUX 2       // cache policy of the activation stores (1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef SWN_BIG_Y_AUX
#define SWN_BIG_Y_AUX SWN_BIG_STORE_AUX      // ... of the chain OUTPUT (read back by the next kernel, unlike the saved activations)
#endif
#define SWN_PIN() __builtin_amdgcn_sched_barrier(0)
// Store-operand hold.  The registers of a 16-byte store must stay ALLOCATED behind it: the value is dead behind its store, the compiler
// hands the registers to the next temporaries, and with a VALU write two instructions behind the store single dwords reached memory with
// the NEW value under store back-pressure (profiles/r04_experiments.md section 5; the documented hazard is one wait state).  An empty asm
// statement that READS the operands keeps them allocated up to the program point where it stands - put it where the next write to the
// registers is structurally far: behind the s_waitcnt vmcnt(0) that retires the stores (S phase, end of kernel), or at the refill of an
// alternating register set a half epilogue step later (E phase).  No idle issue slots anywhere.
#define SWN_KEEP4(a) asm volatile("" :: "v"((a)[0]), "v"((a)[1]), "v"((a)[2]), "v"((a)[3]))
#define SWN_KEEP8(a) asm volatile("" :: "v"((a)[0]), "v"((a)[1]), "v"((a)[2]), "v"((a)[3]), "v"((a)[4]), "v"((a)[5]), "v"((a)[6]), "v"((a)[7]))
#define SWN_KEEP16(a) do { SWN_KEEP8(a); SWN_KEEP8((a) + 8); } while (0)
#define SWN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SWN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// per-wave / per-lane constants of a workgroup
struct Ctx {
  char* smem;
  int w, lane, l31, lhi;
  uint32_t a_base;       // LDS byte address of this lane's activation-fragment row (mi = 0) incl. the swizzle seed; ^ (ks << 5) per step
  uint32_t e_base;       // ... of this lane's epilogue row incl. swizzle seed, half-wave and wave feature offset; ^ ((4 ni + g4) << 4)
  uint32_t e2_base;      // the same for the 16-byte writes after the half-wave exchange: chunk g4 + lhi, no 8-byte half offset
  uint32_t wf_base;      // RING0 + this wave's first feature tile + lane * 16 (add the slot offset)
  int slot_off[3];       // byte offset of the ring slot of K step (16 L + j), j mod 3 - refreshed per layer
};

// LDS byte address of this lane's 16 bytes of write-out piece c (tile rows 2 c, 2 c + 1; lane -> row 2 c + lhi, 16-byte column l31)
__device__ __forceinline__ uint32_t piece_addr(const Ctx& cx, int c) {
  const int r = 2 * c + cx.lhi;
  return (uint32_t)(r * ROWB + ((cx.l31 ^ (r & 15)) << 4));
}

// ---- the K loop of one layer ------------------------------------------------------------------------------------------------
// Roles (vmcnt completes in order per wave, so a wave that both copies weights and stores activations can keep only ~2 stores in
// flight behind the copy it waits for - far too few bytes to cover the HBM write latency):
//   COPY  (waves 0 .. NW/2 - 1): FPC weight copies per K step, counted wait for the copies of step ks + 1;
//   STORE (the other waves):     two 1 KiB pieces of the write-out per K step (SAVE: the input tile of this layer is a saved
//                                activation), never waits for its stores - up to 63 of them in flight per wave.
// One wave of each role per SIMD (G256) / per SIMD pair (G96).
template <typename E, typename G, bool SAVE>
__device__ __forceinline__ void k_loop(f32x16_t (&acc)[G::MI][2], const Ctx& cx, __amdgpu_buffer_rsrc_t rs_cur, __amdgpu_buffer_rsrc_t rs_nxt,
                                       __amdgpu_buffer_rsrc_t rs_save, const bool COPY /* wave-uniform */, Timers& tm) {
  constexpr int MI = G::MI;
  char* smem = cx.smem;
  const int lane16 = cx.lane * 16;
  const bool STORE = SAVE && !COPY;
  u32x4_t fa[2][MI], fw[2][2];
  u32x4_t wo[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  // launder the per-lane bases: stops the compiler from hoisting the 16 swizzled fragment addresses (and friends) of the unrolled
  // steps out of the layer loop into long-lived registers (spills next to the accumulators)
  uint32_t a_base = cx.a_base, wf_base = cx.wf_base;
  asm volatile("" : "+v"(a_base), "+v"(wf_base));
  auto read_w = [&](int ks, int set, int ni) {
    fw[set][ni] = *(const u32x4_t*)(smem + wf_base + cx.slot_off[ks % 3] + ni * 1024);
  };
  auto read_a = [&](int ks, int set, int mi) {
    fa[set][mi] = *(const u32x4_t*)(smem + (a_base ^ (uint32_t)(ks << 5)) + mi * (32 * ROWB));
  };
  auto copy = [&](int ks, int i) {     // feature tile FPC w + i of K step ks + 3 of the stream -> the slot of step ks
    const int nx = ks + 3, t = G::FPC * cx.w + i;
    if (nx < KSTEPS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cur, SWN_LDS(smem + G::RING0 + cx.slot_off[nx % 3] + t * 1024), 16, lane16, (t * KSTEPS + nx) * 1024, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_nxt, SWN_LDS(smem + G::RING0 + cx.slot_off[nx % 3] + t * 1024), 16, lane16, (t * KSTEPS + nx - KSTEPS) * 1024, 0, 0);
  };
  // write-out piece (ks, i) of this store wave: c = ks * NW + 2 (w - NW/2) + i.  G96 has 64 slots for 48 pieces: the surplus ones
  // read piece 0 and their store falls outside the descriptor (dropped).
  auto piece = [&](int ks, int i) -> int { return ks * G::NW + 2 * (cx.w - G::NCOPY) + i; };
  auto store = [&](int ks, int i) {
    __builtin_amdgcn_raw_buffer_store_b128(wo[i], rs_save, lane16, piece(ks, i) * 1024, SWN_BIG_STORE_AUX);
  };
  auto read_piece = [&](int ks, int i) {
    const int c = piece(ks, i);
    wo[i] = *(const u32x4_t*)(smem + piece_addr(cx, c < G::PIECES ? c : 0));
  };
  read_w(0, 0, 0); read_w(0, 0, 1);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) read_a(0, 0, mi);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int cur = ks & 1, nxt = cur ^ 1;
#ifdef SWN_ABL_NOREAD
    const bool more = false;
#else
    const bool more = ks + 1 < KSTEPS;
#endif
    SWN_TM(const long long q0 = TICK();)
    SWN_WAIT_LGKM0();                       // the fragments of step ks (and the write-out pieces read in step ks - 1) are in registers
#ifndef SWN_ABL_NOCOPY
    if (COPY) {                             // this wave's copies of step ks + 1 have landed (younger: the FPC copies of step ks + 2)
      if constexpr (G::FPC == 2) SWN_WAIT_VM(2); else SWN_WAIT_VM(4);
    }
#endif
    SWN_TM(const long long q1 = TICK();)
#ifndef SWN_ABL_NOBAR
    __builtin_amdgcn_s_barrier();           // ... and everybody else's; every wave is done reading the slot of step ks
#endif
    SWN_TM(const long long q2 = TICK(); tm.wait += q1 - q0; tm.bar += q2 - q1;)
    SWN_PIN();
    // The matrix pipe starts at once; the LDS reads of the next step, the copies of step ks + 3 / the stores of the previous
    // write-out pieces and the reads of this step's pieces are spread between the MFMAs (order pinned).
#ifdef SWN_ABL_NOMFMA
#define SWN_MM(mi, ni) asm volatile("" :: "v"(fw[cur][ni]), "v"(fa[cur][mi]))
#else
#define SWN_MM(mi, ni) acc[mi][ni] = E::mfma(fw[cur][ni], fa[cur][mi], acc[mi][ni])
#endif
    SWN_MM(0, 0);
    SWN_PIN();
    if (more) { read_w(ks + 1, nxt, 0); read_w(ks + 1, nxt, 1); }
    SWN_PIN();
    SWN_MM(0, 1);
    SWN_PIN();
    if (more) read_a(ks + 1, nxt, 0);
#ifndef SWN_ABL_NOCOPY
    if (COPY) { copy(ks, 0); if constexpr (G::FPC == 4) copy(ks, 1); }
#else
    if (false) {}
#endif
    else if (SAVE && ks >= 1) store(ks - 1, 0);
    SWN_PIN();
    SWN_MM(1, 0);
    SWN_PIN();
    if (more) read_a(ks + 1, nxt, 1);
#ifndef SWN_ABL_NOCOPY
    if (COPY) { copy(ks, G::FPC - 1); if constexpr (G::FPC == 4) copy(ks, 2); }
#else
    if (false) {}
#endif
    else if (SAVE && ks >= 1) store(ks - 1, 1);
    SWN_PIN();
    SWN_MM(1, 1);
    SWN_PIN();
    if (more) read_a(ks + 1, nxt, 2);
    SWN_PIN();
    SWN_MM(2, 0);
    SWN_PIN();
    if constexpr (MI == 4) { if (more) read_a(ks + 1, nxt, 3); }
    else { if (STORE) read_piece(ks, 0); }
    SWN_PIN();
    SWN_MM(2, 1);
    SWN_PIN();
    if constexpr (MI == 4) {
      if (STORE) read_piece(ks, 0);
      SWN_PIN();
      SWN_MM(3, 0);
      SWN_PIN();
      if (STORE) read_piece(ks, 1);
      SWN_PIN();
      SWN_MM(3, 1);
    } else {
      if (STORE) read_piece(ks, 1);
    }
#undef SWN_MM
    SWN_PIN();
  }
  if (STORE) {
    SWN_WAIT_LGKM0();
    store(KSTEPS - 1, 0);
    store(KSTEPS - 1, 1);
  }
}

// ---- epilogue of one layer: accumulators (+bias, +skip input) -> ReLU (recording the mask) / stored mask -> the tile, in place ----
// A lane owns row (32 MI rg + 32 mi + l31) and, per (ni, g4), features 64 fg + 32 ni + 8 g4 + 4 lhi .. + 3.
// A non-packed VALU instruction costs a wave 4 clocks on CDNA4 (16 lanes per clock), an MFMA 32: ~5 VALU per value were as
// expensive as the K loop.  ReLU and its mask therefore work on the PACKED 16-bit results (two values per instruction; bf16 and
// fp16 are sign-magnitude, so as int16 a negative value or -0 is < 0):
//   forward:  p = cvt_pk(z0, z1);  p = pk_max_i16(p, 0)  [ReLU];  q = pk_min_u16(p, 1)  [1 where the output is > 0];  m |= q << d
//   backward: t = pk_min_u16(m & (0x00010001 << d), 1);  p = pk_mul_lo_u16(cvt_pk(g0, g1), t)
// The mask bit is (rounded output > 0) - what the reference's autocast ReLU backward tests - and equals (z > 0) unless
// 0 < z < 2^-134.  Mask layout: 128 bits per lane and layer = one dword per mi; packed dword d = ni * 8 + g4 * 2 + i of that mi
// has its low half at bit d, its high half at bit d + 16.
// The half-waves exchange 8-byte pieces (v_permlane32_swap) so that a lane writes one whole 16-byte chunk: ds_write_b128 into the
// swizzled tile is conflict-free, the 8-byte form is 2-way conflicted.
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t_;
__device__ __forceinline__ uint32_t pk_relu16(uint32_t p) {
  const i16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, p), z));
}
// (inline asm: hipcc rewrites the vector-extension forms of these two into compare / select / v_perm sequences of five instructions)
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

// bias_off: byte offset of the bias slot behind BIAS0 (geometry 4 double-buffers it); hook(mi) runs after the row tile mi is rewritten.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// HalfHook<F>: a hook that epilogue_p calls after every HALF row tile (h = 2 mi + ni, 8 calls) instead of after every row tile
template <typename F> struct HalfHook {
  F f;
  __device__ __forceinline__ void operator()(int h) const { f(h); }
};
template <typename T> struct hook_halves { static constexpr bool value = false; };
template <typename F> struct hook_halves<HalfHook<F>> { static constexpr bool value = true; };
template <typename E, typename G, int RELU, bool BIAS, bool SKIP, typename HOOK = NoHook>
__device__ __forceinline__ void epilogue(f32x16_t (&acc)[G::MI][2], const Ctx& cx, u32x4_t& mk, int bias_off = 0, HOOK hook = HOOK()) {
  char* smem = cx.smem;
  uint32_t e_base = cx.e_base, e2_base = cx.e2_base;      // laundered: keeps the swizzled addresses inside the layer loop (see k_loop)
  asm volatile("" : "+v"(e_base), "+v"(e2_base));
#pragma unroll
  for (int mi = 0; mi < G::MI; ++mi) {
    f32x4_t bias[2][4];        // (re-read per row tile: 32 registers that would otherwise live through the whole epilogue; broadcast reads)
    if constexpr (BIAS) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          bias[ni][g4] = *(const f32x4_t*)(smem + G::BIAS0 + bias_off + (((cx.w & 3) * 64 + 32 * ni + 8 * g4 + 4 * cx.lhi) << 2));
    }
    u32x2_t xv[2][4];
    if constexpr (SKIP) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          xv[ni][g4] = *(const u32x2_t*)(smem + (e_base ^ (uint32_t)((4 * ni + g4) << 4)) + mi * (32 * ROWB));
    }
    uint32_t mbits = (RELU == 2) ? mk[mi] : 0u;
    u32x2_t pk[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x2_t_ v = {acc[mi][ni][g4 * 4 + 2 * i], acc[mi][ni][g4 * 4 + 2 * i + 1]};
          if constexpr (BIAS) {
            const f32x2_t_ b2 = {bias[ni][g4][2 * i], bias[ni][g4][2 * i + 1]};
            v += b2;
          }
          if constexpr (SKIP) {
            const f32x2_t_ x2 = {E::lo(xv[ni][g4][i]), E::hi(xv[ni][g4][i])};
            v += x2;
          }
          uint32_t p = E::pack2(v[0], v[1]);
          const int d = ni * 8 + g4 * 2 + i;
          if constexpr (RELU == 1) {
            p = pk_relu16(p);
            mbits |= pk_nonzero16(p) << d;
          } else if constexpr (RELU == 2) {
            p = pk_mul16(p, pk_nonzero16(mbits & (0x00010001u << d)));
          }
          pk[ni][g4][i] = p;
        }
      }
    }
    if constexpr (RELU == 1) mk[mi] = mbits;
    if constexpr (SKIP) SWN_PIN();          // every read of the residual rows of this mi is consumed before the tile rows are rewritten
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        // lower half-wave ends up with chunk g (features 8 g .. 8 g + 7 of its row), the upper one with chunk g + 1
        auto r0 = __builtin_amdgcn_permlane32_swap(pk[ni][g][0], pk[ni][g + 1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(pk[ni][g][1], pk[ni][g + 1][1], false, false);
        const u32x4_t o = {(uint32_t)r0[0], (uint32_t)r1[0], (uint32_t)r0[1], (uint32_t)r1[1]};
        *(u32x4_t*)(smem + (e2_base ^ (uint32_t)((4 * ni + g) << 4)) + mi * (32 * ROWB)) = o;
      }
    }
    SWN_PIN();                              // one row tile at a time: bounded register footprint
    hook(mi);
  }
}

template <typename E, typename G, typename HOOK = NoHook>
__device__ __forceinline__ void epilogue_dispatch(f32x16_t (&acc)[G::MI][2], const Ctx& cx, u32x4_t& mk, int relu, bool bias, bool skip,
                                                  int boff = 0, HOOK hook = HOOK()) {
  if (relu == 1) {
    if (skip) { if (bias) epilogue<E, G, 1, true, true, HOOK>(acc, cx, mk, boff, hook); else epilogue<E, G, 1, false, true, HOOK>(acc, cx, mk, boff, hook); }
    else { if (bias) epilogue<E, G, 1, true, false, HOOK>(acc, cx, mk, boff, hook); else epilogue<E, G, 1, false, false, HOOK>(acc, cx, mk, boff, hook); }
  } else if (relu == 2) {
    if (skip) epilogue<E, G, 2, false, true, HOOK>(acc, cx, mk, boff, hook); else epilogue<E, G, 2, false, false, HOOK>(acc, cx, mk, boff, hook);
  } else {
    if (skip) { if (bias) epilogue<E, G, 0, true, true, HOOK>(acc, cx, mk, boff, hook); else epilogue<E, G, 0, false, true, HOOK>(acc, cx, mk, boff, hook); }
    else { if (bias) epilogue<E, G, 0, true, false, HOOK>(acc, cx, mk, boff, hook); else epilogue<E, G, 0, false, false, HOOK>(acc, cx, mk, boff, hook); }
  }
}

// the (gathered) chain input rows -> the swizzled tile: 1 KiB pieces (2 rows), PIECES / NW per wave, global_load ... lds
template <typename G>
__device__ __forceinline__ void stage_input(const Ctx& cx, const char* x) {
  const int* idx = (const int*)(cx.smem + G::IDX0);
#pragma unroll 4
  for (int j = 0; j < G::PIECES / G::NW; ++j) {
    const int c = j * G::NW + cx.w;
    const int r = 2 * c + cx.lhi;
    const long src = idx[r];
    const int q = cx.l31 ^ (r & 15);          // LDS position l31 of row r holds chunk q
    __builtin_amdgcn_global_load_lds(SWN_GLB(x + src * ROWB + q * 16), SWN_LDS(cx.smem + c * 1024), 16, 0, 0);
  }
}

template <typename E, typename G, int TAG>
__global__ __launch_bounds__(G::NT, G::OCC) void chainb_kernel(const Args args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = G::BM, MI = G::MI;
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
    g = ((x + s_) & 7) + 8 * s_;     // rotated per 8 groups: every XCD meets every weight set - an expert that draws more rows than the
                                     // others (unbalanced routing) would otherwise make ITS XCD the long pole of the launch
  }
  int rows_valid = d.group_stride;
  if (d.group_rows) rows_valid = d.group_rows[g];
  if (rows_valid > d.group_rows_clamp) rows_valid = d.group_rows_clamp;
  const int row0 = tile * BM;
  if (row0 >= rows_valid) return;
  const int rows_in_tile = min(BM, rows_valid - row0);
  const long grow0 = (d.group_begin ? (long)d.group_begin[g] : (long)g * d.group_stride) + row0;
  const int wset = g % d.n_wsets;
  const int n_layers = d.n_layers;

  const int lrow = 32 * MI * rg + cx.l31;                 // this lane's tile row for mi = 0
  cx.a_base = (uint32_t)(lrow * ROWB + ((cx.lhi ^ r15) << 4));
  cx.e_base = (uint32_t)(lrow * ROWB + (r15 << 4) + 8 * cx.lhi) ^ (uint32_t)(fg << 7);
  cx.e2_base = (uint32_t)(lrow * ROWB + (r15 << 4)) ^ (uint32_t)((fg << 7) | (cx.lhi << 4));
  cx.wf_base = (uint32_t)(G::RING0 + (2 * fg) * 1024 + cx.lane * 16);
  const int lane16 = cx.lane * 16;
  const bool copy_role = cx.w < G::NCOPY;

  // the weight-fragment stream of layer L, weight set `wset`: 8 feature tiles x 16 K steps of 1 KiB
  auto wrs = [&](int L) -> __amdgpu_buffer_rsrc_t {
    const char* p = (const char*)d.layers[L].w + (size_t)wset * 8 * (KSTEPS * 1024);
    return uniform_rsrc(p, 8 * KSTEPS * 1024);
  };
  auto out_rs = [&](void* base) -> __amdgpu_buffer_rsrc_t {     // rows of this tile in a row-major [*, 256] tensor, clipped to the valid rows
    return uniform_rsrc((char*)base + grow0 * ROWB, rows_in_tile * ROWB);
  };
  auto stage_bias = [&](int L) {        // 1 KiB: four waves copy 256 B each (further waves repeat them: uniform code)
    const float* b = d.layers[L].b;
    if (b) {
      const __amdgpu_buffer_rsrc_t rb = uniform_rsrc(b + (size_t)wset * 256, 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, SWN_LDS(smem + G::BIAS0 + (cx.w & 3) * 256), 4, cx.lane * 4, (cx.w & 3) * 256, 0, 0);
    }
  };

  SWN_TM(const long long t_start = TICK();)
  // ---- prologue: weight ring (steps 0..2 of layer 0), source rows, input tile, bias ----
  cx.slot_off[0] = 0; cx.slot_off[1] = SLOT_B; cx.slot_off[2] = 2 * SLOT_B;
  {
    const __amdgpu_buffer_rsrc_t r0 = wrs(0);
#pragma unroll
    for (int i = 0; i < 24 / G::NW; ++i) {      // 3 steps x 8 feature tiles, spread over the waves
      const int f = i * G::NW + cx.w, s = f >> 3, t = f & 7;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, SWN_LDS(smem + G::RING0 + s * SLOT_B + t * 1024), 16, lane16, (t * KSTEPS + s) * 1024, 0, 0);
    }
  }
  if (tid < BM) {
    const long gr = grow0 + (tid < rows_in_tile ? tid : 0);      // rows past the end repeat the first row (computed, never stored)
    long src = d.x_gather ? (long)d.x_gather[gr] : gr;
    if (src < 0) src = 0;
    ((int*)(smem + G::IDX0))[tid] = (int)src;
  }
  SWN_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();
  stage_input<G>(cx, (const char*)d.x);
  stage_bias(0);
  SWN_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();

  f32x16_t acc[MI][2];
  Timers tm;
  SWN_TM(long long t_a = t_start;)
  for (int L = 0; L < n_layers; ++L) {
    const swn_chain_layer& ly = d.layers[L];
    const bool has_next = (L + 1) < n_layers;
    const __amdgpu_buffer_rsrc_t rs_cur = wrs(L);
    const __amdgpu_buffer_rsrc_t rs_nxt = has_next ? wrs(L + 1) : rs_cur;     // (end of chain: a valid stream, copied and never read)
    void* save_in = L > 0 ? d.layers[L - 1].save : nullptr;
    u32x4_t mk = {0u, 0u, 0u, 0u};
    uint32_t* mkp = ly.mask ? ly.mask + ((size_t)(blockIdx.x * G::NW + cx.w) * 64 + cx.lane) * 4 : nullptr;
    if (ly.relu == 2) mk = *(const u32x4_t*)mkp;
    {   // ring slots of this layer's K steps: step (16 L + j) -> slot (L + j) mod 3
      const int s0 = L % 3;
      cx.slot_off[0] = s0 * SLOT_B;
      cx.slot_off[1] = ((s0 + 1) % 3) * SLOT_B;
      cx.slot_off[2] = ((s0 + 2) % 3) * SLOT_B;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    SWN_TM(const long long p0 = TICK(); if (L == 0) tm.pro = p0 - t_a;)
    if (save_in) k_loop<E, G, true>(acc, cx, rs_cur, rs_nxt, out_rs(save_in), copy_role, tm);
    else k_loop<E, G, false>(acc, cx, rs_cur, rs_nxt, rs_cur, copy_role, tm);
    SWN_TM(const long long p1 = TICK(); tm.kloop += p1 - p0;)

    SWN_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();       // every wave has finished reading the tile (fragments and write-out)
    if (ly.skip) {                      // the residual input: bring the chain input back into the (dead) tile; the epilogue reads it in place
      stage_input<G>(cx, (const char*)d.x);
      SWN_WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
    }
    SWN_TM(const long long p2 = TICK(); tm.mid += p2 - p1;)
    epilogue_dispatch<E, G>(acc, cx, mk, ly.relu, ly.b != nullptr, ly.skip != 0);
    if (ly.relu == 1 && mkp) *(u32x4_t*)mkp = mk;
    SWN_WAIT_LGKM0();                   // the tile is rewritten
    if (copy_role) SWN_WAIT_VM(0);      // the copies of the next layer's first steps landed long ago (store waves: nothing to wait for,
                                        // their activation stores stay in flight)
    __builtin_amdgcn_s_barrier();
    if (has_next) stage_bias(L + 1);    // (the bias slot is free: every wave is past its epilogue)
    SWN_TM(tm.epi += TICK() - p2;)
  }
  SWN_TM(t_a = TICK();)

  // ---- the chain output: row-major, coalesced, + y_add rows ----
  {
    const __amdgpu_buffer_rsrc_t ry = out_rs(d.y);
    const __amdgpu_buffer_rsrc_t ra = d.y_add ? out_rs((void*)d.y_add) : ry;
#pragma unroll 4
    for (int j = 0; j < G::PIECES / G::NW; ++j) {
      const int c = j * G::NW + cx.w;
      u32x4_t v = *(const u32x4_t*)(smem + piece_addr(cx, c));
      const int soff = c * 1024;
      if (d.y_add) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(ra, lane16, soff, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = E::pack2(E::lo(v[q]) + E::lo(a[q]), E::hi(v[q]) + E::hi(a[q]));
      }
      __builtin_amdgcn_raw_buffer_store_b128(v, ry, lane16, soff, SWN_BIG_STORE_AUX);
    }
  }
  SWN_WAIT_VM(0);      // no LDS copy may be in flight when the workgroup retires
#ifdef SWN_BIG_TIMING
  if (d.y_add_gather && tid == 0 && blockIdx.x < 4096) {
    long long* dbg = (long long*)d.y_add_gather + (long)blockIdx.x * 8;
    const long long t_end = TICK();
    dbg[0] = tm.wait; dbg[1] = tm.bar; dbg[2] = tm.kloop; dbg[3] = tm.epi; dbg[4] = tm.mid; dbg[5] = tm.pro; dbg[6] = t_end - t_a; dbg[7] = t_end - t_start;
  }
#endif
}

// =================================================================================================================================
// Geometry 4: the 256-row workgroup with its two row groups HALF A LAYER APART.
//
// In chainb_kernel all 8 waves run the K loop together and then the epilogue together: the matrix pipe idles for the epilogue's 5 k
// clocks per layer and the K loop pays a workgroup barrier per step.  Here row group 0 (waves 0-3) and row group 1 (waves 4-7, the
// SAME SIMDs: a workgroup's waves go to the SIMDs cyclically) alternate roles per PHASE:
//     phase 2 l     : group 0 runs the K loop of layer l          | group 1 runs the epilogue of layer l - 1
//     phase 2 l + 1 : group 0 runs the epilogue of layer l        | group 1 runs the K loop of layer l
// so a SIMD always has one wave issuing MFMAs and one doing the VALU / LDS / store work.  One workgroup barrier per phase (two per
// layer) instead of one per K step:
//   * weights: wave (g, fg) loads the fragments of ITS two feature tiles itself, straight into a 4-deep register ring (2 x 1 KiB per
//     K step, three steps ahead, counted vmcnt; the first three steps of a K loop are requested at the end of the wave's preceding
//     epilogue phase) - no other wave reads them, nothing is handed over.  The weights pass the CU twice per 256 rows (the price of
//     the phase shift: 2x chainb's weight copies);
//   * an epilogue wave first writes out the PARTNER group's rows (the input tile of the K loop the partner is running = a saved
//     activation: 16 x 1 KiB pieces per wave, non-temporal, never waited for inside the phase - the following K phase's first counted
//     vmcnt is where a backlog of the store stream shows), then rewrites its own rows in place; bias and stored masks come straight
//     from global memory into registers at the start of the phase;
//   * the residual (skip) layer re-stages the chain input rows of the group into its own (dead) rows; the four waves of the group
//     meet at an LDS counter (the only place where a group has to synchronise inside a phase).
// Results and the ReLU mask layout are identical to chainb_kernel's (same K order per accumulator, same epilogue code).
// =================================================================================================================================
#define SWN_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

// The K loop of geometry 4: the wave's weight fragments come from global memory STRAIGHT INTO REGISTERS (nobody else reads them: the
// LDS ring of chainb_kernel - and of this kernel's first version - would be a detour): a 4-deep register ring, three K steps ahead, counted vmcnt.  The activation
// fragments are read TWO steps ahead (three register sets): the LDS round trip of a lone wave's reads (~300 clocks with four waves
// reading at once) no longer fits into one 256-clock step.  `wq` holds the fragments of steps 0..2 on entry (issued by the caller before
// the phase barrier) - on exit nothing is in flight.
// (Closed experiments, removed from the source in round 6: s_setprio for the K phase's wave - no change, r04_experiments.md 12; the K
//  loop rolled to four steps per trip - 3-4 % slower, r04 24; row group 1 without its weight loads - the upper bound of any weight
//  sharing scheme, r05 1.  `git show d737e95:switch_nerf_amd/csrc/chain_big.hip` has all three switches.)
#ifndef SWN_WQ_DEPTH
#define SWN_WQ_DEPTH 4            // weight-fragment register sets of a wave's K loop: SWN_WQ_DEPTH - 1 K steps in flight
#endif
constexpr int WQD = SWN_WQ_DEPTH, WQA = SWN_WQ_DEPTH - 1;
template <typename E, int NS = KSTEPS>
__device__ __forceinline__ void k_phase2(f32x16_t (&acc)[4][2], const Ctx& cx, __amdgpu_buffer_rsrc_t rs_cur, u32x4_t (&wq)[WQD][2]) {
  constexpr int MI = 4;           // NS = K steps of this layer (K / 16): 16, or 8 for a 128-feature chain input (geometries 6 / 7)
  char* smem = cx.smem;
  const int lane16 = cx.lane * 16;
  const int fg = cx.w & 3;
  u32x4_t fa[3][MI];
  uint32_t a_base = cx.a_base;
  asm volatile("" : "+v"(a_base));
  auto read_a = [&](int ks, int mi) {
    fa[ks % 3][mi] = *(const u32x4_t*)(smem + (a_base ^ (uint32_t)(ks << 5)) + mi * (32 * ROWB));
  };
  auto load_w = [&](int ks) {          // the two feature tiles of this wave, K step ks
#pragma unroll
    for (int i = 0; i < 2; ++i)
      wq[ks % WQD][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, lane16, ((2 * fg + i) * NS + ks) * 1024, 0);
  };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) read_a(0, mi);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) read_a(1, mi);
#pragma unroll
  for (int ks = 0; ks < NS; ++ks) {
    // weights of step ks: younger are the loads of steps ks + 1, ks + 2 (an older bias copy / mask load only makes the wait longer)
    {   // weights of step ks have landed: younger are the loads of the next min(WQA - 1, NS - 1 - ks) steps, two each
      const int younger = (NS - 1 - ks < WQA - 1 ? NS - 1 - ks : WQA - 1) * 2;
      if (younger >= 8) SWN_WAIT_VM(8); else if (younger == 6) SWN_WAIT_VM(6); else if (younger == 4) SWN_WAIT_VM(4);
      else if (younger == 2) SWN_WAIT_VM(2); else SWN_WAIT_VM(0);
    }
    if (ks + 1 < NS) SWN_WAIT_LGKM(4); else SWN_WAIT_LGKM(0);        // fragments of step ks (younger: those of step ks + 1)
    SWN_PIN();
#ifdef SWN_ABL_NOMFMA
#define SWN_MM(mi, ni) asm volatile("" :: "v"(wq[ks % WQD][ni]), "v"(fa[ks % 3][mi]))
#else
#define SWN_MM(mi, ni) acc[mi][ni] = E::mfma(wq[ks % WQD][ni], fa[ks % 3][mi], acc[mi][ni])
    SWN_MM(0, 0);
    SWN_PIN();
    if (ks + 2 < NS) { read_a(ks + 2, 0); read_a(ks + 2, 1); }
    SWN_PIN();
    SWN_MM(0, 1);
    SWN_PIN();
    if (ks + 2 < NS) { read_a(ks + 2, 2); read_a(ks + 2, 3); }
    SWN_PIN();
    SWN_MM(1, 0);
    SWN_PIN();
    if (ks + WQA < NS) load_w(ks + WQA);    // into the register set of step ks - 1, whose MFMAs were issued a step ago
    SWN_PIN();
    SWN_MM(1, 1);
    SWN_PIN();
    SWN_MM(2, 0);
    SWN_PIN();
    SWN_MM(2, 1);
    SWN_PIN();
    SWN_MM(3, 0);
    SWN_PIN();
    SWN_MM(3, 1);
#undef SWN_MM
    SWN_PIN();
  }
}
#endif

// The epilogue of geometry 4.  Same arithmetic as epilogue() above, value for value - but an epilogue wave of chainp_kernel runs ALONE
// beside its SIMD's MFMA wave: dependent VALU chains that two lockstep waves hide from each other (chainb) cost it ~8 clocks per
// instruction.  The 16 packed pairs of a row tile therefore move through the stages together (16 independent instructions per
// stage, order pinned), the mask accumulates in two registers, and the bias is read once per phase.
template <typename E, int RELU, bool BIAS, bool SKIP, typename HOOK>
__device__ __forceinline__ void epilogue_p(f32x16_t (&acc)[4][2], const Ctx& cx, u32x4_t& mk, int bias_off, HOOK hook) {
  typedef G256 G;
  char* smem = cx.smem;
  uint32_t e_base = cx.e_base, e2_base = cx.e2_base;
  asm volatile("" : "+v"(e_base), "+v"(e2_base));
  f32x4_t bias[2][4];
  if constexpr (BIAS) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        bias[ni][g4] = *(const f32x4_t*)(smem + G::BIAS0 + bias_off + (((cx.w & 3) * 64 + 32 * ni + 8 * g4 + 4 * cx.lhi) << 2));
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    uint32_t m[4] = {0u, 0u, 0u, 0u};
    const uint32_t mbits = (RELU == 2) ? mk[mi] : 0u;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {        // 8 packed pairs per stage: k = g4 * 2 + i, mask bit ni * 8 + k
      u32x2_t xv[4];
      if constexpr (SKIP) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) xv[g4] = *(const u32x2_t*)(smem + (e_base ^ (uint32_t)((4 * ni + g4) << 4)) + mi * (32 * ROWB));
      }
      f32x2_t_ v[8];
      uint32_t pp[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int g4 = k >> 1, i = k & 1;
        v[k] = f32x2_t_{acc[mi][ni][g4 * 4 + 2 * i], acc[mi][ni][g4 * 4 + 2 * i + 1]};
        if constexpr (BIAS) v[k] += f32x2_t_{bias[ni][g4][2 * i], bias[ni][g4][2 * i + 1]};
      }
      SWN_PIN();
      if constexpr (SKIP) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += f32x2_t_{E::lo(xv[k >> 1][k & 1]), E::hi(xv[k >> 1][k & 1])};
        SWN_PIN();
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) pp[k] = E::pack2(v[k][0], v[k][1]);
      SWN_PIN();
      if constexpr (RELU == 1) {
        uint32_t q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pp[k] = pk_relu16(pp[k]);
        SWN_PIN();
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = pk_nonzero16(pp[k]);
        SWN_PIN();
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k & 3] |= q[k] << (ni * 8 + k);
      } else if constexpr (RELU == 2) {
        uint32_t t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = mbits & (0x00010001u << (ni * 8 + k));
        SWN_PIN();
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = pk_nonzero16(t[k]);
        SWN_PIN();
#pragma unroll
        for (int k = 0; k < 8; ++k) pp[k] = pk_mul16(pp[k], t[k]);
      }
      SWN_PIN();                            // (SKIP: the residual values of these chunks are consumed before the chunks are rewritten)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        auto r0 = __builtin_amdgcn_permlane32_swap(pp[g * 2], pp[(g + 1) * 2], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(pp[g * 2 + 1], pp[(g + 1) * 2 + 1], false, false);
        const u32x4_t o = {(uint32_t)r0[0], (uint32_t)r1[0], (uint32_t)r0[1], (uint32_t)r1[1]};
        *(u32x4_t*)(smem + (e2_base ^ (uint32_t)((4 * ni + g) << 4)) + mi * (32 * ROWB)) = o;
      }
      SWN_PIN();
      if constexpr (hook_halves<HOOK>::value) hook(2 * mi + ni);
    }
    if constexpr (RELU == 1) mk[mi] = (m[0] | m[1]) | (m[2] | m[3]);
    if constexpr (!hook_halves<HOOK>::value) hook(mi);
  }
}

template <typename E, typename HOOK>
__device__ __forceinline__ void epilogue_p_dispatch(f32x16_t (&acc)[4][2], const Ctx& cx, u32x4_t& mk, int relu, bool bias, bool skip, int boff, HOOK hook) {
  if (relu == 1) {
    if (skip) { if (bias) epilogue_p<E, 1, true, true, HOOK>(acc, cx, mk, boff, hook); else epilogue_p<E, 1, false, true, HOOK>(acc, cx, mk, boff, hook); }
    else { if (bias) epilogue_p<E, 1, true, false, HOOK>(acc, cx, mk, boff, hook); else epilogue_p<E, 1, false, false, HOOK>(acc, cx, mk, boff, hook); }
  } else if (relu == 2) {
    if (skip) epilogue_p<E, 2, false, true, HOOK>(acc, cx, mk, boff, hook); else epilogue_p<E, 2, false, false, HOOK>(acc, cx, mk, boff, hook);
  } else {
    if (skip) { if (bias) epilogue_p<E, 0, true, true, HOOK>(acc, cx, mk, boff, hook); else epilogue_p<E, 0, false, true, HOOK>(acc, cx, mk, boff, hook); }
    else { if (bias) epilogue_p<E, 0, true, false, HOOK>(acc, cx, mk, boff, hook); else epilogue_p<E, 0, false, false, HOOK>(acc, cx, mk, boff, hook); }
  }
}

// 1 KiB pieces c0, c0 + stride, ... (n of them) of the (gathered) chain input -> the swizzled tile
__device__ __forceinline__ void stage_pieces(const Ctx& cx, const char* x, int c0, int n, int stride) {
  const int* idx = (const int*)(cx.smem + G256::IDX0);
#pragma unroll 4
  for (int j = 0; j < n; ++j) {
    const int c = c0 + j * stride;
    const int r = 2 * c + cx.lhi;
    const long src = idx[r];
    const int q = cx.l31 ^ (r & 15);
    __builtin_amdgcn_global_load_lds(SWN_GLB(x + src * ROWB + q * 16), SWN_LDS(cx.smem + c * 1024), 16, 0, 0);
  }
}

// pieces c0 + 4 j (j < 16) of the tile -> rows of a row-major tensor (descriptor clipped to the valid rows); batches of NB.  The store
// operands live in the CALLER's keep[16] (SWN_KEEP: allocated until the caller has waited for the stores)
template <typename E, bool ADD, int NB>
__device__ __forceinline__ void write_pieces16(const Ctx& cx, int c0, __amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t ra, u32x4_t (&keep)[16]) {
  const int lane16 = cx.lane * 16;
#pragma unroll
  for (int b = 0; b < 16 / NB; ++b) {
    u32x4_t* v = keep + NB * b;
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = *(const u32x4_t*)(cx.smem + piece_addr(cx, c0 + 4 * (NB * b + j)));
    if constexpr (ADD) {
      u32x4_t a[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, lane16, (c0 + 4 * (NB * b + j)) * 1024, 0);
      SWN_WAIT_VM(0);
      SWN_WAIT_LGKM0();
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = E::pack2(E::lo(v[j][q]) + E::lo(a[j][q]), E::hi(v[j][q]) + E::hi(a[j][q]));
    } else {
      SWN_WAIT_LGKM0();
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rs, lane16, (c0 + 4 * (NB * b + j)) * 1024, SWN_BIG_Y_AUX);
    SWN_PIN();
  }
}

template <typename E, int TAG>
__global__ __launch_bounds__(512, 2) void chainp_kernel(const Args args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef G256 G;
  constexpr int BM = G::BM, MI = 4;
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

  int g = blockIdx.x % d.n_groups;
  int tile = blockIdx.x / d.n_groups;
  if ((d.n_wsets & 7) == 0 && (d.n_groups & 7) == 0) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int s_ = q / args.tiles_per_group;
    tile = q - s_ * args.tiles_per_group;
    g = ((x + s_) & 7) + 8 * s_;     // rotated per 8 groups: every XCD meets every weight set - an expert that draws more rows than the
                                     // others (unbalanced routing) would otherwise make ITS XCD the long pole of the launch
  }
  int rows_valid = d.group_stride;
  if (d.group_rows) rows_valid = d.group_rows[g];
  if (rows_valid > d.group_rows_clamp) rows_valid = d.group_rows_clamp;
  const int row0 = tile * BM;
  if (row0 >= rows_valid) return;
  const int rows_in_tile = min(BM, rows_valid - row0);
  const long grow0 = (d.group_begin ? (long)d.group_begin[g] : (long)g * d.group_stride) + row0;
  const int wset = g % d.n_wsets;
  const int n_layers = d.n_layers;

  const int lrow = 32 * MI * rg + cx.l31;
  cx.a_base = (uint32_t)(lrow * ROWB + ((cx.lhi ^ r15) << 4));
  cx.e_base = (uint32_t)(lrow * ROWB + (r15 << 4) + 8 * cx.lhi) ^ (uint32_t)(fg << 7);
  cx.e2_base = (uint32_t)(lrow * ROWB + (r15 << 4)) ^ (uint32_t)((fg << 7) | (cx.lhi << 4));
  cx.wf_base = (uint32_t)(G::RING0 + (2 * fg) * 1024 + cx.lane * 16);
  const int lane16 = cx.lane * 16;

  auto wrs = [&](int L) -> __amdgpu_buffer_rsrc_t {
    const char* p = (const char*)d.layers[L].w + (size_t)wset * 8 * (KSTEPS * 1024);
    return uniform_rsrc(p, 8 * KSTEPS * 1024);
  };
  auto out_rs = [&](void* base) -> __amdgpu_buffer_rsrc_t {
    return uniform_rsrc((char*)base + grow0 * ROWB, rows_in_tile * ROWB);
  };
  SWN_TM(const long long t_start = TICK(); long long tk = 0, tkb = 0, te = 0, teb = 0, two = 0, tpro = 0;)
  int* gcount = (int*)(smem + G::BIAS0 + 3072);       // [2]: arrivals of the waves of a row group at the residual layer's meeting point

  // ---- prologue: this wave's weight fragments of the first three K steps, source rows, the rows of group 0 ----
  u32x4_t wq[WQD][2];
  auto preload_w = [&](int L) {
    const __amdgpu_buffer_rsrc_t r = wrs(L);
#pragma unroll
    for (int ks = 0; ks < WQA; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) wq[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(r, lane16, ((2 * fg + i) * KSTEPS + ks) * 1024, 0);
  };
  if (tid < BM) {
    const long gr = grow0 + (tid < rows_in_tile ? tid : 0);
    long src = d.x_gather ? (long)d.x_gather[gr] : gr;
    if (src < 0) src = 0;
    ((int*)(smem + G::IDX0))[tid] = (int)src;
  }
  if (tid < 2) gcount[tid] = 0;
  // bias of layer L lives in LDS slot L mod 3 from the K phase of layer L - 1 (group 0 copies it there) to the epilogues of layer L
  auto stage_bias = [&](int L) {
    const float* b = d.layers[L].b;
    if (b) {
      const __amdgpu_buffer_rsrc_t rb = uniform_rsrc(b + (size_t)wset * 256, 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, SWN_LDS(smem + G::BIAS0 + (L % 3) * 1024 + fg * 256), 4, cx.lane * 4, fg * 256, 0, 0);
    }
  };
  if (rg == 0) stage_bias(0);
  // geometry 4 starts the accumulators AT the bias (one packed add per pair less in the epilogue, which is the longer phase); geometry 5
  // adds it in the epilogue like every other kernel of the library: bit-identical to them (the test of the phase machinery)
  const bool bias_init = d.geometry == 4;
  SWN_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();
  stage_pieces(cx, (const char*)d.x, cx.w, 8, 8);                 // rows of group 0: pieces 0..63, all waves
  SWN_WAIT_VM(0);
  SWN_PIN();
  if (rg == 0) preload_w(0);
  SWN_PIN();
  __builtin_amdgcn_s_barrier();
  if (rg == 1) {                                                  // phase 0 of group 1: its own rows
    stage_pieces(cx, (const char*)d.x, 64 + fg, 16, 4);
    SWN_WAIT_VM(0);
    preload_w(0);
    __builtin_amdgcn_s_barrier();
  }

  f32x16_t acc[MI][2];
  int n_skip = 0;
  auto load_mask = [&](int L) -> u32x4_t {      // the stored mask of this wave for the epilogue of layer L (backward chains)
    const swn_chain_layer& l_ = d.layers[L];
    if (l_.relu == 2) return *(const u32x4_t*)(l_.mask + ((size_t)(blockIdx.x * G::NW + cx.w) * 64 + cx.lane) * 4);
    return u32x4_t{0u, 0u, 0u, 0u};
  };
  u32x4_t mk_next = load_mask(0);
  SWN_TM(tpro = TICK() - t_start;)
  for (int L = 0; L < n_layers; ++L) {
    const swn_chain_layer& ly = d.layers[L];
    u32x4_t mk = mk_next;
    // ---- K phase ----
    {
      const __amdgpu_buffer_rsrc_t rs_cur = wrs(L);
      // (group 0, for both groups) the next layer's bias -> its LDS slot; the K loop's counted waits cover the copy
      if (rg == 0 && L + 1 < n_layers) stage_bias(L + 1);
      if (bias_init && ly.b) {
        f32x4_t bv[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            bv[ni][g4] = *(const f32x4_t*)(smem + G::BIAS0 + (L % 3) * 1024 + ((fg * 64 + 32 * ni + 8 * g4 + 4 * cx.lhi) << 2));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = bv[ni][r >> 2][r & 3];
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
      }
      SWN_TM(const long long k0 = TICK();)
      k_phase2<E>(acc, cx, rs_cur, wq);
#pragma unroll
      for (int q_ = 0; q_ < WQD; ++q_)        // the ring's registers are dead from here to the preload at the end of the epilogue phase:
#pragma unroll
        for (int i_ = 0; i_ < 2; ++i_) asm volatile("" : "=v"(wq[q_][i_]));      // say so (no code), or they are kept alive through it
      SWN_TM(const long long k1 = TICK();)
      __builtin_amdgcn_s_barrier();     // the group is done reading its rows; the partner has rewritten its own and written ours out
      SWN_TM(const long long k2 = TICK(); tk += k1 - k0; tkb += k2 - k1;)
    }
    // ---- E phase ----
    {
      SWN_TM(const long long e0 = TICK();)
      // per-lane addresses of this phase are re-derived from a laundered lane id: kept alive through the K phase they would spill
      Ctx ce = cx;
      asm volatile("" : "+v"(ce.lane));
      ce.l31 = ce.lane & 31;
      ce.lhi = ce.lane >> 5;
      const int lane16e = ce.lane * 16;
      uint32_t* mkp = ly.mask ? ly.mask + ((size_t)(blockIdx.x * G::NW + cx.w) * 64 + ce.lane) * 4 : nullptr;
      if (L + 1 < n_layers) mk_next = load_mask(L + 1);      // a phase and a half ahead of its use: the next K loop's counted waits cover it
      const bool bias_epi = ly.b != nullptr && !bias_init;
      if (ly.skip) {                    // the residual input of this group -> its (dead) rows; the four waves stage 16 pieces each and meet
        stage_pieces(ce, (const char*)d.x, 64 * rg + fg, 16, 4);
        SWN_WAIT_VM(0);
        ++n_skip;
        if (ce.lane == 0) __hip_atomic_fetch_add(&gcount[rg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&gcount[rg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4 * n_skip)
          __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      // the partner's rows = the input tile of the K loop it is running: group 1's hold the output of layer L - 1, group 0's that of layer L.
      // Their write-out (16 pieces of 1 KiB per wave) is spread over the epilogue: 4 pieces after every row tile.
      void* wo = rg == 0 ? (L >= 1 ? d.layers[L - 1].save : nullptr) : (L + 1 < n_layers ? ly.save : nullptr);
      SWN_TM(two += TICK() - e0;)
      // the write-out's store operands - two alternating register sets: the set stored at row tile mi stays allocated (SWN_KEEP4) until
      // its refill a row tile later, the last two to the end of the phase (behind the mask store and the next K loop's preloads)
      u32x4_t wv[2][4];
      if (wo) {
        const __amdgpu_buffer_rsrc_t rs = out_rs(wo);
        const int c0 = 64 * (1 - rg) + fg;
        auto rd = [&](int b) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wv[b & 1][j] = *(const u32x4_t*)(smem + piece_addr(ce, c0 + 4 * (4 * b + j)));
        };
        rd(0);
        auto hook = [&](int mi) {      // (no explicit wait: the compiler counts the LDS operations behind the piece reads itself)
          if (mi > 0) SWN_KEEP4(wv[(mi + 1) & 1]);       // (the set stored a row tile ago: its next write is the refill below)
#pragma unroll
          for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(wv[mi & 1][j], rs, lane16e, (c0 + 4 * (4 * mi + j)) * 1024, SWN_BIG_STORE_AUX);
          if (mi < 3) rd(mi + 1);
          SWN_PIN();
        };
        epilogue_p_dispatch<E, decltype(hook)>(acc, ce, mk, ly.relu, bias_epi, ly.skip != 0, (L % 3) * 1024, hook);
      } else {
        epilogue_p_dispatch<E, NoHook>(acc, ce, mk, ly.relu, bias_epi, ly.skip != 0, (L % 3) * 1024, NoHook());
      }
      if (ly.relu == 1 && mkp) *(u32x4_t*)mkp = mk;
      SWN_PIN();                                    // (the loads must not be scheduled up into the epilogue: 24 registers)
      preload_w(L + 1 < n_layers ? L + 1 : L);     // the first three K steps of this wave's next K loop (end of chain: loaded, never used)
      SWN_PIN();
      if (wo) { SWN_KEEP4(wv[0]); SWN_KEEP4(wv[1]); }
      SWN_WAIT_LGKM0();
      SWN_TM(const long long e1 = TICK();)
      __builtin_amdgcn_s_barrier();
      SWN_TM(const long long e2 = TICK(); te += e1 - e0; teb += e2 - e1;)
    }
  }
  SWN_TM(const long long t_tail = TICK();)

  // ---- the chain output (+ y_add rows): group 0 writes its own rows while group 1 is in its last epilogue, then both write group 1's ----
  const __amdgpu_buffer_rsrc_t ry = out_rs(d.y);
  const __amdgpu_buffer_rsrc_t ra = d.y_add ? out_rs((void*)d.y_add) : ry;
  u32x4_t keep[16], keep2[8];                      // store operands, allocated until the stores have retired (SWN_KEEP)
  if (rg == 0) {
    if (d.y_add) write_pieces16<E, true, 8>(cx, fg, ry, ra, keep); else write_pieces16<E, false, 8>(cx, fg, ry, ra, keep);
    SWN_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
  }
  {
    // pieces 64 + w + 8 j, j < 8
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 64 + cx.w + 8 * j;
      u32x4_t v = *(const u32x4_t*)(smem + piece_addr(cx, c));
      if (d.y_add) {
        const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(ra, lane16, c * 1024, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = E::pack2(E::lo(v[q]) + E::lo(a[q]), E::hi(v[q]) + E::hi(a[q]));
      }
      keep2[j] = v;
      __builtin_amdgcn_raw_buffer_store_b128(keep2[j], ry, lane16, c * 1024, SWN_BIG_Y_AUX);
    }
  }
  SWN_WAIT_VM(0);                                  // (the tile's last stores have retired: their operands may go)
  if (rg == 0) SWN_KEEP16(keep);
  SWN_KEEP8(keep2);
#ifdef SWN_BIG_TIMING
  if (d.y_add_gather && (tid == 0 || tid == 256) && blockIdx.x < 2048) {     // wave 0 -> row blockIdx.x, wave 4 -> row 2048 + blockIdx.x
    long long* dbg = (long long*)d.y_add_gather + (long)(blockIdx.x + (tid ? 2048 : 0)) * 8;
    const long long t_end = TICK();
    dbg[0] = tk; dbg[1] = tkb; dbg[2] = te; dbg[3] = teb; dbg[4] = two; dbg[5] = tpro; dbg[6] = t_end - t_tail; dbg[7] = t_end - t_start;
  }
#endif
}


// =================================================================================================================================
// Geometry 6 / 7: the phase-shifted 256-row workgroup of geometry 4 / 5 made PERSISTENT.
//
// chainp_kernel pays a fixed price per 256-row tile around its 2 x n_layers phases: the prologue (source rows -> staging of both row
// groups' input, two barriers and an HBM round trip that nothing overlaps: one workgroup per CU), the fill / drain phase of the
// half-layer shift, and the tail (write-out of both row groups) - 14-23 k of ~110 k clocks on the 7-layer expert chain
// (profiles/r03_experiments.md section 9), and as much as the useful phases themselves on a 2-3 layer chain.  Here ONE workgroup per CU
// stays resident and walks a queue of tiles; the two row groups keep their half-layer offset ACROSS tiles:
//     phase (per tile, 2 n + 1 of them):   S    K0   E0   K1   ...  K(n-1)  E(n-1) | S'   K0' ...
//     row group 0                          S    K0   E0   K1        K(n-1)  E(n-1) | S'   K0'
//     row group 1 (one phase behind)     E(n-1)'' S   K0   E0  ...  E(n-2)  K(n-1) | E(n-1) S'
//   * S (stage) = a row group writes ITS output rows of the previous tile out (coalesced 1 KiB pieces) and brings the (gathered) input
//     rows of its next tile into the same LDS rows by LDS-DMA - while its partner is in a K or E phase on the same SIMDs: the HBM round
//     trip of the staging and the write-out are hidden behind the partner's work, there is no fill / drain phase any more;
//   * the tile queue: per XCD x (workgroups b = x mod 8) one atomic counter q -> virtual block 8 q + x of chainp_kernel's grid, i.e. the
//     SAME tile -> (group, rows) mapping (rotated expert per XCD), the same ReLU-mask layout - masks recorded by geometry 4 / 5 serve a
//     geometry 6 / 7 backward and vice versa - but tiles past a ragged group's end cost one atomic instead of a workgroup launch slot, and
//     the queue balances unequal groups by itself.  Wave 0 claims the tile AFTER next while the current one is staged and leaves its
//     (group, first row, valid rows, weight set) in LDS for both row groups; the source rows of the next tile are fetched during the
//     last epilogue phase.  The counters are left zeroed by the last workgroup (swn_chain_desc.sched: 16 ints, zero before the first
//     launch); without them the tiles are dealt round-robin (b, b + grid, ...);
//   * biases live in a 3-slot ring indexed by a running layer counter (the bias of the next tile's first layer arrives during the
//     last K phase of this one), weight fragments as in geometry 4 (register ring, preloaded at the end of the preceding phase).
// Arithmetic, rounding points and masks are chainp_kernel's: geometry 6 is bit-identical to geometry 5 (and to the 64-row kernels),
// geometry 7 to geometry 4 (accumulators start at the bias).
// =================================================================================================================================
struct ArgsQ {
  swn_chain_desc d;
  int tiles_per_group;
  int n_vb;            // virtual blocks = chainp_kernel's grid
  int n_queues;        // 8: one queue per XCD (rotated mapping), 1: a single queue
  int stagger;         // start offset between the workgroups of an XCD in units of ~1 k clocks (0: all start together)
  int per_queue;       // 8 queues over ONE group (dense chains): queue x walks the tiles [x * per_queue, (x + 1) * per_queue) - the
                       // workgroups of an XCD write consecutive rows (as the 64-row kernels' dense mapping does)
  int n_vb_e;          // fused tail (tag 7): virtual blocks [n_vb_e, n_vb) are the tiles of the dropped tokens (= n_vb otherwise)
};

constexpr int Q_IDX1 = G256::BIAS0 + 3072 + 64;      // second source-row table (int32 [256]); the first one is G256::IDX0
constexpr int Q_TINFO = Q_IDX1 + 1024;               // int32 [2][8]: vb (-1 = none), group, first tile row, valid rows, first row (lo, hi), weight set
constexpr int Q_YIDX = Q_TINFO + 64;                 // int32 [2][256]: rows of y_add for the output rows of a tile (y_add_gather), by tile parity
constexpr int Q_LDS = Q_YIDX + 2048;

// NARROW: the chain input has 128 features (256-byte rows) under a first layer whose weights are zero-padded to K = 256 (one K-loop
// instantiation for every layer: a second, 8-step one beside it cost 160 spilled registers): the first 16 chunk positions of a tile row
// receive the row - position p of row r holds chunk p ^ (r & 15) < 16, copied by the lanes that own them (an inactive lane of an
// LDS-DMA writes nothing) - and the other lanes ZERO the upper 16 positions (what is left there from the previous tile would meet zero
// weights, but 0 x inf is not 0)
template <bool NARROW>
__device__ __forceinline__ void stage_pieces_q(const Ctx& cx, const char* x, int c0, int n, int stride, int idx_off) {
  const int* idx = (const int*)(cx.smem + idx_off);
  constexpr int XRB = NARROW ? 256 : ROWB;
  // all source rows first (one LDS round trip), then the copies back to back: with the index read in front of every copy the 16 copies of
  // a wave took ~6 k clocks to ISSUE (16 dependent LDS round trips beside the partner group's LDS traffic)
  int src[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) src[j] = j < n ? idx[2 * (c0 + j * stride) + cx.lhi] : 0;
  if constexpr (NARROW) {       // the zero half FIRST: behind an LDS-DMA every LDS write waits for the copy to land (one HBM round trip each)
    if (cx.l31 >= 16) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < n) *(u32x4_t*)(cx.smem + (c0 + j * stride) * 1024 + cx.lane * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  SWN_WAIT_LGKM0();
  if (!NARROW || cx.l31 < 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < n) {
        const int c = c0 + j * stride;
        const int r = 2 * c + cx.lhi;
        const int q = cx.l31 ^ (r & 15);
        __builtin_amdgcn_global_load_lds(SWN_GLB(x + (long)src[j] * XRB + q * 16), SWN_LDS(cx.smem + c * 1024), 16, 0, 0);
      }
    }
  }
}

// write_pieces16 with the COMBINE BACKWARD applied on the way out (the tail backward chain; include/swn.h, comb_*; the arithmetic of
// chain_kernel's fused write-out, value for value and in its summation order): with z = the output row,
//   t = (z + dsig[row] * wsig) * (y[row] > 0);   out[row] = t * gate[row];   dgate[row] = <y[row], t> / gate[row]
// A half-wave holds one row (lane l31 = its 8 features 8 l31 ..): the row's dot product is 8 sequential fmas per lane, then the
// xor butterfly over the 32 lanes (16, 8, 4, 2, 1) like the 64-row kernel's 32 chunk lanes.
template <typename E>
__device__ __forceinline__ void write_pieces16_comb(const Ctx& cx, int c0, __amdgpu_buffer_rsrc_t rs, const swn_chain_desc& d, long grow0, int rows,
                                                    u32x4_t (&keep)[16]) {
  const int lane16 = cx.lane * 16;
  float wv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wv[e] = d.comb_wsig ? d.comb_wsig[cx.l31 * 8 + e] : 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    u32x4_t* v = keep + 8 * b;        // (store operands: the caller keeps them allocated until the stores have retired)
    u32x4_t yc[8];
    float gt[8], ds[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 2 * (c0 + 4 * (8 * b + j)) + cx.lhi;
      const long gr = grow0 + (r < rows ? r : 0);
      yc[j] = *(const u32x4_t*)((const char*)d.comb_y + gr * ROWB + cx.l31 * 16);
      gt[j] = d.comb_gate[gr];
      ds[j] = d.comb_dsig ? d.comb_dsig[gr] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const u32x4_t*)(cx.smem + piece_addr(cx, c0 + 4 * (8 * b + j)));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 2 * (c0 + 4 * (8 * b + j)) + cx.lhi;
      float z[8], yv[8], dot = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        z[2 * q] = E::lo(v[j][q]); z[2 * q + 1] = E::hi(v[j][q]);
        yv[2 * q] = E::lo(yc[j][q]); yv[2 * q + 1] = E::hi(yc[j][q]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = z[e] + ds[j] * wv[e];          // (one fma, like combine_bwd_kernel)
        t = yv[e] > 0.f ? t : 0.f;
        dot += yv[e] * t;
        z[e] = t * gt[j];
      }
      for (int o = 16; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
      if (cx.l31 == 0 && r < rows) d.comb_dgate[grow0 + r] = dot / gt[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[j][q] = E::pack2(z[2 * q], z[2 * q + 1]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rs, lane16, (c0 + 4 * (8 * b + j)) * 1024, SWN_BIG_Y_AUX);
    SWN_PIN();
  }
}

// write_pieces16 with the rows of y_add fetched through an index (the front backward chain adds the expert path's input gradient through
// tok2row: -1 = nothing to add).  The indices of the tile's rows were put into an LDS table during its last epilogue phase (yidx_off), so
// the 16 row pieces of a wave are requested together: ONE global round trip in the staging phase (index and row fetched back to back
// in two batches of 8 were four: the front backward chain's S phase was twice as long as its other phases)
template <typename E>
__device__ __forceinline__ void write_pieces16_gather(const Ctx& cx, int c0, __amdgpu_buffer_rsrc_t rs, const char* y_add, int yidx_off,
                                                      u32x4_t (&keep)[16]) {
  const int lane16 = cx.lane * 16;
  const int* yidx = (const int*)(cx.smem + yidx_off);
  int ar[16];
  u32x4_t a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) ar[j] = yidx[2 * (c0 + 4 * j) + cx.lhi];
  SWN_WAIT_LGKM0();
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = *(const u32x4_t*)(y_add + (long)(ar[j] < 0 ? 0 : ar[j]) * ROWB + cx.l31 * 16);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    u32x4_t* v = keep + 8 * b;        // (store operands: the caller keeps them allocated until the stores have retired)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const u32x4_t*)(cx.smem + piece_addr(cx, c0 + 4 * (8 * b + j)));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ar[8 * b + j] >= 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = E::pack2(E::lo(v[j][q]) + E::lo(a[8 * b + j][q]), E::hi(v[j][q]) + E::hi(a[8 * b + j][q]));
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rs, lane16, (c0 + 4 * (8 * b + j)) * 1024, SWN_BIG_Y_AUX);
    SWN_PIN();
  }
}

// ---- the dense tail folded into the expert forward chain (TAG 7; include/swn.h, tail_first) ------------------------------------------
// LDS tables of that instantiation, in the 24 KiB that chainb_kernel's weight ring occupies (chainq_kernel streams its weights
// straight into registers):
constexpr int T_GATE = G256::RING0;        // f32 [256]: gate value of every tile row
constexpr int T_WS = T_GATE + 1024;        // f32 [256]: sigma head weights
constexpr int T_SIGP = T_WS + 1024;        // f32 [8][256]: partial sums of the sigma head, [wave quarter * 2 + half-wave][tile row]
constexpr int T_COL = T_SIGP + 8192;       // f32 [3][256]: the colour head's dot products of every tile row
constexpr int T_WC = T_COL + 3072;         // f32 [3][128]: colour head weights
constexpr int T_HB = T_WC + 1536;          // f32 [4]: the heads' biases (colour 0..2, sigma)
static_assert(T_HB + 16 <= G256::IDX0, "tail tables must fit the ring region");
constexpr int T_DSIG = T_COL;              // (tag 8) f32 [256]: the sigma head's gradient of every tile row

// Epilogue of the LAST EXPERT layer of a fused chain: the row becomes relu(gate[row] * z) - z rounded to the 16-bit type first, the
// product rounded again: GatingDecoder's fp32 multiply on the 16-bit expert output, cast back, act relu (tutel_fast_dispatch.py:119-127,
// nerf_moe.py:385; value for value the x_scale / x_relu staging of the 64-row tail chain) - and, with the heads, leaves the lane's share
// of the sigma head's dot product <row, w_sigma> in T_SIGP (a lane holds 32 features of each of its 4 rows).
template <typename E, typename HOOK>
__device__ __forceinline__ void epilogue_q_gate(f32x16_t (&acc)[4][2], const Ctx& cx, bool heads, HOOK hook) {
  char* smem = cx.smem;
  uint32_t e2_base = cx.e2_base;
  asm volatile("" : "+v"(e2_base));
  const int fg = cx.w & 3;
  const int row0 = 128 * (cx.w >> 2) + cx.l31;
  float gt[4], sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) gt[mi] = *(const float*)(smem + T_GATE + (row0 + 32 * mi) * 4);
  int h = 0;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    f32x4_t ws[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) ws[g4] = *(const f32x4_t*)(smem + T_WS + ((fg * 64 + 32 * ni + 8 * g4 + 4 * cx.lhi) << 2));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      uint32_t pp[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pp[k] = E::pack2(acc[mi][ni][(k >> 1) * 4 + 2 * (k & 1)], acc[mi][ni][(k >> 1) * 4 + 2 * (k & 1) + 1]);
      SWN_PIN();
#pragma unroll
      for (int k = 0; k < 8; ++k) pp[k] = pk_relu16(E::pack2(E::lo(pp[k]) * gt[mi], E::hi(pp[k]) * gt[mi]));
      SWN_PIN();
      if (heads) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          sg[mi] = __builtin_fmaf(E::lo(pp[k]), ws[k >> 1][2 * (k & 1)], sg[mi]);
          sg[mi] = __builtin_fmaf(E::hi(pp[k]), ws[k >> 1][2 * (k & 1) + 1], sg[mi]);
        }
        asm volatile("" : "+v"(sg[mi]));     // (computed here, not sunk to the store at the end: see epilogue_q_comb)
      }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        auto r0 = __builtin_amdgcn_permlane32_swap(pp[g * 2], pp[(g + 1) * 2], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(pp[g * 2 + 1], pp[(g + 1) * 2 + 1], false, false);
        const u32x4_t o = {(uint32_t)r0[0], (uint32_t)r1[0], (uint32_t)r0[1], (uint32_t)r1[1]};
        *(u32x4_t*)(smem + (e2_base ^ (uint32_t)((4 * ni + g) << 4)) + mi * (32 * ROWB)) = o;
      }
      SWN_PIN();
      if constexpr (hook_halves<HOOK>::value) hook(h);
      else if (h & 1) hook(h >> 1);
      ++h;
    }
  }
  if (heads) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) *(float*)(smem + T_SIGP + (((fg * 2 + cx.lhi) * 256 + row0 + 32 * mi) << 2)) = sg[mi];
  }
}

// The COMBINE BACKWARD of a fused backward chain (tag 8), in place on a row group's rows of the LDS tile behind the last head layer
// (include/swn.h comb_*; write_pieces16_comb's arithmetic value for value and in its summation order - the result, the gate gradient
// included, is bit-identical to the separate launches): with z = the row (the gradient of the decoded, gate-scaled, ReLU'd expert output y),
//   t = (z + dsig[row] * wsig) * (y[row] > 0);   row <- t * gate[row];   dgate[token] = <y[row], t> / gate[row]
// A wave takes the pieces it stages (fg + 4 j: tile rows 2 c + lhi); a half-wave holds one row, lane l31 its 8 features 8 l31 ..: y is
// read from memory in whole 512-byte rows through the rows' tokens.  (A first version applied the combine in the accumulator layout of
// the layer's epilogue and read y in 8-byte pieces - 32 rows per load instruction: 0.33 ms of the launch, profiles/r04_experiments.md 19.)
// dws_part != NULL: the wave also leaves the sum over ITS 32 rows of dsig[row] * y[row][:] (the sigma head's weight gradient,
// include/swn.h comb_dwsig) in dws_part[256] - a lane adds its 8 features over its 16 rows in loop order, then the two half-waves'
// sums: the same bits whoever runs the tile.
template <typename E>
__device__ __forceinline__ void comb_pieces16_inplace(const Ctx& cx, int c0, const swn_chain_desc& d, int idx_off, int rows, float* dws_part) {
  char* smem = cx.smem;
  const int* idx = (const int*)(smem + idx_off);
  float aw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float wv[8];
  {
    const f32x4_t w0 = *(const f32x4_t*)(smem + T_WS + cx.l31 * 32), w1 = *(const f32x4_t*)(smem + T_WS + cx.l31 * 32 + 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) { wv[e] = w0[e]; wv[4 + e] = w1[e]; }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    u32x4_t v[8], yc[8];
    float gt[8], ds[8];
    long tok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 2 * (c0 + 4 * (8 * b + j)) + cx.lhi;
      tok[j] = idx[r];
      gt[j] = *(const float*)(smem + T_GATE + r * 4);
      ds[j] = *(const float*)(smem + T_DSIG + r * 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) yc[j] = *(const u32x4_t*)((const char*)d.comb_y + tok[j] * ROWB + cx.l31 * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const u32x4_t*)(smem + piece_addr(cx, c0 + 4 * (8 * b + j)));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 2 * (c0 + 4 * (8 * b + j)) + cx.lhi;
      float z[8], yv[8], dot = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        z[2 * q] = E::lo(v[j][q]); z[2 * q + 1] = E::hi(v[j][q]);
        yv[2 * q] = E::lo(yc[j][q]); yv[2 * q + 1] = E::hi(yc[j][q]);
      }
      if (dws_part) {                            // (rows past the tile's end repeat its first row: not counted)
        const float dsm = r < rows ? ds[j] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) aw[e] = __builtin_fmaf(dsm, yv[e], aw[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = z[e] + ds[j] * wv[e];          // (one fma, like combine_bwd_kernel)
        t = yv[e] > 0.f ? t : 0.f;
        dot += yv[e] * t;
        z[e] = t * gt[j];
      }
      for (int o = 16; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
      if (cx.l31 == 0 && r < rows) d.comb_dgate[tok[j]] = dot / gt[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[j][q] = E::pack2(z[2 * q], z[2 * q + 1]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) *(u32x4_t*)(smem + piece_addr(cx, c0 + 4 * (8 * b + j))) = v[j];
    SWN_PIN();
  }
  if (dws_part) {
#pragma unroll
    for (int e = 0; e < 8; ++e) aw[e] += __shfl_xor(aw[e], 32);
    if (cx.lhi == 0) {
      *(f32x4_t*)(dws_part + cx.l31 * 8) = f32x4_t{aw[0], aw[1], aw[2], aw[3]};
      *(f32x4_t*)(dws_part + cx.l31 * 8 + 4) = f32x4_t{aw[4], aw[5], aw[6], aw[7]};
    }
  }
}

// Epilogue of the LAST layer of a fused chain (Linear "2" over cat([h, PE(dir), embedding_a]), nerf_moe.py:419-429): the per-ray half
// of the layer arrives as a row bias (fp32 [rays][nf], the row of token / rpb), then ReLU.  nf = 128: the wave quarters beyond the
// layer's width (zero-padded weights) only run the hooks.
template <typename E, typename HOOK>
__device__ __forceinline__ void epilogue_q_rowbias(f32x16_t (&acc)[4][2], const Ctx& cx, const float* rowbias, int rpb, const int32_t* bias_row,
                                                   int nf, int idx_off, HOOK hook) {
  char* smem = cx.smem;
  const int fg = cx.w & 3;
  if (fg * 64 >= nf) {
    if constexpr (hook_halves<HOOK>::value) {
#pragma unroll
      for (int h = 0; h < 8; ++h) hook(h);
    } else {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) hook(mi);
    }
    return;
  }
  uint32_t e2_base = cx.e2_base;
  asm volatile("" : "+v"(e2_base));
  const int row0 = 128 * (cx.w >> 2) + cx.l31;
  const float* rb[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const uint32_t tok = (uint32_t)((const int*)(smem + idx_off))[row0 + 32 * mi];
    // (bias_row: swn_chain_desc.tail_bias_row - a token space that is not in ray order, ep_owner.py)
    const uint32_t br = bias_row ? (uint32_t)bias_row[tok] : tok / (uint32_t)rpb;
    rb[mi] = rowbias ? rowbias + (size_t)br * nf + fg * 64 + 4 * cx.lhi : nullptr;
  }
  f32x4_t bq[2][4];
  auto fetch = [&](int t) {                 // the bias chunks of half step t = 4 ni + mi, one half step ahead of their use
    const int ni = t >> 2, mi = t & 3;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#ifdef SWN_ABL_NORB     // (experiment: what the per-ray bias gather of the last layer's epilogue costs)
      bq[t & 1][g4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#else
      bq[t & 1][g4] = rb[mi] ? *(const f32x4_t*)(rb[mi] + 32 * ni + 8 * g4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#endif
  };
  fetch(0);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int t = 4 * ni + mi;
      if (t + 1 < 8) fetch(t + 1);
      uint32_t pp[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int g4 = k >> 1, i = k & 1;
        pp[k] = pk_relu16(E::pack2(acc[mi][ni][g4 * 4 + 2 * i] + bq[t & 1][g4][2 * i], acc[mi][ni][g4 * 4 + 2 * i + 1] + bq[t & 1][g4][2 * i + 1]));
      }
      SWN_PIN();
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        auto r0 = __builtin_amdgcn_permlane32_swap(pp[g * 2], pp[(g + 1) * 2], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(pp[g * 2 + 1], pp[(g + 1) * 2 + 1], false, false);
        const u32x4_t o = {(uint32_t)r0[0], (uint32_t)r1[0], (uint32_t)r0[1], (uint32_t)r1[1]};
        *(u32x4_t*)(smem + (e2_base ^ (uint32_t)((4 * ni + g) << 4)) + mi * (32 * ROWB)) = o;
      }
      SWN_PIN();
      if constexpr (hook_halves<HOOK>::value) hook(t);
      else if (t & 1) hook(t >> 1);
    }
  }
}

// The rows of a finished tile whose last layer is 128 features wide -> y[token] (256-byte rows), with the colour head on the way.  A wave
// handles the rows it STAGES (pieces fg + 4 j of its row group: rows 8 j + 2 fg, + 1 - nobody else's copy may land on a row that is
// still to be read), four at a time (lane -> row 8 (2 i + lane / 32) + 2 fg + (lane / 16 & 1), 16-byte chunk lane % 16: every lane
// carries data, 8 stores per wave instead of the 16 half-empty ones of the 512-byte pieces); a lane multiplies its 8 features by the
// three colour rows, a butterfly over the row's 16 lanes (quad swaps, half mirror, mirror) gives <h2, w_c> -> T_COL[c][row].
template <typename E, typename PRE>
__device__ __forceinline__ void write_rows128_tok(const Ctx& cx, void* y, uint32_t oob, const float* wc, int idx_off, int rows, bool heads, PRE pre,
                                                  u32x4_t (&keep)[16]) {
  char* smem = cx.smem;
  const int* idx = (const int*)(smem + idx_off);
  const int l15 = cx.lane & 15, rq = cx.lane >> 4;
  const int rbase = 128 * (cx.w >> 2) + 2 * (cx.w & 3) + 8 * (rq >> 1) + (rq & 1);      // row of step i: rbase + 16 i
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(y ? y : (void*)wc, y ? (int)oob : 0);
  f32x4_t w[3][2];
  if (heads) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int q = 0; q < 2; ++q) w[c][q] = *(const f32x4_t*)(smem + T_WC + ((c * 128 + l15 * 8 + 4 * q) << 2));
  }
  u32x4_t* v = keep;                // (8 store operands: the caller keeps them allocated until the stores have retired)
  int tk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = rbase + 16 * i;
    v[i] = *(const u32x4_t*)(smem + r * ROWB + ((l15 ^ (r & 15)) << 4));
    tk[i] = idx[r];
  }
  SWN_WAIT_LGKM0();
  pre();                 // (the caller's loads go in front of the stores: one in-order counter for both)
  // (no y: a descriptor of 0 bytes drops the stores - unconditional, so that the wait for the caller's load counts them on every path)
#pragma unroll
  for (int i = 0; i < 8; ++i)
    __builtin_amdgcn_raw_buffer_store_b128(v[i], rs, rbase + 16 * i < rows ? (uint32_t)tk[i] * 256u + (uint32_t)(l15 * 16) : oob, 0, SWN_BIG_Y_AUX);
  if (heads) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float f[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { f[2 * q] = E::lo(v[i][q]); f[2 * q + 1] = E::hi(v[i][q]); }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a = __builtin_fmaf(f[e], w[c][e >> 2][e & 3], a);
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x141, 0xF, 0xF, true));     // row_half_mirror
        a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x140, 0xF, 0xF, true));     // row_mirror
        if (l15 == c) ((float*)(smem + T_COL))[c * 256 + rbase + 16 * i] = a;
      }
    }
  }
}

// The lane id, re-derived where it is used (two VALU instructions, nothing to keep alive): a lane id that lives across the phases of
// chainq_kernel is SPILLED, and its reloads sat in front of the K loop with a full vmcnt(0) wait each (four per K phase of row group 0)
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <typename E, int TAG, bool BIAS_INIT>
__global__ __launch_bounds__(512, 2) void chainq_kernel(const ArgsQ args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef G256 G;
  constexpr int BM = G::BM, MI = 4;
  constexpr bool TAIL = TAG == 7;               // the dense tail folded into the expert forward chain (include/swn.h, tail_first)
  constexpr bool HEAD = TAG == 8;               // the tail's backward layers in front of the expert backward chain (head_layers)
  constexpr bool DROPS = TAIL || HEAD;          // tiles of dropped tokens behind the experts' tiles
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
  const int n_layers = d.n_layers;

  // The per-lane LDS bases of a phase (fragment rows of the K loop, epilogue rows) are derived INSIDE the phase from a laundered lane
  // id: nothing but the lane id itself lives across the phases (the K phase holds 208 registers of accumulators and fragment rings).
  cx.a_base = cx.e_base = cx.e2_base = cx.wf_base = 0;
  auto phase_ctx = [&](bool k_phase) -> Ctx {
    Ctx c = cx;
    c.lane = fresh_lane();
    asm volatile("" : "+s"(c.w));
    c.l31 = c.lane & 31;
    c.lhi = c.lane >> 5;
    const int r15_ = c.lane & 15, rg_ = c.w >> 2, fg_ = c.w & 3;
    const int lrow = 32 * MI * rg_ + c.l31;
    if (k_phase) {
      c.a_base = (uint32_t)(lrow * ROWB + ((c.lhi ^ r15_) << 4));
    } else {
      c.e_base = (uint32_t)(lrow * ROWB + (r15_ << 4) + 8 * c.lhi) ^ (uint32_t)(fg_ << 7);
      c.e2_base = (uint32_t)(lrow * ROWB + (r15_ << 4)) ^ (uint32_t)((fg_ << 7) | (c.lhi << 4));
    }
    return c;
  };
  int* gcount = (int*)(smem + G::BIAS0 + 3072);
  const int* tinfo = (const int*)(smem + Q_TINFO);      // (every reader sits behind a barrier that follows an asm memory clobber)

  // ---- the tile queue (wave 0 only) ----
  const int xq = args.n_queues == 8 ? (int)(blockIdx.x & 7) : 0;
  int kq = 0;                                    // (no counters: the kq-th tile of this workgroup is block b + kq * grid)
  // Work stealing between the eight per-XCD queues: a workgroup whose own queue is exhausted moves on to its neighbours' (sq = how many
  // queues it has left behind).  The static group -> XCD map balances over MANY segments (XCD x meets expert (x + s) & 7 in segment s);
  // with the two segments of a 1024-ray share the busiest XCD held 128 tiles, the idlest 3 - the launch ran four rounds where three
  // tile the chip (profiles/r06_experiments.md 5).  Stolen tiles read their expert's weights through a cold L2: last round only.
  // (sq lives in LDS - gcount[2], wave 0 only - not in a register that every wave would carry through all its phases)
  auto cur_sq = [&]() -> int { return __builtin_amdgcn_readfirstlane(*(volatile int*)(gcount + 2)); };
  auto claim = [&]() -> int {                    // one ticket of the queue this workgroup works on (lane 0; NOT waited for: consume with grab)
    int q = 0;
    if (d.sched) {
      const int sq = cur_sq();
      if (fresh_lane() == 0) q = __hip_atomic_fetch_add(d.sched + ((xq + sq) & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return q;
  };
  auto next_queue = [&]() -> bool {              // the current queue is empty: mark it (sched[9]: one bit per queue) and move on to the next
    int m = 0;                                   // queue nobody has marked yet - one round trip, not one claim per dead queue; false: none left
    int sq = cur_sq();
    const int bit = 1 << ((xq + sq) & 7);
    if (fresh_lane() == 0) m = __hip_atomic_fetch_or(d.sched + 9, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | bit;
    m = __builtin_amdgcn_readfirstlane(m);
    do { ++sq; } while (sq < 8 && ((m >> ((xq + sq) & 7)) & 1));
    if (fresh_lane() == 0) *(volatile int*)(gcount + 2) = sq;
    return sq < 8;
  };
  auto grab = [&](int slot, int ticket) {        // the next tile with at least one valid row -> tinfo[slot] (vb = -1: the queue is empty);
                                                 // `ticket`: a claim issued earlier (its round trip hidden behind other work)
    int vb, g = 0, tile = 0, rows_valid = 0;
    bool first = true;
    int n_groups = d.n_groups, tpg = args.tiles_per_group, n_wsets = d.n_wsets;
    asm volatile("" : "+s"(n_groups), "+s"(tpg), "+s"(n_wsets));      // (divisors re-read here: their reciprocals must not live - in
                                                                      //  vector registers - through the phases of the tile loop)
    for (;;) {
      int q;
      if (d.sched) {
        q = first ? ticket : claim();
        first = false;
        q = __builtin_amdgcn_readfirstlane(q);
        const int xs = (xq + cur_sq()) & 7;      // the queue the ticket came from
        if (args.n_queues != 8) vb = q;
        else if (args.per_queue == 0) vb = q * 8 + xs;
        else vb = q < args.per_queue ? xs * args.per_queue + q : args.n_vb;
        if (args.n_queues == 8 && vb >= args.n_vb) {      // this queue is empty: the next live one (its tickets are claimed afresh)
          if (next_queue()) continue;
          vb = -1;
          break;
        }
      } else {
        vb = (int)blockIdx.x + kq * (int)gridDim.x;
        ++kq;
      }
      if (vb >= args.n_vb) { vb = -1; break; }
      if constexpr (DROPS) {
        if (vb >= args.n_vb_e) {                 // a tile of dropped tokens (forward: they enter at the first shared layer as zero rows;
                                                 // backward: they run the shared head layers only)
          rows_valid = min(*d.tail_n_dropped, d.tail_dropped_max);
          tile = vb - args.n_vb_e;
          g = -1;
          if (tile * BM >= rows_valid) {         // (a queue hands its tiles out in order: everything behind this one is empty too)
            if (args.n_queues == 8 && d.sched && next_queue()) continue;
            vb = -1;
          }
          break;
        }
      }
      if ((n_wsets & 7) == 0 && (n_groups & 7) == 0) {           // chainp_kernel's mapping: XCD x meets every weight set in turn
        const int x = vb & 7, qq = vb >> 3;
        const int s_ = qq / tpg;
        tile = qq - s_ * tpg;
        g = ((x + s_) & 7) + 8 * s_;
      } else {
        tile = vb / n_groups;
        g = vb - tile * n_groups;
      }
      rows_valid = d.group_stride;
      if (d.group_rows) rows_valid = d.group_rows[g];
      if (rows_valid > d.group_rows_clamp) rows_valid = d.group_rows_clamp;
      if (tile * BM < rows_valid) break;
    }
    if (fresh_lane() == 0) {
      int* t = (int*)(smem + Q_TINFO) + slot * 8;
      t[0] = vb;
      if (vb >= 0) {
        long grow0 = (long)tile * BM;
        if (g >= 0) grow0 += d.group_begin ? (long)d.group_begin[g] : (long)g * d.group_stride;
        t[1] = g;
        t[2] = min(BM, rows_valid - tile * BM);
        t[3] = (int)(uint32_t)(grow0 & 0xFFFFFFFFl);
        t[4] = (int)(grow0 >> 32);
        t[5] = g >= 0 ? g % n_wsets : 0;
        if constexpr (TAIL) t[6] = g >= 0 ? 0 : d.tail_first;      // first layer of the tile
        if constexpr (HEAD) t[7] = g >= 0 ? d.n_layers : d.head_layers;      // ... one past its last layer
      }
    }
  };
  struct Tile { int vb, rows; long grow0; int wset, l0, l1; };
  auto read_tile = [&](int slot) -> Tile {
    Tile t;
    t.vb = __builtin_amdgcn_readfirstlane(tinfo[slot * 8 + 0]);
    t.rows = __builtin_amdgcn_readfirstlane(tinfo[slot * 8 + 2]);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane(tinfo[slot * 8 + 3]);
    const int hi = __builtin_amdgcn_readfirstlane(tinfo[slot * 8 + 4]);
    t.grow0 = ((long)hi << 32) | (long)lo;
    t.wset = __builtin_amdgcn_readfirstlane(tinfo[slot * 8 + 5]);
    t.l0 = 0;
    if constexpr (TAIL) t.l0 = __builtin_amdgcn_readfirstlane(tinfo[slot * 8 + 6]);
    t.l1 = n_layers;
    if constexpr (HEAD) t.l1 = __builtin_amdgcn_readfirstlane(tinfo[slot * 8 + 7]);
    return t;
  };
  auto finish = [&]() {                          // the last workgroup to leave zeroes the counters for the next launch
    if (d.sched && cx.w == 0 && fresh_lane() == 0) {
      const int done = __hip_atomic_fetch_add(d.sched + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done == (int)gridDim.x - 1) {
#pragma unroll
        for (int i = 0; i < 10; ++i) __hip_atomic_store(d.sched + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  // source rows of the tile's rows of THIS row group (threads 0..127 of the row group: one row each): load, and - later - store into
  // the idx table `idx_off` (two steps: the gather-index load of the NEXT tile is issued at the top of the last epilogue phase and
  // looked at behind it)
  auto load_row = [&](const Tile& t) -> int {
    const int lt = fg * 64 + fresh_lane();
    int src = 0;
    if (lt < 128) {
      const int r = 128 * rg + lt;
      const long gr = t.grow0 + (r < t.rows ? r : 0);          // rows past the end repeat the first row (computed, never stored)
      if ((TAIL && t.l0) || (HEAD && t.l1 < n_layers)) src = d.tail_dropped[gr];
      else src = d.x_gather ? d.x_gather[gr] : (int)gr;
    }
    return src;
  };
  auto store_row = [&](int src, int idx_off) {
    const int lt = fg * 64 + fresh_lane();
    if (lt < 128) ((int*)(smem + idx_off))[128 * rg + lt] = src < 0 ? 0 : src;
  };
  const bool narrow = d.x_features == 128;                  // 128-feature chain input (256-byte rows) under a K = 256 zero-padded first layer
  auto wrs = [&](int L, int wset) -> __amdgpu_buffer_rsrc_t {
    const int bytes = 8 * (d.layers[L].k >> 4) * 1024;      // 8 feature tiles x K / 16 steps of 1 KiB
    const char* p = (const char*)d.layers[L].w + (size_t)wset * bytes;
    return uniform_rsrc(p, bytes);
  };
  auto ws_of = [&](int L, const Tile& t) -> int {      // (shared layers: one weight set)
    return ((TAIL && L >= d.tail_first) || (HEAD && L < d.head_layers)) ? 0 : t.wset;
  };
  auto out_rs = [&](void* base, const Tile& t) -> __amdgpu_buffer_rsrc_t {
    return uniform_rsrc((char*)base + t.grow0 * ROWB, t.rows * ROWB);
  };
  u32x4_t wq[WQD][2];
  auto preload_w = [&](int L, int wset, int lane_) {
    const __amdgpu_buffer_rsrc_t r = wrs(L, wset);
#pragma unroll
    for (int ks = 0; ks < WQA; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) wq[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(r, lane_ * 16, ((2 * fg + i) * (d.layers[L].k >> 4) + ks) * 1024, 0);
  };
  auto stage_bias = [&](int L, int wset, int slot) {     // (the four waves of row group 0: 256 B each)
    const float* b = d.layers[L].b;
    if (b) {
      const __amdgpu_buffer_rsrc_t rb = uniform_rsrc(b + (size_t)wset * 256, 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, SWN_LDS(smem + G::BIAS0 + slot * 1024 + fg * 256), 4, fresh_lane() * 4, fg * 256, 0, 0);
    } else if constexpr (BIAS_INIT) {                    // the accumulators ALWAYS start at the slot's contents: no bias = zeros
      *(float*)(smem + G::BIAS0 + slot * 1024 + fg * 256 + fresh_lane() * 4) = 0.f;
    }
  };
  auto load_mask = [&](int L, int vb) -> u32x4_t {
    const swn_chain_layer& l_ = d.layers[L];
    if (l_.relu == 2) return *(const u32x4_t*)(l_.mask + ((size_t)(vb * G::NW + cx.w) * 64 + fresh_lane()) * 4);
    return u32x4_t{0u, 0u, 0u, 0u};
  };
  constexpr bool bias_init = BIAS_INIT;
  f32x4_t bv[2][4];                              // the bias values of this wave's next K phase (BIAS_INIT: the accumulators start there)
  auto load_bv = [&](int slot) {
    if constexpr (BIAS_INIT) {
      const int lh_ = fresh_lane() >> 5;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          bv[ni][g4] = *(const f32x4_t*)(smem + G::BIAS0 + slot * 1024 + ((fg * 64 + 32 * ni + 8 * g4 + 4 * lh_) << 2));
    }
  };

  // Workgroups that start together on equal tiles reach their S phases together - every CU bursts its 64 KiB out and 64 KiB in at the
  // same moment and the phase is as long as the whole chip's burst takes through HBM (measured: 7-9 k clocks).  A start offset
  // per workgroup spreads the S phases of the CUs over the tile period.
  for (int i = ((int)(blockIdx.x >> 3) & 15) * args.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(16);
  // ---- fused tail: what a row group does with the rows of a finished tile, and how a tile of dropped tokens is staged ----
  const int yf = TAIL ? (d.y_features ? d.y_features : 256) : 256;           // real width of the last layer
  const uint32_t oob_y = (uint32_t)d.tail_tokens * (uint32_t)(yf * 2), oob_s = (uint32_t)d.tail_tokens * (uint32_t)ROWB;
  auto tail_out = [&](const Ctx& c_, const Tile& t, int idx_off, u32x4_t (&keep)[16]) {
    // the last layer's rows -> y[token] (128 features), and the heads: raw[token] = (sigmoid(colour sums + b), softplus(sigma sum + b - 1));
    // a wave finishes the 32 rows it wrote out itself (lanes 0 .. 31: one row each, rows 8 j + 2 fg + {0, 1}) - no one else's LDS
    // writes are involved
    const bool hd = d.heads_raw != nullptr;
    // (the finalizing lane's token and its sigma noise are fetched BEFORE the write-out's stores: loads and stores retire through one
    //  in-order counter on this part - a load issued behind the stores is waited for with all of them, the round trip of the tile's
    //  write-out in front of the staging; issued first it costs a wait that leaves the 8 stores in flight)
    const int r = 128 * (c_.w >> 2) + 8 * (c_.l31 >> 1) + 2 * (c_.w & 3) + (c_.l31 & 1);
    const bool fin = hd && c_.lhi == 0 && r < t.rows;
    const long tok = fin ? ((const int*)(smem + idx_off))[r] : 0;
    float nz = 0.f;
    write_rows128_tok<E>(c_, d.y, oob_y, d.heads_wc, idx_off, t.rows, hd, [&]() { if (hd && d.heads_noise) nz = d.heads_noise[tok]; }, keep);
    if (hd) {
      SWN_WAIT_LGKM0();
      if (fin) {
        const float* sp = (const float*)(smem + T_SIGP) + r;
        const float* cp = (const float*)(smem + T_COL) + r;
        const float sg = ((sp[0] + sp[256]) + (sp[512] + sp[768])) + ((sp[1024] + sp[1280]) + (sp[1536] + sp[1792]));
        const f32x4_t hb = *(const f32x4_t*)(smem + T_HB);
        const float u = sg + hb[3] + nz - 1.f;      // ShiftedSoftplus, models/nerf.py:68-69
        f32x4_t o;
        // (hardware exp2 / log2 / rcp: ~2 ulp each, far inside the 16-bit rows they are computed from; the library forms keep
        //  a dozen constants alive across the whole tile loop - spilled, and reloaded here behind a wait for the stores above)
        o[0] = __builtin_amdgcn_rcpf(1.f + __expf(-(cp[0] + hb[0])));
        o[1] = __builtin_amdgcn_rcpf(1.f + __expf(-(cp[256] + hb[1])));
        o[2] = __builtin_amdgcn_rcpf(1.f + __expf(-(cp[512] + hb[2])));
        const float eu = __expf(fminf(u, 20.f));
        const float lp = eu < 1e-3f ? eu * (1.f - eu * (0.5f - eu * (1.f / 3.f))) : __logf(1.f + eu);     // log1p(e^u)
        o[3] = u > 20.f ? u : lp;
        *(f32x4_t*)(d.heads_raw + tok * 4) = o;
      }
    }
  };
  auto stage_dropped = [&](const Ctx& c_, const Tile& t, int idx_off, u32x4_t& z) {
    // dropped tokens: zero rows into the tile (they meet the shared layers' biases only), zero sigma sums, zero rows of the saved y
    // (z: the zero store operand, in a register set of the caller's - SWN_KEEP)
    const int rg_ = c_.w >> 2, fg_ = c_.w & 3;
    z = u32x4_t{0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(z));           // (one register set for all 16 stores, not a constant re-materialised beside each)
#pragma unroll
    for (int j = 0; j < 16; ++j) *(u32x4_t*)(smem + (64 * rg_ + fg_ + 4 * j) * 1024 + c_.lane * 16) = z;
    if (c_.lhi == 0) {      // (the wave's own rows: the ones its tail_out has just read)
#pragma unroll
      for (int q = 0; q < 8; ++q) ((float*)(smem + T_SIGP))[q * 256 + 128 * rg_ + 8 * (c_.l31 >> 1) + 2 * fg_ + (c_.l31 & 1)] = 0.f;
    }
    void* ys = d.layers[d.tail_first - 1].save;
    if (ys) {
      const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(ys, (int)oob_s);
      const int* idx = (const int*)(smem + idx_off);
      uint32_t off[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) off[j] = (uint32_t)idx[2 * (64 * rg_ + fg_ + 4 * j) + c_.lhi];
      SWN_WAIT_LGKM0();
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = 2 * (64 * rg_ + fg_ + 4 * j) + c_.lhi;
        __builtin_amdgcn_raw_buffer_store_b128(z, rs, r < t.rows ? off[j] * (uint32_t)ROWB + (uint32_t)(c_.l31 * 16) : oob_s, 0, SWN_BIG_STORE_AUX);
      }
    }
  };
  if constexpr (TAIL) {
    if (d.heads_raw) {      // the heads' weights (fp32) for the epilogues of the gate layer and of the last layer
      if (tid < 256) ((float*)(smem + T_WS))[tid] = d.heads_ws[tid];
      if (tid < 384) ((float*)(smem + T_WC))[tid] = d.heads_wc[tid];
      if (tid < 4) ((float*)(smem + T_HB))[tid] = tid < 3 ? d.heads_bc[tid] : d.heads_bs[0];
    }
  }
  if constexpr (HEAD) {
    if (tid < 256) ((float*)(smem + T_WS))[tid] = d.comb_wsig ? d.comb_wsig[tid] : 0.f;
  }
  // ---- prologue: the first tile, its source rows ----
  if (cx.w == 0 && cx.lane < 3) gcount[cx.lane] = 0;      // ([2]: queues this workgroup has left behind - work stealing)
  if (cx.w == 0) grab(0, claim());
  SWN_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();
  Tile cur = read_tile(0);
  if (cur.vb < 0) { finish(); return; }          // (nothing for this workgroup: every wave sees the same)
  store_row(load_row(cur), G::IDX0);
  SWN_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();
  if (rg == 1) __builtin_amdgcn_s_barrier();     // row group 1 runs one phase behind from here on

  f32x16_t acc[MI][2];
  SWN_TM(const long long t_start = TICK(); long long tS = 0, tSb = 0, tK = 0, tKb = 0, tE = 0, tEb = 0, tSw = 0, tSi = 0;)
  int n_skip = 0;
  int it = 0;                                    // tiles this row group has started
  int bc = 0;                                    // running layer counter: the bias of this row group's layer lives in slot bc % 3
  Tile prev = cur;
  for (;;) {
    const int idx_cur = (it & 1) ? Q_IDX1 : G::IDX0, idx_nxt = (it & 1) ? G::IDX0 : Q_IDX1;
    SWN_TM(const long long s0 = TICK();)
    // ================= S phase: write-out of the previous tile's rows of this group, staging of this tile's =================
    {
      // (per-lane / per-wave coordinates re-derived from laundered ids: computed once outside the tile loop, the ~50 swizzled piece
      //  addresses and scalar piece offsets of this phase would live - spilled - through every other phase)
      Ctx cs = cx;
      cs.lane = fresh_lane();
      asm volatile("" : "+s"(cs.w));
      cs.l31 = cs.lane & 31;
      cs.lhi = cs.lane >> 5;
      const int rgs = cs.w >> 2, fgs = cs.w & 3;
      int ticket = 0;
      if (cs.w == 0) ticket = claim();           // (the claim of the tile after this one travels under the write-out and the staging)
      // the store operands of this phase's write-out: allocated (SWN_KEEP) until the s_waitcnt vmcnt(0) at the end of the phase has
      // retired the stores - no register of a store is handed to the staging's temporaries behind it
      u32x4_t keep[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) keep[i] = u32x4_t{0u, 0u, 0u, 0u};
      if (it > 0) {
        if constexpr (TAIL) {
          tail_out(cs, prev, idx_nxt, keep);     // (the other table still holds the previous tile's tokens)
        } else if (HEAD && prev.l1 < n_layers) {
          // (a tile of dropped tokens has no output rows; its gate gradients were zeroed when it was staged)
        } else {
          const __amdgpu_buffer_rsrc_t ry = out_rs(d.y, prev);
          const __amdgpu_buffer_rsrc_t ra = d.y_add ? out_rs((void*)d.y_add, prev) : ry;
          if constexpr (TAG == 5) {      // (only this instantiation carries the fused combine backward)
            if (d.comb_y) write_pieces16_comb<E>(cs, 64 * rgs + fgs, ry, d, prev.grow0, prev.rows, keep);
            else write_pieces16<E, false, 8>(cs, 64 * rgs + fgs, ry, ra, keep);
          } else if (d.y_add && d.y_add_gather) write_pieces16_gather<E>(cs, 64 * rgs + fgs, ry, (const char*)d.y_add, ((it - 1) & 1) ? Q_YIDX + 1024 : Q_YIDX, keep);
          else if (d.y_add) write_pieces16<E, true, 8>(cs, 64 * rgs + fgs, ry, ra, keep);
          else write_pieces16<E, false, 8>(cs, 64 * rgs + fgs, ry, ra, keep);
        }
        SWN_WAIT_LGKM0();                        // (every piece is in registers / on its way: the rows may be overwritten)
      }
      SWN_TM(const long long sw = TICK(); tSw += sw - s0;)
      float gate_v = 0.f, dsig_v = 0.f;
      const int lts = fgs * 64 + cs.lane;
      if constexpr (HEAD) {
        if (lts < 128) {
          const int r_ = 128 * rgs + lts;
          const long tok = ((const int*)(smem + idx_cur))[r_];
          if (cur.l1 < n_layers) {               // dropped tokens: no expert saw them - their gate gradient is zero
            if (r_ < cur.rows) d.comb_dgate[tok] = 0.f;
          } else {
            gate_v = d.comb_gate[tok];
            dsig_v = d.comb_dsig ? d.comb_dsig[tok] : 0.f;
          }
        }
        stage_pieces_q<true>(cs, (const char*)d.x, 64 * rgs + fgs, 16, 4, idx_cur);
      } else if constexpr (TAIL) {
        if (cur.l0) {
          stage_dropped(cs, cur, idx_cur, keep[15]);     // (tail_out holds keep[0..7])
        } else {
          if (lts < 128) gate_v = d.tail_gate[((const int*)(smem + idx_cur))[128 * rgs + lts]];      // the rows' gate values -> T_GATE below
          stage_pieces_q<false>(cs, (const char*)d.x, 64 * rgs + fgs, 16, 4, idx_cur);
        }
      } else if (narrow) stage_pieces_q<true>(cs, (const char*)d.x, 64 * rgs + fgs, 16, 4, idx_cur);
      else stage_pieces_q<false>(cs, (const char*)d.x, 64 * rgs + fgs, 16, 4, idx_cur);
      if (cs.w == 0) grab((it + 1) & 1, ticket);   // the tile after this one (both row groups read it during their last epilogue phase)
      SWN_TM(const long long si = TICK(); tSi += si - s0;)
      // The rows have landed (and the stores before them have retired).  Measured (profiles/r04_experiments.md): the 32 vector memory
      // operations of this phase take 6-9 k clocks to ISSUE whatever their order (copies first and a counted wait: no gain) - the CU's
      // vector memory path carries the partner group's weight stream at the same time and is the bound of this kernel family.
      SWN_WAIT_VM(0);
      SWN_KEEP16(keep);                          // (the write-out's stores have retired)
      if constexpr (TAIL) {
        if (!cur.l0 && lts < 128) ((float*)(smem + T_GATE))[128 * rgs + lts] = gate_v;
      }
      if constexpr (HEAD) {
        if (cur.l1 == n_layers && lts < 128) {
          ((float*)(smem + T_GATE))[128 * rgs + lts] = gate_v;
          ((float*)(smem + T_DSIG))[128 * rgs + lts] = dsig_v;
        }
      }
    }
    if (rg == 0 && it == 0) { stage_bias(cur.l0, ws_of(cur.l0, cur), 0); SWN_WAIT_VM(0); }
    u32x4_t mk_next = load_mask(cur.l0, cur.vb);
    SWN_PIN();
    preload_w(cur.l0, ws_of(cur.l0, cur), fresh_lane());
    SWN_PIN();
    SWN_TM(const long long s1 = TICK(); tS += s1 - s0;)
    Tile nxt = cur;
    const int l_end = HEAD ? cur.l1 : n_layers;
    for (int L = cur.l0; L < l_end; ++L) {
      const swn_chain_layer& ly = d.layers[L];
      const bool last = L + 1 == l_end;
      // The phase boundary in front of every K phase (behind the S phase or the preceding E phase) sits HERE, with the bias values of
      // the K phase read right before it: one program point for them (read at the top of the K phase their LDS round trip was on its
      // critical path; read in the two preceding phases they were carried around the loop's back edge through scratch)
      load_bv(bc % 3);                             // (first tile: a wave reads the quarter of the slot its own copy has just filled)
      SWN_PIN();
      SWN_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      u32x4_t mk = mk_next;
      if (last) nxt = read_tile((it + 1) & 1);   // (written by wave 0 during row group 0's S phase of this tile, barriers ago)
      // ---- K phase ----
      {
        SWN_TM(const long long k0 = TICK();)
        const __amdgpu_buffer_rsrc_t rs_cur = wrs(L, ws_of(L, cur));
        // (row group 0, for both groups) the NEXT layer's bias -> its slot; the K loop's counted waits cover the copy
        if (rg == 0) {
          if (!last) stage_bias(L + 1, ws_of(L + 1, cur), (bc + 1) % 3);
          else if (nxt.vb >= 0) stage_bias(nxt.l0, ws_of(nxt.l0, nxt), (bc + 1) % 3);
        }
        if constexpr (bias_init) {
          // accumulators start at the bias (zeros for a layer without one: stage_bias).  Written as plain copies of two 16-register
          // tuples: the compiler feeds the tuples to the first K step's MFMAs as their C operand - no copy is executed.  The values were
          // read at the end of the preceding phase (load_bv: their LDS round trip is not on this phase's critical path)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = bv[ni][r >> 2][r & 3];
        } else {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }
        {
          const Ctx ck = phase_ctx(true);
          k_phase2<E>(acc, ck, rs_cur, wq);
        }
#pragma unroll
        for (int q_ = 0; q_ < WQD; ++q_)
#pragma unroll
          for (int i_ = 0; i_ < 2; ++i_) asm volatile("" : "=v"(wq[q_][i_]));
        SWN_TM(const long long k1 = TICK();)
        __builtin_amdgcn_s_barrier();
        SWN_TM(const long long k2 = TICK(); tK += k1 - k0; tKb += k2 - k1;)
      }
      // ---- E phase ----
      {
        SWN_TM(const long long e0 = TICK();)
        const Ctx ce = phase_ctx(false);
        const int rge = ce.w >> 2, fge = ce.w & 3;
        const int lane16e = ce.lane * 16;
        uint32_t* mkp = ly.mask ? ly.mask + ((size_t)(cur.vb * G::NW + ce.w) * 64 + ce.lane) * 4 : nullptr;
        // The partner's rows = the input tile of the K loop it is running (same tile: the groups are one phase apart) -> their save
        // tensor: 16 pieces in 8 batches of 2 through TWO alternating register sets.  Batch 0 is READ at the top of the phase and stored
        // behind the first half row tile (its LDS round trip runs under that half tile - storing it in front of the epilogue, as round 5
        // did, put the round trip on the critical path of all 19 phases of a tile: +2-3 % on the two fused launches, same-box A/B in
        // profiles/r06_experiments.md 1); batch h is stored behind half row tile h.  The LDS reads that refill a set are issued a half
        // step after ITS stores, behind the statement that keeps it allocated (SWN_KEEP) - hundreds of clocks; the last batch's set stays
        // allocated past the end of the epilogue (`held` below).
        void* wo = rge == 0 ? (L > cur.l0 ? d.layers[L - 1].save : nullptr) : (!last ? ly.save : nullptr);
        // fused tail: the saves from the gate layer on go to TOKEN order; the gate layer and the last layer have their own epilogues
        const bool wo_tok = (TAIL && (rge == 0 ? L - 1 : L) >= d.tail_first - 1) || (HEAD && (rge == 0 ? L - 1 : L) < d.head_layers - 1);
        const int c0 = 64 * (1 - rge) + fge;
        u32x4_t wv[2][2];
        int tk[2][2];                              // (fused tail: the tokens of the pieces' rows)
        auto rd = [&](int b) {                     // the two pieces of batch b -> set b & 1
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            wv[b & 1][j] = *(const u32x4_t*)(smem + piece_addr(ce, c0 + 4 * (2 * b + j)));
            if constexpr (DROPS) tk[b & 1][j] = ((const int*)(smem + idx_cur))[2 * (c0 + 4 * (2 * b + j)) + ce.lhi];
          }
        };
        if (wo) rd(0);
        if (!last) mk_next = load_mask(L + 1, cur.vb);
        int row_nxt = 0, yrow = -1;
        if (last && nxt.vb >= 0) row_nxt = load_row(nxt);      // (consumed behind the epilogue)
        if (last && d.y_add_gather && d.y_add) {               // the y_add row of this thread's output row of THIS tile (write_pieces16_gather)
          const int lt = fg * 64 + fresh_lane();
          if (lt < 128) { const int r = 128 * rg + lt; yrow = d.y_add_gather[cur.grow0 + (r < cur.rows ? r : 0)]; }
        }
        const bool bias_epi = ly.b != nullptr && !bias_init;
        if (ly.skip) {
          stage_pieces_q<false>(ce, (const char*)d.x, 64 * rge + fge, 16, 4, idx_cur);      // (a residual layer has n = k0 = 256)
          SWN_WAIT_VM(0);
          ++n_skip;
          if (ce.lane == 0) __hip_atomic_fetch_add(&gcount[rge], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&gcount[rge], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4 * n_skip)
            __builtin_amdgcn_s_sleep(2);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const bool gate_l = TAIL && L + 1 == d.tail_first, rb_l = TAIL && last;
        auto run_epi = [&](auto hook) {
          typedef decltype(hook) HK;
          if constexpr (TAIL) {
            if (gate_l) { epilogue_q_gate<E, HK>(acc, ce, d.heads_raw != nullptr, hook); return; }
            if (rb_l) { epilogue_q_rowbias<E, HK>(acc, ce, ly.rowbias, ly.rows_per_bias, d.tail_bias_row, yf, idx_cur, hook); return; }
          }
          epilogue_p_dispatch<E, HK>(acc, ce, mk, ly.relu, bias_epi, ly.skip != 0, (bc % 3) * 1024, hook);
        };
        if (wo) {
          const __amdgpu_buffer_rsrc_t rs = wo_tok ? uniform_rsrc(wo, (int)oob_s) : out_rs(wo, cur);
          auto st = [&](int b) {                   // the two pieces of batch b, from set b & 1
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int c = c0 + 4 * (2 * b + j);
              if constexpr (DROPS) {     // one form for both row spaces: row index = the token, or the tile row under a descriptor of the tile's rows
                const int r = 2 * c + ce.lhi;
                const uint32_t ri = wo_tok ? (uint32_t)tk[b & 1][j] : (uint32_t)r;
                const uint32_t off = r < cur.rows ? ri * (uint32_t)ROWB + (uint32_t)(ce.l31 * 16) : oob_s;
                __builtin_amdgcn_raw_buffer_store_b128(wv[b & 1][j], rs, off, 0, SWN_BIG_STORE_AUX);
              } else {
                __builtin_amdgcn_raw_buffer_store_b128(wv[b & 1][j], rs, lane16e, c * 1024, SWN_BIG_STORE_AUX);
              }
            }
          };
          auto hook_f = [&](int h) {
            // the set stored a half step ago (batch h - 1) stays ALLOCATED up to here (an empty statement that reads it): a value is
            // dead behind its store, and the compiler would hand its registers to the very next epilogue temporaries.  The next write
            // to them is the refill below.
            asm volatile("" :: "v"(wv[(h + 1) & 1][0]), "v"(wv[(h + 1) & 1][1]));
            st(h);                                 // (read a half step ago; batch 0 at the top of the phase)
            if (h < 7) rd(h + 1);                  // (into the OTHER set)
            SWN_PIN();
          };
          const HalfHook<decltype(hook_f)> hook{hook_f};
          run_epi(hook);
          SWN_PIN();
        } else {
          run_epi(NoHook());
        }
        if (ly.relu == 1 && mkp) *(u32x4_t*)mkp = mk;
        SWN_PIN();
        bool held = false;                         // (the last batch's store operands stay allocated past the end of the epilogue)
        if constexpr (HEAD) {
          if (L + 1 == d.head_layers && cur.l1 == n_layers) {
            // ---- the combine backward on this row group's rows, in place (see comb_pieces16_inplace): the four waves have written the
            //      layer's output rows - meet (the row groups' own counter: no workgroup barrier inside a phase), then every wave takes
            //      the rows it stages ----
            ++n_skip;
            SWN_WAIT_LGKM0();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (ce.lane == 0) __hip_atomic_fetch_add(&gcount[rge], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&gcount[rge], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4 * n_skip)
              __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (wo) asm volatile("" :: "v"(wv[1][0]), "v"(wv[1][1]));      // (behind the four waves' meeting: hundreds of clocks)
            held = true;
            comb_pieces16_inplace<E>(ce, 64 * rge + fge, d, idx_cur, cur.rows,
                                     d.comb_dwsig_ws ? d.comb_dwsig_ws + ((long)cur.vb * 8 + ce.w) * 256 : nullptr);
          }
        }
        SWN_PIN();
        if (!last) preload_w(L + 1, ws_of(L + 1, cur), ce.lane);      // (last layer: the S phase that follows loads the next tile's first fragments)
        else if (nxt.vb >= 0) store_row(row_nxt, idx_nxt);
        if (last && d.y_add_gather && d.y_add) {
          const int lt = fg * 64 + fresh_lane();
          if (lt < 128) ((int*)(smem + ((it & 1) ? Q_YIDX + 1024 : Q_YIDX)))[128 * rg + lt] = yrow;
        }
        SWN_PIN();
        // the set stored behind the LAST half row tile: allocated up to here - behind the mask store and the next K loop's six fragment
        // loads, vector memory instructions that issue in order behind its stores
        if (wo && !held) asm volatile("" :: "v"(wv[1][0]), "v"(wv[1][1]));
        SWN_PIN();
        SWN_TM(const long long e1 = TICK();)
        if (last) {                                // (behind the other layers the boundary is the one at the top of the next layer)
          SWN_WAIT_LGKM0();
          __builtin_amdgcn_s_barrier();
        }
        SWN_TM(const long long e2 = TICK(); tE += e1 - e0; tEb += e2 - e1;)
      }
      ++bc;
    }
    prev = cur;
    cur = nxt;
    ++it;
    if (cur.vb < 0) break;
  }
  // ---- the rows of the last tile ----
  Ctx cf = cx;                                   // (lane coordinates re-derived: nothing of them lives through the tile loop)
  cf.lane = fresh_lane();
  cf.l31 = cf.lane & 31;
  cf.lhi = cf.lane >> 5;
  u32x4_t keep[16];                               // (store operands: allocated until the stores have retired - SWN_KEEP)
#pragma unroll
  for (int i = 0; i < 16; ++i) keep[i] = u32x4_t{0u, 0u, 0u, 0u};
  if constexpr (TAIL) {
    tail_out(cf, prev, ((it - 1) & 1) ? Q_IDX1 : G::IDX0, keep);
  } else if (HEAD && prev.l1 < n_layers) {
  } else {
    const __amdgpu_buffer_rsrc_t ry = out_rs(d.y, prev);
    const __amdgpu_buffer_rsrc_t ra = d.y_add ? out_rs((void*)d.y_add, prev) : ry;
    if constexpr (TAG == 5) {
      if (d.comb_y) write_pieces16_comb<E>(cf, 64 * rg + fg, ry, d, prev.grow0, prev.rows, keep);
      else write_pieces16<E, false, 8>(cf, 64 * rg + fg, ry, ra, keep);
    } else if (d.y_add && d.y_add_gather) write_pieces16_gather<E>(cf, 64 * rg + fg, ry, (const char*)d.y_add, ((it - 1) & 1) ? Q_YIDX + 1024 : Q_YIDX, keep);
    else if (d.y_add) write_pieces16<E, true, 8>(cf, 64 * rg + fg, ry, ra, keep);
    else write_pieces16<E, false, 8>(cf, 64 * rg + fg, ry, ra, keep);
  }
  SWN_WAIT_VM(0);
  SWN_KEEP16(keep);
  if (rg == 0) __builtin_amdgcn_s_barrier();      // (row group 1's last phase boundary)
#ifdef SWN_BIG_TIMING
  if (d.y_add_gather && !d.y_add && cx.lane == 0 && blockIdx.x < 512) {   // wave w of workgroup b -> row 512 w + b
    long long* dbg = (long long*)d.y_add_gather + (long)(cx.w * 512 + blockIdx.x) * 8;
    dbg[0] = tS; dbg[1] = tSw; dbg[2] = tK; dbg[3] = tKb; dbg[4] = tE; dbg[5] = tEb + tSb; dbg[6] = it | (tSi << 16); dbg[7] = TICK() - t_start;
  }
#endif
  finish();
}

}  // namespace swn_big

namespace swn {

bool chain_big_eligible(const swn_chain_desc& d) {
  if (d.dtype != SWN_HALF) return false;
#ifdef SWN_BIG_TIMING
  if (d.x_save || d.x_scale) return false;
#else
  if (d.x_save || d.x_scale || d.y_add_gather) return false;
#endif
  if (d.comb_y) return false;
  for (int l = 0; l < d.n_layers; ++l) {
    const swn_chain_layer& ly = d.layers[l];
    if (ly.n != 256 || ly.k != 256 || ly.rowbias || ly.skip > 1) return false;
  }
  return true;
}

// geometries 6 / 7 (chainq_kernel) also take the dense front chains: a 128-feature chain input (x_features = 128 under a first layer
// whose weights are zero-padded to k = 256) and a gathered y_add
bool chain_persistent_eligible(const swn_chain_desc& d) {
  if (d.tail_first > 0) {      // the dense tail folded into the expert forward chain: chainq_kernel<., 7, true> only
    if (d.dtype != SWN_HALF || d.geometry != 7 || d.tag != 7 || d.x_save || d.x_scale || d.y_add || d.comb_y || !d.x_gather) return false;
    if (d.tail_first >= d.n_layers || !d.tail_gate || !d.tail_dropped || !d.tail_n_dropped || d.tail_tokens <= 0) return false;
    if ((long)d.tail_tokens * 512 >= (1L << 32) - 64) return false;
    if (d.x_features != 0 && d.x_features != 256) return false;
    if (d.y_features != 128) return false;
    for (int l = 0; l < d.n_layers; ++l) {
      const swn_chain_layer& ly = d.layers[l];
      const bool last = l + 1 == d.n_layers;
      if (ly.n != 256 || ly.k != 256 || ly.skip > 1 || ly.relu == 2) return false;
      if (ly.rowbias && (!last || (ly.rows_per_bias <= 0 && !d.tail_bias_row))) return false;
      if (l >= d.tail_first - 1 && (ly.skip || ly.mask)) return false;
      if (l == d.tail_first - 1 && ly.relu) return false;      // (the gate layer: ReLU comes with the scaling)
      if (last && ly.relu != 1) return false;                  // (the last layer's epilogue: row bias, then ReLU)
    }
    return true;
  }
  if (d.head_layers > 0) {     // the tail's backward layers in front of the expert backward chain: chainq_kernel<., 8, true> only
    if (d.dtype != SWN_HALF || d.geometry != 7 || d.tag != 8 || d.tail_first || d.x_save || d.x_scale || d.heads_raw || d.y_add_gather || !d.x_gather) return false;
    if (d.head_layers >= d.n_layers || !d.comb_y || !d.comb_gate || !d.comb_dgate || !d.tail_dropped || !d.tail_n_dropped || d.tail_tokens <= 0) return false;
    if ((long)d.tail_tokens * 512 >= (1L << 32) - 64 || d.x_features != 128) return false;
    for (int l = 0; l < d.n_layers; ++l) {
      const swn_chain_layer& ly = d.layers[l];
      if (ly.n != 256 || ly.k != 256 || ly.rowbias || ly.skip || ly.b || ly.relu == 1) return false;
      if (l < d.head_layers && (ly.relu || ly.mask)) return false;
    }
    return true;
  }
  if (d.tag == 7 || d.tag == 8) return false;
  if (d.dtype != SWN_HALF || d.x_save || d.x_scale || d.heads_raw) return false;
  if (d.comb_y && (d.tag != 5 || d.y_add)) return false;      // (the fused combine backward: the tail backward instantiation only)
#ifndef SWN_BIG_TIMING
  if (d.y_add_gather && !d.y_add) return false;
#endif
  for (int l = 0; l < d.n_layers; ++l) {
    const swn_chain_layer& ly = d.layers[l];
    if (ly.n != 256 || ly.k != 256 || ly.rowbias || ly.skip > 1) return false;
    if (ly.skip && d.x_features == 128) return false;
  }
  return d.x_features == 0 || d.x_features == 256 || d.x_features == 128;
}

int chain_big_tile_rows(int geometry) { return geometry == 3 ? swn_big::G96::BM : swn_big::G256::BM; }
int chain_big_mask_words_per_tile(int geometry) { return (geometry == 3 ? swn_big::G96::NW : swn_big::G256::NW) * 256; }

template <typename G>
static int chain_big_launch_g(const swn_chain_desc& d, void* stream) {
  using namespace swn_big;
#ifdef SWN_HALF_F16
  typedef Fp16 HalfT;
#else
  typedef Bf16 HalfT;
#endif
  Args a;
  a.d = d;
  a.tiles_per_group = cdiv(d.group_rows ? (d.group_rows_clamp < d.group_stride ? d.group_rows_clamp : d.group_stride) : d.group_stride, G::BM);
  if (!d.group_rows) a.d.group_rows_clamp = d.group_stride;
  const long grid = (long)a.tiles_per_group * d.n_groups;
  SWN_CHECK(grid > 0 && grid < (1L << 31), "swn_mlp_chain: grid %ld out of range", grid);
  const void* fn = nullptr;
#define SWN_PICKB(TAGV)                                                                                                     \
  case TAGV:                                                                                                                \
    fn = (const void*)chainb_kernel<HalfT, G, TAGV>;       \
    break;
  switch (d.tag) {
    SWN_PICKB(1) SWN_PICKB(2)
    default: fn = (const void*)chainb_kernel<HalfT, G, 0>;
  }
#undef SWN_PICKB
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(G::NT), kargs, G::LDS_BYTES, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_mlp_chain (geometry %d) launch: %s", d.geometry, hipGetErrorString(e));
  return 0;
}

static int chain_phase_launch(const swn_chain_desc& d, void* stream) {
  using namespace swn_big;
  typedef G256 G;
#ifdef SWN_HALF_F16
  typedef Fp16 HalfT;
#else
  typedef Bf16 HalfT;
#endif
  Args a;
  a.d = d;
  a.tiles_per_group = cdiv(d.group_rows ? (d.group_rows_clamp < d.group_stride ? d.group_rows_clamp : d.group_stride) : d.group_stride, G::BM);
  if (!d.group_rows) a.d.group_rows_clamp = d.group_stride;
  const long grid = (long)a.tiles_per_group * d.n_groups;
  SWN_CHECK(grid > 0 && grid < (1L << 31), "swn_mlp_chain: grid %ld out of range", grid);
  const void* fn = d.tag == 1 ? (const void*)chainp_kernel<HalfT, 1> : d.tag == 2 ? (const void*)chainp_kernel<HalfT, 2> : (const void*)chainp_kernel<HalfT, 0>;
  constexpr int LDS_P = G::BIAS0 + 3072 + 64;        // three bias slots + the row groups' meeting counters
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(G::NT), kargs, LDS_P, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_mlp_chain (geometry 4 / 5) launch: %s", hipGetErrorString(e));
  return 0;
}


static int n_compute_units() {
  static int n = 0;
  if (!n) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    n = cus;
  }
  return n;
}

// comb_dwsig (include/swn.h): the per-wave sums of a fused backward launch, [n_rows][256] fp32, added up in a fixed order - runs of
// consecutive rows first (this kernel: one column per thread), then the runs (ordered_reduce_kernel).
__global__ __launch_bounds__(256) void dwsig_runs_kernel(const float* __restrict__ partial, long n_rows, int rows_per_block, float* __restrict__ out) {
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
  const int t = threadIdx.x;
  float s = 0.f;
  long r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(partial + (r + u) * 256 + t);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; r < r1; ++r) s += partial[r * 256 + t];
  out[(long)blockIdx.x * 256 + t] = s;
}
constexpr int DWSIG_RUNS = 1024;
static long dwsig_ws_rows(int n_groups, int group_rows_clamp) { return (long)cdiv(group_rows_clamp, 256) * n_groups * 8; }

static int chain_persistent_launch(const swn_chain_desc& d, void* stream) {
  using namespace swn_big;
  typedef G256 G;
#ifdef SWN_HALF_F16
  typedef Fp16 HalfT;
#else
  typedef Bf16 HalfT;
#endif
  ArgsQ a;
  a.d = d;
  a.tiles_per_group = cdiv(d.group_rows ? (d.group_rows_clamp < d.group_stride ? d.group_rows_clamp : d.group_stride) : d.group_stride, G::BM);
  if (!d.group_rows) a.d.group_rows_clamp = d.group_stride;
  const long n_vb = (long)a.tiles_per_group * d.n_groups;
  SWN_CHECK(n_vb > 0 && n_vb < (1L << 28), "swn_mlp_chain: %ld tiles out of range", n_vb);
  a.n_vb = a.n_vb_e = (int)n_vb;
  if (d.tail_first > 0 || d.head_layers > 0) a.n_vb += cdiv(d.tail_dropped_max, G::BM);      // tiles of the dropped tokens behind the experts' (the count is a
                                                                        // device scalar: tiles beyond it end the queue)
  int grid = n_compute_units();                         // one resident workgroup per CU (157 KiB of LDS each)
  const char* ov = getenv("SWN_CHAINQ_WGS");            // experiments
  if (ov && atoi(ov) > 0) grid = atoi(ov);
  if (grid > n_vb) grid = (int)n_vb;
  const bool rotated = (d.n_wsets & 7) == 0 && (d.n_groups & 7) == 0;
  a.n_queues = (rotated && grid % 8 == 0 && d.sched) ? 8 : 1;
  a.per_queue = 0;
  if (!rotated && d.n_groups == 1 && grid % 8 == 0 && d.sched && n_vb >= 64) {      // one group (dense chains): XCD x walks its own eighth
    a.n_queues = 8;
    a.per_queue = (int)cdiv(n_vb, 8);
  }
  a.stagger = 0;                                        // (a start offset between the workgroups of an XCD: measured, no effect - r04_experiments.md 2)
  const void* fn;
#define SWN_PICKQ(TAGV)                                                                                                       \
  case TAGV:                                                                                                                  \
    fn = d.geometry == 7 ? (const void*)chainq_kernel<HalfT, TAGV, true> : (const void*)chainq_kernel<HalfT, TAGV, false>;    \
    break;
  switch (d.tag) {
    SWN_PICKQ(1) SWN_PICKQ(2) SWN_PICKQ(3) SWN_PICKQ(5) SWN_PICKQ(6)
    case 7: fn = (const void*)chainq_kernel<HalfT, 7, true>; break;
    case 8: fn = (const void*)chainq_kernel<HalfT, 8, true>; break;
    default: fn = d.geometry == 7 ? (const void*)chainq_kernel<HalfT, 0, true> : (const void*)chainq_kernel<HalfT, 0, false>;
  }
#undef SWN_PICKQ
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
  SWN_CHECK(e == hipSuccess, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const long dws_rows = d.comb_dwsig_ws ? (long)a.n_vb_e * 8 : 0;      // (tiles without rows are never visited: their slots stay zero)
  if (dws_rows) {
    e = fill_u32_async(d.comb_dwsig_ws, 0u, (size_t)dws_rows * 1024, as_stream(stream));
    SWN_CHECK(e == hipSuccess, "swn_mlp_chain: comb_dwsig_ws fill: %s", hipGetErrorString(e));
  }
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(G::NT), kargs, Q_LDS, as_stream(stream));
  SWN_CHECK(e == hipSuccess, "swn_mlp_chain (geometry 6 / 7) launch: %s", hipGetErrorString(e));
  if (dws_rows) {
    int rpb = cdiv(dws_rows, DWSIG_RUNS);
    rpb = (rpb + 7) & ~7;
    const int runs = cdiv(dws_rows, rpb);
    float* run_sums = d.comb_dwsig_ws + dws_rows * 256;
    hipLaunchKernelGGL(dwsig_runs_kernel, dim3(runs), dim3(256), 0, as_stream(stream), (const float*)d.comb_dwsig_ws, dws_rows, rpb, run_sums);
    OrdDst od{{d.comb_dwsig, nullptr, nullptr, nullptr}, {256, 0, 0, 0}};
    ordered_reduce_async(run_sums, runs, 256, od, true, as_stream(stream));
    SWN_LAUNCH_CHECK();
  }
  return 0;
}

int chain_big_launch(const swn_chain_desc& d, void* stream) {
  if (d.geometry >= 6) return chain_persistent_launch(d, stream);
  if (d.geometry >= 4) return chain_phase_launch(d, stream);
  if (d.geometry == 3) return chain_big_launch_g<swn_big::G96>(d, stream);
  return chain_big_launch_g<swn_big::G256>(d, stream);
}

}  // namespace swn

extern "C" size_t swn_chain_dwsig_workspace_bytes(int n_groups, int group_rows_clamp) {
  if (n_groups <= 0 || group_rows_clamp <= 0) return 0;
  return (size_t)(swn::dwsig_ws_rows(n_groups, group_rows_clamp) + swn::DWSIG_RUNS) * 1024;
}

extern "C" int swn_chain_big_ok(const swn_chain_desc* desc) { return desc != nullptr && swn::chain_big_eligible(*desc) ? 1 : 0; }

// swn_mlp_chain: fused Linear(+bias,+ReLU,+skip) chains on CDNA4 MFMA with the activation tile resident in LDS.
//
// Replaces ExpertMLP.forward (/root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:887-924,
// a baddbmm per layer with activations round-tripping HBM) and Mlp.forward (models/nerf_moe.py:30-49), and - run
// with transposed weights and the stored ReLU masks - their backward-data passes.
//
// Geometry (one workgroup = 512 threads = 8 waves = 2 (rows) x 4 (features)):
//   bf16: 128-row tile, v_mfma_f32_32x32x16_bf16, weight K-slices of 64 streamed L2 -> regs -> LDS (double buffer)
//   fp32:  64-row tile, v_mfma_f32_32x32x2_f32 (exact fp32 fma chain; the parity mode)
// The MFMA is issued "transposed": the weight fragment is the A operand and the activation fragment the B
// operand, so a lane ends up with 4 consecutive output FEATURES of one row (D[i=feature][j=row]); the epilogue
// can then pack them and write the next layer's input tile row-major with 8-byte LDS stores.
// LDS: activation tile 64 KiB (XOR-swizzled 16-B chunks so the 32 rows of a fragment read hit distinct banks)
//      + 2 x 32 KiB weight slices  = 128 KiB  -> one workgroup per CU, two waves per SIMD.
#include "common.hpp"

namespace swn {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int NT = 512;         // threads per workgroup
constexpr int WN = 4;           // waves along features
constexpr int NI = 2;           // 32-wide feature tiles per wave (4 * 2 * 32 = 256 = max features)
constexpr int ACT_BYTES = 65536;
constexpr int WBUF_BYTES = 32768;

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
  static constexpr int BM = 128, BK = 64, MI = 2;
};
template <> struct Cfg<float> {
  static constexpr int BM = 64, BK = 32, MI = 1;
};

// ---- LDS addressing ---------------------------------------------------------------------------------------------
// activation tile: row-major, row stride 256 elements.
__device__ __forceinline__ int act_off(bf16_t*, int row, int col) {  // byte offset of element (row, col)
  return row * 512 + ((((col >> 3) ^ (row & 15))) << 4) + ((col & 7) << 1);
}
__device__ __forceinline__ int act_off(float*, int row, int col) { return row * 1024 + ((col ^ (row & 31)) << 2); }
// weight slice: [n][BK] with k contiguous (128 B per n-row for both dtypes)
__device__ __forceinline__ int w_off_chunk(bf16_t*, int n, int kc) { return n * 128 + ((kc ^ ((n >> 1) & 7)) << 4); }
__device__ __forceinline__ int w_off_word(int n, int k) { return (n * 32 + (k ^ (n & 31))) << 2; }

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
template <typename T>
__device__ __forceinline__ void load_rows_to_lds(char* dst, const void* src, const int32_t* gather, void* save,
                                                 long grow0, int rows_valid_in_tile, int kfeat, int tid) {
  constexpr int BM = Cfg<T>::BM;
  const int row_bytes = kfeat * (int)sizeof(T);
  const int cpr = row_bytes >> 4;  // 16-byte chunks per row (power of two)
  const int sh = 31 - __builtin_clz(cpr);
  const int total = BM * cpr;
  constexpr int B = 4;
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
      if (in[i]) {
        if (save && row[i] < rows_valid_in_tile) *(uint4*)((char*)save + (grow0 + row[i]) * row_bytes + ch[i] * 16) = v[i];
        store_chunk_to_act<T>(dst, row[i], ch[i], v[i]);
      }
    }
  }
}

template <typename T, int TAG>
__global__ __launch_bounds__(NT) void chain_kernel(const ChainArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = Cfg<T>::BM, BK = Cfg<T>::BK, MI = Cfg<T>::MI;
  const swn_chain_desc& d = args.d;
  char* act = smem;
  char* wbuf = smem + ACT_BYTES;  // 2 x WBUF_BYTES; also reused as a second activation-layout tile (skip input)
  char* bias_lds = smem + ACT_BYTES + 2 * WBUF_BYTES;  // 1 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int g = blockIdx.x % d.n_groups;
  const int tile = blockIdx.x / d.n_groups;
  int rows_valid = d.group_stride;
  if (d.group_rows) rows_valid = d.group_rows[g];
  if (rows_valid > d.group_rows_clamp) rows_valid = d.group_rows_clamp;
  const int row0 = tile * BM;
  if (row0 >= rows_valid) return;
  const int rows_in_tile = min(BM, rows_valid - row0);
  const long grow0 = (long)g * d.group_stride + row0;
  const int wset = g % d.n_wsets;

  // ---- stage the chain input ----
  load_rows_to_lds<T>(act, d.x, d.x_gather, d.x_save, grow0, rows_in_tile, d.layers[0].k, tid);
  __syncthreads();

  f32x16_t acc[MI][NI];

  // Weight K-slices travel global -> registers (W4, named members so they stay in VGPRs) -> LDS.  The loads for the
  // next slice are issued before the MFMAs of the current one and consumed after them; at a layer boundary the
  // "next slice" is slice 0 of the next layer (cross-layer prefetch), parked in registers across the epilogue.
  struct W4 { uint4 a, b, c, e; };
  auto gload = [&](const char* wgp, int n_, int k_, int s_) -> W4 {
    W4 r;
    const size_t kb = (size_t)s_ * BK;
    const int n1 = n_ - 1;
    r.a = *(const uint4*)(wgp + ((size_t)min((tid + NT * 0) >> 3, n1) * k_ + kb) * sizeof(T) + (tid & 7) * 16);
    r.b = *(const uint4*)(wgp + ((size_t)min((tid + NT * 1) >> 3, n1) * k_ + kb) * sizeof(T) + (tid & 7) * 16);
    r.c = *(const uint4*)(wgp + ((size_t)min((tid + NT * 2) >> 3, n1) * k_ + kb) * sizeof(T) + (tid & 7) * 16);
    r.e = *(const uint4*)(wgp + ((size_t)min((tid + NT * 3) >> 3, n1) * k_ + kb) * sizeof(T) + (tid & 7) * 16);
    return r;
  };
  auto lstore1 = [&](char* wb, int c, uint4 v) {
    const int nrow = c >> 3, kc = c & 7;
    if constexpr (sizeof(T) == 2) {
      *(uint4*)(wb + w_off_chunk((bf16_t*)nullptr, nrow, kc)) = v;
    } else {
      *(uint32_t*)(wb + w_off_word(nrow, kc * 4 + 0)) = v.x;
      *(uint32_t*)(wb + w_off_word(nrow, kc * 4 + 1)) = v.y;
      *(uint32_t*)(wb + w_off_word(nrow, kc * 4 + 2)) = v.z;
      *(uint32_t*)(wb + w_off_word(nrow, kc * 4 + 3)) = v.w;
    }
  };
  auto lstore = [&](char* wb, const W4& r) {
    lstore1(wb, tid + NT * 0, r.a);
    lstore1(wb, tid + NT * 1, r.b);
    lstore1(wb, tid + NT * 2, r.c);
    lstore1(wb, tid + NT * 3, r.e);
  };
  auto wptr = [&](int L_) -> const char* {
    return (const char*)d.layers[L_].w + (size_t)wset * d.layers[L_].n * d.layers[L_].k * sizeof(T);
  };
  auto stage_bias = [&](int L_) {
    const swn_chain_layer& q = d.layers[L_];
    if (q.b && tid < (q.n >> 2)) *(float4*)(bias_lds + tid * 16) = *(const float4*)(q.b + (size_t)wset * q.n + tid * 4);
  };

  // slice sequence of the whole chain: (L, s) -> next
  auto next_slice = [&](int& L_, int& s_) -> bool {   // returns false past the end (L_, s_ left on the last slice)
    if (s_ + 1 < d.layers[L_].k / BK) { ++s_; return true; }
    if (L_ + 1 < d.n_layers) { ++L_; s_ = 0; return true; }
    return false;
  };
  auto gload_at = [&](int L_, int s_) -> W4 { return gload(wptr(L_), d.layers[L_].n, d.layers[L_].k, s_); };

  // Software pipeline, two slices deep: at slice u the loads of slice u+2 are issued (-> wC), the MFMAs of slice u run,
  // then slice u+1 (wB, issued one slice earlier: ~2 MFMA phases of latency budget) is written to the other LDS buffer.
  W4 wB, wC;
  int Lb = 0, sb = 0;       // slice held in wB
  int Lc = 0, sc = 0;       // slice held in wC / to be loaded next
  {
    const W4 w0 = gload_at(0, 0);
    stage_bias(0);
    bool ok = next_slice(Lb, sb);          // slice 1 of the chain (or a re-read of slice 0 if there is none)
    wB = gload_at(Lb, sb);
    Lc = Lb; sc = sb;
    (void)ok;
    lstore(wbuf, w0);
  }
  __syncthreads();

  for (int L = 0; L < d.n_layers; ++L) {
    const swn_chain_layer& ly = d.layers[L];
    const int n = ly.n, k = ly.k;
    const int nslices = k / BK;
    const bool wave_active = (wn * 64) < n;
    const bool has_next = (L + 1) < d.n_layers;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // One K-slice: issue the loads of slice u+2 into `ld`, run the MFMAs of slice u from LDS buffer s&1, then write
    // slice u+1 (held in `st`, issued one slice ago) to the other buffer.  Slices are processed in pairs with the
    // roles of the two register sets swapped, so no register copy (and no early wait on the fresh loads) is needed.
    auto slice_body = [&](int s, W4& st, W4& ld) {
      const char* wb = wbuf + (s & 1) * WBUF_BYTES;
      const bool last_slice = (s + 1 == nslices);
      next_slice(Lc, sc);                  // slice u+2 (clamped at the end of the chain: harmless re-read)
      ld = gload_at(Lc, sc);
      if (wave_active) {
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
              bf16x8_t wf[NI], af[MI];
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                const int nn = wn * 64 + ni * 32 + l31;
                wf[ni] = *(const bf16x8_t*)(wb + w_off_chunk((bf16_t*)nullptr, nn, kk * 2 + lhi));
              }
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) {
                const int m = wm * (BM / 2) + mi * 32 + l31;
                af[mi] = *(const bf16x8_t*)(act + act_off((bf16_t*)nullptr, m, s * BK + kk * 16 + lhi * 8));
              }
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
            }
          } else {
#pragma unroll 4
            for (int kk = 0; kk < BK / 2; ++kk) {
              float wf[NI], af[MI];
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                const int nn = wn * 64 + ni * 32 + l31;
                wf[ni] = *(const float*)(wb + w_off_word(nn, kk * 2 + lhi));
              }
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) {
                const int m = wm * (BM / 2) + mi * 32 + l31;
                af[mi] = *(const float*)(act + act_off((float*)nullptr, m, s * BK + kk * 2 + lhi));
              }
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
            }
          }
        }
      if (!last_slice) lstore(wbuf + ((s + 1) & 1) * WBUF_BYTES, st);   // else: st = slice 0 of the next layer, parked
      __syncthreads();
    };
    for (int s = 0; s < nslices; s += 2) {
      slice_body(s, wB, wC);
      if (s + 1 < nslices) slice_body(s + 1, wC, wB);
    }
    // even slice count: next layer's slice 0 is parked in wC and its slice 1 is already in wB (no register moves);
    // odd slice count: parked in wB, the following slice in wC.
    // every wave has finished reading `act` and the weight buffers for this layer.

    if (ly.skip) {  // stage the chain input again (activation layout) in the idle weight buffers
      load_rows_to_lds<T>(wbuf, d.x, d.x_gather, nullptr, grow0, rows_in_tile, d.layers[0].k, tid);
      __syncthreads();
    }

    // ---- epilogue: bias / row-bias / skip / ReLU (or stored mask) -> next layer's input tile ----
    if (wave_active) {
      const bool has_bias = ly.b != nullptr;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = wm * (BM / 2) + mi * 32 + l31;
        uint32_t mbits = 0;
        const size_t midx = ((size_t)(blockIdx.x * 8 + wave) * MI + mi) * 64 + lane;
        if (ly.mask && ly.relu == 2) mbits = ly.mask[midx];
        const float* rb = nullptr;
        if (ly.rowbias) rb = ly.rowbias + ((grow0 + m) / ly.rows_per_bias) * (size_t)n;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (wn * 64 + ni * 32 < n) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int n0 = wn * 64 + ni * 32 + g4 * 8 + lhi * 4;
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][g4 * 4 + j];
              if (has_bias) {
                const float4 b4 = *(const float4*)(bias_lds + n0 * 4);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
              }
              if (rb) {
                const float4 b4 = *(const float4*)(rb + n0);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
              }
              if (ly.skip) {
                if constexpr (sizeof(T) == 2) {
                  const uint2 xv = *(const uint2*)(wbuf + act_off((bf16_t*)nullptr, m, n0));
                  v[0] += bf16_to_f32((bf16_t)(xv.x & 0xFFFF)); v[1] += bf16_to_f32((bf16_t)(xv.x >> 16));
                  v[2] += bf16_to_f32((bf16_t)(xv.y & 0xFFFF)); v[3] += bf16_to_f32((bf16_t)(xv.y >> 16));
                } else {
#pragma unroll
                  for (int j = 0; j < 4; ++j) v[j] += *(const float*)(wbuf + act_off((float*)nullptr, m, n0 + j));
                }
              }
              if (ly.relu == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const bool pos = v[j] > 0.f;
                  mbits |= (pos ? 1u : 0u) << (ni * 16 + g4 * 4 + j);
                  v[j] = pos ? v[j] : 0.f;
                }
              } else if (ly.relu == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ((mbits >> (ni * 16 + g4 * 4 + j)) & 1u) ? v[j] : 0.f;
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
            }
          }
        }
        if (ly.mask && ly.relu == 1) ly.mask[midx] = mbits;
      }
    }
    __syncthreads();
    if (has_next) {  // the skip tile / bias of this layer are dead now: park next layer's slice 0 and bias in LDS
      if (nslices & 1) { lstore(wbuf, wB); wB = wC; }
      else lstore(wbuf, wC);
      stage_bias(L + 1);
    }

    // ---- write-out (row-major, coalesced) ----
    const bool last = (L == d.n_layers - 1);
    void* outp = last ? d.y : ly.save;
    if (outp) {
      const int row_bytes = n * (int)sizeof(T);
      const int cpr = row_bytes >> 4;
      const int total = rows_in_tile * cpr;
      for (int c = tid; c < total; c += NT) {
        const int row = c / cpr, ch = c - row * cpr;
        uint4 v = load_chunk_from_act<T>(act, row, ch);
        if (last && d.y_add) {
          long arow = grow0 + row;
          if (d.y_add_gather) arow = d.y_add_gather[grow0 + row];
          if (arow >= 0) {
            const uint4 a = *(const uint4*)((const char*)d.y_add + arow * row_bytes + ch * 16);
            v = add_chunks<T>(v, a);
          }
        }
        *(uint4*)((char*)outp + (grow0 + row) * row_bytes + ch * 16) = v;
      }
    }
    __syncthreads();  // next layer's slice 0 / bias are visible; `act` is only read until its post-K-loop barrier
  }
}

}  // namespace swn

using namespace swn;

extern "C" int swn_mlp_chain(const swn_chain_desc* desc, void* stream) {
  SWN_CHECK(desc != nullptr, "swn_mlp_chain: null descriptor");
  const swn_chain_desc& d = *desc;
  SWN_CHECK(d.dtype == SWN_F32 || d.dtype == SWN_BF16, "swn_mlp_chain: bad dtype %d", d.dtype);
  SWN_CHECK(d.n_layers >= 1 && d.n_layers <= 8, "swn_mlp_chain: n_layers %d not in [1,8]", d.n_layers);
  SWN_CHECK(d.n_groups >= 1 && d.n_wsets >= 1 && d.group_stride >= 1, "swn_mlp_chain: bad group geometry");
  const int bk = d.dtype == SWN_BF16 ? 64 : 32;
  for (int l = 0; l < d.n_layers; ++l) {
    const swn_chain_layer& ly = d.layers[l];
    SWN_CHECK(ly.n >= 32 && ly.n <= 256 && ly.n % 32 == 0, "swn_mlp_chain: layer %d n=%d must be a multiple of 32 <= 256", l, ly.n);
    SWN_CHECK(ly.k >= bk && ly.k <= 256 && ly.k % bk == 0, "swn_mlp_chain: layer %d k=%d must be a multiple of %d <= 256", l, ly.k, bk);
    if (l > 0) SWN_CHECK(ly.k == d.layers[l - 1].n, "swn_mlp_chain: layer %d k=%d != previous n=%d", l, ly.k, d.layers[l - 1].n);
    SWN_CHECK(ly.w != nullptr, "swn_mlp_chain: layer %d has no weights", l);
    if (ly.skip) SWN_CHECK(ly.n == d.layers[0].k, "swn_mlp_chain: skip layer %d needs n == chain input width", l);
    if (ly.rowbias) SWN_CHECK(ly.rows_per_bias > 0, "swn_mlp_chain: rows_per_bias must be > 0");
    SWN_CHECK(ly.relu >= 0 && ly.relu <= 2, "swn_mlp_chain: relu mode %d", ly.relu);
    if (ly.relu == 2) SWN_CHECK(ly.mask != nullptr, "swn_mlp_chain: relu=2 (apply stored mask) needs a mask");
  }
  SWN_CHECK(d.x != nullptr && d.y != nullptr, "swn_mlp_chain: x / y must not be null");
  ChainArgs a;
  a.d = d;
  const int bm = d.dtype == SWN_BF16 ? 128 : 64;
  a.tiles_per_group = cdiv(d.group_rows ? (d.group_rows_clamp < d.group_stride ? d.group_rows_clamp : d.group_stride) : d.group_stride, bm);
  if (!d.group_rows) a.d.group_rows_clamp = d.group_stride;
  const long grid = (long)a.tiles_per_group * d.n_groups;
  SWN_CHECK(grid > 0 && grid < (1L << 31), "swn_mlp_chain: grid %ld out of range", grid);
  const int lds = ACT_BYTES + 2 * WBUF_BYTES + 1024;
  SWN_CHECK(d.tag >= 0 && d.tag <= 6, "swn_mlp_chain: tag %d not in [0,6]", d.tag);
  const void* fn = nullptr;
#define SWN_PICK(TAGV)                                                                         \
  case TAGV:                                                                                   \
    fn = d.dtype == SWN_BF16 ? (const void*)chain_kernel<bf16_t, TAGV> : (const void*)chain_kernel<float, TAGV>; \
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
  SWN_LAUNCH_CHECK();
  return 0;
}

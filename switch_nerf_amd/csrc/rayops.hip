// Per-RAY work of the training step as single launches (round 3).  The direction / appearance half of layer "2" is constant along a ray
// (models/nerf_moe.py:419-429: cat([h, embedding_dir(d), embedding_a(idx)]) feeds Linear "2"); model.py folds it into a per-ray bias
//     c_ray[n] = [PE(dir_n), emb[idx_n]] @ W2r + b2                       (N_rays x 75 x 128)
// and the loss of the step (runner.py:1099-1111, 646-658) is a reduction over N_rays x 3 values.  Both were ~15 small torch kernels
// (cat, index, addmm, sub, mul, mean, full ...) - nothing at 8192 x 256 points, but launches that the host pays for one by one when it
// is the bottleneck (the first process on a fresh box, the 1024 rays per GPU of an 8-GPU run).  The BACKWARD of the per-ray part stays
// in torch (addmm_ / sum / matmul / index_add_, ~35 us): a fused deterministic version measured 0.8 ms (an ordered reduction over 800
// rays per embedding row is a latency chain), an atomic one 0.12 ms - profiles/r03_experiments.md.
#include "common.hpp"

namespace swn {

// feat[n] = [pe_dir[n][0..in_dir), emb[idx[n]][0..app_dim)] (fp32), c_ray[n][j] = b2[j] + sum_k feat[n][k] w2r[k][j].  One block per ray.
template <typename T>
__global__ __launch_bounds__(256) void ray_feat_fwd_kernel(const T* __restrict__ pe_dir, int dir_stride, int in_dir, const float* __restrict__ emb,
                                                           int app_dim, const void* __restrict__ image_indices, int idx64, const float* __restrict__ w2r,
                                                           const float* __restrict__ b2, int n_rays, int h2, float* __restrict__ feat,
                                                           float* __restrict__ c_ray) {
  __shared__ float f[256];
  const int n = blockIdx.x, F = in_dir + app_dim;
  const long img = idx64 ? (long)((const long long*)image_indices)[n] : (long)((const int32_t*)image_indices)[n];
  for (int k = threadIdx.x; k < F; k += blockDim.x) {
    const float v = k < in_dir ? ElemIO<T>::ld(pe_dir + (long)n * dir_stride + k) : emb[img * app_dim + (k - in_dir)];
    f[k] = v;
    feat[(long)n * F + k] = v;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < h2; j += blockDim.x) {
    float acc = b2[j];
    for (int k = 0; k < F; ++k) acc = fmaf(f[k], w2r[(long)k * h2 + j], acc);
    c_ray[(long)n * h2 + j] = acc;
  }
}

// The loss of the step in ONE block: photo = mean((rgb - target)^2), gate_loss = mean(l_aux_a) [or the mean of both means],
// loss = photo + wt * gate_loss, psnr = -10 log10(photo); d_rgb = 2 (rgb - target) / (3 N) * scale, d_laux_* = wt * share / n_* * scale
// (scale = the loss scale of fp16 training, read from a device scalar, or 1).  out4 = {photo, gate_loss, loss, psnr}.
__global__ __launch_bounds__(1024) void step_loss_kernel(const float* __restrict__ rgb, const float* __restrict__ target, int n_vals,
                                                         const float* __restrict__ la, int na, const float* __restrict__ lb, int nb, float wt,
                                                         const float* __restrict__ scale_dev, float* __restrict__ d_rgb,
                                                         float* __restrict__ d_la, float* __restrict__ d_lb, float* __restrict__ out4) {
  __shared__ double red[1024];
  const float sc = scale_dev ? scale_dev[0] : 1.f;
  const float k = 2.0f / (float)n_vals * sc;
  double s = 0.0;
  for (int i = threadIdx.x; i < n_vals; i += 1024) {
    const float d = rgb[i] - target[i];
    s += (double)d * (double)d;
    d_rgb[i] = d * k;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float share = nb > 0 ? 0.5f : 1.f;
  for (int i = threadIdx.x; i < na; i += 1024) d_la[i] = wt * share / (float)na * sc;
  for (int i = threadIdx.x; i < nb; i += 1024) d_lb[i] = wt * share / (float)nb * sc;
  if (threadIdx.x == 0) {
    const float photo = (float)(red[0] / (double)n_vals);
    double ga = 0.0, gb = 0.0;
    for (int i = 0; i < na; ++i) ga += la[i];
    for (int i = 0; i < nb; ++i) gb += lb[i];
    float gate = (float)(ga / (double)na);
    if (nb > 0) gate = 0.5f * ((float)(gb / (double)nb) + gate);      // runner.py:1104-1111: (mean(fine) + mean(coarse)) / 2
    out4[0] = photo;
    out4[1] = gate;
    out4[2] = photo + wt * gate;
    out4[3] = -10.0f * log10f(photo);
  }
}

// d_emb[a][:] += sum over the rays n with image_indices[n] == a, in ascending n, of d_feat[n][:]   (the appearance embedding's gradient,
// nn.Embedding's backward: models/nerf_moe.py:215-222).  One block per embedding row: the block walks the ray indices in tiles of 1024,
// compacts the matching rays IN ORDER (wave scan of the per-thread match counts), and sums their rows with a fixed association
// (256 / app_dim interleaved partial sums per column, added in order) - the same bits on every run; an index_add with atomics adds in
// arrival order.  A training batch holds a few rays per image, so almost all the time is the scan of the (L2-resident) index array.
template <typename IT>
__global__ __launch_bounds__(256) void emb_grad_kernel(const float* __restrict__ d_feat, int ld, const IT* __restrict__ idx, int n_rays,
                                                       int app_dim, float* __restrict__ d_emb) {
  __shared__ int list[1024];
  __shared__ int wcnt[4];
  __shared__ float red[256];
  const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nsub = 256 / app_dim, sub = tid / app_dim, col = tid - sub * app_dim;
  float tot = 0.f;
  for (int sbase = 0; sbase < n_rays; sbase += 8192) {      // the index loads of 8 tiles are in flight together (one L2 round trip)
    uint32_t mbits = 0;
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = sbase + t8 * 1024 + 4 * tid + u;
        const bool hit = n < n_rays && (long)idx[n < n_rays ? n : 0] == (long)a;
        mbits |= (hit ? 1u : 0u) << (t8 * 4 + u);
      }
    for (int t8 = 0; t8 < 8; ++t8) {
      const int base = sbase + t8 * 1024;
      if (base >= n_rays) break;
      const uint32_t m4 = (mbits >> (t8 * 4)) & 15u;
      const int cnt = __popc(m4);
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wcnt[w] = incl;
      __syncthreads();
      int off = incl - cnt;
      for (int q = 0; q < w; ++q) off += wcnt[q];
      const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if ((m4 >> u) & 1u) list[off++] = base + 4 * tid + u;
      __syncthreads();
      if (total) {                                            // (the same for every thread)
        float s = 0.f;
        if (sub < nsub)
          for (int k = sub; k < total; k += nsub) s += d_feat[(size_t)list[k] * ld + col];
        red[tid] = s;
        __syncthreads();
        if (tid < app_dim)
          for (int q = 0; q < nsub; ++q) tot += red[q * app_dim + tid];
      }
      __syncthreads();
    }
  }
  if (tid < app_dim) d_emb[(size_t)a * app_dim + tid] += tot;
}

// Parameter gradients of the per-ray half of layer "2": d_w2r[k][j] += sum_n feat[n][k] dc_ray[n][j], d_b2[j] += sum_n dc_ray[n][j]
// (N_rays x (F = 75) x (H2 = 128): a GEMM whose only long dimension is the reduction - the library ran it on a handful of tiles, 57 us
// at 8192 rays, + 14 us for the column sums).  Split over the rays: block b sums RB rays into partial[b] (thread = column j and a
// contiguous range of rows k; the block's feat rows in LDS, zero-padded, read as 16-byte broadcasts; the thread's dc column in
// LDS too), ordered_reduce_kernel adds the blocks in a fixed order.  (First version: a guard per FMA on a per-thread row index and a
// dynamically indexed register array - 2000 exec-mask branches and scratch: 166 us.)
constexpr int RFW_RB = 16;       // rays per block
constexpr int RFW_KC = 8;        // rows (accumulators) per thread and pass
__global__ __launch_bounds__(256) void ray_feat_wgrad_kernel(const float* __restrict__ feat, const float* __restrict__ dc_ray, int n_rays, int F,
                                                             int F_pad, int H2, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float fs[];      // [RB][F_pad], zero beyond the valid rays / features: no guards in the loop
  const int b = blockIdx.x, r0 = b * RFW_RB, nr = min(RFW_RB, n_rays - r0);
  for (int i = threadIdx.x; i < RFW_RB * F_pad; i += 256) {
    const int r = i / F_pad, k = i - r * F_pad;
    fs[i] = (r < nr && k < F) ? feat[(long)(r0 + r) * F + k] : 0.f;
  }
  __syncthreads();
  const int KG = 256 / H2;                             // row groups (H2 = 64: 4, 128: 2, 256: 1); a wave lies inside one group
  const int j = threadIdx.x % H2, kg = __builtin_amdgcn_readfirstlane(threadIdx.x / H2);
  const int rpk = F_pad / KG;                          // rows of this group: [kg rpk, (kg + 1) rpk), a multiple of RFW_KC
  float* part = partial + (size_t)b * ((size_t)F * H2 + H2);
  // the block's dc rows go to LDS as well ([RB][H2] behind the feature rows; a register array would have to be indexed by constants,
  // i.e. a fully unrolled ray loop whose 128 hoisted LDS reads spill: measured 96 us): thread j reads its own column
  float* dcs = fs + RFW_RB * F_pad;
  for (int i = threadIdx.x; i < RFW_RB * H2; i += 256) {
    const int r = i / H2;
    dcs[i] = r < nr ? dc_ray[(long)r0 * H2 + i] : 0.f;
  }
  __syncthreads();
  for (int k0 = kg * rpk; k0 < (kg + 1) * rpk; k0 += RFW_KC) {
    float acc[RFW_KC];
#pragma unroll
    for (int i = 0; i < RFW_KC; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int r = 0; r < RFW_RB; ++r) {
      const float4 f0 = *(const float4*)(fs + r * F_pad + k0), f1 = *(const float4*)(fs + r * F_pad + k0 + 4);      // wave-uniform: broadcasts
      const float dc = dcs[r * H2 + j];
      acc[0] = fmaf(f0.x, dc, acc[0]); acc[1] = fmaf(f0.y, dc, acc[1]); acc[2] = fmaf(f0.z, dc, acc[2]); acc[3] = fmaf(f0.w, dc, acc[3]);
      acc[4] = fmaf(f1.x, dc, acc[4]); acc[5] = fmaf(f1.y, dc, acc[5]); acc[6] = fmaf(f1.z, dc, acc[6]); acc[7] = fmaf(f1.w, dc, acc[7]);
    }
#pragma unroll
    for (int i = 0; i < RFW_KC; ++i)
      if (k0 + i < F) part[(size_t)(k0 + i) * H2 + j] = acc[i];
  }
  if (kg == 0) {
    float sb = 0.f;
#pragma unroll 8
    for (int r = 0; r < RFW_RB; ++r) sb += dcs[r * H2 + j];
    part[(size_t)F * H2 + j] = sb;
  }
}

}  // namespace swn

using namespace swn;

extern "C" int swn_ray_feat_fwd(const void* pe_dir, int dtype, int dir_stride, int in_dir, const float* emb, int app_dim,
                                const void* image_indices, int indices_are_int64, const float* w2r, const float* b2, int n_rays, int h2,
                                float* feat, float* c_ray, void* stream) {
  SWN_CHECK(pe_dir && emb && image_indices && w2r && b2 && feat && c_ray, "swn_ray_feat_fwd: null pointer");
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_ray_feat_fwd: bad dtype %d", dtype);
  SWN_CHECK(n_rays > 0 && h2 > 0 && in_dir >= 0 && app_dim >= 0 && in_dir + app_dim > 0 && in_dir + app_dim <= 256,
            "swn_ray_feat_fwd: bad sizes (rays %d, h2 %d, features %d + %d)", n_rays, h2, in_dir, app_dim);
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((ray_feat_fwd_kernel<bf16_t>), dim3(n_rays), dim3(128), 0, as_stream(stream), (const bf16_t*)pe_dir, dir_stride, in_dir, emb,
                       app_dim, image_indices, indices_are_int64, w2r, b2, n_rays, h2, feat, c_ray);
  else
    hipLaunchKernelGGL((ray_feat_fwd_kernel<float>), dim3(n_rays), dim3(128), 0, as_stream(stream), (const float*)pe_dir, dir_stride, in_dir, emb,
                       app_dim, image_indices, indices_are_int64, w2r, b2, n_rays, h2, feat, c_ray);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_step_loss(const float* rgb, const float* target, int n_values, const float* l_aux_a, int n_a, const float* l_aux_b,
                             int n_b, float l_aux_weight, const float* loss_scale_dev, float* d_rgb, float* d_l_aux_a, float* d_l_aux_b,
                             float* out4, void* stream) {
  SWN_CHECK(rgb && target && l_aux_a && d_rgb && d_l_aux_a && out4, "swn_step_loss: null pointer");
  SWN_CHECK(n_values > 0 && n_a > 0 && n_b >= 0 && (n_b == 0 || (l_aux_b && d_l_aux_b)), "swn_step_loss: bad sizes");
  hipLaunchKernelGGL(step_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), rgb, target, n_values, l_aux_a, n_a, l_aux_b, n_b,
                     l_aux_weight, loss_scale_dev, d_rgb, d_l_aux_a, d_l_aux_b, out4);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_emb_grad(const float* d_feat, int ld, const void* image_indices, int indices_are_int64, int n_rays, int app_dim,
                            int n_images, float* d_emb, void* stream) {
  SWN_CHECK(d_feat && image_indices && d_emb, "swn_emb_grad: null pointer");
  SWN_CHECK(n_rays >= 0 && n_images > 0 && app_dim > 0 && app_dim <= 256 && ld >= app_dim, "swn_emb_grad: bad sizes (rays %d, images %d, width %d)",
            n_rays, n_images, app_dim);
  if (n_rays == 0) return 0;
  if (indices_are_int64)
    hipLaunchKernelGGL((emb_grad_kernel<int64_t>), dim3(n_images), dim3(256), 0, as_stream(stream), d_feat, ld, (const int64_t*)image_indices,
                       n_rays, app_dim, d_emb);
  else
    hipLaunchKernelGGL((emb_grad_kernel<int32_t>), dim3(n_images), dim3(256), 0, as_stream(stream), d_feat, ld, (const int32_t*)image_indices,
                       n_rays, app_dim, d_emb);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t swn_ray_feat_wgrad_workspace_bytes(int n_rays, int n_feat, int h2) {
  return (size_t)cdiv(n_rays > 0 ? n_rays : 1, RFW_RB) * ((size_t)n_feat * h2 + h2) * sizeof(float);
}

extern "C" int swn_ray_feat_wgrad(const float* feat, const float* dc_ray, int n_rays, int n_feat, int h2, float* d_w2r, float* d_b2,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  SWN_CHECK(feat && dc_ray && d_w2r && d_b2 && workspace, "swn_ray_feat_wgrad: null pointer");
  SWN_CHECK(n_rays >= 0 && n_feat > 0 && n_feat <= 256 && (h2 == 64 || h2 == 128 || h2 == 256), "swn_ray_feat_wgrad: bad sizes (rays %d, features %d, h2 %d)",
            n_rays, n_feat, h2);
  SWN_CHECK(workspace_bytes >= swn_ray_feat_wgrad_workspace_bytes(n_rays, n_feat, h2), "swn_ray_feat_wgrad: workspace of %zu bytes, need %zu",
            workspace_bytes, swn_ray_feat_wgrad_workspace_bytes(n_rays, n_feat, h2));
  if (n_rays == 0) return 0;
  const int nb = cdiv(n_rays, RFW_RB);
  const int unit = (256 / h2) * RFW_KC, f_pad = cdiv(n_feat, unit) * unit;      // every row group gets a whole number of passes
  const size_t lds = (size_t)RFW_RB * (f_pad + h2) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)ray_feat_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  SWN_CHECK(e == hipSuccess, "swn_ray_feat_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(ray_feat_wgrad_kernel, dim3(nb), dim3(256), lds, as_stream(stream), feat, dc_ray, n_rays, n_feat, f_pad, h2, (float*)workspace);
  OrdDst od{{d_w2r, d_b2, nullptr, nullptr}, {n_feat * h2, h2, 0, 0}};
  ordered_reduce_async((const float*)workspace, nb, n_feat * h2 + h2, od, true, as_stream(stream));
  SWN_LAUNCH_CHECK();
  return 0;
}

// ---- sign bits of a 16-bit activation matrix (expert parallelism with the tail on the expert's rank, ep_owner.py) ------------------------
// The per-ray bias gradient needs of layer "2"'s output only its ReLU mask: the owner returns 1 bit per feature (16 bytes per token at 128
// features) beside raw, the source rebuilds a 0 / 1 stand-in for swn_heads_bwd.  bit j of word q of a row = (h[row][32 q + j] > 0).
namespace swn {
__global__ __launch_bounds__(256) void sign_bits_pack_kernel(const bf16_t* __restrict__ h, long n_words, uint32_t* __restrict__ bits) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_words) return;
  const uint4* p = (const uint4*)(h + t * 32);
  uint32_t w = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = p[q];
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w |= (bf16_to_f32((bf16_t)(u[i] & 0xFFFFu)) > 0.f ? 1u : 0u) << (q * 8 + 2 * i);
      w |= (bf16_to_f32((bf16_t)(u[i] >> 16)) > 0.f ? 1u : 0u) << (q * 8 + 2 * i + 1);
    }
  }
  bits[t] = w;
}
__global__ __launch_bounds__(256) void sign_bits_unpack_kernel(const uint32_t* __restrict__ bits, long n_words, bf16_t* __restrict__ h) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_words) return;
  const uint32_t w = bits[t];
  const uint32_t one = SWN_HALF_ONE_X2 & 0xFFFFu;
  uint4* p = (uint4*)(h + t * 32);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      u[i] = (((w >> (q * 8 + 2 * i)) & 1u) ? one : 0u) | (((w >> (q * 8 + 2 * i + 1)) & 1u) ? (one << 16) : 0u);
    p[q] = make_uint4(u[0], u[1], u[2], u[3]);
  }
}
}  // namespace swn

extern "C" int swn_sign_bits_pack(const void* h, long rows, int features, uint32_t* bits, void* stream) {
  SWN_CHECK(h && bits, "swn_sign_bits_pack: null pointer");
  SWN_CHECK(rows >= 0 && features > 0 && features % 32 == 0, "swn_sign_bits_pack: %ld rows of %d features (a multiple of 32)", rows, features);
  const long n = rows * (features / 32);
  if (n == 0) return 0;
  hipLaunchKernelGGL(swn::sign_bits_pack_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), (const bf16_t*)h, n, bits);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_sign_bits_unpack(const uint32_t* bits, long rows, int features, void* h, void* stream) {
  SWN_CHECK(h && bits, "swn_sign_bits_unpack: null pointer");
  SWN_CHECK(rows >= 0 && features > 0 && features % 32 == 0, "swn_sign_bits_unpack: %ld rows of %d features (a multiple of 32)", rows, features);
  const long n = rows * (features / 32);
  if (n == 0) return 0;
  hipLaunchKernelGGL(swn::sign_bits_unpack_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), bits, n, (bf16_t*)h);
  SWN_LAUNCH_CHECK();
  return 0;
}

// ---- small index kernels of the owner-tail exchange (ep_owner.py) --------------------------------------------------------------------------
namespace swn {
// dst[index[r]] = src[r] (rows of row_bytes, a multiple of 4; index < 0: the row goes nowhere).  The mirror of gather_rows_kernel: what came
// home in send order goes back to token order (every token at most once: no two rows meet).
template <typename V>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const char* __restrict__ src, const int32_t* __restrict__ index, long n_rows,
                                                           int row_bytes, char* __restrict__ dst) {
  const int cpr = row_bytes / (int)sizeof(V);
  const long total = n_rows * cpr;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < total; c += (long)gridDim.x * 256) {
    const long r = c / cpr;
    const int ch = (int)(c - r * cpr);
    const int d_ = index[r];
    if (d_ >= 0) *(V*)(dst + (long)d_ * row_bytes + (long)ch * sizeof(V)) = *(const V*)(src + c * (long)sizeof(V));
  }
}
// aux[r] = (gate[tok], bits of (tok / rows_per_ray + ray_base), noise[tok] or 0, 0), tok = index[r]: the 16-byte record that travels with a
// kept token's row to its expert's rank (zero_gate: the rank's own dropped tokens - their gate value is never used, keep it defined)
__global__ __launch_bounds__(256) void owner_aux_kernel(const float* __restrict__ gate, const float* __restrict__ noise,
                                                        const int32_t* __restrict__ index, long n, int rows_per_ray, int ray_base, int zero_gate,
                                                        float4* __restrict__ aux) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const int tok = index[r];
  float4 v;
  v.x = zero_gate ? 0.f : gate[tok];
  v.y = __int_as_float(tok / rows_per_ray + ray_base);
  v.z = noise ? noise[tok] : 0.f;
  v.w = 0.f;
  aux[r] = v;
}
// the records of an owner's token space -> the planar operands of the fused launch
__global__ __launch_bounds__(256) void owner_aux_split_kernel(const float4* __restrict__ aux, long n, float* __restrict__ gate, int32_t* __restrict__ ray,
                                                              float* __restrict__ noise) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float4 v = aux[r];
  gate[r] = v.x;
  ray[r] = __float_as_int(v.y);
  if (noise) noise[r] = v.z;
}
}  // namespace swn

extern "C" int swn_scatter_rows(const void* src, const int32_t* index, long n_rows, int row_bytes, void* dst, void* stream) {
  SWN_CHECK(src && index && dst, "swn_scatter_rows: null pointer");
  SWN_CHECK(n_rows >= 0 && row_bytes > 0 && row_bytes % 4 == 0, "swn_scatter_rows: %ld rows of %d bytes (a multiple of 4)", n_rows, row_bytes);
  if (n_rows == 0) return 0;
  const bool wide = row_bytes % 16 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
  long blocks = cdiv(n_rows * (row_bytes / (wide ? 16 : 4)), 256);
  if (blocks > 16384) blocks = 16384;
  if (wide)
    hipLaunchKernelGGL(swn::scatter_rows_kernel<uint4>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const char*)src, index, n_rows,
                       row_bytes, (char*)dst);
  else
    hipLaunchKernelGGL(swn::scatter_rows_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const char*)src, index, n_rows,
                       row_bytes, (char*)dst);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_owner_aux(const float* gate, const float* noise, const int32_t* index, long n, int rows_per_ray, int ray_base, int zero_gate,
                             float* aux, void* stream) {
  SWN_CHECK(gate && index && aux, "swn_owner_aux: null pointer");
  SWN_CHECK(n >= 0 && rows_per_ray > 0 && ray_base >= 0, "swn_owner_aux: %ld tokens, %d rows per ray, first ray %d", n, rows_per_ray, ray_base);
  if (n == 0) return 0;
  hipLaunchKernelGGL(swn::owner_aux_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), gate, noise, index, n, rows_per_ray,
                     ray_base, zero_gate, (float4*)aux);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_owner_aux_split(const float* aux, long n, float* gate, int32_t* ray, float* noise, void* stream) {
  SWN_CHECK(aux && gate && ray, "swn_owner_aux_split: null pointer");
  SWN_CHECK(n >= 0, "swn_owner_aux_split: %ld tokens", n);
  if (n == 0) return 0;
  hipLaunchKernelGGL(swn::owner_aux_split_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), (const float4*)aux, n, gate, ray, noise);
  SWN_LAUNCH_CHECK();
  return 0;
}

// ---- the per-ray bias gradient from sign bits (owner-tail expert parallelism, ep_owner.py) ---------------------------------------------------
// dc_ray[ray][f] = sum over the ray's tokens of dh2[token][f], dh2 = (h2 > 0) * (dc0 wc[0][f] + dc1 wc[1][f] + dc2 wc[2][f]) rounded to the
// 16-bit type (what swn_heads_bwd stores and sums: elementwise.hip heads_bwd_kernel), dc_c = d_raw_c raw_c (1 - raw_c) - with (h2 > 0) given as
// the bits swn_sign_bits_pack made on the expert's rank.  One workgroup per ray, a thread per feature; 48 bytes per token.
namespace swn {
__global__ __launch_bounds__(256) void ray_bias_grad_bits_kernel(const uint32_t* __restrict__ bits, const float* __restrict__ raw,
                                                                 const float* __restrict__ d_raw, const float* __restrict__ wc, int rows_per_ray,
                                                                 int features, float* __restrict__ dc_ray) {
  // a ray's tokens in rounds of 256: the workgroup fetches their raw / d_raw / bit words once (coalesced), leaves (dc0, dc1, dc2) and the words in
  // LDS, then thread f walks the tokens (broadcast LDS reads) - one memory round trip per 256 tokens instead of one per token
  __shared__ float4 dcs[256];
  __shared__ uint32_t wds[256 * 8];
  const int tid = threadIdx.x, f = tid;
  const long ray = blockIdx.x;
  const int wpr = features >> 5;                 // bit words per token
  const bool mine = f < features;
  const float w0 = mine ? wc[f] : 0.f, w1 = mine ? wc[features + f] : 0.f, w2 = mine ? wc[2 * features + f] : 0.f;
  const long t0 = ray * rows_per_ray;
  float acc = 0.f;
  for (int s0 = 0; s0 < rows_per_ray; s0 += 256) {
    const int n = min(256, rows_per_ray - s0);
    if (tid < n) {
      const float4 r = *(const float4*)(raw + (t0 + s0 + tid) * 4), d = *(const float4*)(d_raw + (t0 + s0 + tid) * 4);
      dcs[tid] = make_float4(d.x * r.x * (1.f - r.x), d.y * r.y * (1.f - r.y), d.z * r.z * (1.f - r.z), 0.f);
    }
    for (int i = tid; i < n * wpr; i += 256) wds[i] = bits[(t0 + s0) * wpr + i];
    __syncthreads();
    if (mine) {
      const int wq = f >> 5, sh = f & 31;
#pragma unroll 4
      for (int s = 0; s < n; ++s) {
        const float4 dc = dcs[s];
        const float gg = dc.x * w0 + dc.y * w1 + dc.z * w2;
        acc += ((wds[s * wpr + wq] >> sh) & 1u) ? bf16_to_f32(f32_to_bf16(gg)) : 0.f;
      }
    }
    __syncthreads();
  }
  if (mine) dc_ray[ray * features + f] = acc;
}
}  // namespace swn

extern "C" int swn_ray_bias_grad_bits(const uint32_t* bits, const float* raw, const float* d_raw, const float* w_color, int n_rays, int rows_per_ray,
                                      int features, float* dc_ray, void* stream) {
  SWN_CHECK(bits && raw && d_raw && w_color && dc_ray, "swn_ray_bias_grad_bits: null pointer");
  SWN_CHECK(n_rays >= 0 && rows_per_ray > 0 && features > 0 && features % 32 == 0 && features <= 256,
            "swn_ray_bias_grad_bits: %d rays x %d rows, %d features (a multiple of 32 up to 256)", n_rays, rows_per_ray, features);
  if (n_rays == 0) return 0;
  hipLaunchKernelGGL(swn::ray_bias_grad_bits_kernel, dim3((unsigned)n_rays), dim3(256), 0, as_stream(stream), bits,
                     raw, d_raw, w_color, rows_per_ray, features, dc_ray);
  SWN_LAUNCH_CHECK();
  return 0;
}

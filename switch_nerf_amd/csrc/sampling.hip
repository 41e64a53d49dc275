// Hierarchical sampling kernels:
//   swn_sample_pdf    - _sample_pdf / _sample_cdf          (/root/reference/switch_nerf/rendering.py:587-637)
//   swn_merge_samples - sort(cat[z_fine, z_coarse]) + gather of the raw (rgb, sigma) outputs   (rendering.py:419-433)
//   swn_unmerge_grad  - backward of that gather
// One workgroup per ray; the per-ray arrays (<= 1024 floats) live in LDS.
#include "common.hpp"

namespace swn {

// pdf = (w + 1e-8) / sum; cdf = [0, cumsum(pdf)]; inds = searchsorted(cdf, u, right=True); lerp inside the bin.
// z [N,S] coarse depths, w [N,S] coarse weights: bins = mid points of z (S-1), weights used = w[1:-1] (S-2).
__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                         const float* __restrict__ u_in, int N, int S, int F,
                                                         float* __restrict__ z_fine) {
  __shared__ float bins[1024];
  __shared__ float cdf[1024];
  __shared__ float red[256];
  const int ray = blockIdx.x, t = threadIdx.x;
  const float* zr = z + (long)ray * S;
  const float* wr = w + (long)ray * S;
  const int nb = S - 1;   // bins
  const int nw = S - 2;   // weights
  for (int i = t; i < nb; i += 256) bins[i] = 0.5f * (zr[i] + zr[i + 1]);          // rendering.py:238
  float part = 0.f;
  for (int i = t; i < nw; i += 256) part += wr[i + 1] + 1e-8f;                      // :599
  red[t] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float tot = red[0];
  if (t == 0) {   // sequential cumsum like torch.cumsum (:603); nw <= 1022
    float run = 0.f;
    cdf[0] = 0.f;
    for (int i = 0; i < nw; ++i) {
      run += (wr[i + 1] + 1e-8f) / tot;
      cdf[i + 1] = run;
    }
  }
  __syncthreads();
  for (int j = t; j < F; j += 256) {
    // det: torch.linspace(0, 1, F) (symmetric formula of torch: start + i*step for the first half, end - (F-1-i)*step after)
    float u;
    if (u_in) u = u_in[(long)ray * F + j];
    else {
      const float step = 1.f / (float)(F - 1);
      u = (j < F / 2) ? (float)j * step : 1.f - (float)(F - 1 - j) * step;
    }
    // searchsorted(cdf[0..nw], u, right=True): first index with cdf[idx] > u
    int lo = 0, hi = nw + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    const int below = max(lo - 1, 0), above = min(lo, nw);                             // :621-622
    const float cb = cdf[below], ca = cdf[above];
    const float bb = bins[below], ba = bins[above];
    float denom = ca - cb;
    if (denom < 1e-8f) denom = 1.f;                                                    // :631
    z_fine[(long)ray * F + j] = bb + (u - cb) / denom * (ba - bb);                     // :636
  }
}

// Bitonic sort of (key, index) pairs in LDS, ascending; n_pad = power of two >= n, padded with +inf.
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ z_fine, const float* __restrict__ z_coarse,
                                                    const float* __restrict__ raw_fine, const float* __restrict__ raw_coarse,
                                                    int N, int F, int S, int n_pad, float* __restrict__ z_out,
                                                    int32_t* __restrict__ order, float* __restrict__ raw_out) {
  __shared__ float key[1024];
  __shared__ int idx[1024];
  const int ray = blockIdx.x, t = threadIdx.x, T = F + S;
  for (int i = t; i < n_pad; i += 256) {
    float k = __builtin_inff();
    if (i < F) k = z_fine[(long)ray * F + i];
    else if (i < T) k = z_coarse[(long)ray * S + (i - F)];
    key[i] = k;
    idx[i] = i;
  }
  __syncthreads();
  for (int size = 2; size <= n_pad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < n_pad / 2; i += 256) {
        const int lo = (i / stride) * stride * 2 + (i % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const float a = key[lo], b = key[hi];
        const int ia = idx[lo], ib = idx[hi];
        // order by (key, index): a stable total order, so that equal depths keep cat-order (fine before coarse)
        const bool gt = (a > b) || (a == b && ia > ib);
        if (gt == up) { key[lo] = b; key[hi] = a; idx[lo] = ib; idx[hi] = ia; }
      }
      __syncthreads();
    }
  }
  for (int i = t; i < T; i += 256) {
    const int src = idx[i];
    z_out[(long)ray * T + i] = key[i];
    order[(long)ray * T + i] = src;
    const float4 v = (src < F) ? *(const float4*)(raw_fine + ((long)ray * F + src) * 4)
                               : *(const float4*)(raw_coarse + ((long)ray * S + (src - F)) * 4);
    *(float4*)(raw_out + ((long)ray * T + i) * 4) = v;
  }
}

__global__ void unmerge_kernel(const float* __restrict__ d_raw, const int32_t* __restrict__ order, int N, int F, int S,
                               float* __restrict__ d_fine, float* __restrict__ d_coarse) {
  const int T = F + S;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * T) return;
  const long ray = i / T;
  const int src = order[i];
  const float4 v = *(const float4*)(d_raw + i * 4);
  if (src < F) *(float4*)(d_fine + (ray * F + src) * 4) = v;
  else *(float4*)(d_coarse + (ray * S + (src - F)) * 4) = v;
}

}  // namespace swn

using namespace swn;

extern "C" int swn_sample_pdf(const float* z_coarse, const float* weights, const float* u, int n_rays, int n_coarse,
                              int n_fine, float* z_fine, void* stream) {
  SWN_CHECK(z_coarse && weights && z_fine, "swn_sample_pdf: null pointer");
  SWN_CHECK(n_coarse >= 3 && n_coarse <= 1024 && n_fine >= 2, "swn_sample_pdf: 3 <= coarse samples <= 1024, fine >= 2");
  hipLaunchKernelGGL(sample_pdf_kernel, dim3(n_rays), dim3(256), 0, as_stream(stream), z_coarse, weights, u, n_rays, n_coarse,
                     n_fine, z_fine);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_merge_samples(const float* z_fine, const float* z_coarse, const float* raw_fine, const float* raw_coarse,
                                 int n_rays, int n_fine, int n_coarse, float* z_out, int32_t* order, float* raw_out,
                                 void* stream) {
  SWN_CHECK(z_fine && z_coarse && raw_fine && raw_coarse && z_out && order && raw_out, "swn_merge_samples: null pointer");
  const int T = n_fine + n_coarse;
  SWN_CHECK(T >= 2 && T <= 1024, "swn_merge_samples: fine + coarse samples must be <= 1024");
  int n_pad = 2;
  while (n_pad < T) n_pad <<= 1;
  hipLaunchKernelGGL(merge_kernel, dim3(n_rays), dim3(256), 0, as_stream(stream), z_fine, z_coarse, raw_fine, raw_coarse, n_rays,
                     n_fine, n_coarse, n_pad, z_out, order, raw_out);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_unmerge_grad(const float* d_raw, const int32_t* order, int n_rays, int n_fine, int n_coarse, float* d_fine,
                                float* d_coarse, void* stream) {
  SWN_CHECK(d_raw && order && d_fine && d_coarse, "swn_unmerge_grad: null pointer");
  const long total = (long)n_rays * (n_fine + n_coarse);
  hipLaunchKernelGGL(unmerge_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), d_raw, order, n_rays, n_fine,
                     n_coarse, d_fine, d_coarse);
  SWN_LAUNCH_CHECK();
  return 0;
}

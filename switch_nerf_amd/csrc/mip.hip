// mip path (BASELINE configs[3] family): conical-frustum casting + integrated positional
// encoding, and the level hand-over (blurred weights -> piecewise-constant pdf -> fine edges).
//   swn_sample_z     - the interval edges of a level        (/root/reference/switch_nerf/rendering_mip.py:153-157, rendering.py:573-584)
//   swn_mip_encode   - mip_cast_rays + MipEmbedder          (rendering_mip.py:15-25; models/nerf.py:28-56)
//   swn_mip_resample - weights blur / padding + sorted_piecewise_constant_pdf1 (rendering_mip.py:206-223, :75-131)
#include "common.hpp"

namespace swn {

__device__ __forceinline__ float mip_z_of(float near, float far, float t) {
#pragma clang fp contract(off)
  const float a = near * (1.f - t);
  const float b = far * t;
  return a + b;
}

__global__ void sample_z_kernel(const float* __restrict__ rays, const float* __restrict__ tsteps, const float* __restrict__ prand,
                                float perturb, int n_rays, int S, float* __restrict__ z_out) {
#pragma clang fp contract(off)
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)n_rays * S) return;
  const int ray = (int)(p / S), s = (int)(p - (long)ray * S);
  const float near = rays[(long)ray * 8 + 6], far = rays[(long)ray * 8 + 7];
  float z = mip_z_of(near, far, tsteps[s]);
  if (perturb > 0.f && prand) {   // rendering.py:573-584 (_expand_and_perturb_z_vals)
    const float zp = s > 0 ? mip_z_of(near, far, tsteps[s - 1]) : z;
    const float zn = s < S - 1 ? mip_z_of(near, far, tsteps[s + 1]) : z;
    const float lower = s > 0 ? 0.5f * (zp + z) : z;
    const float upper = s < S - 1 ? 0.5f * (z + zn) : z;
    const float span = (upper - lower) * (perturb * prand[p]);
    z = lower + span;
  }
  z_out[p] = z;
}

// One thread per frustum (ray, interval i): edges z[i], z[i+1] -> mean / diagonal covariance -> integrated encoding
// [mean, sin(2^k mean) exp(-4^k var / 2), cos(2^k mean) exp(-4^k var / 2), k < L], zero-padded to pe_stride; the block's rows
// are staged in LDS and written as one contiguous region (as in sample_pe_kernel).
template <typename T, int LMAX>
__global__ __launch_bounds__(128) void mip_encode_kernel(const float* __restrict__ rays, const float* __restrict__ radii,
                                                         const float* __restrict__ z, int n_rays, int S, int L,
                                                         T* __restrict__ pe, int pe_stride) {
#pragma clang fp contract(off)
  constexpr int NT = sizeof(T) == 2 ? 128 : 64;
  constexpr int ROWB = 128 * (int)sizeof(T) + 16;
  __shared__ __attribute__((aligned(16))) char stage[NT * ROWB];
  const int S1 = S - 1;
  const long total = (long)n_rays * S1;
  const long p_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p_raw < total;
  const long p = live ? p_raw : total - 1;
  const int ray = (int)(p / S1), i = (int)(p - (long)ray * S1);
  const float* r = rays + (long)ray * 8;
  const float t0 = z[(long)ray * S + i], t1 = z[(long)ray * S + i + 1];
  const float rad = radii[ray];
  // rendering_mip.py:16-21
  const float c = (t0 + t1) / 2.f, d = (t1 - t0) / 2.f;
  const float c2 = c * c, d2 = d * d, d4 = d2 * d2;
  const float den = 3.f * c2 + d2;
  const float t_mean = c + (2.f * c * d2) / den;
  const float t_var = d2 / 3.f - (4.f / 15.f) * ((d4 * (12.f * c2 - d2)) / (den * den));
  const float r_var = (rad * rad) * (c2 / 4.f + (5.f / 12.f) * d2 - (4.f / 15.f) * d4 / den);
  const float dd[3] = {r[3] * r[3], r[4] * r[4], r[5] * r[5]};
  const float dsum = (dd[0] + dd[1]) + dd[2];
  float v[8 + 6 * LMAX + 8];
  float mean[3], cov[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mean[a] = r[a] + r[3 + a] * t_mean;
    cov[a] = t_var * dd[a] + r_var * (1.f - dd[a] / dsum);
    v[a] = mean[a];
  }
  if constexpr (sizeof(T) == 4) {   // fp32 (parity mode): every octave from its own argument
    float fy = 1.f, fw = 1.f;
#pragma unroll
    for (int k = 0; k < LMAX; ++k) {
      if (k < L) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float sn, cs;
          sincosf(mean[a] * fy, &sn, &cs);
          const float damp = expf((-0.5f * fw) * cov[a]);
          v[3 + 6 * k + a] = sn * damp;
          v[3 + 6 * k + 3 + a] = cs * damp;
        }
      }
      fy *= 2.f;
      fw *= 4.f;
    }
  } else {   // bf16 output: one sincos per coordinate + angle doubling (see sample_pe_kernel); the damping needs its own exp
             // per octave (repeated squaring would quadruple the relative error each octave)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float sn, cs;
      sincosf(mean[a], &sn, &cs);
      float fw = 1.f;
#pragma unroll
      for (int k = 0; k < LMAX; ++k) {
        if (k < L) {
          const float damp = __expf((-0.5f * fw) * cov[a]);
          v[3 + 6 * k + a] = sn * damp;
          v[3 + 6 * k + 3 + a] = cs * damp;
        }
        const float s2 = 2.f * sn * cs, c2_ = 1.f - 2.f * sn * sn;
        sn = s2;
        cs = c2_;
        fw *= 4.f;
      }
    }
  }
  const int used = 3 + 6 * L;
  const int step = 16 / (int)sizeof(T);
  const bool staged = pe_stride <= 128;
  T* dst = staged ? (T*)(stage + threadIdx.x * ROWB) : pe + p * pe_stride;
  if (live || staged) {
    for (int c0 = 0; c0 < pe_stride; c0 += step) {
      float tmp[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) tmp[j] = 0.f;
      for (int j = 0; j < step; ++j) {
        const int cc = c0 + j;
        tmp[j] = cc < used ? v[cc] : 0.f;
      }
      if (live || staged) {
        if constexpr (sizeof(T) == 2) {
          uint4 u;
          u.x = pack_bf16x2(tmp[0], tmp[1]); u.y = pack_bf16x2(tmp[2], tmp[3]);
          u.z = pack_bf16x2(tmp[4], tmp[5]); u.w = pack_bf16x2(tmp[6], tmp[7]);
          *(uint4*)(dst + c0) = u;
        } else {
          *(float4*)(dst + c0) = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    const int cpr = pe_stride / step;
    const long row0 = (long)blockIdx.x * NT;
    const long rows = min((long)NT, total - row0);
    char* out = (char*)(pe + row0 * pe_stride);
    for (int cidx = threadIdx.x; cidx < rows * cpr; cidx += NT) {
      const int row = cidx / cpr, ch = cidx - row * cpr;
      *(uint4*)(out + (long)cidx * 16) = *(const uint4*)(stage + row * ROWB + ch * 16);
    }
  }
}

// One workgroup per ray.  z [N,S] edges, w [N,S-1] weights of the level -> F new edges (non-decreasing: u is increasing and
// the inverse cdf is monotone, so the reference's torch.sort afterwards is the identity).
__global__ __launch_bounds__(256) void mip_resample_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                           const float* __restrict__ u_rand, float padding, int N, int S, int F,
                                                           float* __restrict__ z_out) {
  __shared__ float wb[1056];    // blurred, padded weights (S - 1)
  __shared__ float cdf[1056];   // S entries: 0, clipped cumsum, 1
  __shared__ float red[256];
  const int ray = blockIdx.x, t = threadIdx.x, n = S - 1;
  const float* wr = w + (long)ray * n;
  const float* zr = z + (long)ray * S;
  // weights_pad = [w0, w, w_last]; weights_max = max of neighbours; blur = mean of consecutive maxima (+ padding)  (:207-216)
  float part = 0.f;
  for (int i = t; i < n; i += 256) {
    const float a = wr[max(i - 1, 0)], b = wr[i], c = wr[min(i + 1, n - 1)];
    const float m0 = fmaxf(a, b), m1 = fmaxf(b, c);
    const float v = 0.5f * (m0 + m1) + padding;
    wb[i] = v;
    part += v;
  }
  red[t] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  float wsum = red[0];
  const float pad = fmaxf(0.f, 1e-5f - wsum);           // :83-86
  wsum += pad;
  const float padn = pad / (float)n;
  if (t == 0) {   // sequential cumsum like torch.cumsum (n <= 1024)
    float run = 0.f;
    cdf[0] = 0.f;
    for (int i = 0; i < n - 1; ++i) {
      run += (wb[i] + padn) / wsum;
      cdf[i + 1] = fminf(run, 1.f);                      // :91-92
    }
    cdf[n] = 1.f;
  }
  __syncthreads();
  const float eps = 1.1920928955078125e-07f;             // torch.finfo(float32).eps
  const float s_ = 1.f / (float)F;
  for (int j = t; j < F; j += 256) {
    float u;
    if (u_rand) {                                        // :100-108
      u = (float)j * s_ + u_rand[(long)ray * F + j] * (s_ - eps);
      u = fminf(u, 1.f - eps);
    } else {                                             // torch.linspace(0, 1 - eps, F)
      const float end = 1.f - eps, stp = end / (float)(F - 1);
      u = (j < F / 2) ? (float)j * stp : end - (float)(F - 1 - j) * stp;
      if (F == 1) u = 0.f;
    }
    // last edge with cdf <= u: first index with cdf > u, minus one (cdf[0] = 0 <= u < 1 = cdf[n])
    int lo = 0, hi = n + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    const int i0 = lo - 1, i1 = lo;
    const float c0 = cdf[i0], c1 = cdf[i1];
    float tt = (u - c0) / (c1 - c0);
    if (!(tt == tt)) tt = 0.f;                           // nan_to_num
    tt = fminf(fmaxf(tt, 0.f), 1.f);
    const float b0 = zr[i0], b1 = zr[i1];
    z_out[(long)ray * F + j] = b0 + tt * (b1 - b0);
  }
}

}  // namespace swn

using namespace swn;

extern "C" int swn_sample_z(const float* rays, const float* t_steps, const float* perturb_rand, float perturb, int n_rays,
                            int n_samples, float* z_out, void* stream) {
  SWN_CHECK(rays && t_steps && z_out, "swn_sample_z: null pointer");
  SWN_CHECK(n_rays > 0 && n_samples > 0, "swn_sample_z: bad sizes");
  const long P = (long)n_rays * n_samples;
  hipLaunchKernelGGL(sample_z_kernel, dim3(cdiv(P, 256)), dim3(256), 0, as_stream(stream), rays, t_steps, perturb_rand, perturb, n_rays,
                     n_samples, z_out);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_mip_encode(const float* rays, const float* radii, const float* z, int n_rays, int n_edges, int l_xyz, int dtype,
                              void* pe, int pe_stride, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_mip_encode: bad dtype");
  SWN_CHECK(rays && radii && z && pe, "swn_mip_encode: null pointer");
  SWN_CHECK(n_rays > 0 && n_edges >= 2 && l_xyz >= 0 && l_xyz <= 12, "swn_mip_encode: bad sizes");
  const int epc = dtype == SWN_HALF ? 8 : 4;
  SWN_CHECK(pe_stride >= 3 + 6 * l_xyz && pe_stride % epc == 0, "swn_mip_encode: pe_stride %d too small / unaligned", pe_stride);
  const long P = (long)n_rays * (n_edges - 1);
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((mip_encode_kernel<bf16_t, 12>), dim3(cdiv(P, 128)), dim3(128), 0, as_stream(stream), rays, radii, z, n_rays,
                       n_edges, l_xyz, (bf16_t*)pe, pe_stride);
  else
    hipLaunchKernelGGL((mip_encode_kernel<float, 12>), dim3(cdiv(P, 64)), dim3(64), 0, as_stream(stream), rays, radii, z, n_rays,
                       n_edges, l_xyz, (float*)pe, pe_stride);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_mip_resample(const float* z, const float* weights, const float* u_rand, float padding, int n_rays, int n_edges,
                                int n_fine, float* z_out, void* stream) {
  SWN_CHECK(z && weights && z_out, "swn_mip_resample: null pointer");
  SWN_CHECK(n_edges >= 3 && n_edges <= 1025 && n_fine >= 1, "swn_mip_resample: 3 <= edges <= 1025, fine >= 1");
  hipLaunchKernelGGL(mip_resample_kernel, dim3(n_rays), dim3(256), 0, as_stream(stream), z, weights, u_rand, padding, n_rays, n_edges,
                     n_fine, z_out);
  SWN_LAUNCH_CHECK();
  return 0;
}

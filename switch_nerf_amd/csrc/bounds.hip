// Foreground bound + background sampling of render_rays (/root/reference/switch_nerf/rendering.py:32-78, 497-570):
//   fg_bounds_kernel     where a ray leaves the (ellipsoidal) foreground bound, which rays continue into the background
//                        model, the clipped far plane and the last sample's delta
//   bg_sample_pe_kernel  the background's inverse-distance samples (stratified, flipped), their NeRF++ inverted-sphere
//                        points (unit-sphere point + 1/r), metric depths, and the 4-D positional encoding
// fp32 arithmetic in the reference's operation order (fma contraction off), one thread per ray / per sample.
#include "common.hpp"
#include "pe_store.hpp"

namespace swn {

struct Bound {       // (o - center) / radius, d / radius; has_radius = 0: unit sphere at the origin (sphere_radius None)
  float cx, cy, cz, rx, ry, rz;
  int has_radius;
};

struct RayGeom {     // per-ray part of _intersect_sphere / _depth2pts_outside
  float o[3], d[3];  // normalised to the unit sphere
  float d1, pn, dd;  // depth of the ray's closest point to the centre p_mid, |p_mid|^2, |d|^2
};

__device__ __forceinline__ float sum3(float a, float b, float c) {
#pragma clang fp contract(off)
  return (a + b) + c;
}

__device__ __forceinline__ RayGeom ray_geom(const float* __restrict__ r, const Bound& bd) {
#pragma clang fp contract(off)
  RayGeom g;
  const float c[3] = {bd.cx, bd.cy, bd.cz}, rad[3] = {bd.rx, bd.ry, bd.rz};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g.o[i] = bd.has_radius ? (r[i] - c[i]) / rad[i] : r[i];
    g.d[i] = bd.has_radius ? r[3 + i] / rad[i] : r[3 + i];
  }
  g.dd = sum3(g.d[0] * g.d[0], g.d[1] * g.d[1], g.d[2] * g.d[2]);
  g.d1 = -sum3(g.d[0] * g.o[0], g.d[1] * g.o[1], g.d[2] * g.o[2]) / g.dd;               // rendering.py:510 / :535
  float p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = g.o[i] + g.d1 * g.d[i];
  g.pn = sum3(p[0] * p[0], p[1] * p[1], p[2] * p[2]);
  return g;
}

__global__ __launch_bounds__(256) void fg_bounds_kernel(const float* __restrict__ rays, Bound bd, int N, float* __restrict__ rays_fg,
                                                        float* __restrict__ fg_far, float* __restrict__ last_delta,
                                                        int32_t* __restrict__ has_bg, int32_t* __restrict__ n_outside) {
#pragma clang fp contract(off)
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= N) return;
  const float* r = rays + (long)ray * 8;
  const RayGeom g = ray_geom(r, bd);
  const float cosv = 1.f / sqrtf(g.dd);
  if (g.pn >= 1.f) atomicAdd(n_outside, 1);                                               // :515-517 raises
  const float d2 = sqrtf(1.f - g.pn) * cosv;                                              // :518
  const float near = r[6], far = r[7];
  const float ff = fmaxf(g.d1 + d2, near);                                                // :34-35
  const bool bg = far > ff;                                                               // :36
  fg_far[ray] = ff;
  has_bg[ray] = bg ? 1 : 0;
  last_delta[ray] = bg ? ff : 1e10f;                                                      // :31, :42
  float* o = rays_fg + (long)ray * 8;
#pragma unroll
  for (int i = 0; i < 7; ++i) o[i] = r[i];
  o[7] = fminf(far, ff);                                                                  // :44
}

// z_in == nullptr: coarse pass.  Thread (ray, j) handles ascending sample a = S - 1 - j: the reference evaluates the flipped
//   sequence (:302-304), so z_out / pe are written in descending-depth order j, while depth_real stays in ascending order
//   a (the reference never flips it, :483-484 reads it as is).
// z_in != nullptr: depths supplied (hierarchical pass, :246 xyz_fine_fn): no flip, everything in the order of z_in.
template <typename T, int LMAX>
__global__ __launch_bounds__(128) void bg_sample_pe_kernel(const float* __restrict__ rays, Bound bd, const float* __restrict__ tsteps,
                                                           const float* __restrict__ prand, float perturb, int n_rays, int S, int L,
                                                           const float* __restrict__ z_in, float* __restrict__ z_out,
                                                           float* __restrict__ depth_real, T* __restrict__ pe, int pe_stride) {
#pragma clang fp contract(off)
  const long total = (long)n_rays * S;
  const long p_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p_raw < total;
  const long p = live ? p_raw : total - 1;
  const int ray = (int)(p / S), j = (int)(p - (long)ray * S);
  const int a = z_in ? j : S - 1 - j;
  float z;
  if (z_in) {
    z = z_in[p];
  } else {
    z = tsteps[a];                                                                        // linspace(0, 1, S), :46
    if (perturb > 0.f && prand) {                                                         // :573-584
      const float lower = a > 0 ? 0.5f * (tsteps[a - 1] + z) : z;
      const float upper = a < S - 1 ? 0.5f * (z + tsteps[a + 1]) : z;
      const float pr = perturb * prand[(long)ray * S + a];
      z = lower + (upper - lower) * pr;
    }
  }
  const float* r = rays + (long)ray * 8;
  const RayGeom g = ray_geom(r, bd);
  const float pmn = sqrtf(g.pn);                                                          // :537
  const float cosv = 1.f / sqrtf(g.dd);
  const float d2 = sqrtf(1.f - pmn * pmn) * cosv;                                         // :540
  const float t = g.d1 + d2;
  float ps[3], ax[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ps[i] = g.o[i] + t * g.d[i];                               // p_sphere :541
  ax[0] = g.o[1] * ps[2] - g.o[2] * ps[1];                                                // cross(o, p_sphere) :543
  ax[1] = g.o[2] * ps[0] - g.o[0] * ps[2];
  ax[2] = g.o[0] * ps[1] - g.o[1] * ps[0];
  const float an = sqrtf(sum3(ax[0] * ax[0], ax[1] * ax[1], ax[2] * ax[2])) + 1e-8f;
#pragma unroll
  for (int i = 0; i < 3; ++i) ax[i] = ax[i] / an;
  const float phi = asinf(pmn);
  const float theta = asinf(pmn * z);                                                   // :546
  const float ang = phi - theta;
  const float ca = cosf(ang), sa = sinf(ang);
  float cr[3];
  cr[0] = ax[1] * ps[2] - ax[2] * ps[1];                                                  // cross(axis, p_sphere)
  cr[1] = ax[2] * ps[0] - ax[0] * ps[2];
  cr[2] = ax[0] * ps[1] - ax[1] * ps[0];
  const float dot = sum3(ax[0] * ps[0], ax[1] * ps[1], ax[2] * ps[2]);
  float x[4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {                                                           // Rodrigues :548-550
    const float t1 = ps[i] * ca;
    const float t2 = cr[i] * sa;
    const float t3 = (ax[i] * dot) * (1.f - ca);
    x[i] = (t1 + t2) + t3;
  }
  const float xn = sqrtf(sum3(x[0] * x[0], x[1] * x[1], x[2] * x[2]));
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = x[i] / xn;
  x[3] = z;
  if (live) {
    if (z_out) z_out[p] = z;
    if (depth_real) depth_real[(long)ray * S + a] = (1.f / (z + 1e-8f)) * cosf(theta) + g.d1;   // :554
  }
  // 4-D positional encoding: [x, sin(2^k x), cos(2^k x)]_k  (models/nerf.py:11-25 Embedding over xyz_dim = 4)
  float v[4 + 8 * LMAX + 4];
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = x[c];
  if constexpr (sizeof(T) == 4) {
    float f = 1.f;
#pragma unroll
    for (int k = 0; k < LMAX; ++k) {
      if (k < L) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float sn, cs;
          sincosf(f * x[c], &sn, &cs);
          v[4 + 8 * k + c] = sn;
          v[4 + 8 * k + 4 + c] = cs;
        }
      }
      f *= 2.f;
    }
  } else {   // bf16 output: higher octaves by angle doubling (see sample_pe_kernel)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float sn, cs;
      sincosf(x[c], &sn, &cs);
#pragma unroll
      for (int k = 0; k < LMAX; ++k) {
        if (k < L) {
          v[4 + 8 * k + c] = sn;
          v[4 + 8 * k + 4 + c] = cs;
        }
        const float s2 = 2.f * sn * cs, c2 = 1.f - 2.f * sn * sn;
        sn = s2;
        cs = c2;
      }
    }
  }
  pe_store_rows<T>(v, 4 + 8 * L, pe, pe_stride, p, live, total);
}

}  // namespace swn

using namespace swn;

static inline Bound make_bound(const float* center, const float* radius) {
  Bound b{0.f, 0.f, 0.f, 1.f, 1.f, 1.f, 0};
  if (radius) {
    b.has_radius = 1;
    b.cx = center[0]; b.cy = center[1]; b.cz = center[2];
    b.rx = radius[0]; b.ry = radius[1]; b.rz = radius[2];
  }
  return b;
}

extern "C" int swn_fg_bounds(const float* rays, const float* center_host, const float* radius_host, int n_rays, float* rays_fg,
                             float* fg_far, float* last_delta, int32_t* has_bg, int32_t* n_outside, void* stream) {
  SWN_CHECK(rays && rays_fg && fg_far && last_delta && has_bg && n_outside, "swn_fg_bounds: null pointer");
  SWN_CHECK((center_host == nullptr) == (radius_host == nullptr), "swn_fg_bounds: center / radius must both be given or both NULL");
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(fg_bounds_kernel, dim3(cdiv(n_rays, 256)), dim3(256), 0, as_stream(stream), rays, make_bound(center_host, radius_host),
                     n_rays, rays_fg, fg_far, last_delta, has_bg, n_outside);
  SWN_LAUNCH_CHECK();
  return 0;
}

extern "C" int swn_bg_sample_pe(const float* rays, const float* center_host, const float* radius_host, const float* t_steps,
                                const float* perturb_rand, float perturb, int n_rays, int n_samples, int l_xyz, int dtype,
                                const float* z_in, float* z_out, float* depth_real, void* pe, int pe_stride, void* stream) {
  SWN_CHECK(dtype == SWN_F32 || dtype == SWN_HALF, "swn_bg_sample_pe: bad dtype");
  SWN_CHECK(rays && pe && (z_in || t_steps), "swn_bg_sample_pe: null pointer");
  SWN_CHECK((center_host == nullptr) == (radius_host == nullptr), "swn_bg_sample_pe: center / radius must both be given or both NULL");
  SWN_CHECK(l_xyz >= 0 && l_xyz <= 12, "swn_bg_sample_pe: frequencies must be <= 12");
  const int epc = dtype == SWN_HALF ? 8 : 4;
  SWN_CHECK(pe_stride >= 4 + 8 * l_xyz && pe_stride % epc == 0, "swn_bg_sample_pe: pe_stride %d too small / unaligned", pe_stride);
  if (n_rays <= 0) return 0;
  const long P = (long)n_rays * n_samples;
  const Bound b = make_bound(center_host, radius_host);
  if (dtype == SWN_HALF)
    hipLaunchKernelGGL((bg_sample_pe_kernel<bf16_t, 12>), dim3(cdiv(P, 128)), dim3(128), 0, as_stream(stream), rays, b, t_steps,
                       perturb_rand, perturb, n_rays, n_samples, l_xyz, z_in, z_out, depth_real, (bf16_t*)pe, pe_stride);
  else
    hipLaunchKernelGGL((bg_sample_pe_kernel<float, 12>), dim3(cdiv(P, 64)), dim3(64), 0, as_stream(stream), rays, b, t_steps,
                       perturb_rand, perturb, n_rays, n_samples, l_xyz, z_in, z_out, depth_real, (float*)pe, pe_stride);
  SWN_LAUNCH_CHECK();
  return 0;
}

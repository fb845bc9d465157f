// Device helpers shared by the K-Planes kernels (kplanes.hip, kplanes_sorted.hip).
#pragma once
#include "common.hpp"

namespace snerf {

struct AxisTap {
  int i0, i1;    // texel indices along the axis (i1 clamped for addressing)
  float w0, w1;  // weights of i0 / i1:  (i1 - x), (x - i0); w1 forced to 0 when i0+1 is out of range
};

// ATen grid_sampler semantics for align_corners=True + padding_mode="border":
//   x_pix = ((x + 1) / 2) * (size - 1), clipped to [0, size-1]; taps floor / floor+1.
__device__ __forceinline__ AxisTap axis_tap(float x, int size) {
  float fx = ((x + 1.f) / 2.f) * (float)(size - 1);
  fx = fminf((float)(size - 1), fmaxf(fx, 0.f));
  float f0 = floorf(fx);
  AxisTap t;
  t.i0 = (int)f0;
  t.w0 = (f0 + 1.f) - fx;
  t.w1 = fx - f0;
  bool in = (t.i0 + 1) <= (size - 1);
  t.i1 = in ? t.i0 + 1 : t.i0;
  if (!in) t.w1 = 0.f;  // out-of-range corner contributes nothing (ATen within_bounds_2d)
  return t;
}

// coordinate of sample s of ray r (snerf_coords mode 1: positions derived from rays + euclidean bin edges)
__device__ __forceinline__ void load_coords_ray(const snerf_coords& c, int64_t r, int s, float p[4]) {
  const float* eb = c.ebins + r * (c.S + 1) + s;
  float mid = eb[0] + eb[1];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float pos = c.origins[r * 3 + k] + (c.dirs[r * 3 + k] * mid) / 2.f;
    float q = (pos - c.aabb_min[k]) / (c.aabb_max[k] - c.aabb_min[k]);
    p[k] = c.rescale ? q * 2.f - 1.f : q;
  }
  p[3] = c.times[r] * 2.f - 1.f;
}

// coordinate of sample n along x,y,z,t in grid_sample's [-1,1] convention
template <int NP>
__device__ __forceinline__ void load_coords(const snerf_coords& c, int64_t n, float p[4]) {
  if (c.mode == 0) {
    if (NP == 6) {
      float4 v = *reinterpret_cast<const float4*>(c.pts + n * 4);
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    } else {  // static scene: pts is [N,3]
      p[0] = c.pts[n * 3]; p[1] = c.pts[n * 3 + 1]; p[2] = c.pts[n * 3 + 2]; p[3] = 0.f;
    }
  } else {
    int64_t r = n / c.S;  // a 64-bit software division (~100 instructions): loops over consecutive samples step (r, s) themselves
    int s = (int)(n - r * c.S);
    load_coords_ray(c, r, s, p);
  }
}

template <int NP> struct PlanePairs;
template <> struct PlanePairs<6> {  // XY XZ XT YZ YT ZT
  static constexpr int a[6] = {0, 0, 0, 1, 1, 2};
  static constexpr int b[6] = {1, 2, 3, 2, 3, 3};
};
template <> struct PlanePairs<3> {  // static scene: XY XZ YZ
  static constexpr int a[3] = {0, 0, 1};
  static constexpr int b[3] = {1, 2, 2};
};

// The same tables as arithmetic on the plane index.  Inside an unrolled plane loop q is a literal and these fold; indexing the constexpr
// arrays above did not always (kplanes_gather_bwd_kernel, field_fwd_kernel: the taps were kept in scratch and read back through a
// uniform-but-dynamic index, 16 dwords per sample -- profiles/r02_kernels.md).
template <int NP> __host__ __device__ constexpr int pair_a(int q) { return NP == 6 ? (q < 3 ? 0 : (q < 5 ? 1 : 2)) : (q < 2 ? 0 : 1); }
template <int NP> __host__ __device__ constexpr int pair_b(int q) { return NP == 6 ? (q < 3 ? q + 1 : (q < 5 ? q - 1 : 3)) : (q < 1 ? 1 : 2); }
static_assert(pair_a<6>(0) == 0 && pair_a<6>(2) == 0 && pair_a<6>(3) == 1 && pair_a<6>(4) == 1 && pair_a<6>(5) == 2, "pair_a<6>");
static_assert(pair_b<6>(0) == 1 && pair_b<6>(1) == 2 && pair_b<6>(2) == 3 && pair_b<6>(3) == 2 && pair_b<6>(4) == 3 && pair_b<6>(5) == 3, "pair_b<6>");
static_assert(pair_a<3>(0) == 0 && pair_a<3>(1) == 0 && pair_a<3>(2) == 1 && pair_b<3>(0) == 1 && pair_b<3>(1) == 2 && pair_b<3>(2) == 2, "pair_*<3>");

__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Bilinear blend of the four texels of a tap with a FIXED evaluation order and explicit roundings (no compiler-chosen contraction):
// the forward gather and the quotient form of the sorted scatter (kplanes_sorted.hip, QUOT) must produce the SAME bits for a plane's
// value at a sample -- the quotient G / v_q cancels the forward's v_q exactly only if pass B recomputes that very number (a value that is
// a small difference of large texels would otherwise come back with a large relative error).
__device__ __forceinline__ float bilerp4(float nw, float ne, float sw, float se, float w00, float w10, float w01, float w11) {
  float acc = __fmul_rn(nw, w00);
  acc = __fmaf_rn(ne, w10, acc);
  acc = __fmaf_rn(sw, w01, acc);
  acc = __fmaf_rn(se, w11, acc);
  return acc;
}
// the four corner weights of a tap pair: (x0,y0), (x1,y0), (x0,y1), (x1,y1)
__device__ __forceinline__ float4 tap_weights(const AxisTap& tx, const AxisTap& ty) {
  return make_float4(__fmul_rn(tx.w0, ty.w0), __fmul_rn(tx.w1, ty.w0), __fmul_rn(tx.w0, ty.w1), __fmul_rn(tx.w1, ty.w1));
}

// bilinear value of plane p for this lane's 4 channels
template <int C>
__device__ __forceinline__ float4 plane_sample(const float* __restrict__ base, int W, const AxisTap& tx, const AxisTap& ty, int cg) {
  const float* r0 = base + ((int64_t)ty.i0 * W) * C + cg * 4;
  const float* r1 = base + ((int64_t)ty.i1 * W) * C + cg * 4;
  float4 nw = *reinterpret_cast<const float4*>(r0 + (int64_t)tx.i0 * C);
  float4 ne = *reinterpret_cast<const float4*>(r0 + (int64_t)tx.i1 * C);
  float4 sw = *reinterpret_cast<const float4*>(r1 + (int64_t)tx.i0 * C);
  float4 se = *reinterpret_cast<const float4*>(r1 + (int64_t)tx.i1 * C);
  const float4 w = tap_weights(tx, ty);
  return make_float4(bilerp4(nw.x, ne.x, sw.x, se.x, w.x, w.y, w.z, w.w), bilerp4(nw.y, ne.y, sw.y, se.y, w.x, w.y, w.z, w.w),
                     bilerp4(nw.z, ne.z, sw.z, se.z, w.x, w.y, w.z, w.w), bilerp4(nw.w, ne.w, sw.w, se.w, w.x, w.y, w.z, w.w));
}


}  // namespace snerf

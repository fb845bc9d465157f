// Per-float4 pieces of the dense optimiser sweep (plane_reg_kernel, optim.hip): the K-Planes plane regularisers' analytic gradient (NS/model_components/losses.py:356-452) and torch.optim.Adam
// (NS/configs/method_configs.py:546-557) on one float4 of a channel-last plane.
#pragma once
#include "common.hpp"

namespace snerf {

struct RegArgs {
  snerf_kplanes_desc d;
  int blk_off[SNERF_MAX_SCALES][6];  // first workgroup of each plane (prefix sum), 256 float4-lanes per workgroup
  int n_planes;                      // 6 or 3
  const float* planes;
  float* grad;                       // may be null (values only)
  float c_tv, c_smooth, c_l1;        // loss coefficients folded into the gradient
  float* losses;                     // [n_slots][16]: per-slot partial sums (cols 0..2), UNSCALED
  int n_slots;
  // fused Adam (adam_planes_kernel): the regulariser gradient never touches HBM; parameters ping-pong p_in -> p_out because the
  // sweep reads +-1/+-2 neighbours of the OLD parameters
  float* p_out; float* m; float* v;
  float step_size, b1, b2, inv_sqrt_bc2, eps, grad_scale;
  int zero_grad;
  int overwrite;                     // 1: grad = reg gradient (buffer known to be zero), 0: grad += reg gradient
  // optimiser sharding (one rank updates floats [range_lo, range_hi) of the segment): the grid starts at workgroup blk_base
  int blk_base;
  int64_t range_lo, range_hi;
  snerf_adam_dyn* dyn;               // device-side step state (snerf_adam_prepare); null: step_size / inv_sqrt_bc2 above are used
};

// Device-side optimiser state of one parameter group (snerf.h: snerf_adam_dyn).  The kernels read {step_size, inv_sqrt_bc2, skip} from
// it when given, so that a step can be skipped (the reference's GradScaler semantics) without the host ever reading the flag.
struct DynConsts { float step_size, inv_sqrt_bc2; int skip; };
__device__ __forceinline__ DynConsts load_dyn(const snerf_adam_dyn* dyn, float step_size, float inv_sqrt_bc2) {
  DynConsts c = {step_size, inv_sqrt_bc2, 0};
  if (dyn) { c.step_size = dyn->step_size; c.inv_sqrt_bc2 = dyn->inv_sqrt_bc2; c.skip = dyn->skip; }
  return c;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt4(const float* p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stnt4(float* p, float4 v) {
  nt_f4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<nt_f4*>(p));
}
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float sq4(float4 a) { return a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }


// Regulariser gradient (coefficients folded in) of the float4 at texel (h, w) of a plane and its share of the three loss values.
// `at(hh, ww)` returns the OLD parameters' float4 (same channel group) at another texel; t = at(h, w).
template <int C, typename AT>
__device__ __forceinline__ float4 plane_reg_grad(const AT& at, const float4 t, int h, int w, int H, int W, bool time_plane, float c_tv, float c_smooth,
                                                 float c_l1, float& l_tv, float& l_sm, float& l_l1) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- total variation: w direction always; h direction only on space-only planes ----
  const float n_w = (float)C * (float)H * (float)(W - 1);
  if (W > 1) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w + 1 < W) { float4 d = sub4(at(h, w + 1), t); l_tv += sq4(d) / n_w; acc = sub4(acc, d); }
    if (w > 0) { float4 d = sub4(t, at(h, w - 1)); acc = add4(acc, d); }
    g = add4(g, mul4(acc, 2.f * c_tv / n_w));
  }
  if (!time_plane && H > 1) {
    const float n_h = (float)C * (float)(H - 1) * (float)W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h + 1 < H) { float4 d = sub4(at(h + 1, w), t); l_tv += sq4(d) / n_h; acc = sub4(acc, d); }
    if (h > 0) { float4 d = sub4(t, at(h - 1, w)); acc = add4(acc, d); }
    g = add4(g, mul4(acc, 2.f * c_tv / n_h));
  }
  if (time_plane) {
    // ---- smoothness: second difference along h (= time); d2[k] = t[k+2] - 2 t[k+1] + t[k], k in [0, H-3] ----
    if (H > 2) {
      const float n_s = (float)C * (float)(H - 2) * (float)W;
      auto d2 = [&](int k) {  // valid for 0 <= k <= H-3
        float4 x0 = at(k, w), x1 = at(k + 1, w), x2 = at(k + 2, w);
        return make_float4(x2.x - 2.f * x1.x + x0.x, x2.y - 2.f * x1.y + x0.y, x2.z - 2.f * x1.z + x0.z, x2.w - 2.f * x1.w + x0.w);
      };
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h <= H - 3) { float4 v = d2(h); l_sm += sq4(v) / n_s; acc = add4(acc, v); }          // t[h] enters d2[h] with +1
      if (h >= 1 && h - 1 <= H - 3) { float4 v = d2(h - 1); acc = add4(acc, mul4(v, -2.f)); }  // d2[h-1] with -2
      if (h >= 2) { float4 v = d2(h - 2); acc = add4(acc, v); }                                 // d2[h-2] with +1
      g = add4(g, mul4(acc, 2.f * c_smooth / n_s));
    }
    // ---- sparse transients: mean |1 - t| ----
    const float n_a = (float)C * (float)H * (float)W;
    l_l1 += (fabsf(1.f - t.x) + fabsf(1.f - t.y) + fabsf(1.f - t.z) + fabsf(1.f - t.w)) / n_a;
    const float k = -c_l1 / n_a;
    g = add4(g, make_float4(k * sgn(1.f - t.x), k * sgn(1.f - t.y), k * sgn(1.f - t.z), k * sgn(1.f - t.w)));
  }
  return g;
}

// torch.optim.Adam on one float4: gradient = gdata * grad_scale + greg; a non-finite element is dropped (and counted), never written into
// m / v / p.  Returns the number of dropped elements.
__device__ __forceinline__ int adam_float4(float4& pp, float4& mm, float4& vv, const float4 gdata, const float4 greg, float grad_scale, float b1, float b2,
                                           float eps, const DynConsts& dc) {
  float* P = &pp.x; float* M = &mm.x; float* V = &vv.x;
  const float* G = &gdata.x; const float* RG = &greg.x;
  int ndrop = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float gk = G[k] * grad_scale + RG[k];
    if (!(fabsf(gk) <= 3.402823466e+38f)) { gk = 0.f; ++ndrop; }
    M[k] = b1 * M[k] + (1.f - b1) * gk;
    V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
    P[k] = P[k] - dc.step_size * (M[k] / (sqrtf(V[k]) * dc.inv_sqrt_bc2 + eps));
  }
  return ndrop;
}

// plane index -> (width axis, height axis) of the reference's coordinate pairs XY XZ XT YZ YT ZT (kplanes_field.py:61-65)
__host__ __device__ inline void plane_axes(int n_planes, int p, int& ax, int& bx) {
  constexpr int PA6[6] = {0, 0, 0, 1, 1, 2}, PB6[6] = {1, 2, 3, 2, 3, 3};
  constexpr int PA3[3] = {0, 0, 1}, PB3[3] = {1, 2, 2};
  ax = n_planes == 6 ? PA6[p] : PA3[p];
  bx = n_planes == 6 ? PB6[p] : PB3[p];
}

void adam_consts(float lr, float beta1, float beta2, int step, float& step_size, float& inv_sqrt_bc2);

}  // namespace snerf

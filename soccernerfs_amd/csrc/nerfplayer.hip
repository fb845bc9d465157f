// NeRFPlayer decomposition mixing (NerfplayerField.get_density, NS/fields/nerfplayer_field.py:365-372):
//   probs = softmax(decomposition_mlp(decomposition_field(x, t)))          [N,3]  0 = static, 1 = deforming, 2 = new
//   v     = probs[:,0] v_static + probs[:,1] v_deform + probs[:,2] v_new   [N,F]
// and its backward, including the gradient that reaches probs from outside v (the rendered-probability regulariser,
// NS/models/nerfplayer.py:336-341).  The reference runs this as ~10 ATen elementwise / reduction kernels each way.
// F / 4 lanes per sample, float4 per lane; the three channel reductions of the backward are xor-shuffles inside the lane group.
#include "common.hpp"

namespace snerf {

template <int F>
__global__ __launch_bounds__(256) void mix_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ vs, const float* __restrict__ vd,
                                                     const float* __restrict__ vn, int64_t N, float* __restrict__ probs, float* __restrict__ v) {
  constexpr int LPS = F / 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = gid / LPS;
  const int cg = (int)(gid % LPS);
  if (n >= N) return;
  const float l0 = logits[n * 3], l1 = logits[n * 3 + 1], l2 = logits[n * 3 + 2];
  const float m = fmaxf(l0, fmaxf(l1, l2));
  const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
  const float inv = 1.f / (e0 + e1 + e2);
  const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
  if (cg == 0) { probs[n * 3] = p0; probs[n * 3 + 1] = p1; probs[n * 3 + 2] = p2; }
  const float4 a = *reinterpret_cast<const float4*>(vs + n * F + cg * 4), b = *reinterpret_cast<const float4*>(vd + n * F + cg * 4),
               c = *reinterpret_cast<const float4*>(vn + n * F + cg * 4);
  float4 o;
  o.x = p0 * a.x + p1 * b.x + p2 * c.x; o.y = p0 * a.y + p1 * b.y + p2 * c.y;
  o.z = p0 * a.z + p1 * b.z + p2 * c.z; o.w = p0 * a.w + p1 * b.w + p2 * c.w;
  *reinterpret_cast<float4*>(v + n * F + cg * 4) = o;
}

template <int F>
__global__ __launch_bounds__(256) void mix_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ vs, const float* __restrict__ vd,
                                                     const float* __restrict__ vn, const float* __restrict__ gv, const float* __restrict__ gp_ext, int64_t N,
                                                     float* __restrict__ gvs, float* __restrict__ gvd, float* __restrict__ gvn, float* __restrict__ glogits) {
  constexpr int LPS = F / 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = gid / LPS;
  const int cg = (int)(gid % LPS);
  const bool live = n < N;
  const int64_t nn = live ? n : N - 1;
  const float p0 = probs[nn * 3], p1 = probs[nn * 3 + 1], p2 = probs[nn * 3 + 2];
  const float4 g = *reinterpret_cast<const float4*>(gv + nn * F + cg * 4);
  const float4 a = *reinterpret_cast<const float4*>(vs + nn * F + cg * 4), b = *reinterpret_cast<const float4*>(vd + nn * F + cg * 4),
               c = *reinterpret_cast<const float4*>(vn + nn * F + cg * 4);
  float d0 = g.x * a.x + g.y * a.y + g.z * a.z + g.w * a.w;
  float d1 = g.x * b.x + g.y * b.y + g.z * b.z + g.w * b.w;
  float d2 = g.x * c.x + g.y * c.y + g.z * c.z + g.w * c.w;
#pragma unroll
  for (int off = LPS / 2; off > 0; off >>= 1) {  // lane groups are aligned: LPS divides 64
    d0 += __shfl_xor(d0, off, 64); d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64);
  }
  if (!live) return;
  if (gp_ext) { d0 += gp_ext[n * 3]; d1 += gp_ext[n * 3 + 1]; d2 += gp_ext[n * 3 + 2]; }
  if (cg == 0) {  // softmax backward
    const float dot = p0 * d0 + p1 * d1 + p2 * d2;
    glogits[n * 3] = p0 * (d0 - dot); glogits[n * 3 + 1] = p1 * (d1 - dot); glogits[n * 3 + 2] = p2 * (d2 - dot);
  }
  *reinterpret_cast<float4*>(gvs + n * F + cg * 4) = make_float4(p0 * g.x, p0 * g.y, p0 * g.z, p0 * g.w);
  *reinterpret_cast<float4*>(gvd + n * F + cg * 4) = make_float4(p1 * g.x, p1 * g.y, p1 * g.z, p1 * g.w);
  *reinterpret_cast<float4*>(gvn + n * F + cg * 4) = make_float4(p2 * g.x, p2 * g.y, p2 * g.z, p2 * g.w);
}

// ---------------------------------------------------------------------------------------------
// NeRFPlayer-nerfacto colour head input (NerfplayerNerfactoField.get_outputs, NS/fields/nerfplayer_nerfacto_field.py:350-384): per sample
//   hx = [ SH degree 4 of the ray direction (16) | geometry features h[:, 1:16] (15) | appearance embedding of the ray's camera (32) | 0 ]
// The reference builds it from ~35 ATen launches (the spherical harmonics alone are 25 elementwise kernels on [R] tensors, then stack / expand /
// cat); here one launch each way.  The SH expressions are soccernerfs_amd/sh.py's, evaluated in its order with contraction off, so the values are
// the bits the torch expressions give.  Backward: gh[:, 1:16] = ghx[:, 16:31] and the appearance gradient = sum over the ray's samples of
// ghx[:, 31:63], added to the camera's row (float atomics, or fixed-point cells).
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ float sh4_coeff(int k, float x, float y, float z) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  switch (k) {
    case 0: return 0.28209479177387814f;
    case 1: return -0.48860251190291987f * y;
    case 2: return 0.48860251190291987f * z;
    case 3: return -0.48860251190291987f * x;
    case 4: return 1.0925484305920792f * xy;
    case 5: return -1.0925484305920792f * yz;
    case 6: return 0.94617469575755997f * z2 - 0.31539156525251999f;
    case 7: return -1.0925484305920792f * xz;
    case 8: return 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    case 9: return (0.59004358992664352f * y) * (-3.0f * x2 + y2);
    case 10: return (2.8906114426405538f * xy) * z;
    case 11: return (0.45704579946446572f * y) * (1.0f - 5.0f * z2);
    case 12: return (0.3731763325901154f * z) * (5.0f * z2 - 3.0f);
    case 13: return (0.45704579946446572f * x) * (1.0f - 5.0f * z2);
    case 14: return (1.4453057213202769f * z) * (x2 - y2);
    default: return (0.59004358992664352f * x) * (-x2 + 3.0f * y2);
  }
}

// one thread per (sample, float4 of hx); rows of hx are 64 floats.  (r06: was one thread per column -- a 64-bit division per element and a 16-way divergent
// switch over the SH polynomials; 133 us in the traced step of config 4, on the chain field forward -> field backward -> tile pass.  Same expressions per element.)
__global__ __launch_bounds__(256) void head_input_fwd_kernel(const float* __restrict__ dirs, const float* __restrict__ h, const float* __restrict__ app,
                                                            const int64_t* __restrict__ cams, int S, int64_t N, float* __restrict__ hx) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = gid >> 4;
  const int q = (int)(gid & 15);
  if (n >= N) return;
  const int64_t ray = N < (1LL << 31) ? (int64_t)((uint32_t)n / (uint32_t)S) : n / S;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  float* V = &v.x;
  if (q < 4) {
    const float x = dirs[ray * 3], y = dirs[ray * 3 + 1], z = dirs[ray * 3 + 2];
#pragma unroll
    for (int k = 0; k < 4; ++k) V[k] = sh4_coeff(4 * q + k, x, y, z);
  } else {
    const int64_t cam = (app && cams) ? cams[ray] : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * q + k;
      if (c < 31) V[k] = h[n * 16 + (c - 15)];
      else if (c < 63 && app) V[k] = app[cam * 32 + (c - 31)];
    }
  }
  *reinterpret_cast<float4*>(hx + n * 64 + 4 * q) = v;
}

// one 64-thread group per ray: lanes 0..31 sum the appearance columns over the ray's samples (in sample order), lanes 32..46 copy the geometry columns
__global__ __launch_bounds__(256) void head_input_bwd_kernel(const float* __restrict__ ghx, const int64_t* __restrict__ cams, int S, int64_t R,
                                                            float* __restrict__ gh, float* __restrict__ gapp, long long* __restrict__ gapp_fx) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int l = threadIdx.x & 63;
  if (ray >= R) return;
  const float* g = ghx + ray * S * 64;
  if (l < 32) {
    if (!gapp && !gapp_fx) return;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += g[s * 64 + 31 + l];
    const int64_t e = cams[ray] * 32 + l;
    if (gapp_fx) fx_atomic_add(gapp_fx + e, acc); else if (acc != 0.f) atomicAdd(gapp + e, acc);
  } else if (l < 47) {
    const int j = l - 32;  // geometry column 1 + j of h
    for (int s = 0; s < S; ++s) gh[(ray * S + s) * 16 + 1 + j] = g[s * 64 + 16 + j];
  }
}
#pragma clang fp contract(fast)

}  // namespace snerf

using namespace snerf;

// snerf.h (ABI 13)
extern "C" int snerf_nerfacto_head_input_fwd(const float* dirs, const float* h, const float* appearance, const int64_t* cams, int32_t S, int64_t R,
                                             float* hx, snerf_stream_t stream) {
  SNERF_REQUIRE(S >= 1 && R >= 0, "nerfacto_head_input_fwd: S=%d R=%lld", S, (long long)R);
  if (R == 0) return 0;
  SNERF_REQUIRE(dirs && h && hx, "nerfacto_head_input_fwd: null buffer");
  SNERF_REQUIRE(!cams || appearance, "nerfacto_head_input_fwd: camera indices without an embedding table");
  const int64_t N = R * S, threads = N * 16;
  SNERF_REQUIRE(((uintptr_t)hx & 15) == 0, "nerfacto_head_input_fwd: hx must be 16-byte aligned");
  hipLaunchKernelGGL(head_input_fwd_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, (hipStream_t)stream, dirs, h, appearance, cams, S, N, hx);
  SNERF_LAUNCH_CHECK("nerfacto_head_input_fwd");
  return 0;
}

extern "C" int snerf_nerfacto_head_input_bwd(const float* g_hx, const int64_t* cams, int32_t S, int64_t R, float* g_h, float* g_appearance,
                                             int64_t* g_appearance_fx, snerf_stream_t stream) {
  SNERF_REQUIRE(S >= 1 && R >= 0, "nerfacto_head_input_bwd: S=%d R=%lld", S, (long long)R);
  if (R == 0) return 0;
  SNERF_REQUIRE(g_hx && g_h, "nerfacto_head_input_bwd: null buffer");
  SNERF_REQUIRE(!(g_appearance && g_appearance_fx), "nerfacto_head_input_bwd: give g_appearance or g_appearance_fx, not both");
  SNERF_REQUIRE(cams || !(g_appearance || g_appearance_fx), "nerfacto_head_input_bwd: the appearance gradient needs the camera indices");
  hipLaunchKernelGGL(head_input_bwd_kernel, dim3((unsigned)ceil_div(R * 64, 256)), dim3(256), 0, (hipStream_t)stream, g_hx, cams, S, R, g_h, g_appearance,
                     reinterpret_cast<long long*>(g_appearance_fx));
  SNERF_LAUNCH_CHECK("nerfacto_head_input_bwd");
  return 0;
}

extern "C" int snerf_nerfplayer_mix_fwd(const float* logits, const float* v_static, const float* v_deform, const float* v_new, int64_t N, int32_t F,
                                        float* probs, float* v, snerf_stream_t stream) {
  SNERF_REQUIRE(N >= 0 && (F == 4 || F == 8 || F == 16 || F == 32 || F == 64), "nerfplayer_mix_fwd: N=%lld F=%d (4, 8, 16, 32 or 64)", (long long)N, F);
  if (N == 0) return 0;
  SNERF_REQUIRE(logits && v_static && v_deform && v_new && probs && v, "nerfplayer_mix_fwd: null buffer");
  const dim3 grid((unsigned)ceil_div(N * (F / 4), 256));
  hipStream_t st = (hipStream_t)stream;
  if (F == 4) hipLaunchKernelGGL(mix_fwd_kernel<4>, grid, dim3(256), 0, st, logits, v_static, v_deform, v_new, N, probs, v);
  else if (F == 8) hipLaunchKernelGGL(mix_fwd_kernel<8>, grid, dim3(256), 0, st, logits, v_static, v_deform, v_new, N, probs, v);
  else if (F == 16) hipLaunchKernelGGL(mix_fwd_kernel<16>, grid, dim3(256), 0, st, logits, v_static, v_deform, v_new, N, probs, v);
  else if (F == 32) hipLaunchKernelGGL(mix_fwd_kernel<32>, grid, dim3(256), 0, st, logits, v_static, v_deform, v_new, N, probs, v);
  else hipLaunchKernelGGL(mix_fwd_kernel<64>, grid, dim3(256), 0, st, logits, v_static, v_deform, v_new, N, probs, v);
  SNERF_LAUNCH_CHECK("nerfplayer_mix_fwd");
  return 0;
}

extern "C" int snerf_nerfplayer_mix_bwd(const float* probs, const float* v_static, const float* v_deform, const float* v_new, const float* g_v,
                                        const float* g_probs, int64_t N, int32_t F, float* g_static, float* g_deform, float* g_new, float* g_logits,
                                        snerf_stream_t stream) {
  SNERF_REQUIRE(N >= 0 && (F == 4 || F == 8 || F == 16 || F == 32 || F == 64), "nerfplayer_mix_bwd: N=%lld F=%d (4, 8, 16, 32 or 64)", (long long)N, F);
  if (N == 0) return 0;
  SNERF_REQUIRE(probs && v_static && v_deform && v_new && g_v && g_static && g_deform && g_new && g_logits, "nerfplayer_mix_bwd: null buffer");
  const dim3 grid((unsigned)ceil_div(N * (F / 4), 256));
  hipStream_t st = (hipStream_t)stream;
  if (F == 4) hipLaunchKernelGGL(mix_bwd_kernel<4>, grid, dim3(256), 0, st, probs, v_static, v_deform, v_new, g_v, g_probs, N, g_static, g_deform, g_new, g_logits);
  else if (F == 8) hipLaunchKernelGGL(mix_bwd_kernel<8>, grid, dim3(256), 0, st, probs, v_static, v_deform, v_new, g_v, g_probs, N, g_static, g_deform, g_new, g_logits);
  else if (F == 16) hipLaunchKernelGGL(mix_bwd_kernel<16>, grid, dim3(256), 0, st, probs, v_static, v_deform, v_new, g_v, g_probs, N, g_static, g_deform, g_new, g_logits);
  else if (F == 32) hipLaunchKernelGGL(mix_bwd_kernel<32>, grid, dim3(256), 0, st, probs, v_static, v_deform, v_new, g_v, g_probs, N, g_static, g_deform, g_new, g_logits);
  else hipLaunchKernelGGL(mix_bwd_kernel<64>, grid, dim3(256), 0, st, probs, v_static, v_deform, v_new, g_v, g_probs, N, g_static, g_deform, g_new, g_logits);
  SNERF_LAUNCH_CHECK("nerfplayer_mix_bwd");
  return 0;
}

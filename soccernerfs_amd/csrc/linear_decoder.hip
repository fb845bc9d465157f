// The two pointwise pieces of KPlanesField's linear decoder (NS/fields/kplanes_field.py:219-246 constructor, :305-311 density, :349-354 colour)
// that are not dense layers:
//   density = trunc_exp(sigma)                         forward exp(x), backward g * exp(clamp(x, -15, 15))   (NS/field_components/activations.py:25-41)
//   rgb[c]  = sigmoid(sum_f feat[f] * basis[c * F + f])   the learned basis (color_basis(direction), [N, 3F]) contracted with the plane features
// The reference runs the second as a broadcast multiply + sum + sigmoid over an [N, 3, F] temporary.  Both are HBM streams: 16 lanes per
// sample, float4 per lane, rows of basis read coalesced; the three channel sums are xor-shuffles inside the lane group.
#include "common.hpp"

namespace snerf {

__global__ __launch_bounds__(256) void trunc_exp_fwd_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = expf(x[i]);
}

__global__ __launch_bounds__(256) void trunc_exp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int64_t n, float* __restrict__ gx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gx[i] = g[i] * expf(fminf(fmaxf(x[i], -15.f), 15.f));
}

constexpr int BASIS_LPS = 16;  // lanes per sample

__global__ __launch_bounds__(256) void basis_rgb_fwd_kernel(const float* __restrict__ feat, int ldf, const float* __restrict__ basis, int64_t N, int F,
                                                           float* __restrict__ rgb) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = gid / BASIS_LPS;
  const int lane = (int)(gid % BASIS_LPS);
  const bool live = n < N;
  const int64_t nn = live ? n : N - 1;
  const float* f = feat + nn * ldf;
  const float* b = basis + nn * 3 * (int64_t)F;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int q = lane * 4; q < F; q += BASIS_LPS * 4) {
    const float4 v = *reinterpret_cast<const float4*>(f + q);
    const float4 b0 = *reinterpret_cast<const float4*>(b + q), b1 = *reinterpret_cast<const float4*>(b + F + q),
                 b2 = *reinterpret_cast<const float4*>(b + 2 * F + q);
    s0 += v.x * b0.x + v.y * b0.y + v.z * b0.z + v.w * b0.w;
    s1 += v.x * b1.x + v.y * b1.y + v.z * b1.z + v.w * b1.w;
    s2 += v.x * b2.x + v.y * b2.y + v.z * b2.z + v.w * b2.w;
  }
#pragma unroll
  for (int off = BASIS_LPS / 2; off > 0; off >>= 1) {
    s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64);
  }
  if (live && lane == 0) {
    rgb[n * 3] = 1.f / (1.f + expf(-s0)); rgb[n * 3 + 1] = 1.f / (1.f + expf(-s1)); rgb[n * 3 + 2] = 1.f / (1.f + expf(-s2));
  }
}

// g_z[c] = g_rgb[c] rgb[c] (1 - rgb[c]);  g_feat[f] = sum_c g_z[c] basis[c, f];  g_basis[c, f] = g_z[c] feat[f]
__global__ __launch_bounds__(256) void basis_rgb_bwd_kernel(const float* __restrict__ feat, int ldf, const float* __restrict__ basis, const float* __restrict__ rgb,
                                                           const float* __restrict__ g_rgb, int64_t N, int F, float* __restrict__ g_feat,
                                                           float* __restrict__ g_basis) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = gid / BASIS_LPS;
  const int lane = (int)(gid % BASIS_LPS);
  if (n >= N) return;
  float z[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float y = rgb[n * 3 + c];
    z[c] = g_rgb[n * 3 + c] * y * (1.f - y);
  }
  const float* f = feat + n * ldf;
  const float* b = basis + n * 3 * (int64_t)F;
  float* gb = g_basis + n * 3 * (int64_t)F;
  for (int q = lane * 4; q < F; q += BASIS_LPS * 4) {
    const float4 v = *reinterpret_cast<const float4*>(f + q);
    const float4 b0 = *reinterpret_cast<const float4*>(b + q), b1 = *reinterpret_cast<const float4*>(b + F + q),
                 b2 = *reinterpret_cast<const float4*>(b + 2 * F + q);
    if (g_feat)
      *reinterpret_cast<float4*>(g_feat + n * (int64_t)F + q) =
          make_float4(z[0] * b0.x + z[1] * b1.x + z[2] * b2.x, z[0] * b0.y + z[1] * b1.y + z[2] * b2.y, z[0] * b0.z + z[1] * b1.z + z[2] * b2.z,
                      z[0] * b0.w + z[1] * b1.w + z[2] * b2.w);
    *reinterpret_cast<float4*>(gb + q) = make_float4(z[0] * v.x, z[0] * v.y, z[0] * v.z, z[0] * v.w);
    *reinterpret_cast<float4*>(gb + F + q) = make_float4(z[1] * v.x, z[1] * v.y, z[1] * v.z, z[1] * v.w);
    *reinterpret_cast<float4*>(gb + 2 * F + q) = make_float4(z[2] * v.x, z[2] * v.y, z[2] * v.z, z[2] * v.w);
  }
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_trunc_exp_fwd(const float* x, int64_t n, float* y, snerf_stream_t stream) {
  SNERF_REQUIRE(n >= 0, "trunc_exp_fwd: n=%lld", (long long)n);
  if (n == 0) return 0;
  SNERF_REQUIRE(x && y, "trunc_exp_fwd: null buffer");
  hipLaunchKernelGGL(trunc_exp_fwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, y);
  SNERF_LAUNCH_CHECK("trunc_exp_fwd");
  return 0;
}

extern "C" int snerf_trunc_exp_bwd(const float* x, const float* g, int64_t n, float* gx, snerf_stream_t stream) {
  SNERF_REQUIRE(n >= 0, "trunc_exp_bwd: n=%lld", (long long)n);
  if (n == 0) return 0;
  SNERF_REQUIRE(x && g && gx, "trunc_exp_bwd: null buffer");
  hipLaunchKernelGGL(trunc_exp_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, g, n, gx);
  SNERF_LAUNCH_CHECK("trunc_exp_bwd");
  return 0;
}

extern "C" int snerf_basis_rgb_fwd(const float* feat, int32_t ldf, const float* basis, int64_t N, int32_t F, float* rgb, snerf_stream_t stream) {
  SNERF_REQUIRE(N >= 0 && F >= 4 && F % 4 == 0 && ldf >= F && ldf % 4 == 0, "basis_rgb_fwd: N=%lld F=%d ldf=%d (F, ldf multiples of 4)", (long long)N, F, ldf);
  if (N == 0) return 0;
  SNERF_REQUIRE(feat && basis && rgb, "basis_rgb_fwd: null buffer");
  hipLaunchKernelGGL(basis_rgb_fwd_kernel, dim3((unsigned)ceil_div(N * BASIS_LPS, 256)), dim3(256), 0, (hipStream_t)stream, feat, ldf, basis, N, F, rgb);
  SNERF_LAUNCH_CHECK("basis_rgb_fwd");
  return 0;
}

extern "C" int snerf_basis_rgb_bwd(const float* feat, int32_t ldf, const float* basis, const float* rgb, const float* g_rgb, int64_t N, int32_t F,
                                   float* g_feat, float* g_basis, snerf_stream_t stream) {
  SNERF_REQUIRE(N >= 0 && F >= 4 && F % 4 == 0 && ldf >= F && ldf % 4 == 0, "basis_rgb_bwd: N=%lld F=%d ldf=%d (F, ldf multiples of 4)", (long long)N, F, ldf);
  if (N == 0) return 0;
  SNERF_REQUIRE(feat && basis && rgb && g_rgb && g_basis, "basis_rgb_bwd: null buffer");
  hipLaunchKernelGGL(basis_rgb_bwd_kernel, dim3((unsigned)ceil_div(N * BASIS_LPS, 256)), dim3(256), 0, (hipStream_t)stream, feat, ldf, basis, rgb, g_rgb, N,
                     F, g_feat, g_basis);
  SNERF_LAUNCH_CHECK("basis_rgb_bwd");
  return 0;
}

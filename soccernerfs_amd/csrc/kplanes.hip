// K-Planes multiscale bilinear plane gather (forward) and atomic scatter (backward).
//
// Replaces interpolate_kplanes + grid_sample_wrapper (NS/fields/kplanes_field.py:77-126,
// NS/utils/interpolation.py:5-33): the reference issues 6 F.grid_sample launches per scale on NCHW
// planes (a texel's C features strided by H*W) and materialises an [N,C] tensor per plane.  Here planes
// are channel-last, one texel = C contiguous floats (128 B for C=32 = one cache line), C/4 lanes own one
// sample and each lane moves a float4, so every texel fetch is a single coalesced line; the 6-plane
// Hadamard product and the scale concat stay in registers.  HBM-bound: algorithmic bytes per sample =
// n_scales * 6 planes * 4 texels * C * 4 B.
#include <stdlib.h>

#include "kplanes_common.hpp"

namespace snerf {

template <int C, int NP>
__global__ __launch_bounds__(256) void kplanes_gather_fwd_kernel(snerf_kplanes_desc d, const float* __restrict__ planes,
                                                                snerf_coords c, int64_t N, float* __restrict__ out) {
  constexpr int LPS = C / 4;  // lanes per sample
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n = gid / LPS;
  int cg = (int)(gid % LPS);
  if (n >= N) return;
  float p[4];
  load_coords<NP>(c, n, p);
  const int out_w = d.concat ? C * d.n_scales : C;
  float* orow = out + n * out_w + cg * 4;
  float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < d.n_scales; ++s) {
    AxisTap tap[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tap[k] = axis_tap(p[k], d.res[s][k] > 0 ? d.res[s][k] : 1);
    float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      constexpr auto& A = PlanePairs<NP>::a;
      constexpr auto& B = PlanePairs<NP>::b;
      float4 v = plane_sample<C>(planes + d.off[s][q], d.res[s][A[q]], tap[A[q]], tap[B[q]], cg);
      prod = f4_mul(prod, v);
    }
    if (d.concat) {
      *reinterpret_cast<float4*>(orow + s * C) = prod;
    } else {
      total = f4_add(total, prod);
    }
  }
  if (!d.concat) *reinterpret_cast<float4*>(orow) = total;
}

// ---------------------------------------------------------------------------------------------
// Backward: scatter-add into the plane gradients.
//
// Float atomics on gfx950 execute at the memory side in 64-B requests (MI355X_MICROARCH.md "Global float
// atomics"): cost ~ number of 64-B requests, so (1) every request must be full and (2) there must be few.
// Layout: 2*C lanes own one sample -- lane = half*C + ch, half 0/1 = x-corner x0 / x0+1.  Channel-last
// storage makes texels (y,x0) and (y,x0+1) adjacent, so ONE atomic wave-instruction covers 2*C contiguous
// floats (256 B for C=32: the full-rate shape).  Each lane group walks RUN consecutive samples (consecutive
// along a ray => spatially coherent) and run-length-combines: while the (plane,row) texel key stays the same
// the contribution is summed in a register; it is flushed with one atomic only when the key changes.
// (v1 of this kernel used float4-per-lane atomics at a 16-B lane stride: quarter-full requests, 12.8 ms at
// config-2 size vs. 0.5 ms for the forward -- profiles/r01_kernels.md.)
// ---------------------------------------------------------------------------------------------
// value of lane (l ^ C): the other x-corner's partial sum.  __shfl_xor compiles to ds_bpermute_b32 (an LDS-crossbar access with a
// per-lane address; SQ_LDS_BANK_CONFLICT showed it 87 % conflicted here); the fixed patterns have cheaper forms
template <int C>
__device__ __forceinline__ float xor_lanes(float x) {
  if (C == 8) {  // rotate by 8 inside each row of 16 lanes: DPP row_ror:8, no LDS involved
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
  } else if (C == 16) {  // swap the two 16-lane halves of each 32: ds_swizzle bit mode, xor mask 0x10 (no memory access, no per-lane address)
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401f));
  } else {
    return __shfl_xor(x, C, 64);
  }
}

template <int C, int NP, bool FX, int U>
__global__ __launch_bounds__(256) void kplanes_gather_bwd_kernel(snerf_kplanes_desc d, const float* __restrict__ planes,
                                                                snerf_coords c, int64_t N, const float* __restrict__ gout,
                                                                float* __restrict__ gplanes, long long* __restrict__ gplanes_fx, int run) {
  constexpr int LPS = 2 * C;  // lanes per sample
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t group = gid / LPS;
  const int li = (int)(gid % LPS);
  const int hmask = -(li / C);  // all ones for the lanes of the x0 + 1 corner
  const int ch = li % C;
  const int64_t n0 = group * run;
  if (n0 >= N) return;
  const int cnt = (int)((N - n0) < run ? (N - n0) : run);
  const int out_w = d.concat ? C * d.n_scales : C;

  for (int s = 0; s < d.n_scales; ++s) {
    int pend_key[NP][2];
    float pend_val[NP][2];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      pend_key[q][0] = pend_key[q][1] = -1;
      pend_val[q][0] = pend_val[q][1] = 0.f;
    }
    int res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) res[k] = d.res[s][k] > 0 ? d.res[s][k] : 1;

    // U samples in flight: their coordinate, texel and upstream-gradient loads are all issued before the first one is consumed
    // mode 1: (ray, sample-in-ray) of the run's first sample by one division, then stepped -- the per-sample n / S inside load_coords
    // was ~100 of the walk's ~350 vector instructions per sample
    int64_t ray = 0;
    int sir = 0;
    if (c.mode == 1) { ray = n0 / c.S; sir = (int)(n0 - ray * c.S); }
    for (int i0 = 0; i0 < cnt; i0 += U) {
      AxisTap tap[U][4];
      int xsel[U][4];    // this lane's x-corner along each axis (i0 for half 0, i1 for half 1) and its weight: bit-selected once per axis
      float wsel[U][4];  // (`half ? t.i1 : t.i0` on the structs became an indexed load from a scratch copy of the taps)
      float ta[U][NP], tb[U][NP], gup[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t n = n0 + (i0 + u < cnt ? i0 + u : cnt - 1);
        float p[4];
        if (c.mode == 1) {
          load_coords_ray(c, ray, sir, p);
          if (i0 + u + 1 < cnt && ++sir == c.S) { sir = 0; ++ray; }
        } else {
          load_coords<NP>(c, n, p);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const AxisTap t = axis_tap(p[k], res[k]);
          tap[u][k] = t;
          xsel[u][k] = (t.i1 & hmask) | (t.i0 & ~hmask);
          wsel[u][k] = __int_as_float((__float_as_int(t.w1) & hmask) | (__float_as_int(t.w0) & ~hmask));
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const AxisTap& ty = tap[u][pair_b<NP>(q)];
          const int W = res[pair_a<NP>(q)];
          const int xi = xsel[u][pair_a<NP>(q)];
          const float* base = planes + d.off[s][q] + ch;
          ta[u][q] = base[((int64_t)ty.i0 * W + xi) * C];
          tb[u][q] = base[((int64_t)ty.i1 * W + xi) * C];
        }
        gup[u] = gout[n * out_w + (d.concat ? s * C : 0) + ch];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + u >= cnt) continue;
        float v[NP];
        float wx[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const AxisTap& ty = tap[u][pair_b<NP>(q)];
          wx[q] = wsel[u][pair_a<NP>(q)];
          float part = wx[q] * (ty.w0 * ta[u][q] + ty.w1 * tb[u][q]);
          v[q] = part + xor_lanes<C>(part);
        }
        float suf[NP + 1];
        suf[NP] = 1.f;
#pragma unroll
        for (int q = NP - 1; q >= 0; --q) suf[q] = suf[q + 1] * v[q];
        float pre = gup[u];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const float gq = pre * suf[q + 1] * wx[q];
          pre *= v[q];
          const AxisTap& tx = tap[u][pair_a<NP>(q)];
          const AxisTap& ty = tap[u][pair_b<NP>(q)];
          const int W = res[pair_a<NP>(q)];
          const int64_t gbase = d.off[s][q] + li;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int key = (r ? ty.i1 : ty.i0) * W + tx.i0;
            const float val = gq * (r ? ty.w1 : ty.w0);
            if (key != pend_key[q][r]) {
              if (pend_val[q][r] != 0.f) grad_add<FX>(gplanes, gplanes_fx, gbase + (int64_t)pend_key[q][r] * C, pend_val[q][r]);
              pend_key[q][r] = key;
              pend_val[q][r] = val;
            } else {
              pend_val[q][r] += val;
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int64_t gbase = d.off[s][q] + li;
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (pend_val[q][r] != 0.f) grad_add<FX>(gplanes, gplanes_fx, gbase + (int64_t)pend_key[q][r] * C, pend_val[q][r]);
    }
  }
}

static int validate(const snerf_kplanes_desc* d, const snerf_coords* c, int64_t N) {
  SNERF_REQUIRE(d && c, "kplanes: null descriptor");
  SNERF_REQUIRE(d->n_scales >= 1 && d->n_scales <= SNERF_MAX_SCALES, "kplanes: n_scales=%d out of range", d->n_scales);
  SNERF_REQUIRE(d->C == 8 || d->C == 16 || d->C == 32, "kplanes: C=%d unsupported (8, 16, 32)", d->C);
  SNERF_REQUIRE(d->n_coords == 3 || d->n_coords == 4, "kplanes: n_coords=%d unsupported", d->n_coords);
  SNERF_REQUIRE(N >= 0, "kplanes: negative N");
  for (int s = 0; s < d->n_scales; ++s)
    for (int k = 0; k < d->n_coords; ++k) SNERF_REQUIRE(d->res[s][k] >= 1, "kplanes: res[%d][%d]=%d", s, k, d->res[s][k]);
  if (c->mode == 0) {
    SNERF_REQUIRE(c->pts || N == 0, "kplanes: pts is null");
  } else if (c->mode == 1) {
    SNERF_REQUIRE(c->S >= 1 && N % c->S == 0, "kplanes: N=%lld not a multiple of S=%d", (long long)N, c->S);
    SNERF_REQUIRE((c->origins && c->dirs && c->times && c->ebins) || N == 0, "kplanes: null ray buffers");
  } else {
    SNERF_REQUIRE(false, "kplanes: coords.mode=%d unsupported", c->mode);
  }
  return 0;
}

template <int C, int NP>
static int launch_fwd(const snerf_kplanes_desc* d, const float* planes, const snerf_coords* c, int64_t N, float* out, hipStream_t st) {
  int64_t threads = N * (C / 4);
  hipLaunchKernelGGL((kplanes_gather_fwd_kernel<C, NP>), dim3(ceil_div(threads, 256)), dim3(256), 0, st, *d, planes, *c, N, out);
  SNERF_LAUNCH_CHECK("kplanes_gather_fwd");
  return 0;
}
template <int C, int NP>
static int launch_bwd(const snerf_kplanes_desc* d, const float* planes, const snerf_coords* c, int64_t N, const float* gout, float* gp,
                      long long* gp_fx, hipStream_t st) {
  // consecutive samples walked (and run-length-combined) by one lane group, and samples in flight per group.  profiles/r02_kernels.md:
  // with the taps out of scratch the proposal levels (C = 8) take 0.30 ms at run 64 and 0.29 ms at run 32; 2 or 4 samples in flight
  // change nothing (0.30-0.32 ms) -- the walk is bound by its memory-side atomic requests, not by load latency.
  constexpr int run = C <= 8 ? 32 : 64;
  constexpr int U = 1;
  int64_t groups = (N + run - 1) / run;
  int64_t threads = groups * (2 * C);
  if (gp_fx) hipLaunchKernelGGL((kplanes_gather_bwd_kernel<C, NP, true, U>), dim3(ceil_div(threads, 256)), dim3(256), 0, st, *d, planes, *c, N, gout, gp, gp_fx, run);
  else hipLaunchKernelGGL((kplanes_gather_bwd_kernel<C, NP, false, U>), dim3(ceil_div(threads, 256)), dim3(256), 0, st, *d, planes, *c, N, gout, gp, gp_fx, run);
  SNERF_LAUNCH_CHECK("kplanes_gather_bwd");
  return 0;
}

}  // namespace snerf

using namespace snerf;

#define DISPATCH_C_NP(FN, ...)                                                          \
  do {                                                                                  \
    if (desc->n_coords == 4) {                                                          \
      if (desc->C == 32) return FN<32, 6>(__VA_ARGS__);                                 \
      if (desc->C == 16) return FN<16, 6>(__VA_ARGS__);                                 \
      return FN<8, 6>(__VA_ARGS__);                                                     \
    } else {                                                                            \
      if (desc->C == 32) return FN<32, 3>(__VA_ARGS__);                                 \
      if (desc->C == 16) return FN<16, 3>(__VA_ARGS__);                                 \
      return FN<8, 3>(__VA_ARGS__);                                                     \
    }                                                                                   \
  } while (0)

extern "C" int snerf_kplanes_gather_fwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                                        float* out, snerf_stream_t stream) {
  int rc = validate(desc, coords, N);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && out, "kplanes_gather_fwd: null buffer");
  DISPATCH_C_NP(launch_fwd, desc, planes, coords, N, out, (hipStream_t)stream);
}

extern "C" int snerf_kplanes_gather_bwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                                        const float* grad_out, float* grad_planes, snerf_stream_t stream) {
  int rc = validate(desc, coords, N);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && grad_out && grad_planes, "kplanes_gather_bwd: null buffer");
  DISPATCH_C_NP(launch_bwd, desc, planes, coords, N, grad_out, grad_planes, nullptr, (hipStream_t)stream);
}

extern "C" int snerf_kplanes_gather_bwd_fx(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                                           const float* grad_out, int64_t* grad_planes_fx, snerf_stream_t stream) {
  int rc = validate(desc, coords, N);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && grad_out && grad_planes_fx, "kplanes_gather_bwd_fx: null buffer");
  DISPATCH_C_NP(launch_bwd, desc, planes, coords, N, grad_out, nullptr, reinterpret_cast<long long*>(grad_planes_fx), (hipStream_t)stream);
}

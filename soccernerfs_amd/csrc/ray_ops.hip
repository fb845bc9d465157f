// Per-ray kernels: spaced bins, density->weights (+bwd), PDF (inverse-CDF) resampling.
//
// One 64-lane wavefront owns one ray (S <= 320 samples => <= 5 elements per lane); 4 rays per 256-thread
// workgroup, per-ray scratch in LDS.  Everything order-sensitive (the CDF that decides sample INDICES) is a
// wavefront prefix scan with DOUBLE accumulators rounded to fp32 per element (common.hpp: wave_scan_f64) --
// the arithmetic of torch.cumsum on the CPU (ATen cumsum_cpu_kernel: acc_type<float,false> = double) -- so
// indices are bit-exact against the oracle; FP contraction is disabled in this file for the same reason.
//
// Reference: NS/model_components/ray_samplers.py (SpacedSampler :79-126, PDFSampler :274-369,
// ProposalNetworkSampler :584), NS/cameras/rays.py:127-149 (get_weights).
#include "common.hpp"

#pragma clang fp contract(off)

namespace snerf {

constexpr int RAYS_PER_BLOCK = 4;
constexpr int MAX_S = 320;  // max samples per ray handled by the per-ray kernels

__device__ __forceinline__ float spacing_fn(float x, int kind) {
  // kind 0: uniform (identity); kind 1: UniformLinDispPiecewise  x<1 ? x/2 : 1-1/(2x)  (ray_samplers.py:242)
  if (kind == 0) return x;
  return x < 1.f ? x / 2.f : 1.f - 1.f / (2.f * x);
}
__device__ __forceinline__ float spacing_fn_inv(float x, int kind) {
  if (kind == 0) return x;
  return x < 0.5f ? 2.f * x : 1.f / (2.f - 2.f * x);  // ray_samplers.py:243
}
__device__ __forceinline__ float to_euclid(float x, float s_near, float s_far, int kind) {
  return spacing_fn_inv(x * s_far + (1.f - x) * s_near, kind);  // ray_samplers.py:115
}

// torch.linspace(start, end, steps)[i] in fp32 (ATen RangeFactories: symmetric halves keep both ends exact)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// ------------------------------------------------------------------------------------------------
// P5  spaced bins
// ------------------------------------------------------------------------------------------------
__global__ void spaced_bins_kernel(const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ t_rand,
                                   int rand_cols, int R, int S, int kind, float* __restrict__ sbins, float* __restrict__ ebins) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)R * (S + 1);
  if (gid >= total) return;
  int r = (int)(gid / (S + 1));
  int i = (int)(gid % (S + 1));
  float b = linspace_at(0.f, 1.f, S + 1, i);
  if (t_rand) {
    // bins = lower + (upper - lower) * t_rand, lower/upper = neighbouring bin centres (ray_samplers.py:108-112)
    float lo, up;
    if (i == 0) lo = b; else lo = (b + linspace_at(0.f, 1.f, S + 1, i - 1)) / 2.f;
    if (i == S) up = b; else up = (linspace_at(0.f, 1.f, S + 1, i + 1) + b) / 2.f;
    float t = rand_cols == 1 ? t_rand[r] : t_rand[(int64_t)r * (S + 1) + i];
    b = lo + (up - lo) * t;
  }
  sbins[gid] = b;
  float sn = spacing_fn(nears[r], kind), sf = spacing_fn(fars[r], kind);
  ebins[gid] = to_euclid(b, sn, sf, kind);
}

// ------------------------------------------------------------------------------------------------
// P10 weights forward:  w = (1 - exp(-d*sigma)) * exp(-excl_cumsum(d*sigma)), nan_to_num
// P6  PDF resampling (optionally fused behind the weights)
// ------------------------------------------------------------------------------------------------
struct ResampleArgs {
  // weights stage
  const float* density;   // [R,Sp] or null (then `weights_in` is used)
  const float* weights_in;  // [R,Sp] or null
  const float* ebins_prev;  // [R,Sp+1] (needed when density != null)
  float* weights_out;       // [R,Sp] or null
  // pdf stage (skipped when sbins_out == null)
  const float* sbins_prev;  // [R,Sp+1]
  const float* u_or_rand;   // mode 0: u [R,S+1]; mode 1: rand [R,S+1] or [R,1]; mode 2: unused
  const float* nears;       // [R]
  const float* fars;        // [R]
  float* sbins_out;         // [R,S+1]
  float* ebins_out;         // [R,S+1]
  int64_t* inds_out;        // [R,S+1] or null
  int R, Sp, S;
  int u_mode, rand_cols, kind;
  float anneal, hist_pad, eps;
};

__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
  __shared__ float s_w[RAYS_PER_BLOCK][MAX_S + 1];    // weights, then pdf
  __shared__ float s_cdf[RAYS_PER_BLOCK][MAX_S + 2];  // cdf (Sp+1 entries)
  __shared__ float s_bins[RAYS_PER_BLOCK][MAX_S + 2];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RAYS_PER_BLOCK + wv;
  const bool live = ray < a.R;
  const int r = live ? ray : a.R - 1;
  const int Sp = a.Sp;
  float* w = s_w[wv];
  float* cdf = s_cdf[wv];
  float* bins = s_bins[wv];

  // ---- stage 1: weights (or load them) ----
  if (a.density) {
    for (int i = lane; i < Sp; i += 64) {
      float e0 = a.ebins_prev[(int64_t)r * (Sp + 1) + i], e1 = a.ebins_prev[(int64_t)r * (Sp + 1) + i + 1];
      w[i] = (e1 - e0) * a.density[(int64_t)r * Sp + i];  // delta * sigma
    }
    __syncthreads();
    wave_scan_f64<true, false>(w, cdf, Sp, lane);  // exclusive cumsum of delta * sigma (rays.py:141-145), double accumulator as ATen's CPU cumsum
    __syncthreads();
    for (int i = lane; i < Sp; i += 64) {
      float dd = w[i];
      float alpha = 1.f - expf(-dd);
      float T = expf(-cdf[i]);
      float wt = nan_to_num(alpha * T);
      w[i] = wt;
      if (a.weights_out && live) a.weights_out[(int64_t)r * Sp + i] = wt;
    }
  } else {
    for (int i = lane; i < Sp; i += 64) w[i] = a.weights_in[(int64_t)r * Sp + i];
  }
  if (!a.sbins_out) return;  // block-uniform
  __syncthreads();

  // ---- stage 2: annealed, padded weights -> pdf -> cdf ----
  for (int i = lane; i < Sp; i += 64) {
    float x = w[i];
    if (a.anneal != 1.f) x = powf(x, a.anneal);  // ray_samplers.py:584
    w[i] = x + a.hist_pad;                       // :302
  }
  for (int i = lane; i <= Sp; i += 64) bins[i] = a.sbins_prev[(int64_t)r * (Sp + 1) + i];
  __syncthreads();
  {
    // weights_sum = cumsum(w)[-1] (double accumulator, fp32 result), padding (:305-308)
    const float s = (float)wave_scan_f64<false, false>(w, cdf + 1, Sp, lane);  // (the prefix sums written to cdf are overwritten below)
    const float padding = fmaxf(a.eps - s, 0.f);  // :306
    const float padd = padding / (float)Sp, tot = s + padding;
    __syncthreads();
    for (int i = lane; i < Sp; i += 64) w[i] = (w[i] + padd) / tot;  // pdf (:307-310)
  }
  __syncthreads();
  wave_scan_f64<false, false>(w, cdf + 1, Sp, lane);  // cumsum(pdf)
  __syncthreads();
  for (int i = lane; i <= Sp; i += 64) cdf[i] = i == 0 ? 0.f : fminf(1.f, cdf[i]);  // min(ones, cumsum) then the leading zero (:311-312)
  __syncthreads();

  // ---- stage 3: inverse-CDF sampling ----
  const int nb = a.S + 1;
  const float sn = spacing_fn(a.nears[r], a.kind), sf = spacing_fn(a.fars[r], a.kind);
  for (int j = lane; j < nb; j += 64) {
    float u;
    if (a.u_mode == 0) {
      u = a.u_or_rand[(int64_t)r * nb + j];
    } else {
      u = linspace_at(0.f, 1.f - (1.f / (float)nb), nb, j);  // :316
      if (a.u_mode == 1) {
        float rnd = a.rand_cols == 1 ? a.u_or_rand[r] : a.u_or_rand[(int64_t)r * nb + j];
        u = u + rnd / (float)nb;  // :318-322
      } else {
        u = u + 1.f / (float)(2 * nb);  // :326
      }
    }
    // searchsorted(cdf, u, side="right"): number of entries <= u
    int lo = 0, hi = Sp + 1;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int ind = lo;
    int below = ind - 1; below = below < 0 ? 0 : (below > Sp ? Sp : below);
    int above = ind;     above = above < 0 ? 0 : (above > Sp ? Sp : above);
    float c0 = cdf[below], c1 = cdf[above];
    float b0 = bins[below], b1 = bins[above];
    float t = (u - c0) / (c1 - c0);
    t = nan_to_num(t);  // nan -> 0, +-inf -> +-FLT_MAX
    t = fminf(fmaxf(t, 0.f), 1.f);
    float nbin = b0 + t * (b1 - b0);
    if (live) {
      a.sbins_out[(int64_t)r * nb + j] = nbin;
      a.ebins_out[(int64_t)r * nb + j] = to_euclid(nbin, sn, sf, a.kind);
      if (a.inds_out) a.inds_out[(int64_t)r * nb + j] = ind;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weights backward: g_sigma_k = delta_k * ( g_k * T_k * exp(-dd_k) - sum_{i>k} g_i * w_i )
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weights_bwd_kernel(const float* __restrict__ density, const float* __restrict__ ebins,
                                                         const float* __restrict__ gw, int R, int S, float* __restrict__ gdens,
                                                         int accumulate, int32_t* __restrict__ nonfinite_flag) {
  __shared__ float s_a[RAYS_PER_BLOCK][MAX_S + 1];
  __shared__ float s_b[RAYS_PER_BLOCK][MAX_S + 1];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RAYS_PER_BLOCK + wv;
  const bool live = ray < R;
  const int r = live ? ray : R - 1;
  float* dd = s_a[wv];
  float* aux = s_b[wv];
  for (int i = lane; i < S; i += 64) {
    float e0 = ebins[(int64_t)r * (S + 1) + i], e1 = ebins[(int64_t)r * (S + 1) + i + 1];
    dd[i] = (e1 - e0) * density[(int64_t)r * S + i];
  }
  __syncthreads();
  wave_scan_f64<true, false>(dd, aux, S, lane);  // exclusive cumsum
  __syncthreads();
  // gw_i * w_i (0 where the forward weight was non-finite: nan_to_num passes no gradient there)
  float gwterm[5], Tk[5], ek[5];
  bool bad = false;  // a value that autograd would have turned into a non-finite gradient (the reference's GradScaler then skips the step)
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int i = lane + 64 * k;
    gwterm[k] = 0.f; Tk[k] = 0.f; ek[k] = 0.f;
    if (i < S) {
      float T = expf(-aux[i]);
      float e = expf(-dd[i]);
      float wraw = (1.f - e) * T;
      float g = gw[(int64_t)r * S + i];
      bool fin = (wraw == wraw) && fabsf(wraw) != INFINITY;
      Tk[k] = T; ek[k] = e;
      gwterm[k] = fin ? g * wraw : 0.f;
      // a non-finite weight (density = exp(x) overflowed to inf on a zero-width bin: 0 * inf) passes no gradient, as nan_to_num
      // does -- and must not leave a NaN factor behind: 0 * NaN is NaN (seen once in ~20 k training steps: one NaN here reaches
      // every parameter through the sigma net and the TV stencil within a few steps)
      if (!fin) { Tk[k] = 0.f; ek[k] = 0.f; bad = bad || live; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int i = lane + 64 * k;
    if (i < S) aux[i] = gwterm[k];
  }
  __syncthreads();
  wave_scan_f64<true, true>(aux, aux, S, lane);  // suffix sums: aux[i] = sum_{j > i}
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int i = lane + 64 * k;
    if (i < S && live) {
      float g = gw[(int64_t)r * S + i];
      float e0 = ebins[(int64_t)r * (S + 1) + i], e1 = ebins[(int64_t)r * (S + 1) + i + 1];
      float gdd = g * Tk[k] * ek[k] - aux[i];
      float out = gdd * (e1 - e0);
      if (!(fabsf(out) <= 3.402823466e+38f)) { out = 0.f; bad = true; }  // inf * 0 / NaN: no gradient (the reference's GradScaler skips such a step)
      if (accumulate) gdens[(int64_t)r * S + i] += out; else gdens[(int64_t)r * S + i] = out;
    }
  }
  if (bad && nonfinite_flag) *nonfinite_flag = 1;  // plain store of a constant: racing writers agree
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_spaced_bins(const float* nears, const float* fars, const float* t_rand, int32_t rand_cols, int32_t R, int32_t S,
                                 int32_t kind, float* sbins, float* ebins, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1, "spaced_bins: R=%d S=%d", R, S);
  SNERF_REQUIRE(kind == 0 || kind == 1, "spaced_bins: kind=%d (0 uniform, 1 piecewise)", kind);
  SNERF_REQUIRE(!t_rand || rand_cols == 1 || rand_cols == S + 1, "spaced_bins: t_rand must be [R,1] or [R,S+1] (cols=%d)", rand_cols);
  if (R == 0) return 0;
  SNERF_REQUIRE(nears && fars && sbins && ebins, "spaced_bins: null buffer");
  int64_t total = (int64_t)R * (S + 1);
  hipLaunchKernelGGL(spaced_bins_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, nears, fars, t_rand, rand_cols, R, S,
                     kind, sbins, ebins);
  SNERF_LAUNCH_CHECK("spaced_bins");
  return 0;
}

static int launch_resample(const ResampleArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(resample_kernel, dim3(ceil_div(a.R, RAYS_PER_BLOCK)), dim3(256), 0, st, a);
  SNERF_LAUNCH_CHECK("resample");
  return 0;
}

extern "C" int snerf_weights_fwd(const float* density, const float* ebins, int32_t R, int32_t S, float* weights, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1 && S <= MAX_S, "weights_fwd: R=%d S=%d (S <= %d)", R, S, MAX_S);
  if (R == 0) return 0;
  SNERF_REQUIRE(density && ebins && weights, "weights_fwd: null buffer");
  ResampleArgs a = {};
  a.density = density; a.ebins_prev = ebins; a.weights_out = weights; a.R = R; a.Sp = S;
  return launch_resample(a, (hipStream_t)stream);
}

extern "C" int snerf_weights_bwd(const float* density, const float* ebins, const float* grad_weights, int32_t R, int32_t S,
                                 float* grad_density, int32_t accumulate, int32_t* nonfinite_flag, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1 && S <= MAX_S, "weights_bwd: R=%d S=%d (S <= %d)", R, S, MAX_S);
  if (R == 0) return 0;
  SNERF_REQUIRE(density && ebins && grad_weights && grad_density, "weights_bwd: null buffer");
  hipLaunchKernelGGL(weights_bwd_kernel, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, density, ebins, grad_weights,
                     R, S, grad_density, accumulate, nonfinite_flag);
  SNERF_LAUNCH_CHECK("weights_bwd");
  return 0;
}

extern "C" int snerf_pdf_resample(const snerf_resample_args* p, snerf_stream_t stream) {
  SNERF_REQUIRE(p, "pdf_resample: null args");
  SNERF_REQUIRE(p->R >= 0 && p->S_prev >= 1 && p->S_prev <= MAX_S && p->S >= 1 && p->S + 1 <= MAX_S + 1, "pdf_resample: R=%d S_prev=%d S=%d",
                p->R, p->S_prev, p->S);
  SNERF_REQUIRE(p->u_mode >= 0 && p->u_mode <= 2, "pdf_resample: u_mode=%d", p->u_mode);
  SNERF_REQUIRE(p->kind == 0 || p->kind == 1, "pdf_resample: kind=%d", p->kind);
  if (p->R == 0) return 0;
  SNERF_REQUIRE((p->density && p->ebins_prev) || p->weights_in, "pdf_resample: need density+ebins_prev or weights_in");
  SNERF_REQUIRE(p->sbins_prev && p->nears && p->fars && p->sbins_out && p->ebins_out, "pdf_resample: null buffer");
  SNERF_REQUIRE(p->u_mode == 2 || p->u_or_rand, "pdf_resample: u/rand buffer is null");
  SNERF_REQUIRE(p->u_mode != 1 || p->rand_cols == 1 || p->rand_cols == p->S + 1, "pdf_resample: rand must be [R,1] or [R,S+1]");
  ResampleArgs a = {};
  a.density = p->density; a.weights_in = p->density ? nullptr : p->weights_in; a.ebins_prev = p->ebins_prev; a.weights_out = p->weights_out;
  a.sbins_prev = p->sbins_prev; a.u_or_rand = p->u_or_rand; a.nears = p->nears; a.fars = p->fars;
  a.sbins_out = p->sbins_out; a.ebins_out = p->ebins_out; a.inds_out = p->inds_out;
  a.R = p->R; a.Sp = p->S_prev; a.S = p->S; a.u_mode = p->u_mode; a.rand_cols = p->rand_cols; a.kind = p->kind;
  a.anneal = p->anneal; a.hist_pad = p->histogram_padding; a.eps = p->eps;
  return launch_resample(a, (hipStream_t)stream);
}

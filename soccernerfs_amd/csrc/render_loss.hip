// Per-ray compositing (renderers) and the ray-level losses (distortion, interlevel), forward + backward.
//
// One wavefront per ray, 4 rays per workgroup, per-ray scratch in LDS.
// Reference: NS/model_components/renderers.py (RGBRenderer :58-140, AccumulationRenderer :197-223, DepthRenderer
// :226-287, MedianRGBRenderer :290-362) and NS/model_components/losses.py (outer :46-75, lossfun_outer :78-95,
// interlevel_loss :106-121, lossfun_distortion :125-136).  The reference's distortion loss materialises an
// [R,S,S] tensor (67 MB at the preset); here the S x S interaction stays in LDS/registers.
#include "common.hpp"

#pragma clang fp contract(off)

namespace snerf {

constexpr int RPB = 4;
constexpr int MAXS = 320;

// ------------------------------------------------------------------------------------------------
// render forward
// ------------------------------------------------------------------------------------------------
struct RenderArgs {
  const float* weights;  // [R,S]
  const float* rgb;      // [R,S,3]
  const float* ebins;    // [R,S+1]
  const float* bg;       // bg_mode 0: [R,3]; 2: [3]; 1 (last_sample): unused
  int R, S, bg_mode, training;
  float* rgb_out;        // [R,3]
  float* acc_out;        // [R]
  float* depth_median;   // [R] or null
  float* depth_expected; // [R] or null (unclipped: sum(w*steps)/(sum(w)+1e-10))
  float* median_rgb;     // [R,3] or null
  int64_t* median_index; // [R] or null
};

__global__ __launch_bounds__(256) void render_fwd_kernel(RenderArgs a) {
  __shared__ float s_w[RPB][MAXS];
  __shared__ float s_cum[RPB][MAXS];
  __shared__ int s_med[RPB];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wv;
  const bool live = ray < a.R;
  const int r = live ? ray : a.R - 1;
  const int S = a.S;
  float cr = 0.f, cg = 0.f, cb = 0.f, acc = 0.f, dsum = 0.f;
  for (int i = lane; i < S; i += 64) {
    float w = a.weights[(int64_t)r * S + i];
    s_w[wv][i] = w;
    const float* c = a.rgb + ((int64_t)r * S + i) * 3;
    float x = c[0], y = c[1], z = c[2];
    if (!a.training) { x = nan_to_num(x); y = nan_to_num(y); z = nan_to_num(z); }  // renderers.py:133-134
    cr += w * x; cg += w * y; cb += w * z;
    acc += w;
    float e0 = a.ebins[(int64_t)r * (S + 1) + i], e1 = a.ebins[(int64_t)r * (S + 1) + i + 1];
    dsum += w * ((e0 + e1) / 2.f);
  }
  cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); acc = wave_sum(acc); dsum = wave_sum(dsum);
  __syncthreads();
  {
    // median: first index with cumsum(w) >= 0.5 (searchsorted left), clamped (renderers.py:264-267); cumsum by wavefront scan in double
    wave_scan_f64<false, false>(s_w[wv], s_cum[wv], S, lane);
    __syncthreads();
    int idx = S;
    for (int i = lane; i < S; i += 64)
      if (s_cum[wv][i] >= 0.5f) { idx = i; break; }  // the cumsum of non-negative weights is monotone: the lane's first hit is its smallest
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(idx, off, 64); idx = o < idx ? o : idx; }
    if (idx > S - 1) idx = S - 1;
    if (lane == 0) s_med[wv] = idx;
  }
  __syncthreads();
  if (lane == 0 && live) {
    const float* last = a.rgb + ((int64_t)r * S + (S - 1)) * 3;
    float b0, b1, b2;
    if (a.bg_mode == 0) { b0 = a.bg[(int64_t)r * 3]; b1 = a.bg[(int64_t)r * 3 + 1]; b2 = a.bg[(int64_t)r * 3 + 2]; }
    else if (a.bg_mode == 1) {
      b0 = last[0]; b1 = last[1]; b2 = last[2];
      if (!a.training) { b0 = nan_to_num(b0); b1 = nan_to_num(b1); b2 = nan_to_num(b2); }
    } else { b0 = a.bg[0]; b1 = a.bg[1]; b2 = a.bg[2]; }
    float o0 = cr + b0 * (1.f - acc), o1 = cg + b1 * (1.f - acc), o2 = cb + b2 * (1.f - acc);  // renderers.py:113
    if (!a.training) { o0 = fminf(fmaxf(o0, 0.f), 1.f); o1 = fminf(fmaxf(o1, 0.f), 1.f); o2 = fminf(fmaxf(o2, 0.f), 1.f); }
    a.rgb_out[(int64_t)r * 3] = o0; a.rgb_out[(int64_t)r * 3 + 1] = o1; a.rgb_out[(int64_t)r * 3 + 2] = o2;
    a.acc_out[r] = acc;
    const int m = s_med[wv];
    if (a.median_index) a.median_index[r] = m;
    if (a.depth_median) {
      float e0 = a.ebins[(int64_t)r * (S + 1) + m], e1 = a.ebins[(int64_t)r * (S + 1) + m + 1];
      a.depth_median[r] = (e0 + e1) / 2.f;
    }
    if (a.depth_expected) a.depth_expected[r] = dsum / (acc + 1e-10f);
    if (a.median_rgb) {
      const float* c = a.rgb + ((int64_t)r * S + m) * 3;
      float x = c[0], y = c[1], z = c[2];
      if (!a.training) {
        x = fminf(fmaxf(nan_to_num(x), 0.f), 1.f); y = fminf(fmaxf(nan_to_num(y), 0.f), 1.f); z = fminf(fmaxf(nan_to_num(z), 0.f), 1.f);
      }
      a.median_rgb[(int64_t)r * 3] = x; a.median_rgb[(int64_t)r * 3 + 1] = y; a.median_rgb[(int64_t)r * 3 + 2] = z;
    }
  }
}

// render backward (training): rgb_out = sum_s w_s rgb_s + bg (1 - sum_s w_s); acc = sum_s w_s
__global__ void render_bwd_kernel(const float* __restrict__ weights, const float* __restrict__ rgb, const float* __restrict__ bg, int bg_mode,
                                  const float* __restrict__ g_rgb_out, const float* __restrict__ g_acc, int R, int S,
                                  float* __restrict__ g_weights, float* __restrict__ g_rgb, int accumulate_w) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)R * S) return;
  int r = (int)(gid / S);
  float go0 = g_rgb_out[(int64_t)r * 3], go1 = g_rgb_out[(int64_t)r * 3 + 1], go2 = g_rgb_out[(int64_t)r * 3 + 2];
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
  if (bg_mode == 0) { b0 = bg[(int64_t)r * 3]; b1 = bg[(int64_t)r * 3 + 1]; b2 = bg[(int64_t)r * 3 + 2]; }
  else if (bg_mode == 2) { b0 = bg[0]; b1 = bg[1]; b2 = bg[2]; }
  const float* c = rgb + gid * 3;
  float w = weights[gid];
  float gw = go0 * (c[0] - b0) + go1 * (c[1] - b1) + go2 * (c[2] - b2);
  if (g_acc) gw += g_acc[r];
  if (accumulate_w) g_weights[gid] += gw; else g_weights[gid] = gw;
  if (g_rgb) { g_rgb[gid * 3] = go0 * w; g_rgb[gid * 3 + 1] = go1 * w; g_rgb[gid * 3 + 2] = go2 * w; }
}

// render backward with the MSE image loss folded in: g_rgb_out = go_scale * (rgb_out - target) is formed per lane instead of by four
// elementwise launches (sub, square-mean, scale); sqerr_rays[r] = sum_c (rgb_out - target)^2 for the (lazy) loss value
__global__ void render_mse_bwd_kernel(const float* __restrict__ weights, const float* __restrict__ rgb, const float* __restrict__ bg, int bg_mode,
                                      const float* __restrict__ rgb_out, const float* __restrict__ target, float go_scale, int R, int S,
                                      float* __restrict__ g_weights, float* __restrict__ g_rgb, float* __restrict__ sqerr_rays) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)R * S) return;
  int r = (int)(gid / S);
  const float d0 = rgb_out[(int64_t)r * 3] - target[(int64_t)r * 3], d1 = rgb_out[(int64_t)r * 3 + 1] - target[(int64_t)r * 3 + 1],
              d2 = rgb_out[(int64_t)r * 3 + 2] - target[(int64_t)r * 3 + 2];
  if (sqerr_rays && gid == (int64_t)r * S) sqerr_rays[r] = (d0 * d0 + d1 * d1) + d2 * d2;
  const float go0 = d0 * go_scale, go1 = d1 * go_scale, go2 = d2 * go_scale;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
  if (bg_mode == 0) { b0 = bg[(int64_t)r * 3]; b1 = bg[(int64_t)r * 3 + 1]; b2 = bg[(int64_t)r * 3 + 2]; }
  else if (bg_mode == 2) { b0 = bg[0]; b1 = bg[1]; b2 = bg[2]; }
  const float* c = rgb + gid * 3;
  const float w = weights[gid];
  g_weights[gid] = go0 * (c[0] - b0) + go1 * (c[1] - b1) + go2 * (c[2] - b2);
  if (g_rgb) { g_rgb[gid * 3] = go0 * w; g_rgb[gid * 3 + 1] = go1 * w; g_rgb[gid * 3 + 2] = go2 * w; }
}

// ------------------------------------------------------------------------------------------------
// distortion loss (per-ray value + gradient w.r.t. weights)
//   L_r = sum_i w_i sum_j w_j |m_i - m_j| + (1/3) sum_i w_i^2 (t_{i+1} - t_i),  m = bin midpoints
//   dL_r/dw_i = 2 sum_j w_j |m_i - m_j| + (2/3) w_i (t_{i+1} - t_i)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void distortion_kernel(const float* __restrict__ weights, const float* __restrict__ sbins, int R, int S,
                                                        float grad_scale, float* __restrict__ loss_rays, float* __restrict__ g_weights,
                                                        int accumulate) {
  __shared__ float s_w[RPB][MAXS];
  __shared__ float s_m[RPB][MAXS];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wv;
  const bool live = ray < R;
  const int r = live ? ray : R - 1;
  for (int i = lane; i < S; i += 64) {
    float t0 = sbins[(int64_t)r * (S + 1) + i], t1 = sbins[(int64_t)r * (S + 1) + i + 1];
    s_w[wv][i] = weights[(int64_t)r * S + i];
    s_m[wv][i] = (t1 + t0) / 2.f;
  }
  __syncthreads();
  float total = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float wi = s_w[wv][i], mi = s_m[wv][i];
    float inner = 0.f;
    for (int j = 0; j < S; ++j) inner += s_w[wv][j] * fabsf(mi - s_m[wv][j]);
    float t0 = sbins[(int64_t)r * (S + 1) + i], t1 = sbins[(int64_t)r * (S + 1) + i + 1];
    float dt = t1 - t0;
    total += wi * inner + wi * wi * dt / 3.f;
    if (g_weights && live) {
      float g = (2.f * inner + 2.f * wi * dt / 3.f) * grad_scale;
      if (accumulate) g_weights[(int64_t)r * S + i] += g; else g_weights[(int64_t)r * S + i] = g;
    }
  }
  total = wave_sum(total);
  if (lane == 0 && live && loss_rays) loss_rays[r] = total;
}

// ------------------------------------------------------------------------------------------------
// interlevel (proposal) loss for ONE proposal level against the (detached) nerf level:
//   w_outer_i = cy[hi_i + 1] - cy[lo_i],  cy = [0, cumsum(wp)]
//   lo_i = clamp(searchsorted(tp[:-1], c_i, right) - 1, 0, Sp-1),  hi_i = clamp(searchsorted(tp[1:], c_{i+1}, right), 0, Sp-1)
//   loss_i = max(w_i - w_outer_i, 0)^2 / (w_i + 1e-7)
// outputs per-ray sum of loss_i and d(sum)/d wp (scaled by grad_scale)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interlevel_kernel(const float* __restrict__ c_bins, const float* __restrict__ w_nerf, int S,
                                                        const float* __restrict__ p_bins, const float* __restrict__ w_prop, int Sp, int R,
                                                        float grad_scale, float* __restrict__ loss_rays, float* __restrict__ g_wprop) {
  __shared__ float s_tp[RPB][MAXS + 1];
  __shared__ float s_cy[RPB][MAXS + 1];
  __shared__ float s_g[RPB][MAXS];
  __shared__ int s_lo[RPB][MAXS];
  __shared__ int s_hi[RPB][MAXS];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wv;
  const bool live = ray < R;
  const int r = live ? ray : R - 1;
  float* tp = s_tp[wv];
  float* cy = s_cy[wv];
  for (int i = lane; i <= Sp; i += 64) tp[i] = p_bins[(int64_t)r * (Sp + 1) + i];
  for (int i = lane; i < Sp; i += 64) cy[i + 1] = w_prop[(int64_t)r * Sp + i];
  __syncthreads();
  wave_scan_f64<false, false>(cy + 1, cy + 1, Sp, lane);  // cy = [0, cumsum(wp)], double accumulator
  if (lane == 0) cy[0] = 0.f;
  __syncthreads();
  float total = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float c0 = c_bins[(int64_t)r * (S + 1) + i], c1 = c_bins[(int64_t)r * (S + 1) + i + 1];
    // searchsorted(tp[0..Sp-1], c0, right)
    int lo = 0, hi = Sp;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (tp[mid] <= c0) lo = mid + 1; else hi = mid; }
    int ilo = lo - 1; ilo = ilo < 0 ? 0 : (ilo > Sp - 1 ? Sp - 1 : ilo);
    // searchsorted(tp[1..Sp], c1, right)
    lo = 0; hi = Sp;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (tp[mid + 1] <= c1) lo = mid + 1; else hi = mid; }
    int ihi = lo; ihi = ihi < 0 ? 0 : (ihi > Sp - 1 ? Sp - 1 : ihi);
    const float wo = cy[ihi + 1] - cy[ilo];
    const float w = w_nerf[(int64_t)r * S + i];
    const float d = fmaxf(w - wo, 0.f);
    total += d * d / (w + 1.0e-7f);
    s_lo[wv][i] = ilo; s_hi[wv][i] = ihi;
    s_g[wv][i] = -2.f * d / (w + 1.0e-7f);  // d loss_i / d w_outer_i (0 when clipped)
  }
  total = wave_sum(total);
  if (lane == 0 && live && loss_rays) loss_rays[r] = total;
  __syncthreads();
  if (g_wprop) {
    // d loss / d w_prop[j] = sum of g_i over the nerf intervals whose envelope [lo_i, hi_i] holds j.  Both bounds are searchsorted results of
    // ascending queries, i.e. non-decreasing in i, so those intervals are ONE contiguous range [a_j, b_j]: a_j = first i with hi_i >= j,
    // b_j = last i with lo_i <= j (two binary searches), summed in ascending i -- the order (and therefore the bits) of the dense
    // "for every i: if (lo_i <= j <= hi_i)" loop this replaces, in O(S + Sp) instead of O(S Sp) LDS reads per ray.
    const int* slo = s_lo[wv];
    const int* shi = s_hi[wv];
    for (int j = lane; j < Sp; j += 64) {
      int a = 0, e = S;
      while (a < e) { const int mid = (a + e) >> 1; if (shi[mid] < j) a = mid + 1; else e = mid; }   // a = first i with hi_i >= j
      int b = 0;
      e = S;
      while (b < e) { const int mid = (b + e) >> 1; if (slo[mid] <= j) b = mid + 1; else e = mid; }  // b = first i with lo_i > j
      float acc = 0.f;
      for (int i = a; i < b; ++i) acc += s_g[wv][i];
      if (live) g_wprop[(int64_t)r * Sp + j] = acc * grad_scale;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ds_nerf_depth_loss (losses.py:213-235) behind depth_loss (:261-311, DS_NERF branch) for one sampling level:
//   loss_r = [D_r > 0] * sum_s -log(w_s + 1e-7) * exp(-(t_s - D_r)^2 / (2 sigma)) * (e_{s+1} - e_s),  t_s = (e_s + e_{s+1}) / 2,
//   D_r = termination depth (x directions_norm when the depth maps hold z-distances), mean over rays by the caller's grad_scale.
// One wavefront per ray; g_weights (+)= grad_scale * d loss_r / d w.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_loss_kernel(const float* __restrict__ weights, const float* __restrict__ ebins, const float* __restrict__ term,
                                                        const float* __restrict__ dir_norm, float sigma, int R, int S, float grad_scale,
                                                        float* __restrict__ loss_rays, float* __restrict__ g_weights, int accumulate) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wv;
  if (ray >= R) return;
  float D = term[ray];
  if (dir_norm) D = D * dir_norm[ray];
  const bool on = D > 0.f;
  float total = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float e0 = ebins[(int64_t)ray * (S + 1) + i], e1 = ebins[(int64_t)ray * (S + 1) + i + 1];
    const float w = weights[(int64_t)ray * S + i];
    const float t = (e0 + e1) / 2.f;
    const float k = expf(-((t - D) * (t - D)) / (2.f * sigma)) * (e1 - e0);
    total += -logf(w + 1.0e-7f) * k;
    if (g_weights) {
      const float g = on ? grad_scale * (-k / (w + 1.0e-7f)) : 0.f;
      if (accumulate) g_weights[(int64_t)ray * S + i] += g; else g_weights[(int64_t)ray * S + i] = g;
    }
  }
  total = wave_sum(total);
  if (lane == 0 && loss_rays) loss_rays[ray] = on ? total : 0.f;
}

// ------------------------------------------------------------------------------------------------
// urban_radiance_field_depth_loss (losses.py:238-274) behind depth_loss (:261-311, URF branch) for one sampling level, D as above:
//   loss_r = [D_r > 0] * ( (D_r - d_r)^2                                          d_r = the level's predicted depth
//                        + sum_{|t_s - D_r| <= sigma} (w_s - N(t_s - D_r; 0, sigma / 3))^2   "line of sight", near the surface
//                        + sum_{t_s < D_r - sigma} w_s^2 )                                   ... and in front of it
// with N the normal density (torch.distributions.Normal.log_prob, exponentiated).  g_weights (+)= grad_scale * d loss_r / d w,
// g_pred (may be NULL) = grad_scale * d loss_r / d d_r.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void urf_depth_loss_kernel(const float* __restrict__ weights, const float* __restrict__ ebins, const float* __restrict__ term,
                                                            const float* __restrict__ dir_norm, const float* __restrict__ pred, float sigma, int R, int S,
                                                            float grad_scale, float* __restrict__ loss_rays, float* __restrict__ g_weights,
                                                            float* __restrict__ g_pred, int accumulate) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wv;
  if (ray >= R) return;
  float D = term[ray];
  if (dir_norm) D = D * dir_norm[ray];
  const bool on = D > 0.f;
  const float sd = sigma / 3.0f;  // URF_SIGMA_SCALE_FACTOR (losses.py:36)
  const float log_norm = -logf(sd) - 0.918938533204672742f;  // - log(sd) - log(sqrt(2 pi))
  float total = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float e0 = ebins[(int64_t)ray * (S + 1) + i], e1 = ebins[(int64_t)ray * (S + 1) + i + 1];
    const float w = weights[(int64_t)ray * S + i];
    const float t = (e0 + e1) / 2.f;
    const float x = t - D;
    float term_i = 0.f, g = 0.f;
    if (t <= D + sigma && t >= D - sigma) {
      const float pdf = expf(-(x * x) / (2.f * sd * sd) + log_norm);
      term_i = (w - pdf) * (w - pdf);
      g = 2.f * (w - pdf);
    } else if (t < D - sigma) {
      term_i = w * w;
      g = 2.f * w;
    }
    total += term_i;
    if (g_weights) {
      const float gv = on ? grad_scale * g : 0.f;
      if (accumulate) g_weights[(int64_t)ray * S + i] += gv; else g_weights[(int64_t)ray * S + i] = gv;
    }
  }
  total = wave_sum(total);
  if (lane == 0) {
    const float d = pred[ray];
    if (loss_rays) loss_rays[ray] = on ? total + (D - d) * (D - d) : 0.f;
    if (g_pred) g_pred[ray] = on ? grad_scale * -2.f * (D - d) : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// One launch for the nerf level's per-ray work of a TRAINING step: get_weights -> RGB / accumulation / median depth -> MSE backward ->
// distortion loss + gradient -> get_weights backward.  Same arithmetic, in the same order, as the five kernels it stands for
// (resample_kernel stage 1, render_fwd_kernel, render_mse_bwd_kernel, distortion_kernel, weights_bwd_kernel: results are bit-identical,
// tests/test_gpu_render_loss.py); on the step's critical path those five cost ~0.1 ms of launches and barriers for ~20 us of work.
// ------------------------------------------------------------------------------------------------
struct RayTrainArgs {
  const float* density; const float* ebins; const float* sbins; const float* rgb; const float* bg; const float* target;
  int R, S, bg_mode;
  float go_scale, dist_scale;
  float* weights; float* rgb_out; float* acc_out; float* depth_median; float* sqerr_rays; float* dist_rays;
  float* g_rgb; float* g_density; float* g_weights; int32_t* nonfinite_flag;
};

__global__ __launch_bounds__(256) void ray_train_kernel(RayTrainArgs a) {
  __shared__ float s_dd[RPB][MAXS + 1];   // delta * sigma
  __shared__ float s_aux[RPB][MAXS + 1];  // exclusive cumsum of it; later the suffix sums of the backward
  __shared__ float s_w[RPB][MAXS + 1];    // weights
  __shared__ float s_m[RPB][MAXS + 1];    // cumsum of w (median), then the s-space bin midpoints
  __shared__ float s_g[RPB][MAXS + 1];    // d loss / d weights
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wv;
  const bool live = ray < a.R;
  const int r = live ? ray : a.R - 1;
  const int S = a.S;
  float *dd = s_dd[wv], *aux = s_aux[wv], *w = s_w[wv], *mm = s_m[wv], *gw = s_g[wv];
  const float* eb = a.ebins + (int64_t)r * (S + 1);

  // ---- get_weights (resample_kernel stage 1; rays.py:127-149) ----
  for (int i = lane; i < S; i += 64) {
    const float e0 = eb[i], e1 = eb[i + 1];
    dd[i] = (e1 - e0) * a.density[(int64_t)r * S + i];
  }
  __syncthreads();
  wave_scan_f64<true, false>(dd, aux, S, lane);
  __syncthreads();
  for (int i = lane; i < S; i += 64) {
    const float alpha = 1.f - expf(-dd[i]);
    const float T = expf(-aux[i]);
    const float wt = nan_to_num(alpha * T);
    w[i] = wt;
    if (live) a.weights[(int64_t)r * S + i] = wt;
  }
  __syncthreads();

  // ---- render forward, training mode (render_fwd_kernel) ----
  float cr = 0.f, cg = 0.f, cb = 0.f, acc = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float wi = w[i];
    const float* c = a.rgb + ((int64_t)r * S + i) * 3;
    const float x = c[0], y = c[1], z = c[2];
    cr += wi * x; cg += wi * y; cb += wi * z;
    acc += wi;
  }
  cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); acc = wave_sum(acc);
  if (a.depth_median) {
    wave_scan_f64<false, false>(w, mm, S, lane);
    __syncthreads();
    int idx = S;
    for (int i = lane; i < S; i += 64)
      if (mm[i] >= 0.5f) { idx = i; break; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(idx, off, 64); idx = o < idx ? o : idx; }
    if (idx > S - 1) idx = S - 1;
    if (lane == 0 && live) {
      const float e0 = eb[idx], e1 = eb[idx + 1];
      a.depth_median[r] = (e0 + e1) / 2.f;
    }
    __syncthreads();
  }
  float b0, b1, b2;
  if (a.bg_mode == 0) { b0 = a.bg[(int64_t)r * 3]; b1 = a.bg[(int64_t)r * 3 + 1]; b2 = a.bg[(int64_t)r * 3 + 2]; }
  else { b0 = a.bg[0]; b1 = a.bg[1]; b2 = a.bg[2]; }
  const float o0 = cr + b0 * (1.f - acc), o1 = cg + b1 * (1.f - acc), o2 = cb + b2 * (1.f - acc);
  if (lane == 0 && live) {
    a.rgb_out[(int64_t)r * 3] = o0; a.rgb_out[(int64_t)r * 3 + 1] = o1; a.rgb_out[(int64_t)r * 3 + 2] = o2;
    a.acc_out[r] = acc;
  }

  // ---- MSE folded into the render backward (render_mse_bwd_kernel) ----
  const float d0 = o0 - a.target[(int64_t)r * 3], d1 = o1 - a.target[(int64_t)r * 3 + 1], d2 = o2 - a.target[(int64_t)r * 3 + 2];
  if (lane == 0 && live && a.sqerr_rays) a.sqerr_rays[r] = (d0 * d0 + d1 * d1) + d2 * d2;
  const float go0 = d0 * a.go_scale, go1 = d1 * a.go_scale, go2 = d2 * a.go_scale;
  for (int i = lane; i < S; i += 64) {
    const float* c = a.rgb + ((int64_t)r * S + i) * 3;
    gw[i] = go0 * (c[0] - b0) + go1 * (c[1] - b1) + go2 * (c[2] - b2);
    if (live) {
      float* g = a.g_rgb + ((int64_t)r * S + i) * 3;
      const float wi = w[i];
      g[0] = go0 * wi; g[1] = go1 * wi; g[2] = go2 * wi;
    }
  }

  // ---- distortion loss and its gradient, added to gw (distortion_kernel with accumulate = 1) ----
  const float* sb = a.sbins + (int64_t)r * (S + 1);
  for (int i = lane; i < S; i += 64) {
    const float t0 = sb[i], t1 = sb[i + 1];
    mm[i] = (t1 + t0) / 2.f;
  }
  __syncthreads();
  float total = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float wi = w[i], mi = mm[i];
    float inner = 0.f;
    for (int j = 0; j < S; ++j) inner += w[j] * fabsf(mi - mm[j]);
    const float t0 = sb[i], t1 = sb[i + 1];
    const float dt = t1 - t0;
    total += wi * inner + wi * wi * dt / 3.f;
    const float g = (2.f * inner + 2.f * wi * dt / 3.f) * a.dist_scale;
    gw[i] += g;
  }
  total = wave_sum(total);
  if (lane == 0 && live && a.dist_rays) a.dist_rays[r] = total;
  if (a.g_weights && live)
    for (int i = lane; i < S; i += 64) a.g_weights[(int64_t)r * S + i] = gw[i];

  // ---- get_weights backward (weights_bwd_kernel) ----
  float gwterm[5], Tk[5], ek[5];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = lane + 64 * k;
    gwterm[k] = 0.f; Tk[k] = 0.f; ek[k] = 0.f;
    if (i < S) {
      const float T = expf(-aux[i]);
      const float e = expf(-dd[i]);
      const float wraw = (1.f - e) * T;
      const float g = gw[i];
      const bool fin = (wraw == wraw) && fabsf(wraw) != INFINITY;
      Tk[k] = T; ek[k] = e;
      gwterm[k] = fin ? g * wraw : 0.f;
      if (!fin) { Tk[k] = 0.f; ek[k] = 0.f; bad = bad || live; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = lane + 64 * k;
    if (i < S) aux[i] = gwterm[k];
  }
  __syncthreads();
  wave_scan_f64<true, true>(aux, aux, S, lane);  // suffix sums: aux[i] = sum_{j > i}
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = lane + 64 * k;
    if (i < S && live) {
      const float g = gw[i];
      const float e0 = eb[i], e1 = eb[i + 1];
      const float gdd = g * Tk[k] * ek[k] - aux[i];
      float out = gdd * (e1 - e0);
      if (!(fabsf(out) <= 3.402823466e+38f)) { out = 0.f; bad = true; }
      a.g_density[(int64_t)r * S + i] = out;
    }
  }
  if (bad && a.nonfinite_flag) *a.nonfinite_flag = 1;
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_depth_loss(const float* weights, const float* ebins, const float* termination_depth, const float* directions_norm, float sigma,
                                int32_t R, int32_t S, float grad_scale, float* loss_rays, float* g_weights, int32_t accumulate, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1 && S <= MAXS && sigma > 0.f, "depth_loss: R=%d S=%d sigma=%g", R, S, (double)sigma);
  if (R == 0) return 0;
  SNERF_REQUIRE(weights && ebins && termination_depth, "depth_loss: null buffer");
  hipLaunchKernelGGL(depth_loss_kernel, dim3(ceil_div(R, RPB)), dim3(256), 0, (hipStream_t)stream, weights, ebins, termination_depth, directions_norm, sigma,
                     R, S, grad_scale, loss_rays, g_weights, accumulate);
  SNERF_LAUNCH_CHECK("depth_loss");
  return 0;
}

extern "C" int snerf_urf_depth_loss(const float* weights, const float* ebins, const float* termination_depth, const float* directions_norm,
                                    const float* predicted_depth, float sigma, int32_t R, int32_t S, float grad_scale, float* loss_rays, float* g_weights,
                                    float* g_predicted_depth, int32_t accumulate, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1 && S <= MAXS && sigma > 0.f, "urf_depth_loss: R=%d S=%d sigma=%g", R, S, (double)sigma);
  if (R == 0) return 0;
  SNERF_REQUIRE(weights && ebins && termination_depth && predicted_depth, "urf_depth_loss: null buffer");
  hipLaunchKernelGGL(urf_depth_loss_kernel, dim3(ceil_div(R, RPB)), dim3(256), 0, (hipStream_t)stream, weights, ebins, termination_depth, directions_norm,
                     predicted_depth, sigma, R, S, grad_scale, loss_rays, g_weights, g_predicted_depth, accumulate);
  SNERF_LAUNCH_CHECK("urf_depth_loss");
  return 0;
}

extern "C" int snerf_render_fwd(const snerf_render_args* p, snerf_stream_t stream) {
  SNERF_REQUIRE(p, "render_fwd: null args");
  SNERF_REQUIRE(p->R >= 0 && p->S >= 1 && p->S <= MAXS, "render_fwd: R=%d S=%d (S <= %d)", p->R, p->S, MAXS);
  SNERF_REQUIRE(p->bg_mode >= 0 && p->bg_mode <= 2, "render_fwd: bg_mode=%d", p->bg_mode);
  if (p->R == 0) return 0;
  SNERF_REQUIRE(p->weights && p->rgb && p->ebins && p->rgb_out && p->acc_out, "render_fwd: null buffer");
  SNERF_REQUIRE(p->bg_mode == 1 || p->bg, "render_fwd: background buffer is null");
  RenderArgs a;
  a.weights = p->weights; a.rgb = p->rgb; a.ebins = p->ebins; a.bg = p->bg; a.R = p->R; a.S = p->S; a.bg_mode = p->bg_mode;
  a.training = p->training; a.rgb_out = p->rgb_out; a.acc_out = p->acc_out; a.depth_median = p->depth_median;
  a.depth_expected = p->depth_expected; a.median_rgb = p->median_rgb; a.median_index = p->median_index;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(ceil_div(p->R, RPB)), dim3(256), 0, (hipStream_t)stream, a);
  SNERF_LAUNCH_CHECK("render_fwd");
  return 0;
}

extern "C" int snerf_render_bwd(const float* weights, const float* rgb, const float* bg, int32_t bg_mode, const float* g_rgb_out,
                                const float* g_acc, int32_t R, int32_t S, float* g_weights, float* g_rgb, int32_t accumulate_w,
                                snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1, "render_bwd: R=%d S=%d", R, S);
  SNERF_REQUIRE(bg_mode == 0 || bg_mode == 2, "render_bwd: bg_mode=%d has no training backward", bg_mode);
  if (R == 0) return 0;
  SNERF_REQUIRE(weights && rgb && bg && g_rgb_out && g_weights, "render_bwd: null buffer");
  hipLaunchKernelGGL(render_bwd_kernel, dim3(ceil_div((int64_t)R * S, 256)), dim3(256), 0, (hipStream_t)stream, weights, rgb, bg, bg_mode,
                     g_rgb_out, g_acc, R, S, g_weights, g_rgb, accumulate_w);
  SNERF_LAUNCH_CHECK("render_bwd");
  return 0;
}

extern "C" int snerf_render_mse_bwd(const float* weights, const float* rgb, const float* bg, int32_t bg_mode, const float* rgb_out,
                                    const float* target, float go_scale, int32_t R, int32_t S, float* g_weights, float* g_rgb, float* sqerr_rays,
                                    snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1, "render_mse_bwd: R=%d S=%d", R, S);
  SNERF_REQUIRE(bg_mode == 0 || bg_mode == 2, "render_mse_bwd: bg_mode %d has no training backward", bg_mode);
  if (R == 0) return 0;
  SNERF_REQUIRE(weights && rgb && bg && rgb_out && target && g_weights, "render_mse_bwd: null buffer");
  hipLaunchKernelGGL(render_mse_bwd_kernel, dim3(ceil_div((int64_t)R * S, 256)), dim3(256), 0, (hipStream_t)stream, weights, rgb, bg, bg_mode,
                     rgb_out, target, go_scale, R, S, g_weights, g_rgb, sqerr_rays);
  SNERF_LAUNCH_CHECK("render_mse_bwd");
  return 0;
}

extern "C" int snerf_distortion(const float* weights, const float* sbins, int32_t R, int32_t S, float grad_scale, float* loss_rays,
                                float* g_weights, int32_t accumulate, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1 && S <= MAXS, "distortion: R=%d S=%d (S <= %d)", R, S, MAXS);
  if (R == 0) return 0;
  SNERF_REQUIRE(weights && sbins, "distortion: null buffer");
  hipLaunchKernelGGL(distortion_kernel, dim3(ceil_div(R, RPB)), dim3(256), 0, (hipStream_t)stream, weights, sbins, R, S, grad_scale, loss_rays,
                     g_weights, accumulate);
  SNERF_LAUNCH_CHECK("distortion");
  return 0;
}

extern "C" int snerf_interlevel(const float* c_bins, const float* w_nerf, int32_t S, const float* p_bins, const float* w_prop, int32_t Sp,
                                int32_t R, float grad_scale, float* loss_rays, float* g_wprop, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && S >= 1 && S <= MAXS && Sp >= 1 && Sp <= MAXS, "interlevel: R=%d S=%d Sp=%d", R, S, Sp);
  if (R == 0) return 0;
  SNERF_REQUIRE(c_bins && w_nerf && p_bins && w_prop, "interlevel: null buffer");
  hipLaunchKernelGGL(interlevel_kernel, dim3(ceil_div(R, RPB)), dim3(256), 0, (hipStream_t)stream, c_bins, w_nerf, S, p_bins, w_prop, Sp, R,
                     grad_scale, loss_rays, g_wprop);
  SNERF_LAUNCH_CHECK("interlevel");
  return 0;
}

extern "C" int snerf_ray_train_fwd_bwd(const snerf_ray_train_args* p, snerf_stream_t stream) {
  SNERF_REQUIRE(p, "ray_train_fwd_bwd: null args");
  SNERF_REQUIRE(p->R >= 0 && p->S >= 1 && p->S <= MAXS, "ray_train_fwd_bwd: R=%d S=%d (S <= %d)", p->R, p->S, MAXS);
  SNERF_REQUIRE(p->bg_mode == 0 || p->bg_mode == 2, "ray_train_fwd_bwd: bg_mode %d has no training backward", p->bg_mode);
  if (p->R == 0) return 0;
  SNERF_REQUIRE(p->density && p->ebins && p->sbins && p->rgb && p->bg && p->target && p->weights && p->rgb_out && p->acc_out && p->g_rgb && p->g_density,
                "ray_train_fwd_bwd: null buffer");
  RayTrainArgs a = {};
  a.density = p->density; a.ebins = p->ebins; a.sbins = p->sbins; a.rgb = p->rgb; a.bg = p->bg; a.target = p->target;
  a.R = p->R; a.S = p->S; a.bg_mode = p->bg_mode; a.go_scale = p->go_scale; a.dist_scale = p->dist_scale;
  a.weights = p->weights; a.rgb_out = p->rgb_out; a.acc_out = p->acc_out; a.depth_median = p->depth_median; a.sqerr_rays = p->sqerr_rays;
  a.dist_rays = p->dist_rays; a.g_rgb = p->g_rgb; a.g_density = p->g_density; a.g_weights = p->g_weights; a.nonfinite_flag = p->nonfinite_flag;
  hipLaunchKernelGGL(ray_train_kernel, dim3(ceil_div(p->R, RPB)), dim3(256), 0, (hipStream_t)stream, a);
  SNERF_LAUNCH_CHECK("ray_train_fwd_bwd");
  return 0;
}

// Dense per-step sweeps: K-Planes plane regularisers (value + gradient) and fused Adam.
//
// Reference: NS/model_components/losses.py compute_plane_tv :356-366, compute_plane_smoothness :369-380,
// space_tv_loss :383-406, time_smoothness_loss :409-428, sparse_transients_loss :431-452 (autograd supplies
// the gradients there: ~10 elementwise kernels per plane, forward and backward, every step) and
// torch.optim.Adam as configured at NS/configs/method_configs.py:546-557 (lr 1e-2, eps 1e-12).
// Both are HBM-bound streaming passes over every parameter: float4 per lane, channel-last planes so the
// +-1 row/column neighbours are other lanes' lines (L2 hits).
#include "plane_adam_common.hpp"

namespace snerf {

__global__ void adam_prepare_kernel(snerf_adam_dyn* dyn, float lr, float b1, float b2, int policy, int force_nonfinite) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int nf = dyn->nonfinite | force_nonfinite;
  const int skip = (policy == 1 && nf) ? 1 : 0;
  dyn->nonfinite = 0;
  dyn->skip = skip;
  if (skip) { dyn->skipped += 1; return; }
  const int t = dyn->t + 1;
  dyn->t = t;
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
  dyn->step_size = (float)((double)lr / bc1);
  dyn->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

template <int C, bool ADAM>
__global__ __launch_bounds__(256) void plane_reg_kernel(RegArgs a) {
  // locate this workgroup's plane
  int s = 0, p = 0;
  const int blk = (int)blockIdx.x + a.blk_base;
  {
    const int b = blk;
    bool found = false;
    for (int ss = a.d.n_scales - 1; ss >= 0 && !found; --ss)
      for (int pp = a.n_planes - 1; pp >= 0; --pp)
        if (b >= a.blk_off[ss][pp]) { s = ss; p = pp; found = true; break; }
  }
  int ax, bx;
  plane_axes(a.n_planes, p, ax, bx);
  const int W = a.d.res[s][ax], H = a.d.res[s][bx];
  const bool time_plane = (a.n_planes == 6) && (bx == 3);  // planes 2,4,5: H = time
  constexpr int C4 = C / 4;
  // 32-bit index math only (64-bit div/mod is a ~100-instruction software routine on the GPU); a plane has < 2^31 float4s
  const uint32_t n4 = (uint32_t)H * (uint32_t)W * C4;
  const uint32_t e = (uint32_t)(blk - a.blk_off[s][p]) * 256u + threadIdx.x;
  float l_tv = 0.f, l_sm = 0.f, l_l1 = 0.f;
  const int64_t o_lane = a.d.off[s][p] + (int64_t)e * 4;  // this lane's float4 inside the segment
  if (e < n4 && o_lane >= a.range_lo && o_lane < a.range_hi) {
    const int c4 = (int)(e % C4);       // C4 is a compile-time power of two
    const uint32_t hw = e / C4;
    const int h = (int)(hw / (uint32_t)W);
    const int w = (int)(hw - (uint32_t)h * (uint32_t)W);
    const float* base = a.planes + a.d.off[s][p] + c4 * 4;
    auto at = [&](int hh, int ww) { return ld4(base + ((int64_t)hh * W + ww) * C); };
    const float4 t = at(h, w);
    const float4 g = plane_reg_grad<C>(at, t, h, w, H, W, time_plane, a.c_tv, a.c_smooth, a.c_l1, l_tv, l_sm, l_l1);
    if (ADAM) {
      const int64_t o = a.d.off[s][p] + ((int64_t)h * W + w) * C + c4 * 4;
      // g, m, v are touched exactly once per step: stream them past the caches (nontemporal) so that L2 keeps the parameter lines
      // the neighbouring lanes re-read for the regulariser stencil
      const DynConsts dc = load_dyn(a.dyn, a.step_size, a.inv_sqrt_bc2);
      if (dc.skip) {  // skipped step (non-finite gradient somewhere in this parameter group): p, m, v unchanged, gradient cleared
        stnt4(a.p_out + o, t);
        if (a.zero_grad) stnt4(a.grad + o, make_float4(0.f, 0.f, 0.f, 0.f));
      } else {
      float4 mm = ldnt4(a.m + o), vv = ldnt4(a.v + o), pp = t;
      const float4 gz = ldnt4(a.grad + o);
      const int ndrop = adam_float4(pp, mm, vv, gz, g, a.grad_scale, a.b1, a.b2, a.eps, dc);
      if (ndrop && a.dyn) atomicAdd(&a.dyn->dropped, ndrop);
      stnt4(a.p_out + o, pp);
      stnt4(a.m + o, mm);
      stnt4(a.v + o, vv);
      // clear only what is not zero already: at the finest scale about half of the texels receive no sample in a step (gg was read before
      // the loop changed it: compare the loaded value)
      if (a.zero_grad && (gz.x != 0.f || gz.y != 0.f || gz.z != 0.f || gz.w != 0.f)) stnt4(a.grad + o, make_float4(0.f, 0.f, 0.f, 0.f));
      }
    } else if (a.grad) {
      float* gp = a.grad + a.d.off[s][p] + ((int64_t)h * W + w) * C + c4 * 4;
      if (a.overwrite) {
        *reinterpret_cast<float4*>(gp) = g;  // caller guarantees the gradient buffer is zero here: skip the read
      } else {
        float4 old = ld4(gp);
        *reinterpret_cast<float4*>(gp) = add4(old, g);
      }
    }
  }
  // ---- loss values: workgroup reduction, one atomic per workgroup per term ----
  __shared__ float red[3][4];
  l_tv = wave_sum(l_tv); l_sm = wave_sum(l_sm); l_l1 = wave_sum(l_l1);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[0][wv] = l_tv; red[1][wv] = l_sm; red[2][wv] = l_l1; }
  __syncthreads();
  // one add per workgroup per term, spread over n_slots 64-B lines: 150k workgroups adding to ONE address serialise at
  // the memory side (measured: 1.9 ms for this kernel with a single slot, profiles/r01_kernels.md)
  if (threadIdx.x < 3 && a.losses) {
    float v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (v != 0.f) atomicAdd(a.losses + (size_t)(blk % a.n_slots) * 16 + threadIdx.x, v);
  }
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, no weight decay / amsgrad):  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps),  step_size = lr / bc1
// Optionally zeroes g afterwards (saves a separate memset sweep) and scales g first (gradient mean over ranks).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(const float* p, float* p_out, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  int64_t n, float step_size_h, float b1, float b2, float inv_sqrt_bc2_h, float eps,
                                                  float grad_scale, int zero_grad, snerf_adam_dyn* dyn) {
  // one float4 per lane, every buffer streamed past the caches once (nontemporal): as plane_reg_kernel's Adam path, which reaches
  // 6.1 TB/s against 4.6 TB/s for a 4096-workgroup grid-stride loop with cached accesses
  const int64_t n4 = n / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const DynConsts dc = load_dyn(dyn, step_size_h, inv_sqrt_bc2_h);
  const float step_size = dc.step_size, inv_sqrt_bc2 = dc.inv_sqrt_bc2;
  if (dc.skip) {  // skipped step: p, m, v unchanged, gradient cleared
    if (i < n4) {
      if (p_out != p) stnt4(p_out + i * 4, ldnt4(p + i * 4));
      if (zero_grad) stnt4(g + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
      const int64_t j = n4 * 4 + threadIdx.x;
      p_out[j] = p[j];
      if (zero_grad) g[j] = 0.f;
    }
    return;
  }
  if (i < n4) {
    float4 pp = ldnt4(p + i * 4), gg = ldnt4(g + i * 4), mm = ldnt4(m + i * 4), vv = ldnt4(v + i * 4);
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
    int ndrop = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = G[k] * grad_scale;
      if (!(fabsf(gk) <= 3.402823466e+38f)) { gk = 0.f; ++ndrop; }  // a non-finite gradient element is dropped (and counted), never written into m / v / p
      M[k] = b1 * M[k] + (1.f - b1) * gk;
      V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
      float denom = sqrtf(V[k]) * inv_sqrt_bc2 + eps;
      P[k] = P[k] - step_size * (M[k] / denom);
    }
    if (ndrop && dyn) atomicAdd(&dyn->dropped, ndrop);
    stnt4(p_out + i * 4, pp);
    stnt4(m + i * 4, mm);
    stnt4(v + i * 4, vv);
    // clear only what is not zero already (hash / temporal tables: most of a row's columns receive nothing in a step)
    if (zero_grad && (gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f)) stnt4(g + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  // tail (n % 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = n4 * 4 + threadIdx.x;
    float gk = g[i] * grad_scale;
    if (!(fabsf(gk) <= 3.402823466e+38f)) { gk = 0.f; if (dyn) atomicAdd(&dyn->dropped, 1); }
    float mk = b1 * m[i] + (1.f - b1) * gk;
    float vk = b2 * v[i] + (1.f - b2) * gk * gk;
    m[i] = mk; v[i] = vk;
    p_out[i] = p[i] - step_size * (mk / (sqrtf(vk) * inv_sqrt_bc2 + eps));
    if (zero_grad) g[i] = 0.f;
  }
}

// Adam over a temporal-grid table [rows][grid_C] with the temporal-TV gradient of two of its columns folded in
// (TemporalGridEncoder.get_temporal_tv_loss, temporal_grid.py:352-376): srow[r] = weight * sign(E[r,a] - E[r,b]) / rows was
// written by tgrid_tv_sign_kernel from the OLD table, so the in-place update has no read-after-write hazard and the dense
// gradient buffer is never read-modified-written for the TV term.
__global__ __launch_bounds__(256) void adam_tv_kernel(float* p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float step_size_h,
                                                     float b1, float b2, float inv_sqrt_bc2_h, float eps, float grad_scale, int zero_grad, int grid_C, int col_a,
                                                     int col_b, const float* __restrict__ srow, snerf_adam_dyn* dyn) {
  const int64_t n4 = n / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;  // n is a multiple of 4 here (rows * grid_C with rows a multiple of 8)
  const DynConsts dc = load_dyn(dyn, step_size_h, inv_sqrt_bc2_h);
  const float step_size = dc.step_size, inv_sqrt_bc2 = dc.inv_sqrt_bc2;
  if (dc.skip) {
    if (zero_grad) stnt4(g + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  float4 pp = ldnt4(p + i * 4), gg = ldnt4(g + i * 4), mm = ldnt4(m + i * 4), vv = ldnt4(v + i * 4);
  float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
  const int64_t e0 = i * 4;
  const int64_t row0 = e0 / grid_C;
  const int c0 = (int)(e0 - row0 * grid_C);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int col = c0 + k;
    int64_t row = row0;
    if (col >= grid_C) { col -= grid_C; row += 1; }
    float gk = G[k] * grad_scale;
    if (col == col_a) gk += srow[row];
    else if (col == col_b) gk -= srow[row];
    if (!(fabsf(gk) <= 3.402823466e+38f)) { gk = 0.f; if (dyn) atomicAdd(&dyn->dropped, 1); }
    M[k] = b1 * M[k] + (1.f - b1) * gk;
    V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
    P[k] = P[k] - step_size * (M[k] / (sqrtf(V[k]) * inv_sqrt_bc2 + eps));
  }
  stnt4(p + i * 4, pp);
  stnt4(m + i * 4, mm);
  stnt4(v + i * 4, vv);
  if (zero_grad && (gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f)) stnt4(g + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_plane_reg(const snerf_kplanes_desc* desc, const float* planes, float* grad, float c_space_tv, float c_time_smooth,
                               float c_sparse, float* losses, int32_t n_slots, int32_t overwrite, snerf_stream_t stream) {
  SNERF_REQUIRE(desc && planes, "plane_reg: null argument");
  SNERF_REQUIRE(desc->n_scales >= 1 && desc->n_scales <= SNERF_MAX_SCALES, "plane_reg: n_scales=%d", desc->n_scales);
  SNERF_REQUIRE(desc->C == 8 || desc->C == 16 || desc->C == 32, "plane_reg: C=%d unsupported", desc->C);
  SNERF_REQUIRE(desc->n_coords == 3 || desc->n_coords == 4, "plane_reg: n_coords=%d", desc->n_coords);
  RegArgs a = {};
  a.d = *desc;
  a.n_planes = desc->n_coords == 4 ? 6 : 3;
  a.planes = planes; a.grad = grad; a.c_tv = c_space_tv; a.c_smooth = c_time_smooth; a.c_l1 = c_sparse; a.losses = losses;
  a.overwrite = overwrite;
  a.n_slots = n_slots;
  SNERF_REQUIRE(!losses || n_slots >= 1, "plane_reg: n_slots=%d", n_slots);
  static const int PA6[6] = {0, 0, 0, 1, 1, 2}, PB6[6] = {1, 2, 3, 2, 3, 3}, PA3[3] = {0, 0, 1}, PB3[3] = {1, 2, 2};
  int64_t blocks = 0;
  for (int s = 0; s < desc->n_scales; ++s)
    for (int p = 0; p < a.n_planes; ++p) {
      const int ax = a.n_planes == 6 ? PA6[p] : PA3[p], bx = a.n_planes == 6 ? PB6[p] : PB3[p];
      int64_t n4 = (int64_t)desc->res[s][ax] * desc->res[s][bx] * (desc->C / 4);
      a.blk_off[s][p] = (int)blocks;
      blocks += (n4 + 255) / 256;
    }
  SNERF_REQUIRE(blocks < (1LL << 31), "plane_reg: too many workgroups");
  a.blk_base = 0; a.range_lo = 0; a.range_hi = INT64_MAX;
  hipStream_t st = (hipStream_t)stream;
  if (desc->C == 32) hipLaunchKernelGGL((plane_reg_kernel<32, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else if (desc->C == 16) hipLaunchKernelGGL((plane_reg_kernel<16, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((plane_reg_kernel<8, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  SNERF_LAUNCH_CHECK("plane_reg");
  return 0;
}

void snerf::adam_consts(float lr, float beta1, float beta2, int step, float& step_size, float& inv_sqrt_bc2) {
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  step_size = (float)((double)lr / bc1);
  inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

extern "C" int snerf_adam_planes_step_range(const snerf_kplanes_desc* desc, const float* p_in, float* p_out, float* g, float* m, float* v,
                                            float c_space_tv, float c_time_smooth, float c_sparse, float* losses, int32_t n_slots, float lr, float beta1,
                                            float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, int64_t range_lo,
                                            int64_t range_hi, snerf_adam_dyn* dyn, snerf_stream_t stream) {
  SNERF_REQUIRE(desc && p_in && p_out && g && m && v, "adam_planes_step: null argument");
  SNERF_REQUIRE(p_in != p_out, "adam_planes_step: parameters must ping-pong (p_in != p_out): the regulariser reads neighbours of the old values");
  SNERF_REQUIRE(desc->n_scales >= 1 && desc->n_scales <= SNERF_MAX_SCALES, "adam_planes_step: n_scales=%d", desc->n_scales);
  SNERF_REQUIRE(desc->C == 8 || desc->C == 16 || desc->C == 32, "adam_planes_step: C=%d unsupported", desc->C);
  SNERF_REQUIRE(desc->n_coords == 3 || desc->n_coords == 4, "adam_planes_step: n_coords=%d", desc->n_coords);
  SNERF_REQUIRE((step >= 1 || dyn) && (!losses || n_slots >= 1), "adam_planes_step: step=%d n_slots=%d", step, n_slots);
  SNERF_REQUIRE(range_lo >= 0 && (range_lo & 3) == 0 && (range_hi & 3) == 0, "adam_planes_step: range [%lld, %lld) must be float4-aligned",
                (long long)range_lo, (long long)range_hi);
  RegArgs a = {};
  a.d = *desc;
  a.n_planes = desc->n_coords == 4 ? 6 : 3;
  a.planes = p_in; a.grad = g; a.c_tv = c_space_tv; a.c_smooth = c_time_smooth; a.c_l1 = c_sparse; a.losses = losses; a.n_slots = n_slots;
  a.p_out = p_out; a.m = m; a.v = v; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.grad_scale = grad_scale; a.zero_grad = zero_grad;
  a.range_lo = range_lo; a.range_hi = range_hi; a.dyn = dyn;
  if (!dyn) adam_consts(lr, beta1, beta2, step, a.step_size, a.inv_sqrt_bc2);
  static const int PA6[6] = {0, 0, 0, 1, 1, 2}, PB6[6] = {1, 2, 3, 2, 3, 3}, PA3[3] = {0, 0, 1}, PB3[3] = {1, 2, 2};
  int64_t blocks = 0, b_lo = INT64_MAX, b_hi = -1;  // workgroups [b_lo, b_hi] hold every float4 of the range (memory order == grid order)
  for (int s = 0; s < desc->n_scales; ++s)
    for (int p = 0; p < a.n_planes; ++p) {
      const int ax = a.n_planes == 6 ? PA6[p] : PA3[p], bx = a.n_planes == 6 ? PB6[p] : PB3[p];
      const int64_t n4 = (int64_t)desc->res[s][ax] * desc->res[s][bx] * (desc->C / 4);
      a.blk_off[s][p] = (int)blocks;
      const int64_t lo = range_lo > desc->off[s][p] ? range_lo - desc->off[s][p] : 0;                 // floats, relative to the plane
      const int64_t hi = (range_hi < desc->off[s][p] + n4 * 4 ? range_hi : desc->off[s][p] + n4 * 4) - desc->off[s][p];
      if (hi > lo) {
        const int64_t first = blocks + lo / 1024, last = blocks + (hi - 1) / 1024;
        b_lo = first < b_lo ? first : b_lo;
        b_hi = last > b_hi ? last : b_hi;
      }
      blocks += (n4 + 255) / 256;
    }
  SNERF_REQUIRE(blocks < (1LL << 31), "adam_planes_step: too many workgroups");
  if (b_hi < b_lo) return 0;  // the range holds none of this set's parameters
  a.blk_base = (int)b_lo;
  const unsigned grid = (unsigned)(b_hi - b_lo + 1);
  hipStream_t st = (hipStream_t)stream;
  if (desc->C == 32) hipLaunchKernelGGL((plane_reg_kernel<32, true>), dim3(grid), dim3(256), 0, st, a);
  else if (desc->C == 16) hipLaunchKernelGGL((plane_reg_kernel<16, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((plane_reg_kernel<8, true>), dim3(grid), dim3(256), 0, st, a);
  SNERF_LAUNCH_CHECK("adam_planes_step");
  return 0;
}

extern "C" int snerf_adam_planes_step(const snerf_kplanes_desc* desc, const float* p_in, float* p_out, float* g, float* m, float* v,
                                      float c_space_tv, float c_time_smooth, float c_sparse, float* losses, int32_t n_slots, float lr, float beta1,
                                      float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, snerf_adam_dyn* dyn,
                                      snerf_stream_t stream) {
  return snerf_adam_planes_step_range(desc, p_in, p_out, g, m, v, c_space_tv, c_time_smooth, c_sparse, losses, n_slots, lr, beta1, beta2, eps, step,
                                      grad_scale, zero_grad, 0, INT64_MAX & ~(int64_t)3, dyn, stream);
}

extern "C" int snerf_adam_step_tv(float* p, float* g, float* m, float* v, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, const float* srow,
                                  float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, snerf_adam_dyn* dyn,
                                  snerf_stream_t stream) {
  SNERF_REQUIRE(rows >= 1 && grid_C >= 4 && (step >= 1 || dyn), "adam_step_tv: rows=%lld grid_C=%d step=%d", (long long)rows, grid_C, step);
  SNERF_REQUIRE(col_a >= 0 && col_a < grid_C && col_b >= 0 && col_b < grid_C && col_a != col_b, "adam_step_tv: columns (%d, %d) of %d", col_a, col_b, grid_C);
  SNERF_REQUIRE(p && g && m && v && srow, "adam_step_tv: null buffer");
  const int64_t n = rows * grid_C;
  SNERF_REQUIRE((n & 3) == 0, "adam_step_tv: rows * grid_C = %lld must be a multiple of 4", (long long)n);
  SNERF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step_tv: buffers must be 16-byte aligned");
  float step_size = 0.f, inv_sqrt_bc2 = 0.f;
  if (!dyn) adam_consts(lr, beta1, beta2, step, step_size, inv_sqrt_bc2);
  const int64_t blocks = (n / 4 + 255) / 256;
  SNERF_REQUIRE(blocks < (1LL << 31), "adam_step_tv: table too large for one launch");
  hipLaunchKernelGGL(adam_tv_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_size, beta1, beta2, inv_sqrt_bc2, eps,
                     grad_scale, zero_grad, grid_C, col_a, col_b, srow, dyn);
  SNERF_LAUNCH_CHECK("adam_step_tv");
  return 0;
}

extern "C" int snerf_adam_step(const float* p, float* p_out, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               int32_t step, float grad_scale, int32_t zero_grad, snerf_adam_dyn* dyn, snerf_stream_t stream) {
  SNERF_REQUIRE(n >= 0 && (step >= 1 || dyn), "adam_step: n=%lld step=%d (1-based)", (long long)n, step);
  if (n == 0) return 0;
  SNERF_REQUIRE(p && p_out && g && m && v, "adam_step: null buffer");
  SNERF_REQUIRE((((uintptr_t)p | (uintptr_t)p_out | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  float step_size = 0.f, inv_sqrt_bc2 = 0.f;
  if (!dyn) adam_consts(lr, beta1, beta2, step, step_size, inv_sqrt_bc2);
  int64_t n4 = (n + 3) / 4;
  int64_t blocks = (n4 + 255) / 256;
  SNERF_REQUIRE(blocks < (1LL << 31), "adam_step: n=%lld too large for one launch", (long long)n);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, p_out, g, m, v, n, step_size, beta1, beta2, inv_sqrt_bc2,
                     eps, grad_scale, zero_grad, dyn);
  SNERF_LAUNCH_CHECK("adam_step");
  return 0;
}

extern "C" int snerf_adam_prepare(snerf_adam_dyn* dyn, float lr, float beta1, float beta2, int32_t policy, int32_t force_nonfinite,
                                  snerf_stream_t stream) {
  SNERF_REQUIRE(dyn, "adam_prepare: null state");
  SNERF_REQUIRE(policy == 0 || policy == 1, "adam_prepare: policy=%d (0 drop elements, 1 skip the step)", policy);
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dyn, lr, beta1, beta2, policy, force_nonfinite);
  SNERF_LAUNCH_CHECK("adam_prepare");
  return 0;
}

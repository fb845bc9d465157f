// Owner-computes plane-gradient scatter fused with the optimiser sweep, for the scale whose texel grid IS the sort grid (the finest).
//
// What it replaces: for 72 % of the k-planes preset's parameters (the finest of the five scales) the step used to run pass B of the sorted
// scatter (kplanes_sorted.hip: memory-side float atomics into a gradient plane in HBM) and then the optimiser sweep (optim.hip: that
// gradient plane read back and cleared, 8 of the sweep's 32 B / parameter).  Reference semantics are unchanged: the autograd of
// interpolate_kplanes' grid_sample calls (NS/fields/kplanes_field.py:77-126), the plane regularisers (NS/model_components/losses.py:356-452)
// and torch.optim.Adam (NS/configs/method_configs.py:546-557, NS/engine/optimizers.py:119-139).
//
// How: the counting sort (snerf_kplanes_sort_samples) keys every plane's samples by their CELL at the finest scale and leaves the scanned
// histogram behind -- a CSR row pointer: the entries of cell k are sorted_rec[scan[k] .. scan[k+1]).  A workgroup OWNS a TW x TH tile of
// texels of one plane.  It looks up the (TW+1) x (TH+1) cells whose bilinear footprint touches its texels, walks their entries (quotient
// form: g_q = G / v_q with v_q re-interpolated bit-identically to the forward from the parameter tile staged in LDS), sums into a
// gradient tile in LDS -- no atomics: cell rows of equal parity touch disjoint texel rows, so two barrier-separated half-passes are
// race-free -- and then applies regulariser gradient + Adam to its texels straight from LDS.  The data gradient of this scale never
// exists in HBM: 24 B / parameter instead of 32, no memory-side atomic requests, and (most tiles receive no sample at all in a step)
// an empty tile skips everything but the streaming update.  Cells on a tile border are walked by both neighbours (each keeps the
// corners it owns): (TW+1)(TH+1) / (TW TH) of the entries are touched, 1.2x for 16 x 8.
//
// Exact zeros of the quotient form (kplanes_sorted.hip: quotient_fixup_kernel) still go through the gradient plane in HBM: the fix-up
// runs first, and when its device-side count is non-zero this kernel adds (and clears) what it left there.
#include "kplanes_sort_common.hpp"
#include "plane_adam_common.hpp"

namespace snerf {

struct TileArgs {
  snerf_kplanes_desc d;
  int s;                 // the scale (its resolutions equal the sort grid)
  int tile_off[7];       // first workgroup of each plane; [n_planes] = grid size
  int tiles_x[6];
  int cell_off[7];       // first histogram cell of each plane; [n_planes] = number of cells
  int row_major[6];
  int total_entries;     // n_planes * N
  const float* G;        // [N][row_stride] = gfeat .* feat
  int row_stride;
  const float4* rec;     // sorted records {sample id, coord a, coord b, -}
  const int32_t* scan;   // scanned histogram of the sort
  const int32_t* fix_count;
  const float* p_in; float* p_out; float* m; float* v; float* grad;
  float c_tv, c_smooth, c_l1;
  float* losses; int n_slots;
  float step_size, b1, b2, inv_sqrt_bc2, eps, grad_scale;
  snerf_adam_dyn* dyn;
};

template <int TW, int TH>
struct TilePlan {
  static constexpr int C = 32, C4 = 8;
  static constexpr int PW = TW + 2, PH = TH + 2;    // parameter tile with a halo of one texel (clamped at the plane border)
  static constexpr int CW = TW + 1, CH = TH + 1;    // cells whose footprint touches the tile: x0 in [tx-1, tx+TW-1], y0 likewise
  static constexpr int GT = 0;                       // float offsets into dynamic LDS
  static constexpr int PT = GT + TH * TW * C;
  static constexpr int CS = PT + PH * PW * C;        // int32 [CH][CW] entry start of each cell
  static constexpr int CC = CS + CH * CW;            // int32 [CH][CW] entry count
  static constexpr int RC = (CC + CH * CW + 3) / 4 * 4;  // uint32 [4 waves][64][8] walk records
  static constexpr int TOTAL = RC + 4 * 64 * 8;
  static constexpr size_t BYTES = (size_t)TOTAL * 4;
  static_assert(CW <= 64, "a cell row's counts are scanned across one wavefront");
};

template <int NP, int TW, int TH>
__global__ __launch_bounds__(256) void tile_scatter_adam_kernel(TileArgs a) {
  using P = TilePlan<TW, TH>;
  constexpr int C = P::C, C4 = P::C4, PW = P::PW, PH = P::PH, CW = P::CW, CH = P::CH;
  extern __shared__ __align__(16) float lds[];
  float* gt = lds + P::GT;
  float* pt = lds + P::PT;
  int* cstart = reinterpret_cast<int*>(lds + P::CS);
  int* ccount = reinterpret_cast<int*>(lds + P::CC);
  __shared__ int s_total;
  __shared__ float red[3][4];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int half = lane >> 5, ch = lane & 31;
  // ---- this workgroup's plane and tile ----
  int q = 0;
#pragma unroll
  for (int k = 1; k < NP; ++k) q = (int)blockIdx.x >= a.tile_off[k] ? k : q;
  int ax, bx;
  plane_axes(NP, q, ax, bx);
  const int s = a.s;
  const int W = a.d.res[s][ax], H = a.d.res[s][bx] > 0 ? a.d.res[s][bx] : 1;
  const bool time_plane = NP == 6 && bx == 3;
  const int tl = (int)blockIdx.x - a.tile_off[q];
  const int tyi = tl / a.tiles_x[q], txi = tl - tyi * a.tiles_x[q];
  const int tx0 = txi * TW, ty0 = tyi * TH;
  const int64_t poff = a.d.off[s][q];
  const float* __restrict__ pin = a.p_in + poff;

  // ---- phase 0: the cells that touch this tile, from the sort's scanned histogram ----
  if (tid == 0) s_total = 0;
  __syncthreads();
  {
    int mine = 0;
    const int n_cells = a.cell_off[NP];
    for (int c = tid; c < CH * CW; c += 256) {
      const int cy = c / CW, cx = c - cy * CW;
      const int x0 = tx0 - 1 + cx, y0 = ty0 - 1 + cy;
      int st = 0, cnt = 0;
      if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) {
        const int key = a.cell_off[q] + (a.row_major[q] ? y0 * W + x0 : (int)morton2((uint32_t)x0, (uint32_t)y0));
        st = a.scan[key];
        cnt = (key + 1 < n_cells ? a.scan[key + 1] : a.total_entries) - st;
      }
      cstart[c] = st;
      ccount[c] = cnt;
      mine += cnt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0 && mine) atomicAdd(&s_total, mine);
  }
  __syncthreads();
  const int total = s_total;

  if (total > 0) {
    // ---- phase 1: parameter tile (+ halo, clamped: a clamped corner carries weight 0, as in the forward's taps) and a zeroed gradient tile ----
    for (int f = tid; f < PH * PW * C4; f += 256) {
      const int c4 = f % C4, t = f / C4;
      const int py = t / PW, px = t - py * PW;
      int gx = tx0 - 1 + px, gy = ty0 - 1 + py;
      gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
      gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
      *reinterpret_cast<float4*>(pt + t * C + c4 * 4) = ld4(pin + ((int64_t)gy * W + gx) * C + c4 * 4);
    }
    for (int f = tid; f < TH * TW * C4; f += 256) *reinterpret_cast<float4*>(gt + f * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- phase 2: walk the entries.  Cell row r (y0 = ty0 - 1 + r) adds into texel rows r - 1 and r of the tile: rows of equal parity are
    //      disjoint, so the waves take the even rows, meet at a barrier, then take the odd rows -- plain LDS read-add-write, no atomics ----
    uint32_t* R = reinterpret_cast<uint32_t*>(lds + P::RC) + wave * (64 * 8);
    const float* __restrict__ Gs = a.G + s * C + ch;
    for (int par = 0; par < 2; ++par) {
      for (int r = par + 2 * wave; r < CH; r += 8) {
        const int cnt = lane < CW ? ccount[r * CW + lane] : 0;
        const int cst = lane < CW ? cstart[r * CW + lane] : 0;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(incl, off, 64);
          if (lane >= off) incl += t;
        }
        const int row_total = __shfl(incl, 63, 64);
        if (row_total == 0) continue;
        const int excl = incl - cnt;
        int cur = -1;          // lx of the cell being accumulated
        float p0 = 0.f, p1 = 0.f;
        auto flush = [&]() {  // cell lx = cur: this lane's texel column is x0 + half = cur - 1 + half (tile-local), rows r - 1 and r
          const int x = cur - 1 + half;
          if (cur >= 0 && x >= 0 && x < TW) {
            if (r >= 1 && p0 != 0.f) gt[((r - 1) * TW + x) * C + ch] += p0;
            if (r < TH && p1 != 0.f) gt[(r * TW + x) * C + ch] += p1;
          }
        };
        for (int base = 0; base < row_total; base += 64) {
          // per-entry preparation, lane-parallel (one entry per lane): which cell, its record, taps, LDS offsets, weights
          const int e = base + lane;
          int cell = 0;
#pragma unroll
          for (int j = 0; j < CW; ++j) cell += e >= __builtin_amdgcn_readlane(incl, j) ? 1 : 0;
          const bool valid = e < row_total;
          cell = cell > CW - 1 ? CW - 1 : cell;
          const int idx = __shfl(cst, cell, 64) + (e - __shfl(excl, cell, 64));
          uint4 hd = make_uint4(0u, (uint32_t)cell, 0u, 0u);
          float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid) {
            const float4 rc = a.rec[idx];
            const AxisTap tx = axis_tap(rc.y, W);
            const AxisTap ty = axis_tap(rc.z, H);
            const int lx = tx.i0 - (tx0 - 1), ly = ty.i0 - (ty0 - 1);
            if (lx == cell && ly == r) {  // by construction of the sort key; anything else would index outside the staged tile
              hd.x = (uint32_t)__float_as_int(rc.x) * (uint32_t)a.row_stride;
              hd.z = (uint32_t)((ly * PW + lx) * C);
              hd.w = (tx.i1 != tx.i0 ? (uint32_t)C : 0u) | ((ty.i1 != ty.i0 ? (uint32_t)(PW * C) : 0u) << 16);  // offsets of the x1 column / y1 row
              const float4 tw = tap_weights(tx, ty);    // (x0y0, x1y0, x0y1, x1y1): the forward's products, bit for bit
              wt = make_float4(tw.x, tw.z, tw.y, tw.w);  // stored (x0y0, x0y1 | x1y0, x1y1): one float2 per x-corner
            }
          }
          *reinterpret_cast<uint4*>(R + lane * 8) = hd;
          *reinterpret_cast<float4*>(R + lane * 8 + 4) = wt;
          const int mcount = (row_total - base) < 64 ? (row_total - base) : 64;
          constexpr int UN = 8;
          for (int u0 = 0; u0 < mcount; u0 += UN) {
            float g[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
              const int uu = (u0 + u) < mcount ? (u0 + u) : (mcount - 1);
              g[u] = Gs[R[uu * 8]];  // wave-uniform LDS address: broadcast; 128 B of G per entry and half-wave
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
              const int uu = u0 + u;
              if (uu < mcount) {
                const uint32_t lxu = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[uu * 8 + 1]);
                const uint32_t toff = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[uu * 8 + 2]);
                const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[uu * 8 + 3]);
                const float4 w4 = *reinterpret_cast<const float4*>(R + uu * 8 + 4);
                const float* tp = pt + toff + ch;
                const uint32_t dx = fl & 0xffffu, dy = fl >> 16;
                // the plane's value at the entry with the forward's own formula and order (bilerp4): cancels v_q exactly in G / v_q
                const float vq = bilerp4(tp[0], tp[dx], tp[dy], tp[dy + dx], w4.x, w4.z, w4.y, w4.w);
                const float gq = fabsf(vq) >= QUOT_TINY ? g[u] * __builtin_amdgcn_rcpf(vq) : 0.f;  // zero / subnormal: the fix-up supplies the exact term
                const float wy0 = half ? w4.z : w4.x, wy1 = half ? w4.w : w4.y;
                const float v0 = gq * wy0, v1 = gq * wy1;
                if ((int)lxu == cur) {
                  p0 += v0; p1 += v1;
                } else {
                  flush();
                  cur = (int)lxu; p0 = v0; p1 = v1;
                }
              }
            }
          }
        }
        flush();
      }
      __syncthreads();
    }
  }

  // ---- phase 3: regulariser gradient + Adam on the tile's texels (plane_reg_kernel's arithmetic; the data gradient comes from LDS) ----
  const DynConsts dc = load_dyn(a.dyn, a.step_size, a.inv_sqrt_bc2);
  const bool has_fix = a.grad != nullptr && a.fix_count != nullptr && *a.fix_count > 0;  // exact-zero rows left their terms in the gradient plane
  float l_tv = 0.f, l_sm = 0.f, l_l1 = 0.f;
  float* __restrict__ pout = a.p_out + poff;
  float* __restrict__ pm = a.m + poff;
  float* __restrict__ pv = a.v + poff;
  int ndrop = 0;
#pragma unroll 2
  for (int f = tid; f < TH * TW * C4; f += 256) {
    const int c4 = f % C4, t = f / C4;
    const int ly = t / TW, lx = t - ly * TW;
    const int h = ty0 + ly, w = tx0 + lx;
    if (h < H && w < W) {
      const float* base = pin + c4 * 4;
      auto at = [&](int hh, int ww) { return ld4(base + ((int64_t)hh * W + ww) * C); };
      const float4 tt = at(h, w);
      const float4 greg = plane_reg_grad<C>(at, tt, h, w, H, W, time_plane, a.c_tv, a.c_smooth, a.c_l1, l_tv, l_sm, l_l1);
      const int64_t o = ((int64_t)h * W + w) * C + c4 * 4;
      float4 gd = total > 0 ? *reinterpret_cast<const float4*>(gt + f * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_fix) {
        float* gp = a.grad + poff + o;
        const float4 gf = ld4(gp);
        if (gf.x != 0.f || gf.y != 0.f || gf.z != 0.f || gf.w != 0.f) {
          gd = add4(gd, gf);
          *reinterpret_cast<float4*>(gp) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      if (dc.skip) {  // skipped step (non-finite gradient somewhere in this parameter group): p, m, v unchanged
        stnt4(pout + o, tt);
      } else {
        float4 mm = ldnt4(pm + o), vv = ldnt4(pv + o), pp = tt;
        ndrop += adam_float4(pp, mm, vv, gd, greg, a.grad_scale, a.b1, a.b2, a.eps, dc);
        stnt4(pout + o, pp);
        stnt4(pm + o, mm);
        stnt4(pv + o, vv);
      }
    }
  }
  if (ndrop && a.dyn) atomicAdd(&a.dyn->dropped, ndrop);
  // ---- loss values: workgroup reduction, one atomic per workgroup per term ----
  l_tv = wave_sum(l_tv); l_sm = wave_sum(l_sm); l_l1 = wave_sum(l_l1);
  if (lane == 0) { red[0][wave] = l_tv; red[1][wave] = l_sm; red[2][wave] = l_l1; }
  __syncthreads();
  if (tid < 3 && a.losses) {
    const float vsum = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    if (vsum != 0.f) atomicAdd(a.losses + (size_t)(blockIdx.x % a.n_slots) * 16 + tid, vsum);
  }
}

template <int NP, int TW, int TH>
static int launch_tile(TileArgs& a, hipStream_t st) {
  using P = TilePlan<TW, TH>;
  int64_t blocks = 0;
  for (int q = 0; q < NP; ++q) {
    int ax, bx;
    plane_axes(NP, q, ax, bx);
    const int W = a.d.res[a.s][ax], H = a.d.res[a.s][bx] > 0 ? a.d.res[a.s][bx] : 1;
    a.tile_off[q] = (int)blocks;
    a.tiles_x[q] = (W + TW - 1) / TW;
    blocks += (int64_t)a.tiles_x[q] * ((H + TH - 1) / TH);
  }
  a.tile_off[NP] = (int)blocks;
  SNERF_REQUIRE(blocks < (1LL << 31), "kplanes_scatter_adam: too many tiles");
  auto k = tile_scatter_adam_kernel<NP, TW, TH>;
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::BYTES), "kplanes_scatter_adam LDS size");
    if (rc) return rc;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), P::BYTES, st, a);
  SNERF_LAUNCH_CHECK("kplanes_scatter_adam");
  return 0;
}

static int scale_is_sort_grid(const snerf_kplanes_desc* d, int s, const SegTable& st) {
  for (int k = 0; k < d->n_coords; ++k) {
    const int r = d->res[s][k] > 0 ? d->res[s][k] : 1;
    if (r != st.fine[k] || r != st.fine_rm[k]) return 0;
  }
  return 1;
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_kplanes_scatter_adam_supported(const snerf_kplanes_desc* desc, int32_t scale, int64_t N) {
  if (!desc || desc->C != 32 || desc->concat != 1 || (desc->n_coords != 3 && desc->n_coords != 4) || scale < 0 || scale >= desc->n_scales || N < 0) return 0;
  if ((int64_t)N * desc->C * desc->n_scales >= (1LL << 31) || N * (desc->n_coords == 4 ? 6 : 3) >= (1LL << 31)) return 0;
  SegTable st;
  if (build_segs(desc, st) != 0 || st.per_scale) return 0;
  return scale_is_sort_grid(desc, scale, st);
}

extern "C" int snerf_kplanes_scatter_adam_scale(const snerf_kplanes_desc* desc, int32_t scale, int64_t N, const float* G, const float* sorted_rec,
                                                const int32_t* sort_hist, const int32_t* fix_count, const float* p_in, float* p_out, float* g, float* m,
                                                float* v, float c_space_tv, float c_time_smooth, float c_sparse, float* losses, int32_t n_slots, float lr,
                                                float beta1, float beta2, float eps, int32_t step, float grad_scale, snerf_adam_dyn* dyn,
                                                int32_t tile_shape, snerf_stream_t stream) {
  SNERF_REQUIRE(desc && p_in && p_out && m && v, "kplanes_scatter_adam: null argument");
  SNERF_REQUIRE(p_in != p_out, "kplanes_scatter_adam: parameters must ping-pong (p_in != p_out): the regulariser reads neighbours of the old values");
  SNERF_REQUIRE(snerf_kplanes_scatter_adam_supported(desc, scale, N) == 1,
                "kplanes_scatter_adam: built for C = 32, concatenated scales and the scale whose resolutions are the sort grid (the finest); scale=%d C=%d", scale,
                desc->C);
  SNERF_REQUIRE((step >= 1 || dyn) && (!losses || n_slots >= 1), "kplanes_scatter_adam: step=%d n_slots=%d", step, n_slots);
  SNERF_REQUIRE(N == 0 || (G && sorted_rec && sort_hist), "kplanes_scatter_adam: null scatter input");
  SNERF_REQUIRE(!fix_count || g, "kplanes_scatter_adam: fix_count needs the gradient plane the fix-up wrote into");
  SegTable stb;
  int rc = build_segs(desc, stb);
  if (rc) return rc;
  const int NP = desc->n_coords == 4 ? 6 : 3;
  TileArgs a = {};
  a.d = *desc; a.s = scale;
  for (int q = 0; q <= NP; ++q) a.cell_off[q] = stb.cell_off[q];
  for (int q = 0; q < NP; ++q) a.row_major[q] = stb.row_major[q];
  a.total_entries = (int)(N * NP);
  a.G = G; a.row_stride = desc->C * desc->n_scales; a.rec = reinterpret_cast<const float4*>(sorted_rec); a.scan = sort_hist; a.fix_count = fix_count;
  a.p_in = p_in; a.p_out = p_out; a.m = m; a.v = v; a.grad = g;
  a.c_tv = c_space_tv; a.c_smooth = c_time_smooth; a.c_l1 = c_sparse; a.losses = losses; a.n_slots = n_slots;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.grad_scale = grad_scale; a.dyn = dyn;
  if (!dyn) adam_consts(lr, beta1, beta2, step, a.step_size, a.inv_sqrt_bc2);
  hipStream_t st = (hipStream_t)stream;
  // tile_shape: 0 = default (16 x 8 texels), 1 = 32 x 8, 2 = 16 x 16, 3 = 16 x 4 (A-B)
  if (NP == 6) {
    if (tile_shape == 1) return launch_tile<6, 32, 8>(a, st);
    if (tile_shape == 2) return launch_tile<6, 16, 16>(a, st);
    if (tile_shape == 3) return launch_tile<6, 16, 4>(a, st);
    return launch_tile<6, 16, 8>(a, st);
  }
  return launch_tile<3, 16, 8>(a, st);
}

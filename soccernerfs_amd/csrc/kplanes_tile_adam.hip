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
// form: g_q = G / v_q with v_q re-interpolated bit-identically to the forward from the cell's four texels), sums each cell's run
// in registers and adds it into a gradient tile in LDS (LDS float atomics: no memory-side traffic) -- and then applies regulariser
// gradient + Adam to its texels straight from LDS.  The data gradient of this scale never
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
  int order[6];          // grid order of the planes: the ones with a time axis first
  int tile_off[7];       // first workgroup of the k-th plane IN GRID ORDER; [n_planes] = grid size
  int tiles_x[6];        // (grid order)
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
  int debug;             // timing bisection only (tile_shape >> 8): 1 = no walk at all, 2 = records only, 3 = walk without the LDS adds
};

template <int TW, int TH, int NT>
struct TilePlan {
  static constexpr int C = 32, C4 = 8;
  static constexpr int CW = TW + 1, CH = TH + 1;    // cells whose footprint touches the tile: x0 in [tx-1, tx+TW-1], y0 likewise
  static constexpr int NW = NT / 64;                 // wavefronts per workgroup
  static constexpr int NF = TH * TW * C4 / NT;       // float4s per thread in the optimiser phase
  static constexpr int GT = 0;                       // float offsets into dynamic LDS: gradient tile [TH][TW][C]
  static constexpr int CS = GT + TH * TW * C;        // int32 [CH][CW] entry start of each cell
  static constexpr int CC = CS + CH * CW;            // int32 [CH][CW] entry count
  static constexpr int CI = CC + CH * CW;            // int32 [CH][CW] inclusive prefix of the counts along the cell row
  static constexpr int RT = CI + CH * CW;            // int32 [CH] entries per cell row
  static constexpr int RC = (RT + CH + 3) / 4 * 4;   // uint32 [NW waves][64][8] walk records
  static constexpr int TOTAL = RC + NW * 64 * 8;
  static constexpr size_t BYTES = (size_t)TOTAL * 4;
  static_assert(CW <= 64 && CH <= 64, "a cell row's counts / the rows' totals are scanned across one wavefront");
  static_assert(TH * TW * C4 % NT == 0 && NT % 64 == 0 && NW <= 8, "the optimiser phase gives every thread the same number of float4s");
};

// MINW: waves per SIMD the register allocation must leave room for; PRELOAD: the optimiser's p / m / v loads are issued at kernel start (their
// latency under the walk, 12 registers per float4) instead of after it (relying on occupancy, as plane_reg_kernel does); UN: entries per chunk.
template <int NP, int TW, int TH, int NT, int MINW, bool PRELOAD, int UN>
__global__ __launch_bounds__(NT, MINW) void tile_scatter_adam_kernel(TileArgs a) {
  using P = TilePlan<TW, TH, NT>;
  constexpr int C = P::C, C4 = P::C4, CW = P::CW, CH = P::CH, NF = P::NF, NW = P::NW;
  extern __shared__ __align__(16) float lds[];
  float* gt = lds + P::GT;
  int* cstart = reinterpret_cast<int*>(lds + P::CS);
  int* ccount = reinterpret_cast<int*>(lds + P::CC);
  int* cincl = reinterpret_cast<int*>(lds + P::CI);
  int* rowtot = reinterpret_cast<int*>(lds + P::RT);
  __shared__ float red[3][8];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int half = lane >> 5, ch = lane & 31;
  // ---- this workgroup's plane and tile.  Grid order = a.order: the planes with a time axis come first (they hold the fullest tiles -- every
  //      sample of a frame lands in two texel rows -- and a full tile started last would be the kernel's tail) ----
  int qi = 0;
#pragma unroll
  for (int k = 1; k < NP; ++k) qi = (int)blockIdx.x >= a.tile_off[k] ? k : qi;
  const int q = a.order[qi];
  int ax, bx;
  plane_axes(NP, q, ax, bx);
  const int s = a.s;
  const int W = a.d.res[s][ax], H = a.d.res[s][bx] > 0 ? a.d.res[s][bx] : 1;
  const bool time_plane = NP == 6 && bx == 3;
  const int tl = (int)blockIdx.x - a.tile_off[qi];
  const int tyi = tl / a.tiles_x[qi], txi = tl - tyi * a.tiles_x[qi];
  const int tx0 = txi * TW, ty0 = tyi * TH;
  const int64_t poff = a.d.off[s][q];
  const float* __restrict__ pin = a.p_in + poff;
  float* __restrict__ pout = a.p_out + poff;
  float* __restrict__ pm = a.m + poff;
  float* __restrict__ pv = a.v + poff;

  // ---- the optimiser's streams are issued FIRST: their HBM latency runs under the cell lookup and the walk.  A third of the tiles receive no
  //      sample at all in a step and most of the others a few dozen: for them this kernel is the plain sweep, minus the gradient plane ----
  float4 tt[NF], mm[NF], vv[NF];
  int64_t oo[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int f = tid + NT * k;
    const int c4 = f % C4, t = f / C4;
    const int ly = t / TW, lx = t - ly * TW;
    const int h = ty0 + ly, w = tx0 + lx;
    oo[k] = (h < H && w < W) ? ((int64_t)h * W + w) * C + c4 * 4 : (int64_t)-1;
    if (PRELOAD && oo[k] >= 0) {
      tt[k] = ld4(pin + oo[k]);
      mm[k] = ldnt4(pm + oo[k]);
      vv[k] = ldnt4(pv + oo[k]);
    }
    *reinterpret_cast<float4*>(gt + f * 4) = make_float4(0.f, 0.f, 0.f, 0.f);  // the gradient tile starts at zero
  }

  // ---- phase 0: the cells that touch this tile, from the sort's scanned histogram ----
  {
    const int n_cells = a.cell_off[NP];
    for (int c = tid; c < CH * CW; c += NT) {
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
    }
  }
  __syncthreads();
  for (int r = wave; r < CH; r += NW) {  // inclusive prefix of the counts along each cell row; the row's total
    const int cnt = lane < CW ? ccount[r * CW + lane] : 0;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane < CW) cincl[r * CW + lane] = incl;
    if (lane == 63) rowtot[r] = incl;
  }
  __syncthreads();
  // The tile's entries, cell row after cell row, are handed out in batches of 64, round-robin over the waves -- by ENTRIES, not by rows or cells:
  // the median tile holds a few dozen (one batch), while a surface seen edge-on puts thousands of a plane's samples into one cell row.
  int rinc = lane < CH ? rowtot[lane] : 0;  // lane r: entries of cell row r, then the inclusive prefix over the rows
  const int my_rt = rinc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(rinc, off, 64);
    if (lane >= off) rinc += t;
  }
  const int total = __shfl(rinc, 63, 64);
  const int rexc = rinc - my_rt;
  const int n_batches = (total + 63) >> 6;

  if (n_batches > 0 && a.debug != 1) {
    // ---- phase 2: walk the entries, one per wave instruction, lane = (x-corner, channel); a cell's run is summed in registers and added to the
    //      gradient tile with LDS float atomics (neighbouring cells, rows and the other waves' batches meet in the same texels).  The plane's
    //      value v_q is re-interpolated from the cell's four texels with the forward's own arithmetic (bilerp4): each half-wave fetches its
    //      x-column, v_permlane32_swap hands both columns to both halves -- in sorted order these reads stay in cache ----
    uint32_t* R = reinterpret_cast<uint32_t*>(lds + P::RC) + wave * (64 * 8);
    const float* __restrict__ Gs = a.G + s * C + ch;
    const float* __restrict__ pch = pin + ch;
    const uint32_t rowC = (uint32_t)W * (uint32_t)C;
    // the sorted record of lane l's entry of batch fb: issued one batch AHEAD, so that the load's latency runs under the previous batch's walk
    struct Batch { int r, cell; bool valid; float4 rc; };
    auto locate = [&](int fb) {
      Batch B;
      const int e = fb * 64 + lane;
      B.valid = e < total;
      int r = 0;
#pragma unroll
      for (int j = 0; j < CH - 1; ++j) r += e >= __builtin_amdgcn_readlane(rinc, j) ? 1 : 0;  // the entry's cell row
      B.r = r;
      const int er = e - __shfl(rexc, r, 64);
      const int* ci = cincl + r * CW;
      int lo = 0, hi = CW - 1;  // first cell of the row whose inclusive count exceeds er
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ci[mid] <= er) lo = mid + 1; else hi = mid;
      }
      B.cell = lo;
      B.rc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B.valid) B.rc = a.rec[cstart[r * CW + lo] + (er - (ci[lo] - ccount[r * CW + lo]))];
      return B;
    };
    Batch nxt = {};
    if (wave < n_batches) nxt = locate(wave);
    for (int fb = wave; fb < n_batches; fb += NW) {
      const Batch B = nxt;
      // per-entry preparation, lane-parallel (one entry per lane): taps, offsets, weights
      uint4 hd = make_uint4(0u, 0xffffu, 0u, 0u);
      float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B.valid) {
        const AxisTap tx = axis_tap(B.rc.y, W);
        const AxisTap ty = axis_tap(B.rc.z, H);
        const int lx = tx.i0 - (tx0 - 1), ly = ty.i0 - (ty0 - 1);
        if (lx == B.cell && ly == B.r) {  // by construction of the sort key (anything else would land outside the gradient tile)
          hd.x = (uint32_t)__float_as_int(B.rc.x) * (uint32_t)a.row_stride;
          hd.y = (uint32_t)((ly << 8) | lx);                                  // the run key: cell row, cell column
          hd.z = (uint32_t)(ty.i0 * W + tx.i0) * (uint32_t)C;                 // element offset of texel (x0, y0) inside the plane
          hd.w = (tx.i1 != tx.i0 ? 1u : 0u) | (ty.i1 != ty.i0 ? 2u : 0u);    // x0 + 1 / y0 + 1 exist (a clamped corner re-reads the unclamped texel: weight 0)
          const float4 tw = tap_weights(tx, ty);    // (x0y0, x1y0, x0y1, x1y1): the forward's products, bit for bit
          wt = make_float4(tw.x, tw.z, tw.y, tw.w);  // stored (x0y0, x0y1 | x1y0, x1y1): one float2 per x-corner
        }
      }
      *reinterpret_cast<uint4*>(R + lane * 8) = hd;
      *reinterpret_cast<float4*>(R + lane * 8 + 4) = wt;
      if (fb + NW < n_batches) nxt = locate(fb + NW);
      if (a.debug == 2) continue;
      const int mcount = (total - fb * 64) < 64 ? (total - fb * 64) : 64;
      int cur = 0xffff;  // run key of the cell being accumulated
      float p0 = 0.f, p1 = 0.f;
      auto flush = [&]() {  // cell (r, lx) = cur: this lane's texel column is x0 + half = lx - 1 + half (tile-local), rows r - 1 and r
        const int r = cur >> 8, x = (cur & 0xff) - 1 + half;
        if (cur != 0xffff && x >= 0 && x < TW && a.debug != 3) {
          if (r >= 1 && p0 != 0.f) atomicAdd(gt + ((r - 1) * TW + x) * C + ch, p0);
          if (r < TH && p1 != 0.f) atomicAdd(gt + (r * TW + x) * C + ch, p1);
        }
      };
      for (int u0 = 0; u0 < mcount; u0 += UN) {
        float g[UN], t0[UN], t1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int uu = (u0 + u) < mcount ? (u0 + u) : (mcount - 1);
          g[u] = Gs[R[uu * 8]];  // wave-uniform LDS address: broadcast; 128 B of G per entry and half-wave
          const uint32_t toff = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[uu * 8 + 2]);
          const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[uu * 8 + 3]);
          const uint32_t o0 = toff + (half ? (fl & 1u) * (uint32_t)C : 0u);
          t0[u] = pch[o0];                                // this half's x-column, row y0
          t1[u] = pch[o0 + ((fl & 2u) ? rowC : 0u)];      // ... row y0 + 1
        }
        // the chunk's values, branch-free (a chunk's tail repeats the batch's last entry; the accumulation below skips it)
        float v0[UN], v1[UN];
        int keys[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int uu = (u0 + u) < mcount ? (u0 + u) : (mcount - 1);
          keys[u] = __builtin_amdgcn_readfirstlane((int)R[uu * 8 + 1]);
          const float4 w4 = *reinterpret_cast<const float4*>(R + uu * 8 + 4);  // wave-uniform address: broadcast
          const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t0[u]), __float_as_uint(t0[u]), false, false);  // row y0: [0] = x0 column, [1] = x1
          const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t1[u]), __float_as_uint(t1[u]), false, false);  // row y0 + 1
          const float vq = bilerp4(__uint_as_float(s0[0]), __uint_as_float(s0[1]), __uint_as_float(s1[0]), __uint_as_float(s1[1]), w4.x, w4.z, w4.y, w4.w);
          const float gq = fabsf(vq) >= QUOT_TINY ? g[u] * __builtin_amdgcn_rcpf(vq) : 0.f;  // zero / subnormal: the fix-up supplies the exact term
          v0[u] = gq * (half ? w4.z : w4.x);
          v1[u] = gq * (half ? w4.w : w4.y);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          if (u0 + u < mcount) {
            if (keys[u] == cur) {
              p0 += v0[u]; p1 += v1[u];
            } else {
              flush();
              cur = keys[u]; p0 = v0[u]; p1 = v1[u];
            }
          }
        }
      }
      flush();
    }
  }
  __syncthreads();

  // ---- phase 3: regulariser gradient + Adam on the tile's texels (plane_reg_kernel's arithmetic; the data gradient comes from LDS) ----
  const DynConsts dc = load_dyn(a.dyn, a.step_size, a.inv_sqrt_bc2);
  const bool has_fix = a.grad != nullptr && a.fix_count != nullptr && *a.fix_count > 0;  // exact-zero rows left their terms in the gradient plane
  float l_tv = 0.f, l_sm = 0.f, l_l1 = 0.f;
  int ndrop = 0;
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    if (oo[k] >= 0) {
      const int f = tid + NT * k;
      const int c4 = f % C4, t = f / C4;
      const int ly = t / TW, lx = t - ly * TW;
      const int h = ty0 + ly, w = tx0 + lx;
      if (!PRELOAD) {
        tt[k] = ld4(pin + oo[k]);
        mm[k] = ldnt4(pm + oo[k]);
        vv[k] = ldnt4(pv + oo[k]);
      }
      const float* base = pin + c4 * 4;
      auto at = [&](int hh, int ww) { return ld4(base + ((int64_t)hh * W + ww) * C); };
      const float4 greg = plane_reg_grad<C>(at, tt[k], h, w, H, W, time_plane, a.c_tv, a.c_smooth, a.c_l1, l_tv, l_sm, l_l1);
      float4 gd = *reinterpret_cast<const float4*>(gt + f * 4);
      if (has_fix) {
        float* gp = a.grad + poff + oo[k];
        const float4 gf = ld4(gp);
        if (gf.x != 0.f || gf.y != 0.f || gf.z != 0.f || gf.w != 0.f) {
          gd = add4(gd, gf);
          *reinterpret_cast<float4*>(gp) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      if (dc.skip) {  // skipped step (non-finite gradient somewhere in this parameter group): p, m, v unchanged
        stnt4(pout + oo[k], tt[k]);
      } else {
        float4 pp = tt[k];
        ndrop += adam_float4(pp, mm[k], vv[k], gd, greg, a.grad_scale, a.b1, a.b2, a.eps, dc);
        stnt4(pout + oo[k], pp);
        stnt4(pm + oo[k], mm[k]);
        stnt4(pv + oo[k], vv[k]);
      }
    }
  }
  if (ndrop && a.dyn) atomicAdd(&a.dyn->dropped, ndrop);
  // ---- loss values: workgroup reduction, one atomic per workgroup per term ----
  l_tv = wave_sum(l_tv); l_sm = wave_sum(l_sm); l_l1 = wave_sum(l_l1);
  if (lane == 0) { red[0][wave] = l_tv; red[1][wave] = l_sm; red[2][wave] = l_l1; }
  __syncthreads();
  if (tid < 3 && a.losses) {
    float vsum = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) vsum += red[tid][k];
    if (vsum != 0.f) atomicAdd(a.losses + (size_t)(blockIdx.x % a.n_slots) * 16 + tid, vsum);
  }
}

template <int NP, int TW, int TH, int NT, int MINW, bool PRELOAD, int UN>
static int launch_tile(TileArgs& a, hipStream_t st) {
  using P = TilePlan<TW, TH, NT>;
  int64_t blocks = 0;
  int n = 0;
  for (int pass = 0; pass < 2; ++pass)  // time planes first
    for (int q = 0; q < NP; ++q) {
      int ax, bx;
      plane_axes(NP, q, ax, bx);
      if ((NP == 6 && bx == 3) != (pass == 0)) continue;
      a.order[n++] = q;
    }
  for (int k = 0; k < NP; ++k) {
    int ax, bx;
    plane_axes(NP, a.order[k], ax, bx);
    const int W = a.d.res[a.s][ax], H = a.d.res[a.s][bx] > 0 ? a.d.res[a.s][bx] : 1;
    a.tile_off[k] = (int)blocks;
    a.tiles_x[k] = (W + TW - 1) / TW;
    blocks += (int64_t)a.tiles_x[k] * ((H + TH - 1) / TH);
  }
  a.tile_off[NP] = (int)blocks;
  SNERF_REQUIRE(blocks < (1LL << 31), "kplanes_scatter_adam: too many tiles");
  auto k = tile_scatter_adam_kernel<NP, TW, TH, NT, MINW, PRELOAD, UN>;
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::BYTES), "kplanes_scatter_adam LDS size");
    if (rc) return rc;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NT), P::BYTES, st, a);
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
  {
    int64_t mx = 1;
    for (int k = 0; k < desc->n_coords; ++k) mx = mx > desc->res[scale][k] ? mx : desc->res[scale][k];
    SNERF_REQUIRE(mx * mx * desc->C < (1LL << 31), "kplanes_scatter_adam: plane of %lld^2 texels too large for 32-bit offsets", (long long)mx);
  }
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
  a.debug = tile_shape >> 8;
  tile_shape &= 0xff;
  // tile_shape: 0 = default; the others are A-B variants (texels x threads, registers for MINW waves / SIMD, early optimiser loads, chunk)
  if (NP == 6) {
    if (tile_shape == 1) return launch_tile<6, 16, 4, 256, 8, false, 4>(a, st);
    if (tile_shape == 2) return launch_tile<6, 16, 8, 512, 4, true, 8>(a, st);
    if (tile_shape == 3) return launch_tile<6, 16, 8, 256, 6, false, 8>(a, st);
    if (tile_shape == 4) return launch_tile<6, 16, 4, 256, 4, true, 8>(a, st);
    return launch_tile<6, 16, 8, 256, 8, false, 4>(a, st);
  }
  return launch_tile<3, 16, 8, 256, 8, false, 4>(a, st);
}

// Temporal hash grid, backward w.r.t. the table in OWNER-COMPUTES form (round 6), optionally with the optimiser step of the table fused in.
//
// What it replaces: NS/field_components/cuda/csrc/temporal_gridencoder.cu:283-370 (kernel_grid_backward: one atomicAdd per sample, level, corner and
// live column) followed by torch.optim.Adam over the dense table (NS/configs/method_configs.py:648-657) and, in between, the temporal-TV gradient of two
// columns (NS/field_components/temporal_grid.py:352-376).
//
// Why: on camera rays tgrid_bwd_runs_kernel (tgrid.hip) already runs AT the chip-wide rate of memory-side float atomics -- 22 M 64-B atomic requests of
// 12 useful bytes each in 1.26 ms for config 4's main grid (profiles/r06_tgrid_levels.json) -- and above the coarsest levels no two samples share a cell
// (196 608 samples -> 196 5xx distinct cells per level from level 11 on), so combining across rays cannot remove requests.  What can go is the atomic
// itself: every table row gets ONE owner.
//
//   1. bin (count, two scans, fill): every (sample, level, corner pair) is filed under the TILE of 2^k consecutive table rows its corner rows fall into
//      -- a counting sort with one LDS histogram per (sample chunk, level) workgroup and NO global atomics (per-chunk counts go into a [chunks, tiles]
//      matrix whose column prefix sums are the write offsets).  A record is 4 bytes: sample index, which (y, z) corner pair, which of its two x corners.
//   2. tiles: one workgroup per tile holds the tile's gradient rows in LDS (256 rows x 66 columns x 4 B = 67.6 KB: two workgroups per CU), walks the
//      tile's records (re-deriving cell, weights and the <= C + 1 live columns from the ray and its time, exactly as the scatter kernels do), adds with
//      LDS atomics, and then either
//        MODE 0: adds the tile into the dense gradient buffer with plain loads / stores (it is the only writer of those rows), or
//        MODE 1: runs Adam for its rows straight from LDS -- p, m, v are read and written once, the dense gradient buffer is not touched at all
//                (32 -> 24 B per parameter for the sweep, and the scatter's read-modify-write traffic is gone).
//
// The coarsest levels (few rows, thousands of samples per row: one tile would receive 10^4..10^5 records) stay with the run-length atomic kernel, where
// consecutive samples of a ray share cells and requests are few; in MODE 1 their tiles read (and clear) what that kernel left in the gradient buffer.
#include <stdlib.h>

#include "plane_adam_common.hpp"  // adam_float4, ldnt4 / stnt4: compiled with the optimiser sweep's own contraction setting
#include "tgrid_common.hpp"       // no contraction from here on: cells and weights exactly as tgrid.hip derives them

namespace snerf {

constexpr int TT_NT = 512;                  // threads of a tile workgroup
constexpr int TT_BIN_NT = 256;              // threads of a binning workgroup
constexpr int TT_MAX_LEVEL_TILES = 8192;    // LDS histogram of the binning kernels: 2 ints per tile of one level

struct TileArgs {
  snerf_tgrid_desc d;
  snerf_coords c;
  snerf_tgrid_tile_plan pl;
  const float* times;
  int spr;
  int64_t B;
  const float* gout;
  float4* pos4;        // [B]: (x, y, z in [0,1]^3 as tg_sample_x derives them, time) -- written by the binning entry, read by every later pass, so that the
                       // tile kernels never touch the caller's ray buffers (they may run on another stream while the next step's head rewrites those)
  int32_t* counts;     // [n_chunks][n_tiles]: per-chunk record counts, then (scan) the chunk's write offset inside the tile
  int32_t* tile_base;  // [n_tiles + 1]
  uint32_t* records;
  float* gemb;         // MODE 0: accumulated into; MODE 1: optional contribution of the coarse levels (read and cleared)
  float* p; float* m; float* v;
  float step_size, b1, b2, inv_sqrt_bc2, eps;
  int col_a, col_b;    // temporal-TV columns (MODE 1; col_a < 0: none)
  const float* srow;   // [rows]: signed TV step per table row
  int tile0;           // first tile of the launch
};

// the records of sample b at one level: (tile in level, record) per (y, z) corner pair -- its two x corners share a tile unless a tile boundary lies between
// their rows (hashed levels: rows r and r ^ 1 mostly), then one record each
template <int C, typename F>
__device__ __forceinline__ void tt_for_records(const TileArgs& a, const TgLevel& lv, int level, int64_t b, F&& emit) {
  const float4 ps = a.pos4[b];
  const float x[3] = {ps.x, ps.y, ps.z};
  if ((x[0] < 0.f) || (x[0] > 1.f) || (x[1] < 0.f) || (x[1] > 1.f) || (x[2] < 0.f) || (x[2] > 1.f)) return;  // .cu:119-124
  if (a.gout) {  // binning may run before the gradient exists (grad_out = NULL: every in-range sample is filed; a zero gradient then adds nothing in the tile pass)
    const float* g = a.gout + b * (a.d.L * C) + level * C;
    bool any = false;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) any |= g[ch] != 0.f;
    if (!any) return;  // nothing to add for this (sample, level)
  }
  uint32_t pg[3];
  float fr[3];
  tg_cell(lv, a.d.align_corners != 0, x, pg, fr);
#pragma unroll
  for (int yz = 0; yz < 4; ++yz) {
    const uint32_t cy = pg[1] + (uint32_t)(yz & 1), cz = pg[2] + (uint32_t)(yz >> 1);
    const uint32_t t0 = lv.row_of(pg[0], cy, cz) >> a.pl.tile_rows_log2, t1 = lv.row_of(pg[0] + 1u, cy, cz) >> a.pl.tile_rows_log2;
    const uint32_t base = ((uint32_t)b << 4) | ((uint32_t)yz << 2);
    if (t0 == t1) emit(t0, base | 3u);
    else { emit(t0, base | 1u); emit(t1, base | 2u); }
  }
}

__global__ __launch_bounds__(256) void tt_positions_kernel(TileArgs a) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  float x[3];
  tg_sample_x(a.c, b, x);
  a.pos4[b] = make_float4(x[0], x[1], x[2], a.times[(uint32_t)b / (uint32_t)a.spr]);
}

// count (FILL = false) / fill (FILL = true): grid (chunks, tiled levels)
template <int C, bool FILL>
__global__ __launch_bounds__(TT_BIN_NT) void tt_bin_kernel(TileArgs a) {
  extern __shared__ int tt_hist[];
  const int level = (int)blockIdx.y + a.pl.first_tiled_level, chunk = (int)blockIdx.x;
  const int T0 = a.pl.tile_start[level], nt = a.pl.tile_start[level + 1] - T0;
  int* hist = tt_hist;
  int* base = tt_hist + nt;
  int32_t* mine = a.counts + (int64_t)chunk * a.pl.n_tiles + T0;
  for (int i = threadIdx.x; i < nt; i += TT_BIN_NT) {
    hist[i] = 0;
    if (FILL) base[i] = a.tile_base[T0 + i] + mine[i];
  }
  __syncthreads();
  const TgLevel lv = tg_level(a.d, level);
  const int64_t b0 = (int64_t)chunk * a.pl.chunk;
  const int64_t b1 = b0 + a.pl.chunk < a.B ? b0 + a.pl.chunk : a.B;
  for (int64_t b = b0 + threadIdx.x; b < b1; b += TT_BIN_NT)
    tt_for_records<C>(a, lv, level, b, [&](uint32_t t, uint32_t rec) {
      const int rank = atomicAdd(&hist[t], 1);
      if (FILL) a.records[base[t] + rank] = rec;
    });
  if (!FILL) {
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += TT_BIN_NT) mine[i] = hist[i];
  }
}

// per tile: counts[chunk][tile] -> the chunk's offset inside the tile (exclusive prefix over the chunks); totals[tile] = the tile's records
__global__ __launch_bounds__(256) void tt_scan_chunks_kernel(int32_t* __restrict__ counts, int n_chunks, int n_tiles, int t_first, int32_t* __restrict__ totals) {
  const int t = t_first + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= n_tiles) return;
  int run = 0;
  for (int c = 0; c < n_chunks; ++c) {
    const int v = counts[(int64_t)c * n_tiles + t];
    counts[(int64_t)c * n_tiles + t] = run;
    run += v;
  }
  totals[t] = run;
}

// exclusive prefix over the tiles (one workgroup): tile_base[t] = records in front of tile t; tile_base[n_tiles] = all of them.  In place.
__global__ __launch_bounds__(1024) void tt_scan_tiles_kernel(int32_t* __restrict__ tile_base, int n_tiles, int t_first) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int t = tid; t < t_first; t += 1024) tile_base[t] = 0;  // tiles of the coarse (atomic) levels hold no records
  const int n = n_tiles - t_first;
  const int per = (n + 1023) / 1024;
  const int i0 = t_first + tid * per;
  int local = 0;
  for (int k = 0; k < per; ++k)
    if (i0 + k < n_tiles) local += tile_base[i0 + k];
  int incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int before = 0;
  for (int k = 0; k < w; ++k) before += wsum[k];
  int run = before + incl - local;
  for (int k = 0; k < per; ++k)
    if (i0 + k < n_tiles) {
      const int v = tile_base[i0 + k];
      tile_base[i0 + k] = run;
      run += v;
    }
  if (tid == 1023) {
    int total = 0;
    for (int k = 0; k < 16; ++k) total += wsum[k];
    tile_base[n_tiles] = total;
    tile_base[n_tiles + 1] = 0;  // the wave-specialised tile kernel's ticket and exit counter (it leaves them at zero itself; this covers an aborted launch)
    tile_base[n_tiles + 2] = 0;
  }
}

// MODE 0: dense gradient += tile; MODE 1: Adam (+ temporal TV) for the tile's rows
template <int C, int MODE>
__global__ __launch_bounds__(TT_NT) void tt_tiles_kernel(TileArgs a) {
  extern __shared__ float tt_acc[];
  const int tile = (int)blockIdx.x + a.tile0;
  int level = 0;
  while (level + 1 < a.d.L && tile >= a.pl.tile_start[level + 1]) ++level;
  const TgLevel lv = tg_level(a.d, level);
  const int sh = a.pl.tile_rows_log2;
  const uint32_t row0 = (uint32_t)(tile - a.pl.tile_start[level]) << sh;
  const uint32_t nrows = (lv.rows - row0) < (1u << sh) ? (lv.rows - row0) : (1u << sh);
  const int gc = a.d.grid_C;
  const int64_t gb = ((int64_t)lv.off0 + row0) * gc, ge = gb + (int64_t)nrows * gc;  // this tile's floats of the table
  const int64_t q0 = gb >> 2;
  const int nq = (int)(((ge + 3) >> 2) - q0);  // float4 groups that overlap the tile; the first / last may belong to a neighbour in part
  const int ph = (int)(gb - (q0 << 2));
  // the float4 groups that lie wholly inside the tile: [qa, qb)
  const int qa = ph ? 1 : 0, qb = nq - ((ge & 3) ? 1 : 0);
  const bool coarse = level < a.pl.first_tiled_level;  // its gradient came through the atomic kernel into gemb
  // MODE 1, streaming part (below), software-pipelined: TT_U groups per thread and stage, the next stage's 3 x TT_U loads in flight while this one is
  // computed and stored; the FIRST stage is requested here, in front of the record walk, so that the memory system works while the tile is being summed
  constexpr int TT_U = 2;
  constexpr int STRIDE = TT_U * TT_NT;
  float4 PA[TT_U], MA[TT_U], VA[TT_U], PB[TT_U], MB[TT_U], VB[TT_U];
  auto load = [&](float4* P, float4* M, float4* V, int qbase) {
#pragma unroll
    for (int u = 0; u < TT_U; ++u) {
      const int q = qbase + u * TT_NT;
      if (q < qb) {
        const int64_t f0 = (q0 + q) << 2;
        P[u] = ldnt4(a.p + f0); M[u] = ldnt4(a.m + f0); V[u] = ldnt4(a.v + f0);
      }
    }
  };
  const bool stream = MODE == 1 && !(coarse && a.gemb);
  int cur = qa + (int)threadIdx.x;
  if (stream) load(PA, MA, VA, cur);

  // ---- the tile's records: this thread's first record and its position are requested before the LDS image is cleared (two dependent round trips) ----
  const int rec0 = a.tile_base[tile], rec1 = a.tile_base[tile + 1];
  int i_next = rec0 + (int)threadIdx.x;
  uint32_t rec_next = i_next < rec1 ? a.records[i_next] : 0u;
  float4 ps_next = i_next < rec1 ? a.pos4[rec_next >> 4] : make_float4(0.f, 0.f, 0.f, 0.f);

  for (int q = threadIdx.x; q < nq; q += TT_NT) *reinterpret_cast<float4*>(tt_acc + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  lds_barrier();

  const int n_trows = gc - C - 1;
  const int gstride = a.d.L * C;
  for (int i = i_next; i < rec1; i += TT_NT) {
    const uint32_t rec = rec_next;
    const float4 ps = ps_next;
    if (i + TT_NT < rec1) {  // the next record of this thread (tiles of the coarse levels hold thousands)
      rec_next = a.records[i + TT_NT];
      ps_next = a.pos4[rec_next >> 4];
    }
    const int64_t b = (int64_t)(rec >> 4);
    const int yz = (int)(rec >> 2) & 3, xm = (int)(rec & 3u);
    const float x[3] = {ps.x, ps.y, ps.z};
    uint32_t pg[3];
    float fr[3];
    tg_cell(lv, a.d.align_corners != 0, x, pg, fr);
    const float t = ps.w;
    const float tv = t * (float)(n_trows - 1);
    int r = (int)tv;
    if (t == 1.f) r = n_trows - 1;
    const int pch = r % C;
    const float wa_p = (float)(r + 1) - tv, wb_p = tv - (float)r;  // tg_slot_from_time: the blending channel's two weights
    const float* g = a.gout + b * gstride + level * C;
    float gch[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) gch[ch] = g[ch];
    const uint32_t cy = pg[1] + (uint32_t)(yz & 1), cz = pg[2] + (uint32_t)(yz >> 1);
#pragma unroll
    for (int xb = 0; xb < 2; ++xb) {
      if (!((xm >> xb) & 1)) continue;
      float w = 1.f;  // the corner's weight, factors in axis order as tgrid_kernel multiplies them
      w *= xb ? fr[0] : 1.f - fr[0];
      w *= (yz & 1) ? fr[1] : 1.f - fr[1];
      w *= (yz >> 1) ? fr[2] : 1.f - fr[2];
      const uint32_t row = lv.row_of(pg[0] + (uint32_t)xb, cy, cz);
      float* rowp = tt_acc + ph + (int)(row - row0) * gc;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const int occ = r > ch ? C + ch + C * ((r - 1 - ch) / C) : ch;
        const float wt = ch == pch ? wa_p : 1.f;
        if (wt != 0.f) {
          const float val = w * (gch[ch] * wt);
          if (val != 0.f) atomicAdd(rowp + occ, val);
        }
        if (ch == pch && wb_p != 0.f) {
          const float val = w * (gch[ch] * wb_p);
          if (val != 0.f) atomicAdd(rowp + C + r, val);
        }
      }
    }
  }
  if (MODE == 1 && a.col_a >= 0) {
    // temporal TV (temporal_grid.py:352-376): srow[row] = weight / rows * sign(E[row, a] - E[row, b]) from the OLD table (tgrid_tv_sign_kernel); added with
    // LDS atomics like the records' terms, in the same phase (no barrier of its own)
    for (uint32_t lr = threadIdx.x; lr < nrows; lr += TT_NT) {
      const float s = a.srow[(int64_t)lv.off0 + row0 + lr];
      if (s != 0.f) {
        atomicAdd(tt_acc + ph + (int)lr * gc + a.col_a, s);
        atomicAdd(tt_acc + ph + (int)lr * gc + a.col_b, -s);
      }
    }
  }
  lds_barrier();

  // ---- epilogue over the float4 groups; a group that straddles the tile's first / last float is handled element by element ----
  const DynConsts dc = {a.step_size, a.inv_sqrt_bc2, 0};
  if (MODE == 0) {
    for (int q = threadIdx.x; q < nq; q += TT_NT) {
      const int64_t f0 = (q0 + q) << 2;
      const float4 gq = *reinterpret_cast<const float4*>(tt_acc + 4 * q);
      if (gq.x == 0.f && gq.y == 0.f && gq.z == 0.f && gq.w == 0.f) continue;
      if (f0 >= gb && f0 + 4 <= ge) {
        float4 o = ld4(a.gemb + f0);
        o.x += gq.x; o.y += gq.y; o.z += gq.z; o.w += gq.w;
        *reinterpret_cast<float4*>(a.gemb + f0) = o;
      } else {
        const float* G = &gq.x;
        for (int k = 0; k < 4; ++k)
          if (f0 + k >= gb && f0 + k < ge && G[k] != 0.f) a.gemb[f0 + k] += G[k];
      }
    }
    return;
  }
  if (!stream) {
    for (int q = qa + (int)threadIdx.x; q < qb; q += TT_NT) {
      const int64_t f0 = (q0 + q) << 2;
      const float4 gq = *reinterpret_cast<const float4*>(tt_acc + 4 * q);
      float4 pp = ldnt4(a.p + f0), mm = ldnt4(a.m + f0), vv = ldnt4(a.v + f0);
      const float4 extra = ldnt4(a.gemb + f0);
      if (extra.x != 0.f || extra.y != 0.f || extra.z != 0.f || extra.w != 0.f) stnt4(a.gemb + f0, make_float4(0.f, 0.f, 0.f, 0.f));
      adam_float4(pp, mm, vv, gq, extra, 1.f, a.b1, a.b2, a.eps, dc);
      stnt4(a.p + f0, pp);
      stnt4(a.m + f0, mm);
      stnt4(a.v + f0, vv);
    }
  } else {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto step = [&](float4* P, float4* M, float4* V, int qbase) {
#pragma unroll
      for (int u = 0; u < TT_U; ++u) {
        const int q = qbase + u * TT_NT;
        if (q < qb) {
          const int64_t f0 = (q0 + q) << 2;
          const float4 gq = *reinterpret_cast<const float4*>(tt_acc + 4 * q);
          adam_float4(P[u], M[u], V[u], gq, zero, 1.f, a.b1, a.b2, a.eps, dc);
          stnt4(a.p + f0, P[u]);
          stnt4(a.m + f0, M[u]);
          stnt4(a.v + f0, V[u]);
        }
      }
    };
    while (cur < qb) {
      load(PB, MB, VB, cur + STRIDE);
      step(PA, MA, VA, cur);
      cur += STRIDE;
      if (cur >= qb) break;
      load(PA, MA, VA, cur + STRIDE);
      step(PB, MB, VB, cur);
      cur += STRIDE;
    }
  }
  // the (at most two) groups shared with a neighbouring tile: element by element, this tile's floats only
  if (threadIdx.x < 2) {
    const int q = threadIdx.x == 0 ? 0 : nq - 1;
    const bool partial = threadIdx.x == 0 ? (qa == 1) : (qb == nq - 1 && nq - 1 >= qa);
    if (partial && (threadIdx.x == 0 || nq > 1 || qa == 0)) {
      const int64_t f0 = (q0 + q) << 2;
      const float4 gq = *reinterpret_cast<const float4*>(tt_acc + 4 * q);
      float4 pp = make_float4(0.f, 0.f, 0.f, 0.f), mm = pp, vv = pp, extra = pp;
      float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; float* E = &extra.x;
      for (int k = 0; k < 4; ++k)
        if (f0 + k >= gb && f0 + k < ge) {
          P[k] = a.p[f0 + k]; M[k] = a.m[f0 + k]; V[k] = a.v[f0 + k];
          if (coarse && a.gemb) { E[k] = a.gemb[f0 + k]; if (E[k] != 0.f) a.gemb[f0 + k] = 0.f; }
        }
      adam_float4(pp, mm, vv, gq, extra, 1.f, a.b1, a.b2, a.eps, dc);
      for (int k = 0; k < 4; ++k)
        if (f0 + k >= gb && f0 + k < ge) { a.p[f0 + k] = P[k]; a.m[f0 + k] = M[k]; a.v[f0 + k] = V[k]; }
    }
  }
}

// ---- MODE 1, persistent and wave-specialised (round 6, second form) ----
// tt_tiles_kernel<C, 1> walks a tile's records and THEN streams its rows through Adam: with two workgroups per CU the memory system idles whenever both sit
// in their record walks (measured: 24 B / parameter at 5.0 TB/s, against 5.85 TB/s for the plain sweep).  Here one workgroup per CU lives for the whole launch
// and splits its waves by role: TT_WS_BUILD waves sum tile k + 1 into one LDS image while TT_WS_STREAM waves stream tile k from the other one through Adam
// (and leave it zeroed behind them, so the builders never clear anything).  One workgroup barrier per tile; tiles are handed out by a global ticket
// (heavy tiles -- the coarse levels, first in tile order -- then simply cost their workgroup a few tickets).
constexpr int TT_WS_BUILD = 4, TT_WS_STREAM = 8, TT_WS_NT = (TT_WS_BUILD + TT_WS_STREAM) * 64;

struct TileGeo {
  int level;
  uint32_t row0, nrows;
  int64_t gb, ge, q0;
  int nq, ph, qa, qb;
};
__device__ __forceinline__ TileGeo tt_geo(const TileArgs& a, int tile, TgLevel& lv) {
  TileGeo g;
  g.level = 0;
  while (g.level + 1 < a.d.L && tile >= a.pl.tile_start[g.level + 1]) ++g.level;
  lv = tg_level(a.d, g.level);
  const int sh = a.pl.tile_rows_log2;
  g.row0 = (uint32_t)(tile - a.pl.tile_start[g.level]) << sh;
  g.nrows = (lv.rows - g.row0) < (1u << sh) ? (lv.rows - g.row0) : (1u << sh);
  g.gb = ((int64_t)lv.off0 + g.row0) * a.d.grid_C;
  g.ge = g.gb + (int64_t)g.nrows * a.d.grid_C;
  g.q0 = g.gb >> 2;
  g.nq = (int)(((g.ge + 3) >> 2) - g.q0);
  g.ph = (int)(g.gb - (g.q0 << 2));
  g.qa = g.ph ? 1 : 0;
  g.qb = g.nq - ((g.ge & 3) ? 1 : 0);
  return g;
}

// the record walk + temporal-TV rows of one tile by `nthr` threads (this one is number `tid`), into an LDS image that is ZERO on entry
template <int C>
__device__ __forceinline__ void tt_build_tile(const TileArgs& a, int tile, float* acc, int tid, int nthr) {
  TgLevel lv;
  const TileGeo g = tt_geo(a, tile, lv);
  const int gc = a.d.grid_C, level = g.level;
  const int rec0 = a.tile_base[tile], rec1 = a.tile_base[tile + 1];
  const int n_trows = gc - C - 1;
  const int gstride = a.d.L * C;
  int i = rec0 + tid;
  uint32_t rec_next = i < rec1 ? a.records[i] : 0u;
  float4 ps_next = i < rec1 ? a.pos4[rec_next >> 4] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (; i < rec1; i += nthr) {
    const uint32_t rec = rec_next;
    const float4 ps = ps_next;
    if (i + nthr < rec1) {
      rec_next = a.records[i + nthr];
      ps_next = a.pos4[rec_next >> 4];
    }
    const int64_t b = (int64_t)(rec >> 4);
    const int yz = (int)(rec >> 2) & 3, xm = (int)(rec & 3u);
    const float x[3] = {ps.x, ps.y, ps.z};
    uint32_t pg[3];
    float fr[3];
    tg_cell(lv, a.d.align_corners != 0, x, pg, fr);
    const float t = ps.w;
    const float tv = t * (float)(n_trows - 1);
    int r = (int)tv;
    if (t == 1.f) r = n_trows - 1;
    const int pch = r % C;
    const float wa_p = (float)(r + 1) - tv, wb_p = tv - (float)r;
    const float* gp = a.gout + b * gstride + level * C;
    float gch[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) gch[ch] = gp[ch];
    const uint32_t cy = pg[1] + (uint32_t)(yz & 1), cz = pg[2] + (uint32_t)(yz >> 1);
#pragma unroll
    for (int xb = 0; xb < 2; ++xb) {
      if (!((xm >> xb) & 1)) continue;
      float w = 1.f;
      w *= xb ? fr[0] : 1.f - fr[0];
      w *= (yz & 1) ? fr[1] : 1.f - fr[1];
      w *= (yz >> 1) ? fr[2] : 1.f - fr[2];
      const uint32_t row = lv.row_of(pg[0] + (uint32_t)xb, cy, cz);
      float* rowp = acc + g.ph + (int)(row - g.row0) * gc;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const int occ = r > ch ? C + ch + C * ((r - 1 - ch) / C) : ch;
        const float wt = ch == pch ? wa_p : 1.f;
        if (wt != 0.f) {
          const float val = w * (gch[ch] * wt);
          if (val != 0.f) atomicAdd(rowp + occ, val);
        }
        if (ch == pch && wb_p != 0.f) {
          const float val = w * (gch[ch] * wb_p);
          if (val != 0.f) atomicAdd(rowp + C + r, val);
        }
      }
    }
  }
  if (a.col_a >= 0) {
    for (uint32_t lr = (uint32_t)tid; lr < g.nrows; lr += (uint32_t)nthr) {
      const float s = a.srow[(int64_t)lv.off0 + g.row0 + lr];
      if (s != 0.f) {
        atomicAdd(acc + g.ph + (int)lr * gc + a.col_a, s);
        atomicAdd(acc + g.ph + (int)lr * gc + a.col_b, -s);
      }
    }
  }
}

template <int C>
__global__ __launch_bounds__(TT_WS_NT) void tt_tiles_ws_kernel(TileArgs a, int* ticket, int img_floats) {
  extern __shared__ float tt_acc[];
  __shared__ int s_tile[3];
  const int wave = threadIdx.x >> 6;
  const bool builder = wave < TT_WS_BUILD;
  const int n_tiles = a.pl.n_tiles;
  float* img[2] = {tt_acc, tt_acc + img_floats};
  if (threadIdx.x == 0) {
    s_tile[0] = atomicAdd(ticket, 1);
    s_tile[1] = atomicAdd(ticket, 1);
  }
  for (int q = threadIdx.x; q < img_floats / 2; q += TT_WS_NT) *reinterpret_cast<float4*>(tt_acc + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);  // both images
  lds_barrier();
  if (builder && s_tile[0] < n_tiles) tt_build_tile<C>(a, s_tile[0], img[0], threadIdx.x, TT_WS_BUILD * 64);
  lds_barrier();

  constexpr int SNT = TT_WS_STREAM * 64;
  constexpr int TT_U = 2;
  constexpr int STRIDE = TT_U * SNT;
  const DynConsts dc = {a.step_size, a.inv_sqrt_bc2, 0};
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const int stid = (int)threadIdx.x - TT_WS_BUILD * 64;  // streamer thread number
  float4 PA[TT_U], MA[TT_U], VA[TT_U], PB[TT_U], MB[TT_U], VB[TT_U];
  TgLevel lv_cur;
  TileGeo g = {};
  int cur_tile = s_tile[0];
  auto load = [&](float4* P, float4* M, float4* V, const TileGeo& gg, int qbase) {
#pragma unroll
    for (int u = 0; u < TT_U; ++u) {
      const int q = qbase + u * SNT;
      if (q < gg.qb) {
        const int64_t f0 = (gg.q0 + q) << 2;
        P[u] = ldnt4(a.p + f0); M[u] = ldnt4(a.m + f0); V[u] = ldnt4(a.v + f0);
      }
    }
  };
  bool have_first = false;  // streamers: stage 0 of the current tile is in PA / MA / VA already (requested at the end of the previous tile)
  if (!builder && cur_tile < n_tiles) {
    g = tt_geo(a, cur_tile, lv_cur);
    if (!(g.level < a.pl.first_tiled_level && a.gemb)) { load(PA, MA, VA, g, g.qa + stid); have_first = true; }
  }
  for (int k = 0;; ++k) {
    const int cur = s_tile[k % 3], nxt = s_tile[(k + 1) % 3];
    if (cur >= n_tiles) break;
    float* acc = img[k & 1];
    if (builder) {
      if (threadIdx.x == 0) s_tile[(k + 2) % 3] = atomicAdd(ticket, 1);  // nobody reads this slot during iteration k
      if (nxt < n_tiles) tt_build_tile<C>(a, nxt, img[(k + 1) & 1], threadIdx.x, TT_WS_BUILD * 64);
    } else {
      const bool coarse = g.level < a.pl.first_tiled_level && a.gemb;
      if (coarse) {
        for (int q = g.qa + stid; q < g.qb; q += SNT) {
          const int64_t f0 = (g.q0 + q) << 2;
          const float4 gq = *reinterpret_cast<const float4*>(acc + 4 * q);
          *reinterpret_cast<float4*>(acc + 4 * q) = zero;
          float4 pp = ldnt4(a.p + f0), mm = ldnt4(a.m + f0), vv = ldnt4(a.v + f0);
          const float4 extra = ldnt4(a.gemb + f0);
          if (extra.x != 0.f || extra.y != 0.f || extra.z != 0.f || extra.w != 0.f) stnt4(a.gemb + f0, zero);
          adam_float4(pp, mm, vv, gq, extra, 1.f, a.b1, a.b2, a.eps, dc);
          stnt4(a.p + f0, pp); stnt4(a.m + f0, mm); stnt4(a.v + f0, vv);
        }
      } else {
        auto step = [&](float4* P, float4* M, float4* V, int qbase) {
#pragma unroll
          for (int u = 0; u < TT_U; ++u) {
            const int q = qbase + u * SNT;
            if (q < g.qb) {
              const int64_t f0 = (g.q0 + q) << 2;
              const float4 gq = *reinterpret_cast<const float4*>(acc + 4 * q);
              *reinterpret_cast<float4*>(acc + 4 * q) = zero;  // the image is clean again when the tile has been streamed
              adam_float4(P[u], M[u], V[u], gq, zero, 1.f, a.b1, a.b2, a.eps, dc);
              stnt4(a.p + f0, P[u]); stnt4(a.m + f0, M[u]); stnt4(a.v + f0, V[u]);
            }
          }
        };
        int c = g.qa + stid;
        if (!have_first) load(PA, MA, VA, g, c);
        while (c < g.qb) {
          load(PB, MB, VB, g, c + STRIDE);
          step(PA, MA, VA, c);
          c += STRIDE;
          if (c >= g.qb) break;
          load(PA, MA, VA, g, c + STRIDE);
          step(PB, MB, VB, c);
          c += STRIDE;
        }
      }
      // the (at most two) float4 groups shared with a neighbouring tile: element by element, this tile's floats only
      if (stid < 2) {
        const int q = stid == 0 ? 0 : g.nq - 1;
        const bool partial = stid == 0 ? (g.qa == 1) : (g.qb == g.nq - 1 && g.nq - 1 >= g.qa);
        if (partial) {
          const int64_t f0 = (g.q0 + q) << 2;
          const float4 gq = *reinterpret_cast<const float4*>(acc + 4 * q);
          *reinterpret_cast<float4*>(acc + 4 * q) = zero;
          float4 pp = zero, mm = zero, vv = zero, extra = zero;
          float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; float* E = &extra.x;
          for (int e = 0; e < 4; ++e)
            if (f0 + e >= g.gb && f0 + e < g.ge) {
              P[e] = a.p[f0 + e]; M[e] = a.m[f0 + e]; V[e] = a.v[f0 + e];
              if (coarse) { E[e] = a.gemb[f0 + e]; if (E[e] != 0.f) a.gemb[f0 + e] = 0.f; }
            }
          adam_float4(pp, mm, vv, gq, extra, 1.f, a.b1, a.b2, a.eps, dc);
          for (int e = 0; e < 4; ++e)
            if (f0 + e >= g.gb && f0 + e < g.ge) { a.p[f0 + e] = P[e]; a.m[f0 + e] = M[e]; a.v[f0 + e] = V[e]; }
        }
      }
      // stage 0 of the NEXT tile goes out before the barrier: the round trip overlaps the hand-over
      have_first = false;
      if (nxt < n_tiles) {
        g = tt_geo(a, nxt, lv_cur);
        if (!(g.level < a.pl.first_tiled_level && a.gemb)) { load(PA, MA, VA, g, g.qa + stid); have_first = true; }
      }
    }
    lds_barrier();
  }
  // the last workgroup to leave puts the ticket back to zero for the next launch over these tiles
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ticket + 1, 1) == (int)gridDim.x - 1) {
      ticket[0] = 0;
      ticket[1] = 0;
    }
  }
}

static int tiles_lds_bytes(const snerf_tgrid_desc* d, int sh) { return (((1 << sh) * d->grid_C + 6) / 4 + 1) * 16; }

static int validate_tiles(const snerf_tgrid_desc* d, const snerf_tgrid_tile_plan* pl, int64_t B) {
  SNERF_REQUIRE(d && pl, "tgrid tiles: null descriptor");
  SNERF_REQUIRE(d->D == 3, "tgrid tiles: D=%d (3 only)", d->D);
  SNERF_REQUIRE(d->C == 1 || d->C == 2 || d->C == 4 || d->C == 8, "tgrid tiles: level_dim C=%d unsupported (1,2,4,8)", d->C);
  SNERF_REQUIRE(d->L >= 1 && d->L <= 32 && d->grid_C > d->C + 1 && (d->grid_C & 1) == 0, "tgrid tiles: L=%d grid_C=%d (even row length needed)", d->L, d->grid_C);
  SNERF_REQUIRE(B >= 0 && B < (1LL << 28), "tgrid tiles: B=%lld (< 2^28)", (long long)B);
  SNERF_REQUIRE(pl->tile_rows_log2 >= 2 && pl->tile_rows_log2 <= 16 && pl->n_tiles == pl->tile_start[d->L] && pl->chunk >= 1 &&
                    pl->n_chunks == (int)((B + pl->chunk - 1) / pl->chunk) && pl->first_tiled_level >= 0 && pl->first_tiled_level <= d->L,
                "tgrid tiles: the plan does not belong to this descriptor / batch (snerf_tgrid_tile_plan_make)");
  return 0;
}

template <int C>
static int bin_launch(const TileArgs& a, hipStream_t st) {
  const int L = a.d.L, Lc = a.pl.first_tiled_level;
  if (a.B > 0) hipLaunchKernelGGL(tt_positions_kernel, dim3((unsigned)ceil_div(a.B, 256)), dim3(256), 0, st, a);
  if (Lc >= L || a.B == 0) {
    hipLaunchKernelGGL(tt_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, a.tile_base, a.pl.n_tiles, a.pl.n_tiles);
    SNERF_LAUNCH_CHECK("tgrid_bwd_bin (empty)");
    return 0;
  }
  int max_nt = 0;
  for (int l = Lc; l < L; ++l) max_nt = a.pl.tile_start[l + 1] - a.pl.tile_start[l] > max_nt ? a.pl.tile_start[l + 1] - a.pl.tile_start[l] : max_nt;
  const size_t lds = (size_t)max_nt * 2 * sizeof(int);
  const dim3 grid((unsigned)a.pl.n_chunks, (unsigned)(L - Lc));
  const int t_first = a.pl.tile_start[Lc];
  hipLaunchKernelGGL((tt_bin_kernel<C, false>), grid, dim3(TT_BIN_NT), lds, st, a);
  hipLaunchKernelGGL(tt_scan_chunks_kernel, dim3((unsigned)ceil_div(a.pl.n_tiles - t_first, 256)), dim3(256), 0, st, a.counts, a.pl.n_chunks, a.pl.n_tiles, t_first,
                     a.tile_base);
  hipLaunchKernelGGL(tt_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, a.tile_base, a.pl.n_tiles, t_first);
  hipLaunchKernelGGL((tt_bin_kernel<C, true>), grid, dim3(TT_BIN_NT), lds, st, a);
  SNERF_LAUNCH_CHECK("tgrid_bwd_bin");
  return 0;
}

// SNERF_TGRID_TILES_WS=1: the persistent wave-specialised form of the fused pass (read per call).  OFF by default: alone it takes the same 1.90-1.94 ms as one
// workgroup per tile (both sit at what this GPU sustains for three read and three write streams), and inside config 4's step it is SLOWER (4.04-4.17 against
// 3.79-3.83 ms per step, profiles/r06_tgrid_tiles_ab.txt): its one workgroup per CU holds 135 KB of LDS for the whole launch, which keeps the proposal
// backward and the next step's head -- running beside it on purpose -- off those CUs.
static bool tiles_ws_on() {
  const char* e = getenv("SNERF_TGRID_TILES_WS");
  return e && atoi(e) == 1;
}

template <int C>
static int tiles_ws_launch(TileArgs& a, int* ticket, hipStream_t st) {
  const int one = tiles_lds_bytes(&a.d, a.pl.tile_rows_log2);  // a multiple of 16
  const int lds = 2 * one;
  SNERF_ALLOW_LDS((tt_tiles_ws_kernel<C>), lds);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = a.pl.n_tiles < cus ? a.pl.n_tiles : cus;
  if (grid <= 0) return 0;
  hipLaunchKernelGGL((tt_tiles_ws_kernel<C>), dim3((unsigned)grid), dim3(TT_WS_NT), (size_t)lds, st, a, ticket, one / 4);
  SNERF_LAUNCH_CHECK("tgrid_bwd_tiles_adam (wave-specialised)");
  return 0;
}

template <int C, int MODE>
static int tiles_launch(TileArgs& a, hipStream_t st) {
  const int lds = tiles_lds_bytes(&a.d, a.pl.tile_rows_log2);
  SNERF_ALLOW_LDS((tt_tiles_kernel<C, MODE>), lds);
  a.tile0 = MODE == 0 ? a.pl.tile_start[a.pl.first_tiled_level] : 0;  // MODE 0: tiles of the coarse levels hold no records
  const int n = a.pl.n_tiles - a.tile0;
  if (n <= 0) return 0;
  hipLaunchKernelGGL((tt_tiles_kernel<C, MODE>), dim3((unsigned)n), dim3(TT_NT), (size_t)lds, st, a);
  SNERF_LAUNCH_CHECK(MODE == 0 ? "tgrid_bwd_tiles" : "tgrid_bwd_tiles_adam");
  return 0;
}

#define TT_DISPATCH_C(C_, CALL)          \
  switch (C_) {                          \
    case 1: return CALL(1);              \
    case 2: return CALL(2);              \
    case 4: return CALL(4);              \
    default: return CALL(8);             \
  }

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_tgrid_tile_plan_make(const snerf_tgrid_desc* desc, int64_t B, int32_t tile_rows_log2, int32_t first_tiled_level,
                                          snerf_tgrid_tile_plan* plan) {
  SNERF_REQUIRE(desc && plan, "tgrid_tile_plan_make: null argument");
  SNERF_REQUIRE(desc->L >= 1 && desc->L <= 32 && desc->grid_C >= 2 && B >= 0, "tgrid_tile_plan_make: L=%d grid_C=%d B=%lld", desc->L, desc->grid_C, (long long)B);
  int64_t max_rows = 0;
  for (int l = 0; l < desc->L; ++l) max_rows = desc->offsets[l + 1] - desc->offsets[l] > max_rows ? desc->offsets[l + 1] - desc->offsets[l] : max_rows;
  int sh = tile_rows_log2;
  if (sh <= 0) {
    // the largest tile whose LDS image lets two workgroups share a CU's 160 KB, but no more tiles per level than the binning histogram holds
    sh = 2;
    while (sh < 16 && tiles_lds_bytes(desc, sh + 1) <= 72 * 1024) ++sh;
  }
  while (sh < 16 && ((max_rows + (1LL << sh) - 1) >> sh) > TT_MAX_LEVEL_TILES) ++sh;
  SNERF_REQUIRE(sh >= 2 && sh <= 16 && tiles_lds_bytes(desc, sh) <= 156 * 1024, "tgrid_tile_plan_make: a tile of 2^%d rows x %d columns does not fit LDS", sh, desc->grid_C);
  plan->tile_rows_log2 = sh;
  int t = 0;
  for (int l = 0; l < desc->L; ++l) {
    plan->tile_start[l] = t;
    t += (int)((desc->offsets[l + 1] - desc->offsets[l] + (1LL << sh) - 1) >> sh);
  }
  for (int l = desc->L; l < 33; ++l) plan->tile_start[l] = t;
  plan->n_tiles = t;
  plan->chunk = 4096;
  plan->n_chunks = (int)((B + plan->chunk - 1) / plan->chunk);
  int lc = first_tiled_level;
  if (lc < 0) {
    // levels with fewer than 2^16 rows (the coarsest dense ones: 10^4..10^5 records per tile) stay with the run-length atomic kernel
    lc = 0;
    while (lc < desc->L && desc->offsets[lc + 1] - desc->offsets[lc] < (1 << 16)) ++lc;
  }
  plan->first_tiled_level = lc > desc->L ? desc->L : lc;
  plan->lds_bytes = tiles_lds_bytes(desc, sh);
  plan->count_ints = (int64_t)(plan->n_chunks > 0 ? plan->n_chunks : 1) * plan->n_tiles;
  plan->record_capacity = B * (desc->L - plan->first_tiled_level) * 8;
  return 0;
}

extern "C" int snerf_tgrid_bwd_bin(const snerf_tgrid_desc* desc, const snerf_tgrid_tile_plan* plan, const snerf_coords* coords, const float* times,
                                   int32_t samples_per_row, int64_t B, const float* grad_out, float* pos4, int32_t* counts, int32_t* tile_base,
                                   uint32_t* records, snerf_stream_t stream) {
  int rc = validate_tiles(desc, plan, B);
  if (rc) return rc;
  SNERF_REQUIRE(coords && times && samples_per_row >= 1, "tgrid_bwd_bin: coords / times / samples_per_row");
  SNERF_REQUIRE(coords->mode == 0 || coords->mode == 1, "tgrid_bwd_bin: coords.mode=%d", coords->mode);
  if (coords->mode == 0) SNERF_REQUIRE(coords->pts || B == 0, "tgrid_bwd_bin: pts is null");
  if (coords->mode == 1) SNERF_REQUIRE(coords->S >= 1 && B % coords->S == 0 && coords->origins && coords->dirs && coords->ebins, "tgrid_bwd_bin: bad ray coords");
  SNERF_REQUIRE(counts && tile_base && (pos4 || B == 0) && (records || plan->record_capacity == 0), "tgrid_bwd_bin: null buffer");
  SNERF_REQUIRE(((uintptr_t)pos4 & 15) == 0, "tgrid_bwd_bin: pos4 must be 16-byte aligned");
  TileArgs a = {};
  a.d = *desc; a.c = *coords; a.pl = *plan; a.times = times; a.spr = samples_per_row; a.B = B; a.gout = grad_out;
  a.pos4 = reinterpret_cast<float4*>(pos4); a.counts = counts; a.tile_base = tile_base; a.records = records;
#define TT_CALL(C_) bin_launch<C_>(a, (hipStream_t)stream)
  TT_DISPATCH_C(desc->C, TT_CALL)
#undef TT_CALL
}

extern "C" int snerf_tgrid_bwd_tiles(const snerf_tgrid_desc* desc, const snerf_tgrid_tile_plan* plan, int64_t B, const float* grad_out, const float* pos4,
                                     const int32_t* tile_base, const uint32_t* records, float* grad_embeddings, snerf_stream_t stream) {
  int rc = validate_tiles(desc, plan, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(tile_base && records && grad_out && pos4 && grad_embeddings, "tgrid_bwd_tiles: null buffer");
  SNERF_REQUIRE((((uintptr_t)grad_embeddings | (uintptr_t)pos4) & 15) == 0, "tgrid_bwd_tiles: grad_embeddings / pos4 must be 16-byte aligned");
  TileArgs a = {};
  a.d = *desc; a.pl = *plan; a.B = B; a.gout = grad_out; a.pos4 = reinterpret_cast<float4*>(const_cast<float*>(pos4));
  a.tile_base = const_cast<int32_t*>(tile_base); a.records = const_cast<uint32_t*>(records); a.gemb = grad_embeddings;
  a.col_a = -1;
#define TT_CALL(C_) tiles_launch<C_, 0>(a, (hipStream_t)stream)
  TT_DISPATCH_C(desc->C, TT_CALL)
#undef TT_CALL
}

extern "C" int snerf_tgrid_bwd_tiles_adam(const snerf_tgrid_desc* desc, const snerf_tgrid_tile_plan* plan, int64_t B, const float* grad_out, const float* pos4,
                                          int32_t* tile_base, const uint32_t* records, float* grad_embeddings, float* p, float* m, float* v, float lr,
                                          float beta1, float beta2, float eps, int32_t step, int32_t col_a, int32_t col_b, const float* srow,
                                          snerf_stream_t stream) {
  int rc = validate_tiles(desc, plan, B);
  if (rc) return rc;
  SNERF_REQUIRE(tile_base && (records || B == 0) && (grad_out || B == 0) && (pos4 || B == 0) && p && m && v, "tgrid_bwd_tiles_adam: null buffer");
  SNERF_REQUIRE(step >= 1, "tgrid_bwd_tiles_adam: step=%d (1-based)", step);
  SNERF_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad_embeddings | (uintptr_t)pos4) & 15) == 0,
                "tgrid_bwd_tiles_adam: buffers must be 16-byte aligned");
  SNERF_REQUIRE(plan->first_tiled_level == 0 || grad_embeddings, "tgrid_bwd_tiles_adam: levels [0, %d) go through the atomic kernel: pass their gradient buffer",
                plan->first_tiled_level);
  SNERF_REQUIRE(col_a < 0 || (srow && col_b >= 0 && col_a < desc->grid_C && col_b < desc->grid_C && col_a != col_b), "tgrid_bwd_tiles_adam: TV columns (%d,%d)", col_a,
                col_b);
  TileArgs a = {};
  a.d = *desc; a.pl = *plan; a.B = B; a.gout = grad_out; a.pos4 = reinterpret_cast<float4*>(const_cast<float*>(pos4));
  a.tile_base = tile_base; a.records = const_cast<uint32_t*>(records); a.gemb = grad_embeddings;
  a.p = p; a.m = m; a.v = v; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  adam_consts(lr, beta1, beta2, step, a.step_size, a.inv_sqrt_bc2);
  a.col_a = col_a; a.col_b = col_b; a.srow = srow;
  if (tiles_ws_on()) {
    int* ticket = a.tile_base + plan->n_tiles + 1;
#define TT_CALL(C_) tiles_ws_launch<C_>(a, ticket, (hipStream_t)stream)
    TT_DISPATCH_C(desc->C, TT_CALL)
#undef TT_CALL
  }
#define TT_CALL(C_) tiles_launch<C_, 1>(a, (hipStream_t)stream)
  TT_DISPATCH_C(desc->C, TT_CALL)
#undef TT_CALL
}

// Temporal multi-level hash/tiled grid encoder (NeRFPlayer): forward gather and backward atomic scatter.
//
// gfx950 equivalent of the reference's only native code, NS/field_components/cuda/csrc/temporal_gridencoder.cu
// (fast_hash :46-59, get_grid_index :62-88, kernel_grid :91-280, kernel_grid_backward :283-370) as driven by
// NS/field_components/temporal_grid.py (TemporalGridEncodeFunc :33-156, get_temporal_index :320-330).
//
// Differences in design (not in results):
//  * the reference materialises a [B, 4C] float `temporal_row_index` per SAMPLE; time is a per-ray quantity and the
//    channel table has a closed form (oracle/tgrid_oracle.py::channel_table), so the kernel can take times[R] and derive
//    (column, weight) per lane arithmetically -- the explicit-rows form is kept for API parity;
//  * 2*C lanes own one (sample, level): lane = (channel, a|b column).  The <= 3 live columns of a corner row are read /
//    atomically added by ADJACENT lanes, so a corner costs one 64-B request instead of up to three;
//  * outputs are written [B, L*C] directly (the reference writes [L,B,C] and permutes in Python);
//  * sample coordinates can be derived in-kernel from rays (snerf_coords mode 1), as for the K-Planes gather.
#include <stdlib.h>

#include "tgrid_common.hpp"  // tg_slot_from_time (shared with tgrid_tiles.hip)

// No fused multiply-adds formed by contraction in this file: the run-length kernels below promise the per-sample kernels' results BIT FOR BIT, and
// which a * b + c the compiler fuses depends on the code around it (the first run-length forward differed from tgrid_kernel<false> by 1 ulp in ray
// mode for that reason alone).  With contraction off every kernel here evaluates the IEEE operations exactly as written.  These kernels are bound by
// their table accesses; the few extra VALU instructions do not show.
#pragma clang fp contract(off)

namespace snerf {

struct TgridArgs {
  snerf_tgrid_desc d;
  snerf_coords c;
  const float* emb;
  const float* trow;   // [B/spr, 4C] or null
  const float* times;  // [B/spr] or null (exactly one of trow/times)
  int spr;             // samples per row of trow/times (1 = per sample; S = per ray)
  int64_t B;
  float* out;          // fwd: [B, L*C]
  float* dy_dx;        // fwd, optional: [B, L, D, C] d out / d x (temporal_gridencoder.cu:204-273, calc_grad_inputs)
  const float* gout;   // bwd
  float* gemb;         // bwd
  int level0, level1;  // levels [level0, level1) of this launch (level1 = 0: all; snerf_tgrid_encode_bwd_levels, dev switch SNERF_TGRID_LEVELS)
  long long* gemb_fx;  // bwd, deterministic mode: 2^50-scaled fixed-point cells instead of gemb (common.hpp: integer addition is associative)
};

__device__ __forceinline__ void tg_grad_add(const TgridArgs& a, size_t e, float v) {
  if (a.gemb_fx) fx_atomic_add(a.gemb_fx + e, v); else atomicAdd(a.gemb + e, v);
}

template <bool BWD, bool DYDX = false>
__global__ __launch_bounds__(256) void tgrid_kernel(TgridArgs a) {
  const int C = a.d.C, D = a.d.D;
  const int LPG = 2 * C;  // lanes per (sample, level)
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid / LPG;
  const int k = (int)(gid - b * LPG);
  const int ch = k >> 1, ab = k & 1;
  const int level = blockIdx.y + a.level0;
  const bool live = b < a.B;
  const int64_t bb = live ? b : a.B - 1;

  // ---- coordinates in [0,1] ----
  float x[3] = {0.f, 0.f, 0.f};
  if (a.c.mode == 0) {
    for (int d = 0; d < D; ++d) x[d] = a.c.pts[bb * D + d];
  } else {
    const int64_t r = bb / a.c.S;
    const int s = (int)(bb - r * a.c.S);
    const float* eb = a.c.ebins + r * (a.c.S + 1) + s;
    const float mid = eb[0] + eb[1];
    for (int d = 0; d < 3; ++d) {
      float pos = a.c.origins[r * 3 + d] + (a.c.dirs[r * 3 + d] * mid) / 2.f;
      x[d] = (pos - a.c.aabb_min[d]) / (a.c.aabb_max[d] - a.c.aabb_min[d]);
    }
  }
  bool oob = false;
  for (int d = 0; d < D; ++d) oob |= (x[d] < 0.f) || (x[d] > 1.f);  // .cu:119-124

  // ---- this lane's column and temporal weight ----
  int col;
  float wt;
  const int64_t trow_i = bb / a.spr;
  if (a.trow) {
    const float* tr = a.trow + trow_i * (4 * C) + ch * 4;
    const float wa = tr[0];
    if (ab == 0) { wt = wa; col = (int)__float2uint_rn(tr[1]); }
    else { wt = (wa == 1.f) ? 0.f : tr[2]; col = (int)__float2uint_rn(tr[3]); }  // .cu:182-195: w_a == 1 => single column
  } else {
    tg_slot_from_time(a.times[trow_i], C, a.d.grid_C - C - 1, ch, ab, col, wt);
  }

  const uint32_t off0 = (uint32_t)a.d.offsets[level];
  const uint32_t hashmap_size = (uint32_t)(a.d.offsets[level + 1] - a.d.offsets[level]);
  const float scale = exp2f((float)level * a.d.S) * (float)a.d.H - 1.0f;  // .cu:146-148
  const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
  float pos[3];
  uint32_t pg[3];
  for (int d = 0; d < D; ++d) {
    pos[d] = x[d] * scale + (a.d.align_corners ? 0.0f : 0.5f);
    float f = floorf(pos[d]);
    pg[d] = (uint32_t)f;
    pos[d] -= f;
  }

  float g = 0.f;
  if (BWD) g = a.gout[bb * (a.d.L * C) + level * C + ch] * wt;
  const bool active = live && !oob && wt != 0.f;
  float acc = 0.f;
  float dacc[3] = {0.f, 0.f, 0.f};  // DYDX: d (this lane's column share of the output) / d x_d
  const int ncorner = 1 << D;
  // Row index of a corner = get_grid_index (.cu:62-88): fast_hash (XOR of coordinate * prime, .cu:46-59) on hashed levels, the
  // strided sum on dense ones, modulo the level's table size.  Per-axis terms, shared by the 2^D corners: t[d][bit] = (pg[d] + bit) * (prime[d] | dense stride[d]); a corner's row is
  // their XOR (hashed level) or sum (dense level), then one reduction modulo the table size -- a mask when the size is a power of two
  // (every hashed level: 2^log2_hashmap_size).  Same values as get_grid_index (.cu:62-88), a third of the integer work.
  const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
  uint32_t term[3][2];
  bool hashed;
  {
    uint32_t stride = 1;
    for (int d = 0; d < D && stride <= hashmap_size; ++d) stride *= a.d.align_corners ? resolution : (resolution + 1);
    hashed = a.d.gridtype == 0 && stride > hashmap_size;
    uint32_t st = 1;
    for (int d = 0; d < D; ++d) {
      const uint32_t m = hashed ? primes[d] : (st <= hashmap_size ? st : 0u);  // dense: axes beyond the overflowing stride do not contribute (.cu:70-74)
      term[d][0] = pg[d] * m;
      term[d][1] = (pg[d] + 1u) * m;
      if (st <= hashmap_size) st *= a.d.align_corners ? resolution : (resolution + 1);
    }
  }
  const bool pow2 = (hashmap_size & (hashmap_size - 1u)) == 0u;
  for (int idx = 0; idx < ncorner; ++idx) {
    float w = 1.f;
    uint32_t index = 0;
    for (int d = 0; d < D; ++d) {
      const int bit = (idx >> d) & 1;
      w *= bit ? pos[d] : 1.f - pos[d];
      index = hashed ? (index ^ term[d][bit]) : (index + term[d][bit]);
    }
    const uint32_t row = pow2 ? (index & (hashmap_size - 1u)) : (index % hashmap_size);
    const size_t e = ((size_t)off0 + row) * (size_t)a.d.grid_C + (size_t)col;
    if (active) {
      if (BWD) {
        float v = w * g;
        if (v != 0.f) tg_grad_add(a, e, v);
      } else {
        const float val = a.emb[e] * wt;
        acc += w * val;
        if (DYDX) {
          // d/d x_gd of the D-linear interpolation: scale * sum over the corners of (+1 on the far side, -1 on the near side of axis gd) x the
          // other axes' weights x the corner's value (.cu:211-267 pairs the corners up as right - left)
          for (int gd = 0; gd < D; ++gd) {
            float wo = scale;
            for (int d = 0; d < D; ++d)
              if (d != gd) wo *= ((idx >> d) & 1) ? pos[d] : 1.f - pos[d];
            dacc[gd] += (((idx >> gd) & 1) ? wo : -wo) * val;
          }
        }
      }
    }
  }
  if (!BWD) {
    acc += __shfl_xor(acc, 1, 64);  // column a + column b of this channel
    if (live && ab == 0) a.out[b * (a.d.L * C) + level * C + ch] = oob ? 0.f : acc;
    if (DYDX) {
      for (int gd = 0; gd < D; ++gd) {
        float v = dacc[gd];
        v += __shfl_xor(v, 1, 64);
        // out-of-range inputs: the reference returns before dy_dx is written (.cu:119-124) into a zero-initialised buffer
        if (live && ab == 0) a.dy_dx[((b * a.d.L + level) * D + gd) * C + ch] = oob ? 0.f : v;
      }
    }
  }
}

// ---- backward, run-length form (round 4): samples of ONE ray, per-ray times (what the fused trainers hand over) ----
// tgrid_kernel<true> issues one float atomic per (sample, level, corner, live column): 20.7 M 64-B requests per step of config 4, and the pass runs AT the
// chip-wide float-atomic rate (1.2 TB/s of written bytes, profiles/r04_kernels.md section 7).  Consecutive samples of a ray fall into the same cell on
// every level whose cells are wider than the sample spacing -- all of them for the proposal grids (max_res 64 / 256 against 256 / 96 samples per ray),
// the coarser half for the main grid -- and a ray has ONE time, hence one set of live columns.  Here a lane group (2 C lanes = (channel, a | b column),
// as before) walks a SEGMENT of consecutive samples of one ray at one level and sums the eight corner contributions in registers while the cell stays
// the same; it sends them when the cell changes.  Same additions as before in another association (float atomics are order-dependent anyway).
constexpr int TG_RUN = 32;  // samples per segment (a ray of S samples = ceil(S / 32) segments of equal length, the last one shorter)

// RAYS: snerf_coords mode 1 (a segment lies inside one ray: origin, direction and the time slot are loaded once); otherwise explicit points [B,3] in
// sample order with times[b / spr] -- the full NeRFPlayer's deformed positions: consecutive points are still consecutive samples of a ray with one time,
// and where they are not, the (cell, column) key changes and the sum is sent, so any input order is handled correctly.
template <bool RAYS>
__global__ __launch_bounds__(256) void tgrid_bwd_runs_kernel(TgridArgs a, int segs, int run) {
  const int C = a.d.C, LPG = 2 * C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t grp = gid / LPG;
  const int k = (int)(gid - grp * LPG);
  const int ch = k >> 1, ab = k & 1;
  const int level = blockIdx.y + a.level0;
  int64_t b0, b1, r = 0;
  if (RAYS) {
    const int S = a.c.S;
    r = grp / segs;
    const int seg = (int)(grp - r * segs);
    if (r >= a.B / S) return;
    const int s0 = seg * run, s1 = (s0 + run) < S ? (s0 + run) : S;
    b0 = r * S + s0; b1 = r * S + s1;
  } else {
    b0 = grp * run;
    if (b0 >= a.B) return;
    b1 = (b0 + run) < a.B ? (b0 + run) : a.B;
  }
  const int n_rows = a.d.grid_C - C - 1;
  int col = 0;
  float wt = 0.f;
  if (RAYS) {
    tg_slot_from_time(a.times[r], C, n_rows, ch, ab, col, wt);
    if (wt == 0.f) return;  // this (channel, column) slot is dead for the whole ray
  }

  const uint32_t off0 = (uint32_t)a.d.offsets[level];
  const uint32_t hashmap_size = (uint32_t)(a.d.offsets[level + 1] - a.d.offsets[level]);
  const float scale = exp2f((float)level * a.d.S) * (float)a.d.H - 1.0f;
  const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
  const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
  uint32_t mult[3];
  bool hashed;
  {
    uint32_t stride = 1;
    for (int d = 0; d < 3 && stride <= hashmap_size; ++d) stride *= a.d.align_corners ? resolution : (resolution + 1);
    hashed = a.d.gridtype == 0 && stride > hashmap_size;
    uint32_t st = 1;
    for (int d = 0; d < 3; ++d) {
      mult[d] = hashed ? primes[d] : (st <= hashmap_size ? st : 0u);
      if (st <= hashmap_size) st *= a.d.align_corners ? resolution : (resolution + 1);
    }
  }
  const bool pow2 = (hashmap_size & (hashmap_size - 1u)) == 0u;
  float o[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 0.f}, inv[3] = {1.f, 1.f, 1.f};
  const float* eb = nullptr;
  if (RAYS) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      o[d] = a.c.origins[r * 3 + d];
      dir[d] = a.c.dirs[r * 3 + d];
      inv[d] = a.c.aabb_max[d] - a.c.aabb_min[d];
    }
    eb = a.c.ebins + r * (a.c.S + 1) - r * a.c.S;  // eb[b] = edge s of ray r for b = r S + s
  }
  const int gstride = a.d.L * C;
  const float* gp = a.gout + level * C + ch;

  uint32_t ppg[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};  // no cell: floor() of a coordinate in [0, scale + 0.5] never gives this
  int pcol = col;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  auto flush = [&]() {
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      if (acc[idx] != 0.f) {
        uint32_t index = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const uint32_t t = (ppg[d] + ((idx >> d) & 1)) * mult[d];
          index = hashed ? (index ^ t) : (index + t);
        }
        const uint32_t row = pow2 ? (index & (hashmap_size - 1u)) : (index % hashmap_size);
        tg_grad_add(a, ((size_t)off0 + row) * (size_t)a.d.grid_C + (size_t)pcol, acc[idx]);
        acc[idx] = 0.f;
      }
    }
  };
  for (int64_t b = b0; b < b1; ++b) {
    if (!RAYS) tg_slot_from_time(a.times[b / a.spr], C, n_rows, ch, ab, col, wt);
    const float g = gp[b * gstride] * wt;
    float pos[3];
    uint32_t pg[3];
    bool oob = false;
    float mid = 0.f;
    if (RAYS) mid = eb[b] + eb[b + 1];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float x;
      if (RAYS) {
        const float p = o[d] + (dir[d] * mid) / 2.f;
        x = (p - a.c.aabb_min[d]) / inv[d];
      } else {
        x = a.c.pts[b * 3 + d];
      }
      oob |= (x < 0.f) || (x > 1.f);
      pos[d] = x * scale + (a.d.align_corners ? 0.0f : 0.5f);
      const float f = floorf(pos[d]);
      pg[d] = (uint32_t)f;
      pos[d] -= f;
    }
    if (oob || g == 0.f) continue;  // out-of-range samples get no gradient (.cu:119-124); nothing to add
    if (pg[0] != ppg[0] || pg[1] != ppg[1] || pg[2] != ppg[2] || col != pcol) {
      flush();
      ppg[0] = pg[0]; ppg[1] = pg[1]; ppg[2] = pg[2];
      pcol = col;
    }
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
      float w = 1.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) w *= ((idx >> d) & 1) ? pos[d] : 1.f - pos[d];
      acc[idx] += w * g;
    }
  }
  flush();
}

// ---- forward, run-length form (round 5): the same walk as tgrid_bwd_runs_kernel ----
// tgrid_kernel<false> fetches 8 corner rows per (sample, level) even where consecutive samples of a ray stay in one cell: 3.50 GB of counter traffic for
// 0.99 GB algorithmic on config 4 (profiles/r04_nerfplayer_fused_pmc.csv), each corner its own 64-B sector of a 264-B row for <= 3 live floats.  Here the
// lane group that walks a segment of <= 32 consecutive samples keeps the eight corner values of the current cell in
// registers and reloads them only when the cell or the live column changes.  Per sample the arithmetic is tgrid_kernel<false>'s own -- same weights, same
// products, same order of the eight additions, the same pair sum of the a | b columns -- so the outputs are bit-identical (tests/test_gpu_tgrid.py).
template <bool RAYS>
__global__ __launch_bounds__(256) void tgrid_fwd_runs_kernel(TgridArgs a, int segs, int run) {
  const int C = a.d.C, LPG = 2 * C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t grp = gid / LPG;
  const int k = (int)(gid - grp * LPG);
  const int ch = k >> 1, ab = k & 1;
  const int level = blockIdx.y + a.level0;
  // a group past the end keeps running with an empty segment: the pair sum below is a cross-lane operation of a full wave
  int64_t b0 = 0, b1 = 0, r = 0;
  if (RAYS) {
    const int S = a.c.S;
    r = grp / segs;
    const int seg = (int)(grp - r * segs);
    if (r < a.B / S) {
      const int s0 = seg * run, s1 = (s0 + run) < S ? (s0 + run) : S;
      b0 = r * S + s0; b1 = r * S + s1;
    } else {
      r = 0;
    }
  } else {
    b0 = grp * run;
    if (b0 < a.B) b1 = (b0 + run) < a.B ? (b0 + run) : a.B; else b0 = 0;
  }
  const int n_rows = a.d.grid_C - C - 1;
  int col = 0;
  float wt = 0.f;
  if (RAYS) tg_slot_from_time(a.times[r], C, n_rows, ch, ab, col, wt);

  const uint32_t off0 = (uint32_t)a.d.offsets[level];
  const uint32_t hashmap_size = (uint32_t)(a.d.offsets[level + 1] - a.d.offsets[level]);
  const float scale = exp2f((float)level * a.d.S) * (float)a.d.H - 1.0f;
  const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
  const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
  uint32_t mult[3];
  bool hashed;
  {
    uint32_t stride = 1;
    for (int d = 0; d < 3 && stride <= hashmap_size; ++d) stride *= a.d.align_corners ? resolution : (resolution + 1);
    hashed = a.d.gridtype == 0 && stride > hashmap_size;
    uint32_t st = 1;
    for (int d = 0; d < 3; ++d) {
      mult[d] = hashed ? primes[d] : (st <= hashmap_size ? st : 0u);
      if (st <= hashmap_size) st *= a.d.align_corners ? resolution : (resolution + 1);
    }
  }
  const bool pow2 = (hashmap_size & (hashmap_size - 1u)) == 0u;
  float o[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 0.f}, inv[3] = {1.f, 1.f, 1.f};
  const float* eb = nullptr;
  if (RAYS) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      o[d] = a.c.origins[r * 3 + d];
      dir[d] = a.c.dirs[r * 3 + d];
      inv[d] = a.c.aabb_max[d] - a.c.aabb_min[d];
    }
    eb = a.c.ebins + r * (a.c.S + 1) - r * a.c.S;  // eb[b] = edge s of ray r for b = r S + s
  }
  const int ostride = a.d.L * C;
  float* op = a.out + level * C + ch;

  uint32_t ppg[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};  // no cell yet
  int pcol = -1;
  float val[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) val[i] = 0.f;
  for (int64_t b = b0; b < b1; ++b) {
    if (!RAYS) tg_slot_from_time(a.times[b / a.spr], C, n_rows, ch, ab, col, wt);
    float pos[3];
    uint32_t pg[3];
    bool oob = false;
    float mid = 0.f;
    if (RAYS) mid = eb[b] + eb[b + 1];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float x;
      if (RAYS) {
        const float p = o[d] + (dir[d] * mid) / 2.f;
        x = (p - a.c.aabb_min[d]) / inv[d];
      } else {
        x = a.c.pts[b * 3 + d];
      }
      oob |= (x < 0.f) || (x > 1.f);
      pos[d] = x * scale + (a.d.align_corners ? 0.0f : 0.5f);
      const float f = floorf(pos[d]);
      pg[d] = (uint32_t)f;
      pos[d] -= f;
    }
    float acc = 0.f;
    if (!oob && wt != 0.f) {  // tgrid_kernel's `active`: an out-of-range sample reads nothing, a dead (channel, column) slot adds nothing
      if (pg[0] != ppg[0] || pg[1] != ppg[1] || pg[2] != ppg[2] || col != pcol) {
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
          uint32_t index = 0;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const uint32_t t = (pg[d] + ((idx >> d) & 1)) * mult[d];
            index = hashed ? (index ^ t) : (index + t);
          }
          const uint32_t row = pow2 ? (index & (hashmap_size - 1u)) : (index % hashmap_size);
          val[idx] = a.emb[((size_t)off0 + row) * (size_t)a.d.grid_C + (size_t)col];
        }
        ppg[0] = pg[0]; ppg[1] = pg[1]; ppg[2] = pg[2];
        pcol = col;
      }
#pragma unroll
      for (int idx = 0; idx < 8; ++idx) {
        float w = 1.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) w *= ((idx >> d) & 1) ? pos[d] : 1.f - pos[d];
        const float v = val[idx] * wt;  // the temporal weight is applied per sample (explicit points: every sample has its own time), as tgrid_kernel does
        acc += w * v;
      }
    }
    acc += __shfl_xor(acc, 1, 64);  // column a + column b of this channel (segments of a wave's groups have the same length: see the launcher)
    if (ab == 0) op[b * ostride] = oob ? 0.f : acc;
  }
}

static int validate(const snerf_tgrid_desc* d, const snerf_coords* c, const float* trow, const float* times, int spr, int64_t B) {
  SNERF_REQUIRE(d && c, "tgrid: null descriptor");
  SNERF_REQUIRE(d->D >= 1 && d->D <= 3, "tgrid: D=%d unsupported (1..3)", d->D);
  SNERF_REQUIRE(d->C == 1 || d->C == 2 || d->C == 4 || d->C == 8, "tgrid: level_dim C=%d unsupported (1,2,4,8)", d->C);
  SNERF_REQUIRE(d->L >= 1 && d->L <= 32, "tgrid: L=%d (<= 32)", d->L);
  SNERF_REQUIRE(d->grid_C > d->C + 1, "tgrid: grid_C=%d must be level_dim + temporal_dim (temporal_dim >= 2)", d->grid_C);
  SNERF_REQUIRE(d->gridtype == 0 || d->gridtype == 1, "tgrid: gridtype=%d", d->gridtype);
  SNERF_REQUIRE(B >= 0 && spr >= 1, "tgrid: B=%lld samples_per_row=%d", (long long)B, spr);
  SNERF_REQUIRE((trow != nullptr) != (times != nullptr), "tgrid: pass exactly one of temporal_row_index / times");
  SNERF_REQUIRE(c->mode == 0 || (c->mode == 1 && d->D == 3), "tgrid: coords.mode=%d with D=%d", c->mode, d->D);
  if (c->mode == 0) SNERF_REQUIRE(c->pts || B == 0, "tgrid: pts is null");
  if (c->mode == 1) SNERF_REQUIRE(c->S >= 1 && B % c->S == 0 && c->origins && c->dirs && c->ebins, "tgrid: bad ray coords");
  return 0;
}

// SNERF_TGRID_RUNS=0: dev A-B switch back to the per-sample kernels (read per call)
static bool tgrid_runs_off() {
  const char* e = getenv("SNERF_TGRID_RUNS");
  return e && atoi(e) == 0;
}

// coordinate gradient (kernel_input_backward, .cu:373-398): grad_inputs[b, d] = sum_{l, ch} grad[b, l, ch] * dy_dx[b, l, d, ch]; one lane per (b, d)
__global__ __launch_bounds__(256) void tgrid_input_bwd_kernel(const float* __restrict__ grad, const float* __restrict__ dy_dx, int64_t B, int D, int C, int L,
                                                             float* __restrict__ grad_inputs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * D) return;
  const int64_t b = t / D;
  const int d = (int)(t - b * D);
  float r = 0.f;
  for (int l = 0; l < L; ++l)
    for (int ch = 0; ch < C; ++ch) r += grad[(b * L + l) * C + ch] * dy_dx[((b * L + l) * D + d) * C + ch];
  grad_inputs[t] = r;
}

// SNERF_TGRID_LEVELS=lo:hi -- dev switch (tools/tgrid_levels.py): only levels [lo, hi) are launched; read per call
static void tgrid_level_range(int L, int& lo, int& hi) {
  lo = 0; hi = L;
  const char* e = getenv("SNERF_TGRID_LEVELS");
  if (e) { int x = 0, y = L; if (sscanf(e, "%d:%d", &x, &y) == 2 && x >= 0 && y <= L && x < y) { lo = x; hi = y; } }
}

template <bool BWD>
static int launch(const TgridArgs& a_in, hipStream_t st) {
  TgridArgs a = a_in;
  int lv_lo, lv_hi;
  tgrid_level_range(a.d.L, lv_lo, lv_hi);
  if (a.level1 > 0) { lv_lo = a.level0; lv_hi = a.level1; }
  a.level0 = lv_lo;
  const unsigned n_lv = (unsigned)(lv_hi - lv_lo);
  if (BWD && a.d.D == 3 && a.times && !a.trow && !tgrid_runs_off()) {
    // times instead of explicit temporal rows (every caller but the reference-shaped API test): the run-length form
    if (a.c.mode == 1 && a.spr == a.c.S && a.B % a.c.S == 0) {  // rays: segments inside a ray
      const int S = a.c.S;
      const int segs = (S + TG_RUN - 1) / TG_RUN, run = (S + segs - 1) / segs;  // (segments of 8 ... 256 samples: 4.03 - 4.07 ms per step of config 4, no trend)
      const int64_t threads = (a.B / S) * segs * 2 * a.d.C;
      hipLaunchKernelGGL(tgrid_bwd_runs_kernel<true>, dim3((unsigned)ceil_div(threads, 256), n_lv), dim3(256), 0, st, a, segs, run);
      SNERF_LAUNCH_CHECK("tgrid_encode_bwd (runs)");
      return 0;
    }
    if (a.c.mode == 0) {  // explicit points in sample order
      const int64_t threads = ((a.B + TG_RUN - 1) / TG_RUN) * 2 * a.d.C;
      hipLaunchKernelGGL(tgrid_bwd_runs_kernel<false>, dim3((unsigned)ceil_div(threads, 256), n_lv), dim3(256), 0, st, a, 1, TG_RUN);
      SNERF_LAUNCH_CHECK("tgrid_encode_bwd (runs, points)");
      return 0;
    }
  }
  if (!BWD && !a.dy_dx && a.d.D == 3 && a.times && !a.trow && !tgrid_runs_off()) {
    // the forward walks the same segments.  The pair sum (__shfl_xor) needs both lanes of a (channel) pair in the same loop iteration: a pair
    // shares its group and hence its segment, so its trip count -- whatever other groups of the wave do.
    if (a.c.mode == 1 && a.spr == a.c.S && a.B % a.c.S == 0) {
      const int S = a.c.S;
      const int segs = (S + TG_RUN - 1) / TG_RUN, run = (S + segs - 1) / segs;
      const int64_t threads = (a.B / S) * segs * 2 * a.d.C;
      hipLaunchKernelGGL(tgrid_fwd_runs_kernel<true>, dim3((unsigned)ceil_div(threads, 256), n_lv), dim3(256), 0, st, a, segs, run);
      SNERF_LAUNCH_CHECK("tgrid_encode_fwd (runs)");
      return 0;
    }
    if (a.c.mode == 0) {
      const int64_t threads = ((a.B + TG_RUN - 1) / TG_RUN) * 2 * a.d.C;
      hipLaunchKernelGGL(tgrid_fwd_runs_kernel<false>, dim3((unsigned)ceil_div(threads, 256), n_lv), dim3(256), 0, st, a, 1, TG_RUN);
      SNERF_LAUNCH_CHECK("tgrid_encode_fwd (runs, points)");
      return 0;
    }
  }
  const int64_t threads = a.B * 2 * a.d.C;
  dim3 grid((unsigned)ceil_div(threads, 256), n_lv);
  if (!BWD && a.dy_dx) hipLaunchKernelGGL((tgrid_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(tgrid_kernel<BWD>, grid, dim3(256), 0, st, a);
  SNERF_LAUNCH_CHECK(BWD ? "tgrid_encode_bwd" : "tgrid_encode_fwd");
  return 0;
}

}  // namespace snerf

using namespace snerf;

// ---- temporal TV term (TemporalGridEncoder.get_temporal_tv_loss, temporal_grid.py:352-376): mean_r |E[r, a] - E[r, b]| ----
// One lane per table row; both columns of a row sit in the same 264-B row, so a pass reads one or two sectors per row.
namespace snerf {
__global__ __launch_bounds__(256) void tgrid_tv_fwd_kernel(const float* __restrict__ E, int64_t rows, int grid_C, int a, int b, float* __restrict__ partial,
                                                          int n_slots) {
  float acc = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x)
    acc += fabsf(E[r * grid_C + a] - E[r * grid_C + b]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) atomicAdd(partial + (blockIdx.x % n_slots) * 16, acc);
}
__global__ __launch_bounds__(256) void tgrid_tv_bwd_kernel(const float* __restrict__ E, int64_t rows, int grid_C, int a, int b, const float* __restrict__ g_tv,
                                                          float* __restrict__ gE) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float d = E[r * grid_C + a] - E[r * grid_C + b];
  const float s = ((d > 0.f) - (d < 0.f)) * (g_tv[0] / (float)rows);  // d|x|/dx = sign(x) (0 at 0, as torch.abs), mean over rows
  if (s != 0.f) { gE[r * grid_C + a] += s; gE[r * grid_C + b] -= s; }
}
// value and gradient in one pass over the two columns (the fused trainer knows the upstream gradient -- the loss weight -- up front)
__global__ __launch_bounds__(256) void tgrid_tv_fwd_bwd_kernel(const float* __restrict__ E, int64_t rows, int grid_C, int a, int b, float g_over_rows,
                                                              float* __restrict__ partial, int n_slots, float* __restrict__ gE) {
  float acc = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float d = E[r * grid_C + a] - E[r * grid_C + b];
    acc += fabsf(d);
    const float s = ((d > 0.f) - (d < 0.f)) * g_over_rows;
    if (s != 0.f) { gE[r * grid_C + a] += s; gE[r * grid_C + b] -= s; }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) atomicAdd(partial + (blockIdx.x % n_slots) * 16, acc);
}
// value + per-row signed step for snerf_adam_step_tv: srow[r] = g_over_rows * sign(E[r,a] - E[r,b])
__global__ __launch_bounds__(256) void tgrid_tv_sign_kernel(const float* __restrict__ E, int64_t rows, int grid_C, int a, int b, float g_over_rows,
                                                           float* __restrict__ partial, int n_slots, float* __restrict__ srow) {
  float acc = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float d = E[r * grid_C + a] - E[r * grid_C + b];
    acc += fabsf(d);
    srow[r] = ((d > 0.f) - (d < 0.f)) * g_over_rows;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) atomicAdd(partial + (blockIdx.x % n_slots) * 16, acc);
}
}  // namespace snerf

extern "C" int snerf_tgrid_tv_sign(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, float g_tv, float* partial,
                                   int32_t n_slots, float* srow, snerf_stream_t stream) {
  SNERF_REQUIRE(rows >= 1 && grid_C >= 1 && col_a >= 0 && col_a < grid_C && col_b >= 0 && col_b < grid_C && col_a != col_b && n_slots >= 1,
                "tgrid_tv_sign: rows=%lld grid_C=%d cols=(%d,%d) n_slots=%d", (long long)rows, grid_C, col_a, col_b, n_slots);
  SNERF_REQUIRE(embeddings && partial && srow, "tgrid_tv_sign: null buffer");
  const int64_t blocks = (rows + 255) / 256;
  hipLaunchKernelGGL(snerf::tgrid_tv_sign_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, embeddings, rows, grid_C,
                     col_a, col_b, g_tv / (float)rows, partial, n_slots, srow);
  SNERF_LAUNCH_CHECK("tgrid_tv_sign");
  return 0;
}

extern "C" int snerf_tgrid_tv_fwd_bwd(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, float g_tv, float* partial,
                                      int32_t n_slots, float* grad_embeddings, snerf_stream_t stream) {
  SNERF_REQUIRE(rows >= 1 && grid_C >= 1 && col_a >= 0 && col_a < grid_C && col_b >= 0 && col_b < grid_C && col_a != col_b && n_slots >= 1,
                "tgrid_tv_fwd_bwd: rows=%lld grid_C=%d cols=(%d,%d) n_slots=%d", (long long)rows, grid_C, col_a, col_b, n_slots);
  SNERF_REQUIRE(embeddings && partial && grad_embeddings, "tgrid_tv_fwd_bwd: null buffer");
  const int64_t blocks = (rows + 255) / 256;
  hipLaunchKernelGGL(snerf::tgrid_tv_fwd_bwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, embeddings, rows,
                     grid_C, col_a, col_b, g_tv / (float)rows, partial, n_slots, grad_embeddings);
  SNERF_LAUNCH_CHECK("tgrid_tv_fwd_bwd");
  return 0;
}

extern "C" int snerf_tgrid_tv_fwd(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, float* partial, int32_t n_slots,
                                  snerf_stream_t stream) {
  SNERF_REQUIRE(rows >= 1 && grid_C >= 1 && col_a >= 0 && col_a < grid_C && col_b >= 0 && col_b < grid_C && n_slots >= 1,
                "tgrid_tv_fwd: rows=%lld grid_C=%d cols=(%d,%d) n_slots=%d", (long long)rows, grid_C, col_a, col_b, n_slots);
  SNERF_REQUIRE(embeddings && partial, "tgrid_tv_fwd: null buffer");
  const int64_t blocks = (rows + 255) / 256;
  hipLaunchKernelGGL(snerf::tgrid_tv_fwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, embeddings, rows, grid_C,
                     col_a, col_b, partial, n_slots);
  SNERF_LAUNCH_CHECK("tgrid_tv_fwd");
  return 0;
}

extern "C" int snerf_tgrid_tv_bwd(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, const float* g_tv,
                                  float* grad_embeddings, snerf_stream_t stream) {
  SNERF_REQUIRE(rows >= 1 && grid_C >= 1 && col_a >= 0 && col_a < grid_C && col_b >= 0 && col_b < grid_C && col_a != col_b,
                "tgrid_tv_bwd: rows=%lld grid_C=%d cols=(%d,%d)", (long long)rows, grid_C, col_a, col_b);
  SNERF_REQUIRE(embeddings && g_tv && grad_embeddings, "tgrid_tv_bwd: null buffer");
  hipLaunchKernelGGL(snerf::tgrid_tv_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, embeddings, rows, grid_C, col_a, col_b,
                     g_tv, grad_embeddings);
  SNERF_LAUNCH_CHECK("tgrid_tv_bwd");
  return 0;
}

extern "C" int snerf_tgrid_encode_fwd(const snerf_tgrid_desc* desc, const float* embeddings, const snerf_coords* coords,
                                      const float* temporal_row_index, const float* times, int32_t samples_per_row, int64_t B, float* out,
                                      snerf_stream_t stream) {
  int rc = validate(desc, coords, temporal_row_index, times, samples_per_row, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(embeddings && out, "tgrid_encode_fwd: null buffer");
  TgridArgs a = {};
  a.d = *desc; a.c = *coords; a.emb = embeddings; a.trow = temporal_row_index; a.times = times; a.spr = samples_per_row; a.B = B; a.out = out;
  return launch<false>(a, (hipStream_t)stream);
}

extern "C" int snerf_tgrid_encode_fwd_dydx(const snerf_tgrid_desc* desc, const float* embeddings, const snerf_coords* coords,
                                           const float* temporal_row_index, const float* times, int32_t samples_per_row, int64_t B, float* out,
                                           float* dy_dx, snerf_stream_t stream) {
  int rc = validate(desc, coords, temporal_row_index, times, samples_per_row, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(embeddings && out && dy_dx, "tgrid_encode_fwd_dydx: null buffer");
  TgridArgs a = {};
  a.d = *desc; a.c = *coords; a.emb = embeddings; a.trow = temporal_row_index; a.times = times; a.spr = samples_per_row; a.B = B; a.out = out;
  a.dy_dx = dy_dx;
  return launch<false>(a, (hipStream_t)stream);
}

extern "C" int snerf_tgrid_input_bwd(const float* grad_out, const float* dy_dx, int64_t B, int32_t D, int32_t C, int32_t L, float* grad_inputs,
                                     snerf_stream_t stream) {
  SNERF_REQUIRE(B >= 0 && D >= 1 && D <= 3 && C >= 1 && L >= 1, "tgrid_input_bwd: B=%lld D=%d C=%d L=%d", (long long)B, D, C, L);
  if (B == 0) return 0;
  SNERF_REQUIRE(grad_out && dy_dx && grad_inputs, "tgrid_input_bwd: null buffer");
  hipLaunchKernelGGL(tgrid_input_bwd_kernel, dim3((unsigned)ceil_div(B * D, 256)), dim3(256), 0, (hipStream_t)stream, grad_out, dy_dx, B, D, C, L, grad_inputs);
  SNERF_LAUNCH_CHECK("tgrid_input_bwd");
  return 0;
}

extern "C" int snerf_tgrid_encode_bwd(const snerf_tgrid_desc* desc, const snerf_coords* coords, const float* temporal_row_index,
                                      const float* times, int32_t samples_per_row, int64_t B, const float* grad_out, float* grad_embeddings,
                                      snerf_stream_t stream) {
  int rc = validate(desc, coords, temporal_row_index, times, samples_per_row, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(grad_out && grad_embeddings, "tgrid_encode_bwd: null buffer");
  TgridArgs a = {};
  a.d = *desc; a.c = *coords; a.trow = temporal_row_index; a.times = times; a.spr = samples_per_row; a.B = B; a.gout = grad_out; a.gemb = grad_embeddings;
  return launch<true>(a, (hipStream_t)stream);
}

extern "C" int snerf_tgrid_encode_bwd_levels(const snerf_tgrid_desc* desc, const snerf_coords* coords, const float* temporal_row_index, const float* times,
                                             int32_t samples_per_row, int64_t B, const float* grad_out, float* grad_embeddings, int32_t level_begin,
                                             int32_t level_end, snerf_stream_t stream) {
  int rc = validate(desc, coords, temporal_row_index, times, samples_per_row, B);
  if (rc) return rc;
  SNERF_REQUIRE(level_begin >= 0 && level_begin <= level_end && level_end <= desc->L, "tgrid_encode_bwd_levels: levels [%d, %d) of %d", level_begin, level_end, desc->L);
  if (B == 0 || level_begin == level_end) return 0;
  SNERF_REQUIRE(grad_out && grad_embeddings, "tgrid_encode_bwd_levels: null buffer");
  TgridArgs a = {};
  a.d = *desc; a.c = *coords; a.trow = temporal_row_index; a.times = times; a.spr = samples_per_row; a.B = B; a.gout = grad_out; a.gemb = grad_embeddings;
  a.level0 = level_begin; a.level1 = level_end;
  return launch<true>(a, (hipStream_t)stream);
}

// deterministic mode: the same scatter into 2^50-scaled 64-bit integer cells (snerf_fx_to_float converts them once per step)
extern "C" int snerf_tgrid_encode_bwd_fx(const snerf_tgrid_desc* desc, const snerf_coords* coords, const float* temporal_row_index,
                                         const float* times, int32_t samples_per_row, int64_t B, const float* grad_out, int64_t* grad_embeddings_fx,
                                         snerf_stream_t stream) {
  int rc = validate(desc, coords, temporal_row_index, times, samples_per_row, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(grad_out && grad_embeddings_fx, "tgrid_encode_bwd_fx: null buffer");
  TgridArgs a = {};
  a.d = *desc; a.c = *coords; a.trow = temporal_row_index; a.times = times; a.spr = samples_per_row; a.B = B; a.gout = grad_out;
  a.gemb_fx = reinterpret_cast<long long*>(grad_embeddings_fx);
  return launch<true>(a, (hipStream_t)stream);
}

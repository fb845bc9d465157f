// Pieces shared by the temporal-grid kernels (tgrid.hip: per-sample and run-length kernels; tgrid_tiles.hip: the tiled, owner-computes backward):
// the closed form of the temporal channel table, the per-level geometry of NS/field_components/cuda/csrc/temporal_gridencoder.cu:146-176 and the
// sample position of snerf_coords.  Everything here is evaluated exactly as written (no contraction): the kernels of both files must land in the
// same cell for the same sample.
#pragma once
#include "common.hpp"

#pragma clang fp contract(off)

namespace snerf {

// (column, weight) of slot (ch, ab) at a time row; closed form of the reference's sampling_index table + get_temporal_index
__device__ __forceinline__ void tg_slot_from_time(float t, int C, int n_rows, int ch, int ab, int& col, float& w) {
  const float v = t * (float)(n_rows - 1);
  int r = (int)v;  // floor for t >= 0
  if (t == 1.f) r = n_rows - 1;
  auto occ = [&](int q) { return r > q ? C + q + C * ((r - 1 - q) / C) : q; };
  const int p = r % C;
  if (ch == p) {
    if (ab == 0) { col = occ(p); w = (float)(r + 1) - v; }
    else { col = C + r; w = v - (float)r; }
  } else {
    col = occ(ch);
    w = ab == 0 ? 1.f : 0.f;
  }
}

// One level of a D = 3 grid: table rows [off0, off0 + rows), position scale, and the per-axis multipliers whose XOR (hashed level) or sum
// (dense level) over the corner's integer coordinates, reduced modulo `rows`, is get_grid_index (.cu:62-88).
struct TgLevel {
  uint32_t off0, rows, mult[3];
  float scale;
  bool hashed, pow2;
  __device__ __forceinline__ uint32_t row_of(uint32_t cx, uint32_t cy, uint32_t cz) const {
    const uint32_t a = cx * mult[0], b = cy * mult[1], c = cz * mult[2];
    const uint32_t index = hashed ? (a ^ b ^ c) : (a + b + c);
    return pow2 ? (index & (rows - 1u)) : (index % rows);
  }
};

__device__ __forceinline__ TgLevel tg_level(const snerf_tgrid_desc& d, int level) {
  TgLevel lv;
  lv.off0 = (uint32_t)d.offsets[level];
  lv.rows = (uint32_t)(d.offsets[level + 1] - d.offsets[level]);
  lv.scale = exp2f((float)level * d.S) * (float)d.H - 1.0f;  // .cu:146-148
  const uint32_t resolution = (uint32_t)ceilf(lv.scale) + 1;
  const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
  uint32_t stride = 1;
  for (int k = 0; k < 3 && stride <= lv.rows; ++k) stride *= d.align_corners ? resolution : (resolution + 1);
  lv.hashed = d.gridtype == 0 && stride > lv.rows;
  uint32_t st = 1;
  for (int k = 0; k < 3; ++k) {
    lv.mult[k] = lv.hashed ? primes[k] : (st <= lv.rows ? st : 0u);  // dense: axes beyond the overflowing stride do not contribute (.cu:70-74)
    if (st <= lv.rows) st *= d.align_corners ? resolution : (resolution + 1);
  }
  lv.pow2 = (lv.rows & (lv.rows - 1u)) == 0u;
  return lv;
}

// Sample b of a D = 3 batch in [0,1]^3 (snerf_coords mode 0: explicit points [B,3]; mode 1: the midpoint of bin b % S of ray b / S); returns "out of range"
// (.cu:119-124: such a sample reads nothing and receives no gradient).  The same expressions as tgrid_kernel's.
__device__ __forceinline__ bool tg_sample_x(const snerf_coords& c, int64_t b, float x[3]) {
  if (c.mode == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = c.pts[b * 3 + k];
  } else {
    const uint32_t r = (uint32_t)b / (uint32_t)c.S;  // B < 2^31 here: 32-bit division (the 64-bit one is a ~100-instruction routine)
    const int s = (int)((uint32_t)b - r * (uint32_t)c.S);
    const float* eb = c.ebins + (int64_t)r * (c.S + 1) + s;
    const float mid = eb[0] + eb[1];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float pos = c.origins[(int64_t)r * 3 + k] + (c.dirs[(int64_t)r * 3 + k] * mid) / 2.f;
      x[k] = (pos - c.aabb_min[k]) / (c.aabb_max[k] - c.aabb_min[k]);
    }
  }
  bool oob = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) oob |= (x[k] < 0.f) || (x[k] > 1.f);
  return oob;
}

// cell and in-cell position of x at a level
__device__ __forceinline__ void tg_cell(const TgLevel& lv, bool align_corners, const float x[3], uint32_t pg[3], float frac[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float pos = x[k] * lv.scale + (align_corners ? 0.0f : 0.5f);
    const float f = floorf(pos);
    pg[k] = (uint32_t)f;
    frac[k] = pos - f;
  }
}

}  // namespace snerf

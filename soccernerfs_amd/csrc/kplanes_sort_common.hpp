// The sort order of the sorted plane-gradient scatter (kplanes_sorted.hip: counting sort + pass B): segment table, Morton keys, the zero
// threshold of the quotient form.  (Split out in round 3 for an owner-computes scatter + Adam kernel that looked cells up in the sort's scanned
// histogram; that kernel measured slower than pass B + sweep and was removed -- profiles/r03_tile_adam_experiment.md.)
#pragma once
#include "kplanes_common.hpp"

namespace snerf {

struct SegTable {
  int n_planes;
  int per_scale;    // 1: every (scale, plane) segment has its own order, keyed by that scale's own texel -> runs == cells exactly
  int n_segs;       // n_planes (shared order) or n_scales * n_planes (per-scale)
  int n_scales;
  int res[SNERF_MAX_SCALES][4];
  int cell_off[SNERF_MAX_SCALES * 6 + 1];  // first histogram cell of each segment; [n_segs] = total cells
  int fine[4];      // sort-grid resolution of each axis = its finest resolution over the scales
  int fine_rm[4];   // finer sort grid for the minor axis of row-major (time) planes: several samples share a (time row, texel)
                    // there, and only a (near-)true sort by x keeps every scale's texel index monotone inside a row
  int row_major[6]; // 1: key = i0_b * fine_rm[a] + i0_a(fine_rm) (time planes: the time row is exact at every scale); 0: Morton(i0_a, i0_b)
};

// Morton interleave of two 16-bit integers (x -> even bits, y -> odd bits)
__device__ __forceinline__ uint32_t part1by1(uint32_t v) {
  v &= 0x0000ffffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__device__ __forceinline__ uint32_t morton2(uint32_t x, uint32_t y) { return part1by1(x) | (part1by1(y) << 1); }

// (QUOT_TINY, the "vanished" threshold of the quotient form, lives in common.hpp: the sigma_net backward's epilogue uses it too)


inline int build_segs(const snerf_kplanes_desc* d, SegTable& st) {
  const int NP = d->n_coords == 4 ? 6 : 3;
  static const int PA6[6] = {0, 0, 0, 1, 1, 2}, PB6[6] = {1, 2, 3, 2, 3, 3}, PA3[3] = {0, 0, 1}, PB3[3] = {1, 2, 2};
  st.n_planes = NP;
  for (int k = 0; k < 4; ++k) {
    int fine = 1, coarse = 1 << 30;
    for (int s = 0; s < d->n_scales; ++s) {
      const int r = d->res[s][k] > 0 ? d->res[s][k] : 1;
      fine = r > fine ? r : fine;
      coarse = r < coarse ? r : coarse;
    }
    (void)coarse;
    st.fine[k] = fine;  // measured: aligning the sort grid to the coarsest scale, or a finer grid for the time planes, does not pay
    SNERF_REQUIRE(st.fine[k] <= 32768, "kplanes_sort: resolution %d too large for the Morton key", st.fine[k]);
    st.fine_rm[k] = fine;
  }
  // one shared order per plane (a separate global sort per scale costs +0.5 ms of sorting for -0.3 ms of pass B: profiles/r01_kernels.md;
  // the per-scale branches of the kernels are kept for the descriptor field `per_scale`, which stays 0)
  st.per_scale = 0;
  st.n_scales = d->n_scales;
  for (int s = 0; s < d->n_scales; ++s)
    for (int k = 0; k < 4; ++k) st.res[s][k] = d->res[s][k];
  int64_t off = 0;
  if (st.per_scale) {
    st.n_segs = d->n_scales * NP;
    for (int s = 0; s < d->n_scales; ++s)
      for (int q = 0; q < NP; ++q) {
        const int a = NP == 6 ? PA6[q] : PA3[q], b = NP == 6 ? PB6[q] : PB3[q];
        const int ra = d->res[s][a] > 0 ? d->res[s][a] : 1, rb = d->res[s][b] > 0 ? d->res[s][b] : 1;
        SNERF_REQUIRE(ra <= 32768 && rb <= 32768, "kplanes_sort: resolution too large for the Morton key");
        st.cell_off[s * NP + q] = (int)off;
        st.row_major[q] = (NP == 6 && b == 3) ? 1 : 0;
        if (st.row_major[q]) {
          off += (int64_t)ra * rb;
        } else {
          const int m = ra > rb ? ra : rb;
          int bits = 0;
          while ((1 << bits) < m) ++bits;
          off += (int64_t)1 << (2 * bits);
        }
      }
    SNERF_REQUIRE(off < (1LL << 30), "kplanes_sort: too many Morton cells (%lld)", (long long)off);
    st.cell_off[st.n_segs] = (int)off;
    return 0;
  }
  st.n_segs = NP;
  for (int q = 0; q < NP; ++q) {
    const int a = NP == 6 ? PA6[q] : PA3[q], b = NP == 6 ? PB6[q] : PB3[q];
    st.cell_off[q] = (int)off;
    // planes whose row axis is time: time is not multiscale, so (time row, fine x) row-major keeps every scale's runs whole
    st.row_major[q] = (NP == 6 && b == 3) ? 1 : 0;
    if (st.row_major[q]) {
      off += (int64_t)st.fine_rm[a] * st.fine[b];
    } else {
      const int m = st.fine[a] > st.fine[b] ? st.fine[a] : st.fine[b];
      int bits = 0;
      while ((1 << bits) < m) ++bits;
      off += (int64_t)1 << (2 * bits);  // Morton codes of a (2^bits)^2 square
    }
  }
  SNERF_REQUIRE(off < (1LL << 30), "kplanes_sort: too many Morton cells (%lld)", (long long)off);
  st.cell_off[NP] = (int)off;
  return 0;
}


}  // namespace snerf

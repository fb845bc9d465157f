// Static hash grid (the full NeRFPlayer's "stationary field"), backward w.r.t. the table in OWNER-COMPUTES form with the optimiser step fused in (round 6).
//
// What it replaces: the table half of tcnn's HashGrid backward (Mueller et al. 2022, grid.h kernel_grid_backward: one atomicAdd per sample, level, corner and
// feature; tiny-cuda-nn v1.6 is an un-vendored dependency of the reference, NS/fields/nerfplayer_field.py:242-252) followed by torch.optim.Adam over the table
// (NS/configs/method_configs.py:598-607).  hashgrid_kernel<F, true, 1> (hashgrid.hip) runs at the chip's float-atomic request rate: 2 x 196 608 points x 16
// levels x 4..8 requests = 1.7 ms per step of the full NeRFPlayer.  Same construction as tgrid_tiles.hip: the batch's (point, level, corner pair) touches are
// filed under tiles of 2^k consecutive table rows by a counting sort without global atomics; one workgroup per tile sums its rows in LDS (ds_add_f32) and runs
// Adam over them from there (MODE 1: no dense gradient for this table) or adds them into the dense gradient with plain stores (MODE 0).  The coordinate gradient
// (the deformation net's input) stays with hashgrid_kernel, called without a table gradient: it is a gather.
#include <stdlib.h>

#include "plane_adam_common.hpp"

#pragma clang fp contract(off)

namespace snerf {

constexpr int HT_NT = 512, HT_BIN_NT = 256, HT_MAX_LEVEL_TILES = 8192;

struct HtArgs {
  snerf_hashgrid_desc d;
  snerf_hashgrid_tile_plan pl;
  const float* x;      // [B, 3]
  int64_t B;
  const float* gout;   // [B, L*F]
  int32_t* counts;
  int32_t* tile_base;
  uint32_t* records;
  float* gtable;       // MODE 0: accumulated into; MODE 1: the coarse (atomic) levels' contribution, read and cleared (may be null when every level is tiled)
  int tile0;           // first tile of the launch
  float* p; float* m; float* v;
  float step_size, b1, b2, inv_sqrt_bc2, eps;
};

struct HtLevel {
  uint32_t off0, rows, mult[3];
  float scale;
  bool hashed, pow2;
  __device__ __forceinline__ uint32_t row_of(uint32_t cx, uint32_t cy, uint32_t cz) const {
    const uint32_t a = cx * mult[0], b = cy * mult[1], c = cz * mult[2];
    const uint32_t index = hashed ? (a ^ b ^ c) : (a + b + c);
    return pow2 ? (index & (rows - 1u)) : (index % rows);
  }
};
// grid_index of hashgrid_kernel: per-axis multipliers (dense stride or hash prime), one reduction modulo the level's rows
__device__ __forceinline__ HtLevel ht_level(const snerf_hashgrid_desc& d, int level) {
  HtLevel lv;
  lv.off0 = (uint32_t)d.offsets[level];
  lv.rows = (uint32_t)(d.offsets[level + 1] - d.offsets[level]);
  lv.scale = d.scale[level];
  const uint32_t resolution = (uint32_t)d.resolution[level];
  const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
  uint64_t stride = 1;
  for (int k = 0; k < 3 && stride <= lv.rows; ++k) stride *= resolution;
  lv.hashed = lv.rows < stride;
  uint64_t st = 1;
  for (int k = 0; k < 3; ++k) {
    lv.mult[k] = lv.hashed ? primes[k] : (st <= lv.rows ? (uint32_t)st : 0u);
    if (st <= lv.rows) st *= resolution;
  }
  lv.pow2 = (lv.rows & (lv.rows - 1u)) == 0u;
  return lv;
}
__device__ __forceinline__ void ht_cell(const HtLevel& lv, const float* x, uint32_t pg[3], float fr[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float p = fmaf(lv.scale, x[k], 0.5f);  // pos_fract, as hashgrid_kernel
    const float f = floorf(p);
    pg[k] = (uint32_t)(int)f;
    fr[k] = p - f;
  }
}

template <int F, typename EMIT>
__device__ __forceinline__ void ht_for_records(const HtArgs& a, const HtLevel& lv, int level, int64_t b, EMIT&& emit) {
  if (a.gout) {  // grad_out = NULL: binning before the gradient exists files every point (a zero gradient then adds nothing in the tile pass)
    const float* g = a.gout + b * (a.d.L * F) + level * F;
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; ++f) any |= g[f] != 0.f;
    if (!any) return;
  }
  const float x[3] = {a.x[b * 3], a.x[b * 3 + 1], a.x[b * 3 + 2]};
  uint32_t pg[3];
  float fr[3];
  ht_cell(lv, x, pg, fr);
#pragma unroll
  for (int yz = 0; yz < 4; ++yz) {
    const uint32_t cy = pg[1] + (uint32_t)(yz & 1), cz = pg[2] + (uint32_t)(yz >> 1);
    const uint32_t t0 = lv.row_of(pg[0], cy, cz) >> a.pl.tile_rows_log2, t1 = lv.row_of(pg[0] + 1u, cy, cz) >> a.pl.tile_rows_log2;
    const uint32_t base = ((uint32_t)b << 4) | ((uint32_t)yz << 2);
    if (t0 == t1) emit(t0, base | 3u);
    else { emit(t0, base | 1u); emit(t1, base | 2u); }
  }
}

template <int F, bool FILL>
__global__ __launch_bounds__(HT_BIN_NT) void ht_bin_kernel(HtArgs a) {
  extern __shared__ int ht_hist[];
  const int level = (int)blockIdx.y + a.pl.first_tiled_level, chunk = (int)blockIdx.x;
  const int T0 = a.pl.tile_start[level], nt = a.pl.tile_start[level + 1] - T0;
  int* hist = ht_hist;
  int* base = ht_hist + nt;
  int32_t* mine = a.counts + (int64_t)chunk * a.pl.n_tiles + T0;
  for (int i = threadIdx.x; i < nt; i += HT_BIN_NT) {
    hist[i] = 0;
    if (FILL) base[i] = a.tile_base[T0 + i] + mine[i];
  }
  __syncthreads();
  const HtLevel lv = ht_level(a.d, level);
  const int64_t b0 = (int64_t)chunk * a.pl.chunk;
  const int64_t b1 = b0 + a.pl.chunk < a.B ? b0 + a.pl.chunk : a.B;
  for (int64_t b = b0 + threadIdx.x; b < b1; b += HT_BIN_NT)
    ht_for_records<F>(a, lv, level, b, [&](uint32_t t, uint32_t rec) {
      const int rank = atomicAdd(&hist[t], 1);
      if (FILL) a.records[base[t] + rank] = rec;
    });
  if (!FILL) {
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += HT_BIN_NT) mine[i] = hist[i];
  }
}

__global__ __launch_bounds__(256) void ht_scan_chunks_kernel(int32_t* __restrict__ counts, int n_chunks, int n_tiles, int t_first, int32_t* __restrict__ totals) {
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

__global__ __launch_bounds__(1024) void ht_scan_tiles_kernel(int32_t* __restrict__ tile_base, int n_tiles, int t_first) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int t = tid; t < t_first; t += 1024) tile_base[t] = 0;  // tiles of the coarse (atomic) levels hold no records
  const int per = (n_tiles - t_first + 1023) / 1024;
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
  }
}

// MODE 0: dense gradient += tile; MODE 1: Adam for the tile's rows
template <int F, int MODE>
__global__ __launch_bounds__(HT_NT) void ht_tiles_kernel(HtArgs a) {
  extern __shared__ float ht_acc[];
  const int tile = (int)blockIdx.x + a.tile0;
  int level = 0;
  while (level + 1 < a.d.L && tile >= a.pl.tile_start[level + 1]) ++level;
  const HtLevel lv = ht_level(a.d, level);
  const bool coarse = level < a.pl.first_tiled_level;  // its gradient came through the atomic kernel into gtable
  const int sh = a.pl.tile_rows_log2;
  const uint32_t row0 = (uint32_t)(tile - a.pl.tile_start[level]) << sh;
  const uint32_t nrows = (lv.rows - row0) < (1u << sh) ? (lv.rows - row0) : (1u << sh);
  const int64_t gb = ((int64_t)lv.off0 + row0) * F;   // a multiple of 4: levels and tiles are multiples of 8 rows
  const int nq = (int)(nrows * F) >> 2;
  const int rec0 = a.tile_base[tile], rec1 = a.tile_base[tile + 1];
  int i_next = rec0 + (int)threadIdx.x;
  uint32_t rec_next = i_next < rec1 ? a.records[i_next] : 0u;
  for (int q = threadIdx.x; q < nq; q += HT_NT) *reinterpret_cast<float4*>(ht_acc + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  lds_barrier();
  const int gstride = a.d.L * F;
  for (int i = i_next; i < rec1; i += HT_NT) {
    const uint32_t rec = rec_next;
    if (i + HT_NT < rec1) rec_next = a.records[i + HT_NT];
    const int64_t b = (int64_t)(rec >> 4);
    const int yz = (int)(rec >> 2) & 3, xm = (int)(rec & 3u);
    const float x[3] = {a.x[b * 3], a.x[b * 3 + 1], a.x[b * 3 + 2]};
    uint32_t pg[3];
    float fr[3];
    ht_cell(lv, x, pg, fr);
    const float* gp = a.gout + b * gstride + level * F;
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = gp[f];
    const uint32_t cy = pg[1] + (uint32_t)(yz & 1), cz = pg[2] + (uint32_t)(yz >> 1);
#pragma unroll
    for (int xb = 0; xb < 2; ++xb) {
      if (!((xm >> xb) & 1)) continue;
      float w = 1.f;  // factors in axis order, as hashgrid_kernel multiplies them
      w *= xb ? fr[0] : 1.f - fr[0];
      w *= (yz & 1) ? fr[1] : 1.f - fr[1];
      w *= (yz >> 1) ? fr[2] : 1.f - fr[2];
      const uint32_t row = lv.row_of(pg[0] + (uint32_t)xb, cy, cz);
      float* rowp = ht_acc + (int)(row - row0) * F;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float val = w * g[f];
        if (val != 0.f) atomicAdd(rowp + f, val);
      }
    }
  }
  lds_barrier();
  const DynConsts dc = {a.step_size, a.inv_sqrt_bc2, 0};
  for (int q = threadIdx.x; q < nq; q += HT_NT) {
    const int64_t f0 = gb + 4 * (int64_t)q;
    const float4 gq = *reinterpret_cast<const float4*>(ht_acc + 4 * q);
    if (MODE == 0) {
      if (gq.x == 0.f && gq.y == 0.f && gq.z == 0.f && gq.w == 0.f) continue;
      float4 o = ld4(a.gtable + f0);
      o.x += gq.x; o.y += gq.y; o.z += gq.z; o.w += gq.w;
      *reinterpret_cast<float4*>(a.gtable + f0) = o;
    } else {
      float4 pp = ldnt4(a.p + f0), mm = ldnt4(a.m + f0), vv = ldnt4(a.v + f0);
      float4 extra = make_float4(0.f, 0.f, 0.f, 0.f);
      if (coarse && a.gtable) {
        extra = ldnt4(a.gtable + f0);
        if (extra.x != 0.f || extra.y != 0.f || extra.z != 0.f || extra.w != 0.f) stnt4(a.gtable + f0, make_float4(0.f, 0.f, 0.f, 0.f));
      }
      adam_float4(pp, mm, vv, gq, extra, 1.f, a.b1, a.b2, a.eps, dc);
      stnt4(a.p + f0, pp);
      stnt4(a.m + f0, mm);
      stnt4(a.v + f0, vv);
    }
  }
}

static int ht_lds_bytes(const snerf_hashgrid_desc* d, int sh) { return (1 << sh) * d->F * 4; }

static int ht_validate(const snerf_hashgrid_desc* d, const snerf_hashgrid_tile_plan* pl, int64_t B) {
  SNERF_REQUIRE(d && pl, "hashgrid tiles: null descriptor");
  SNERF_REQUIRE(d->D == 3, "hashgrid tiles: D=%d (3 only)", d->D);
  SNERF_REQUIRE(d->F == 1 || d->F == 2 || d->F == 4 || d->F == 8, "hashgrid tiles: F=%d", d->F);
  SNERF_REQUIRE(d->L >= 1 && d->L <= 32 && d->offsets[d->L] > 0, "hashgrid tiles: descriptor not laid out");
  SNERF_REQUIRE(B >= 0 && B < (1LL << 28), "hashgrid tiles: B=%lld (< 2^28)", (long long)B);
  SNERF_REQUIRE(pl->tile_rows_log2 >= 3 && pl->tile_rows_log2 <= 16 && pl->n_tiles == pl->tile_start[d->L] && pl->chunk >= 1 &&
                    pl->n_chunks == (int)((B + pl->chunk - 1) / pl->chunk) && pl->first_tiled_level >= 0 && pl->first_tiled_level <= d->L,
                "hashgrid tiles: the plan does not belong to this descriptor / batch (snerf_hashgrid_tile_plan_make)");
  return 0;
}

template <int F>
static int ht_bin_launch(const HtArgs& a, hipStream_t st) {
  const int L = a.d.L, Lc = a.pl.first_tiled_level;
  if (Lc >= L || a.B == 0) {
    hipLaunchKernelGGL(ht_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, a.tile_base, a.pl.n_tiles, a.pl.n_tiles);
    SNERF_LAUNCH_CHECK("hashgrid_bwd_bin (empty)");
    return 0;
  }
  int max_nt = 0;
  for (int l = Lc; l < L; ++l) max_nt = a.pl.tile_start[l + 1] - a.pl.tile_start[l] > max_nt ? a.pl.tile_start[l + 1] - a.pl.tile_start[l] : max_nt;
  const size_t lds = (size_t)max_nt * 2 * sizeof(int);
  const dim3 grid((unsigned)a.pl.n_chunks, (unsigned)(L - Lc));
  const int t_first = a.pl.tile_start[Lc];
  hipLaunchKernelGGL((ht_bin_kernel<F, false>), grid, dim3(HT_BIN_NT), lds, st, a);
  hipLaunchKernelGGL(ht_scan_chunks_kernel, dim3((unsigned)ceil_div(a.pl.n_tiles - t_first, 256)), dim3(256), 0, st, a.counts, a.pl.n_chunks, a.pl.n_tiles, t_first,
                     a.tile_base);
  hipLaunchKernelGGL(ht_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, a.tile_base, a.pl.n_tiles, t_first);
  hipLaunchKernelGGL((ht_bin_kernel<F, true>), grid, dim3(HT_BIN_NT), lds, st, a);
  SNERF_LAUNCH_CHECK("hashgrid_bwd_bin");
  return 0;
}

template <int F, int MODE>
static int ht_tiles_launch(HtArgs& a, hipStream_t st) {
  const int lds = ht_lds_bytes(&a.d, a.pl.tile_rows_log2);
  SNERF_ALLOW_LDS((ht_tiles_kernel<F, MODE>), lds);
  a.tile0 = MODE == 0 ? a.pl.tile_start[a.pl.first_tiled_level] : 0;  // MODE 0: tiles of the coarse levels hold no records
  const int n = a.pl.n_tiles - a.tile0;
  if (n <= 0) return 0;
  hipLaunchKernelGGL((ht_tiles_kernel<F, MODE>), dim3((unsigned)n), dim3(HT_NT), (size_t)lds, st, a);
  SNERF_LAUNCH_CHECK(MODE == 0 ? "hashgrid_bwd_tiles" : "hashgrid_bwd_tiles_adam");
  return 0;
}

#define HT_DISPATCH_F(F_, CALL) \
  switch (F_) {                 \
    case 1: return CALL(1);     \
    case 2: return CALL(2);     \
    case 4: return CALL(4);     \
    default: return CALL(8);    \
  }

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_hashgrid_tile_plan_make(const snerf_hashgrid_desc* desc, int64_t B, int32_t tile_rows_log2, int32_t first_tiled_level,
                                             snerf_hashgrid_tile_plan* plan) {
  SNERF_REQUIRE(desc && plan, "hashgrid_tile_plan_make: null argument");
  SNERF_REQUIRE(desc->L >= 1 && desc->L <= 32 && desc->offsets[desc->L] > 0 && B >= 0, "hashgrid_tile_plan_make: descriptor not laid out / B=%lld", (long long)B);
  int64_t max_rows = 0;
  for (int l = 0; l < desc->L; ++l) max_rows = desc->offsets[l + 1] - desc->offsets[l] > max_rows ? desc->offsets[l + 1] - desc->offsets[l] : max_rows;
  int sh = tile_rows_log2;
  if (sh <= 0) {
    // 2^11 rows: a level of 2^19 rows is 256 tiles, so that a batch's ~4.5 records per (point, level) spread over enough workgroups (a tile of 2^13 rows
    // of the NeRFPlayer preset receives 27 k records: 54 dependent rounds of its 512 threads)
    sh = 3;
    while (sh < 11 && ht_lds_bytes(desc, sh + 1) <= 64 * 1024) ++sh;
  }
  while (sh < 16 && ((max_rows + (1LL << sh) - 1) >> sh) > HT_MAX_LEVEL_TILES) ++sh;
  SNERF_REQUIRE(sh >= 3 && sh <= 16 && ht_lds_bytes(desc, sh) <= 156 * 1024, "hashgrid_tile_plan_make: a tile of 2^%d rows x %d features does not fit LDS", sh, desc->F);
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
    // levels with fewer than 2^18 rows stay with the atomic kernel: every point of the batch lands in their few tiles (level 0 of the preset: 4096 rows)
    lc = 0;
    while (lc < desc->L && desc->offsets[lc + 1] - desc->offsets[lc] < (1 << 18)) ++lc;
  }
  plan->first_tiled_level = lc > desc->L ? desc->L : lc;
  plan->lds_bytes = ht_lds_bytes(desc, sh);
  plan->count_ints = (int64_t)(plan->n_chunks > 0 ? plan->n_chunks : 1) * plan->n_tiles;
  plan->record_capacity = B * (desc->L - plan->first_tiled_level) * 8;
  return 0;
}

extern "C" int snerf_hashgrid_bwd_bin(const snerf_hashgrid_desc* desc, const snerf_hashgrid_tile_plan* plan, const float* x, int64_t B, const float* grad_out,
                                      int32_t* counts, int32_t* tile_base, uint32_t* records, snerf_stream_t stream) {
  int rc = ht_validate(desc, plan, B);
  if (rc) return rc;
  SNERF_REQUIRE(counts && tile_base && (records || B == 0) && (x || B == 0), "hashgrid_bwd_bin: null buffer");
  HtArgs a = {};
  a.d = *desc; a.pl = *plan; a.x = x; a.B = B; a.gout = grad_out; a.counts = counts; a.tile_base = tile_base; a.records = records;
#define HT_CALL(F_) ht_bin_launch<F_>(a, (hipStream_t)stream)
  HT_DISPATCH_F(desc->F, HT_CALL)
#undef HT_CALL
}

extern "C" int snerf_hashgrid_bwd_tiles(const snerf_hashgrid_desc* desc, const snerf_hashgrid_tile_plan* plan, const float* x, int64_t B, const float* grad_out,
                                        const int32_t* tile_base, const uint32_t* records, float* grad_table, snerf_stream_t stream) {
  int rc = ht_validate(desc, plan, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(tile_base && records && x && grad_out && grad_table && ((uintptr_t)grad_table & 15) == 0, "hashgrid_bwd_tiles: null / misaligned buffer");
  HtArgs a = {};
  a.d = *desc; a.pl = *plan; a.x = x; a.B = B; a.gout = grad_out; a.tile_base = const_cast<int32_t*>(tile_base); a.records = const_cast<uint32_t*>(records);
  a.gtable = grad_table;
#define HT_CALL(F_) ht_tiles_launch<F_, 0>(a, (hipStream_t)stream)
  HT_DISPATCH_F(desc->F, HT_CALL)
#undef HT_CALL
}

extern "C" int snerf_hashgrid_bwd_tiles_adam(const snerf_hashgrid_desc* desc, const snerf_hashgrid_tile_plan* plan, const float* x, int64_t B, const float* grad_out,
                                             const int32_t* tile_base, const uint32_t* records, float* grad_table, float* p, float* m, float* v, float lr,
                                             float beta1, float beta2, float eps, int32_t step, snerf_stream_t stream) {
  int rc = ht_validate(desc, plan, B);
  if (rc) return rc;
  SNERF_REQUIRE(tile_base && (records || B == 0) && (x || B == 0) && (grad_out || B == 0) && p && m && v, "hashgrid_bwd_tiles_adam: null buffer");
  SNERF_REQUIRE(step >= 1 && (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad_table) & 15) == 0, "hashgrid_bwd_tiles_adam: step=%d (1-based) / 16-byte alignment", step);
  SNERF_REQUIRE(plan->first_tiled_level == 0 || grad_table, "hashgrid_bwd_tiles_adam: levels [0, %d) go through the atomic kernel: pass their gradient buffer", plan->first_tiled_level);
  HtArgs a = {};
  a.d = *desc; a.pl = *plan; a.x = x; a.B = B; a.gout = grad_out; a.tile_base = const_cast<int32_t*>(tile_base); a.records = const_cast<uint32_t*>(records);
  a.gtable = grad_table; a.p = p; a.m = m; a.v = v; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  adam_consts(lr, beta1, beta2, step, a.step_size, a.inv_sqrt_bc2);
#define HT_CALL(F_) ht_tiles_launch<F_, 1>(a, (hipStream_t)stream)
  HT_DISPATCH_F(desc->F, HT_CALL)
#undef HT_CALL
}

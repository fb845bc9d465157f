// Static multiresolution hash grid (the full NeRFPlayer's "stationary field"): forward gather, backward atomic scatter into the
// table and -- because NeRFPlayer evaluates it at DEFORMED positions -- the gradient w.r.t. the coordinates.
//
// gfx950 equivalent of tcnn.Encoding(3, {"otype": "HashGrid", n_levels, n_features_per_level, log2_hashmap_size, base_resolution,
// per_level_scale}) at NS/fields/nerfplayer_field.py:242-252 (called :341-342).  tiny-cuda-nn v1.6 is a third-party dependency of the
// reference (Dockerfile:121), not vendored: this follows the published algorithm (Mueller et al. 2022 §3; grid.h: grid_scale,
// grid_resolution, pos_fract, grid_index, coherent_prime_hash), restated on the CPU in oracle/hashgrid_oracle.py.
//
// Layout: table [rows_total][F] fp32, levels back to back at desc.offsets; out [B][L*F] level-major (tcnn's to-row-major output).
// level = blockIdx.y: all lanes of a launch row work on one level's table (<= 4 MB at 2^19 x 2 floats), which stays in the XCD L2s
// while that level is processed.  Forward: one lane per (sample, level), a corner is one F*4-byte read (8 B at F = 2).  Backward: 2F
// adjacent lanes per (sample, level) (see the kernel).
// The level geometry (scale, resolution, rows) is computed ONCE on the host by snerf_hashgrid_layout and carried in the descriptor,
// so host, kernel and oracle agree on it bit for bit.
#include <math.h>

#include "common.hpp"

namespace snerf {

struct HgArgs {
  snerf_hashgrid_desc d;
  const float* x;       // [B, D]
  int64_t B;
  const float* table;
  float* out;           // fwd [B, L*F]
  const float* gout;    // bwd [B, L*F]
  float* gtable;        // bwd, may be null
  float* gx;            // bwd, may be null: [B, D], accumulated over levels
  long long* gtable_fx; // deterministic mode: the same two accumulators as 2^50-scaled 64-bit cells (common.hpp: fx_atomic_add)
  long long* gx_fx;
  int level0, level1;   // levels [level0, level1) of this launch (level1 = 0: all; snerf_hashgrid_encode_bwd_levels)
};

template <int F, bool BWD, int HG_CB>
__global__ __launch_bounds__(256) void hashgrid_kernel(HgArgs a) {
  const int D = a.d.D;
  // forward: one lane per (sample, level), F features in registers.
  // backward: 2F adjacent lanes per (sample, level) = (x-corner, feature).  The scatter is bound by atomic REQUESTS (~20 G/s chip-wide;
  // profiles/r01_kernels.md), not by the duplicated index arithmetic: a corner's F features leave the wave as one request (adjacent
  // addresses), and so do the two x-corners whenever their rows are neighbours -- always on dense levels (row = x + ...), and on hashed
  // levels when x is even (prime_x = 1: (x+1) ^ h = (x ^ h) ^ 1).
  constexpr int LPS = BWD ? (1 << HG_CB) * F : 1;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = gid / LPS;
  const int li = (int)(gid - b * LPS);
  const int fl = li % F;   // this lane's feature (backward)
  const int xb = li / F;   // this lane's low corner bits (backward): x first
  const int level = blockIdx.y + a.level0;
  if (b >= a.B) return;  // B * F is a multiple of F: the F lanes of a sample leave together
  const float scale = a.d.scale[level];
  const uint32_t resolution = (uint32_t)a.d.resolution[level];
  const uint32_t off0 = (uint32_t)a.d.offsets[level];
  const uint32_t rows = (uint32_t)(a.d.offsets[level + 1] - a.d.offsets[level]);

  float pos[3];
  uint32_t pg[3];
  for (int d = 0; d < D; ++d) {
    const float p = fmaf(scale, a.x[b * D + d], 0.5f);  // pos_fract
    const float f = floorf(p);
    pg[d] = (uint32_t)(int)f;
    pos[d] = p - f;
  }
  // grid_index: per-axis terms shared by the 2^D corners (dense stride or hash prime), XOR / sum per corner, one reduction mod rows
  const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
  uint32_t term[3][2];
  bool hashed;
  {
    uint64_t stride = 1;
    for (int d = 0; d < D && stride <= rows; ++d) stride *= resolution;
    hashed = rows < stride;
    uint64_t st = 1;
    for (int d = 0; d < D; ++d) {
      const uint32_t m = hashed ? primes[d] : (st <= rows ? (uint32_t)st : 0u);
      term[d][0] = pg[d] * m;
      term[d][1] = (pg[d] + 1u) * m;
      if (st <= rows) st *= resolution;
    }
  }
  const bool pow2 = (rows & (rows - 1u)) == 0u;

  float g = 0.f;
  if (BWD) g = a.gout[b * (a.d.L * F) + level * F + fl];
  float acc[F] = {};
  float gxd[3] = {0.f, 0.f, 0.f};
  const int cb = BWD ? (HG_CB < D ? HG_CB : D) : 0;  // corner bits spread over lanes
  const int ncorner = 1 << (D - cb);
  if (BWD && xb >= (1 << cb)) g = 0.f;  // D < HG_CB: surplus lanes contribute nothing (they stay for the lane exchanges below)
  for (int sub = 0; sub < ncorner; ++sub) {
    const int idx = (sub << cb) | (xb & ((1 << cb) - 1));
    float w = 1.f;
    uint32_t index = 0;
    for (int d = 0; d < D; ++d) {
      const int bit = (idx >> d) & 1;
      w *= bit ? pos[d] : 1.f - pos[d];
      index = hashed ? (index ^ term[d][bit]) : (index + term[d][bit]);
    }
    const uint32_t row = pow2 ? (index & (rows - 1u)) : (index % rows);
    const size_t e = ((size_t)off0 + row) * F;
    if (!BWD) {
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] += w * a.table[e + f];
    } else {
      if (a.gtable || a.gtable_fx) {
        const float v = w * g;
        if (v != 0.f) {
          if (a.gtable_fx) fx_atomic_add(a.gtable_fx + e + fl, v); else atomicAdd(a.gtable + e + fl, v);
        }
      }
      if (a.gx || a.gx_fx) {
        // dy/dx_d = scale * sum over corners of (+-1 along d) * prod_{e != d} w_e * value (grid.h dy_dx, linear interpolation)
        float dot = a.table[e + fl] * g;
#pragma unroll
        for (int o = 1; o < F; o <<= 1) dot += __shfl_xor(dot, o, 64);  // over the sample's F lanes
        for (int d = 0; d < D; ++d) {
          float wo = 1.f;
          for (int q = 0; q < D; ++q)
            if (q != d) wo *= ((idx >> q) & 1) ? pos[q] : 1.f - pos[q];
          gxd[d] += (((idx >> d) & 1) ? wo : -wo) * dot;
        }
      }
    }
  }
  if (!BWD) {
#pragma unroll
    for (int f = 0; f < F; ++f) a.out[b * (a.d.L * F) + level * F + f] = acc[f];
  } else if (a.gx || a.gx_fx) {
    for (int o = F; o < LPS; o <<= 1)
      for (int d = 0; d < D; ++d) gxd[d] += __shfl_xor(gxd[d], o, 64);  // the other corners' lanes
    if (li == 0) {
      for (int d = 0; d < D; ++d) {
        const float v = gxd[d] * scale;
        if (v != 0.f) {
          if (a.gx_fx) fx_atomic_add(a.gx_fx + b * D + d, v); else atomicAdd(a.gx + b * D + d, v);  // one add per level: the levels are separate workgroups
        }
      }
    }
  }
}

static int validate(const snerf_hashgrid_desc* d, int64_t B) {
  SNERF_REQUIRE(d, "hashgrid: null descriptor");
  SNERF_REQUIRE(d->D >= 1 && d->D <= 3, "hashgrid: D=%d unsupported (1..3)", d->D);
  SNERF_REQUIRE(d->F == 1 || d->F == 2 || d->F == 4 || d->F == 8, "hashgrid: n_features_per_level=%d unsupported (1,2,4,8)", d->F);
  SNERF_REQUIRE(d->L >= 1 && d->L <= 32, "hashgrid: n_levels=%d (<= 32)", d->L);
  SNERF_REQUIRE(d->offsets[d->L] > 0, "hashgrid: descriptor not laid out (call snerf_hashgrid_layout)");
  SNERF_REQUIRE(B >= 0, "hashgrid: B=%lld", (long long)B);
  return 0;
}

template <bool BWD, int CB>
static int launch_cb(const HgArgs& a, hipStream_t st) {
  dim3 grid((unsigned)ceil_div(a.B * (BWD ? (1 << CB) * a.d.F : 1), 256), (unsigned)(a.level1 > 0 ? a.level1 - a.level0 : a.d.L));
  switch (a.d.F) {
    case 1: hipLaunchKernelGGL((hashgrid_kernel<1, BWD, CB>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((hashgrid_kernel<2, BWD, CB>), grid, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((hashgrid_kernel<4, BWD, CB>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((hashgrid_kernel<8, BWD, CB>), grid, dim3(256), 0, st, a); break;
  }
  SNERF_LAUNCH_CHECK(BWD ? "hashgrid_encode_bwd" : "hashgrid_encode_fwd");
  return 0;
}

template <bool BWD>
static int launch(const HgArgs& a, hipStream_t st) {
  // backward: the two x-corners on adjacent lanes; spreading the y / z corners over lanes as well measured no faster (0.70 / 0.71 ms
  // against 0.69 for the table scatter, and a slower coordinate gradient) -- the request rate is the bound by then
  return BWD ? launch_cb<BWD, 1>(a, st) : launch_cb<false, 0>(a, st);
}

}  // namespace snerf

using namespace snerf;

// Host only: the encoding's constructor arithmetic (level scale / resolution / offset table).  Returns the total row count.
extern "C" int64_t snerf_hashgrid_layout(snerf_hashgrid_desc* d, int32_t base_resolution, float per_level_scale, int32_t log2_hashmap_size) {
  if (!d || d->D < 1 || d->D > 3 || d->L < 1 || d->L > 32 || base_resolution < 1 || !(per_level_scale > 0.f) || log2_hashmap_size < 3 ||
      log2_hashmap_size > 30) {
    set_error("hashgrid_layout: D=%d L=%d base_resolution=%d per_level_scale=%g log2_hashmap_size=%d", d ? d->D : -1, d ? d->L : -1, base_resolution,
              (double)per_level_scale, log2_hashmap_size);
    return -1;
  }
  const float log2_pls = log2f(per_level_scale);
  int64_t offset = 0;
  for (int l = 0; l < d->L; ++l) {
    const float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;  // grid_scale
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;                                // grid_resolution
    const uint32_t max_params = 0xFFFFFFFFu / 2;
    uint64_t params = 1;
    for (int k = 0; k < d->D; ++k) { params *= res; if (params > max_params) { params = max_params; break; } }
    params = (params + 7) / 8 * 8;                                                   // aligned levels
    const uint64_t cap = 1ull << log2_hashmap_size;
    if (params > cap) params = cap;
    d->scale[l] = scale;
    d->resolution[l] = (int32_t)res;
    d->offsets[l] = (int32_t)offset;
    offset += (int64_t)params;
    if (offset > 0x7FFFFFFF) { set_error("hashgrid_layout: table exceeds 2^31 rows"); return -1; }
  }
  d->offsets[d->L] = (int32_t)offset;
  return offset;
}

extern "C" int snerf_hashgrid_encode_fwd(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, float* out,
                                         snerf_stream_t stream) {
  int rc = validate(desc, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(table && x && out, "hashgrid_encode_fwd: null buffer");
  HgArgs a = {};
  a.d = *desc; a.x = x; a.B = B; a.table = table; a.out = out;
  return launch<false>(a, (hipStream_t)stream);
}

extern "C" int snerf_hashgrid_encode_bwd(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, const float* grad_out,
                                         float* grad_table, float* grad_x, snerf_stream_t stream) {
  int rc = validate(desc, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(x && grad_out && (grad_table || grad_x), "hashgrid_encode_bwd: null buffer");
  SNERF_REQUIRE(!grad_x || table, "hashgrid_encode_bwd: the coordinate gradient needs the table");
  HgArgs a = {};
  a.d = *desc; a.x = x; a.B = B; a.table = table; a.gout = grad_out; a.gtable = grad_table; a.gx = grad_x;
  return launch<true>(a, (hipStream_t)stream);
}

// snerf.h (ABI 14): the same for levels [level_begin, level_end) only (the coarse levels beside the tiled form, csrc/hashgrid_tiles.hip)
extern "C" int snerf_hashgrid_encode_bwd_levels(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, const float* grad_out,
                                                float* grad_table, float* grad_x, int32_t level_begin, int32_t level_end, snerf_stream_t stream) {
  int rc = validate(desc, B);
  if (rc) return rc;
  SNERF_REQUIRE(level_begin >= 0 && level_begin <= level_end && level_end <= desc->L, "hashgrid_encode_bwd_levels: levels [%d, %d) of %d", level_begin, level_end, desc->L);
  if (B == 0 || level_begin == level_end) return 0;
  SNERF_REQUIRE(x && grad_out && (grad_table || grad_x), "hashgrid_encode_bwd_levels: null buffer");
  SNERF_REQUIRE(!grad_x || table, "hashgrid_encode_bwd_levels: the coordinate gradient needs the table");
  HgArgs a = {};
  a.d = *desc; a.x = x; a.B = B; a.table = table; a.gout = grad_out; a.gtable = grad_table; a.gx = grad_x;
  a.level0 = level_begin; a.level1 = level_end;
  return launch<true>(a, (hipStream_t)stream);
}

// snerf.h (ABI 13): both accumulators as fixed-point cells
extern "C" int snerf_hashgrid_encode_bwd_fx(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, const float* grad_out,
                                            int64_t* grad_table_fx, int64_t* grad_x_fx, snerf_stream_t stream) {
  int rc = validate(desc, B);
  if (rc) return rc;
  if (B == 0) return 0;
  SNERF_REQUIRE(x && grad_out && (grad_table_fx || grad_x_fx), "hashgrid_encode_bwd_fx: null buffer");
  SNERF_REQUIRE(!grad_x_fx || table, "hashgrid_encode_bwd_fx: the coordinate gradient needs the table");
  HgArgs a = {};
  a.d = *desc; a.x = x; a.B = B; a.table = table; a.gout = grad_out;
  a.gtable_fx = reinterpret_cast<long long*>(grad_table_fx); a.gx_fx = reinterpret_cast<long long*>(grad_x_fx);
  return launch<true>(a, (hipStream_t)stream);
}

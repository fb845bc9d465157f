// K-Planes plane-gradient scatter, sorted variant: ~6x fewer atomic requests than the sample-major scatter.
//
// Why: fp32 atomics on gfx950 cost ~one memory-side 64-B request each (MI355X_MICROARCH.md "Global float atomics"), and the
// sample-major scatter (kplanes.hip) can only merge contributions of CONSECUTIVE samples of one ray.  Counted on real
// training batches (tools/scatter_stats.py, 262 144 samples, 30 planes): 6.8 M (plane,row) flushes early in training, 3.7 M
// after 3 k steps -- but only 0.7-0.9 M DISTINCT (plane, row, x0) keys: the three time planes have just n_times x W
// texels, and coarse spatial planes a few thousand.  Sorting the samples of every (scale, plane) segment by texel key
// makes equal keys adjacent, so run-length combining removes ~85 % of the requests.
//
// ONE sort per plane serves every scale: the key is the Z-order (Morton) code of the sample's texel at the FINEST scale, so
// the samples of any coarser texel (a ~power-of-two block of fine texels) are (almost always) contiguous too.
//
// Pipeline (NP plane segments, each with exactly N entries):
//   K1 rank    : one lane per (sample, plane): key = cell(plane) + morton(x0, row0 at the finest scale);
//                rank = atomicAdd(&hist[key], 1)   (int atomics)
//   K2 scan    : exclusive prefix sum of hist (3 small kernels)
//   K3 reorder : sorted_rec[scan[key] + rank] = {n, coord_a, coord_b} (sample id + its two normalised coordinates on that
//                plane) -- a counting sort; positions depend only on the sample coordinates, so K1-K3 run on a side stream
//                under the forward / MLP backward, and pass B never has to touch the ray buffers again
//   A  gradvec : sample-major, float4 per lane (like the forward gather): g_q = dL/d(interp of plane q) -> gvec[seg][n][C]
//   B  scatter : per segment, lane groups of 2*C lanes (x-corner, channel) walk RUN consecutive SORTED entries, multiply by
//                the bilinear weights, run-length-combine per row and flush with one 256-B atomic instruction per run.
#include <stdlib.h>

#include "kplanes_sort_common.hpp"

namespace snerf {

template <int NP>
__device__ __forceinline__ void seg_axes(int q, int& a, int& b) {
  constexpr auto& A = PlanePairs<NP>::a;
  constexpr auto& B = PlanePairs<NP>::b;
  a = A[q]; b = B[q];
}

// ---- K1 / K3: one lane per sample, loops over the segments ----
// pixel coordinate of axis_tap (the clipped, un-normalised grid_sample coordinate); floor of it is the tap's i0
__device__ __forceinline__ float axis_pix(float x, int size) {
  float fx = ((x + 1.f) / 2.f) * (float)(size - 1);
  return fminf((float)(size - 1), fmaxf(fx, 0.f));
}
__device__ __forceinline__ AxisTap tap_from_pix(float fx, int size) {
  float f0 = floorf(fx);
  AxisTap t;
  t.i0 = (int)f0;
  t.w0 = (f0 + 1.f) - fx;
  t.w1 = fx - f0;
  bool in = (t.i0 + 1) <= (size - 1);
  t.i1 = in ? t.i0 + 1 : t.i0;
  if (!in) t.w1 = 0.f;
  return t;
}

template <int NP, bool REORDER>
__global__ __launch_bounds__(256) void sort_keys_kernel(snerf_coords c, SegTable st, int64_t N, int32_t* __restrict__ hist,
                                                       int32_t* __restrict__ rank, const int32_t* __restrict__ scan, float4* __restrict__ sorted_rec) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float p[4];
  load_coords<NP>(c, n, p);
  constexpr auto& A = PlanePairs<NP>::a;
  constexpr auto& B = PlanePairs<NP>::b;
  if (st.per_scale) {
    for (int s = 0; s < st.n_scales; ++s) {
      int i0[4], rs[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rs[k] = st.res[s][k] > 0 ? st.res[s][k] : 1;
        i0[k] = (int)floorf(axis_pix(p[k], rs[k]));  // == axis_tap(p, res).i0: the run key pass B compares
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int seg = s * NP + q;
        const int key = st.cell_off[seg] + (st.row_major[q] ? i0[B[q]] * rs[A[q]] + i0[A[q]] : (int)morton2((uint32_t)i0[A[q]], (uint32_t)i0[B[q]]));
        if (!REORDER) rank[(int64_t)seg * N + n] = atomicAdd(hist + key, 1);
        else sorted_rec[scan[key] + rank[(int64_t)seg * N + n]] = make_float4(__int_as_float((int)n), p[A[q]], p[B[q]], 0.f);
      }
    }
    return;
  }
  int i0[4], i0f[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    i0[k] = (int)floorf(axis_pix(p[k], st.fine[k] > 0 ? st.fine[k] : 1));
    i0f[k] = (int)floorf(axis_pix(p[k], st.fine_rm[k] > 0 ? st.fine_rm[k] : 1));
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int key = st.cell_off[q] + (st.row_major[q] ? i0[B[q]] * st.fine_rm[A[q]] + i0f[A[q]] : (int)morton2((uint32_t)i0[A[q]], (uint32_t)i0[B[q]]));
    if (!REORDER) {
      rank[(int64_t)q * N + n] = atomicAdd(hist + key, 1);
    } else {
      // sorted record: sample id + its two normalised coordinates on this plane, so pass B never touches the ray buffers
      sorted_rec[scan[key] + rank[(int64_t)q * N + n]] = make_float4(__int_as_float((int)n), p[A[q]], p[B[q]], 0.f);
    }
  }
}

// ---- K2: exclusive scan of int32 data[n] in place (block = 1024 elements) ----
__global__ __launch_bounds__(256) void scan_block_kernel(int32_t* __restrict__ data, int64_t n, int32_t* __restrict__ block_sums) {
  __shared__ int32_t s_part[256];
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  int32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = base + k < n ? data[base + k] : 0;
  const int32_t tsum = v[0] + v[1] + v[2] + v[3];
  s_part[threadIdx.x] = tsum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele over the 256 thread sums
    int32_t t = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  int32_t run = s_part[threadIdx.x] - tsum;  // exclusive prefix of this thread inside the block
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) block_sums[blockIdx.x] = s_part[255];
}
__global__ void scan_sums_kernel(int32_t* __restrict__ block_sums, int nb) {  // single workgroup, sequential over chunks of 1024
  __shared__ int32_t s_part[1024];
  __shared__ int32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int32_t v = i < nb ? block_sums[i] : 0;
    s_part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int32_t t = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
      __syncthreads();
      s_part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) block_sums[i] = s_carry + s_part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry += s_part[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void scan_add_kernel(int32_t* __restrict__ data, int64_t n, const int32_t* __restrict__ block_sums) {
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  const int32_t add = block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n) data[base + k] += add;
}

// gvec rows are fp32 (exact: the parity path) or bf16 (half the 2 GB round trip between pass A and pass B; the scatter still accumulates
// in fp32, each contribution carries a 2^-9 relative rounding)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_gq(float* gv, int64_t idx, float4 g) { *reinterpret_cast<float4*>(gv + idx) = g; }
__device__ __forceinline__ void store_gq(__bf16* gv, int64_t idx, float4 g) {
  const bf16x4 b = {(__bf16)g.x, (__bf16)g.y, (__bf16)g.z, (__bf16)g.w};
  *reinterpret_cast<bf16x4*>(gv + idx) = b;
}
__device__ __forceinline__ float load_g(const float* gv, int64_t idx) { return gv[idx]; }
__device__ __forceinline__ float load_g(const __bf16* gv, int64_t idx) { return (float)gv[idx]; }

// ---- pass A: gradient w.r.t. each plane's interpolated value, sample-major ----
template <int C, int NP, typename GV>
__global__ __launch_bounds__(256) void gradvec_kernel(snerf_kplanes_desc d, const float* __restrict__ planes, snerf_coords c, int64_t N,
                                                     const float* __restrict__ gout, GV* __restrict__ gvec) {
  constexpr int LPS = C / 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = gid / LPS;
  const int cg = (int)(gid % LPS);
  if (n >= N) return;
  float p[4];
  load_coords<NP>(c, n, p);
  const int out_w = d.concat ? C * d.n_scales : C;
  const float* grow = gout + n * out_w + cg * 4;
  constexpr auto& A = PlanePairs<NP>::a;
  constexpr auto& B = PlanePairs<NP>::b;
  for (int s = 0; s < d.n_scales; ++s) {
    AxisTap tap[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tap[k] = axis_tap(p[k], d.res[s][k] > 0 ? d.res[s][k] : 1);
    float4 v[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) v[q] = plane_sample<C>(planes + d.off[s][q], d.res[s][A[q]], tap[A[q]], tap[B[q]], cg);
    const float4 g = *reinterpret_cast<const float4*>(grow + (d.concat ? s * C : 0));
    float4 suf[NP + 1];
    suf[NP] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int q = NP - 1; q >= 0; --q) suf[q] = f4_mul(suf[q + 1], v[q]);
    float4 pre = g;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const float4 gq = f4_mul(pre, suf[q + 1]);
      pre = f4_mul(pre, v[q]);
      store_gq(gvec, ((int64_t)(s * NP + q) * N + n) * C + cg * 4, gq);
    }
  }
}

// ---- pass B: sorted run-length scatter ----
// One lane group (2*C lanes = (x-corner, channel)) walks RUN consecutive SORTED records {n, fx, fy}.  Records are read
// UNROLL at a time (wave-uniform addresses), the UNROLL gvec rows are fetched together, then combined in order.
template <int C, int NP, typename GV>
__global__ __launch_bounds__(256) void scatter_sorted_kernel(snerf_kplanes_desc d, int64_t N, const GV* __restrict__ gvec,
                                                            const float4* __restrict__ sorted_rec, float* __restrict__ gplanes, int run,
                                                            int64_t groups_per_seg, int seg_begin, int per_scale) {
  constexpr int LPS = 2 * C;
  constexpr int UNROLL = 8;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // LPS == 64: one lane group = one wave, so the group index (and everything derived from it) is wave-uniform; readfirstlane
  // lets the compiler keep it in SGPRs and branch scalar
  const int64_t group = LPS == 64 ? (int64_t)blockIdx.x * (blockDim.x / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : gid / LPS;
  const int li = (int)(gid % LPS);
  const int half = li / C, ch = li % C;
  const int seg = seg_begin + (int)(group / groups_per_seg);
  if (seg >= d.n_scales * NP) return;
  const int64_t i0 = (group - (int64_t)(seg - seg_begin) * groups_per_seg) * run;
  if (i0 >= N) return;
  const int cnt = (int)((N - i0) < run ? (N - i0) : run);
  const int s = seg / NP, q = seg % NP;
  int a, b;
  seg_axes<NP>(q, a, b);
  const int W = d.res[s][a], H = d.res[s][b] > 0 ? d.res[s][b] : 1;
  const float4* rec = sorted_rec + (int64_t)(per_scale ? seg : q) * N + i0;  // the plane's order is shared by all scales unless per_scale
  const GV* gv = gvec + (int64_t)seg * N * C + ch;
  float* gbase = gplanes + d.off[s][q] + li;
  int pend_key[2] = {-1, -1};
  float pend_val[2] = {0.f, 0.f};
  if (LPS == 64) {
    // The per-entry work (two bilinear taps, keys) is the same for all 64 lanes: do it LANE-PARALLEL for 64 entries at a time
    // (lane l handles entry base + l), then walk the 64 entries in order broadcasting the precomputed values with
    // v_readlane -- ~3x fewer vector instructions per entry than recomputing the taps in every lane.
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < cnt; base += 64) {
      const int e = base + lane;
      const float4 r = rec[e < cnt ? e : cnt - 1];
      const AxisTap tx = axis_tap(r.y, W);
      const AxisTap ty = axis_tap(r.z, H);
      const int nn = __float_as_int(r.x);
      const int k0 = ty.i0 * W + tx.i0, k1 = ty.i1 * W + tx.i0;
      const float wxa = tx.w0, wxb = tx.w1, wya = ty.w0, wyb = ty.w1;
      const int m = (cnt - base) < 64 ? (cnt - base) : 64;
      for (int u0 = 0; u0 < m; u0 += UNROLL) {
        float g[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          const int uu = (u0 + u) < m ? (u0 + u) : (m - 1);
          g[u] = load_g(gv, (int64_t)__builtin_amdgcn_readlane(nn, uu) * C);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          const int uu = u0 + u;
          if (uu < m) {
            const float wx0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wxa), uu));
            const float wx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wxb), uu));
            const float wx = half ? wx1 : wx0;
            const float gx = g[u] * wx;
            const int key[2] = {__builtin_amdgcn_readlane(k0, uu), __builtin_amdgcn_readlane(k1, uu)};
            const float wy[2] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wya), uu)),
                                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wyb), uu))};
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const float val = gx * wy[rr];
              if (key[rr] != pend_key[rr]) {
                if (pend_val[rr] != 0.f) atomicAdd(gbase + (int64_t)pend_key[rr] * C, pend_val[rr]);
                pend_key[rr] = key[rr];
                pend_val[rr] = val;
              } else {
                pend_val[rr] += val;
              }
            }
          }
        }
      }
    }
  } else
  for (int i = 0; i < cnt; i += UNROLL) {
    float4 r[UNROLL];
    float g[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r[u] = rec[i + u < cnt ? i + u : cnt - 1];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) g[u] = load_g(gv, (int64_t)__float_as_int(r[u].x) * C);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (i + u < cnt) {
        const AxisTap tx = axis_tap(r[u].y, W);
        const AxisTap ty = axis_tap(r[u].z, H);
        const float gx = g[u] * (half ? tx.w1 : tx.w0);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int key = (rr ? ty.i1 : ty.i0) * W + tx.i0;
          const float val = gx * (rr ? ty.w1 : ty.w0);
          if (key != pend_key[rr]) {
            if (pend_val[rr] != 0.f) atomicAdd(gbase + (int64_t)pend_key[rr] * C, pend_val[rr]);
            pend_key[rr] = key;
            pend_val[rr] = val;
          } else {
            pend_val[rr] += val;
          }
        }
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
    if (pend_val[rr] != 0.f) atomicAdd(gbase + (int64_t)pend_key[rr] * C, pend_val[rr]);
}

// ---- pass B, C = 32: chunk-local regrouping by THIS scale's texel key, run-length combining, x-carry ----
// The shared Z-order of the finest scale keeps a chunk of 256 consecutive entries spatially compact at every scale, but the
// `res-1` scaling of align_corners=True does not nest coarse texel edges in fine ones, so a coarse cell's samples arrive in
// several runs (2.0 M flushes for 0.93 M distinct cells; a separate global sort per scale brings pass B from 0.81 to 0.53 ms but
// costs 0.5 ms more sorting).  Here every wave re-sorts ITS 256 entries by the segment's own key (row * W + x0) with an in-register
// bitonic network (36 rounds, ~550 instructions, no memory traffic) and walks them in that order: equal keys are adjacent, and
// inside a row x0 ascends, so for x-adjacent cells the shared texel (x0+1 of one cell = x0 of the next) is CARRIED in registers
// from half 1 to half 0 instead of being flushed twice.  What bounds this pass is the number of lane-level atomic operations
// (profiles/r01_kernels.md); regrouping + carry cut them by more than half.
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
  if (J >= 32) return (uint32_t)__shfl_xor((int)v, J, 64);
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (J << 10) | 0x1f);  // bit mode: lane ^ J inside each group of 32
}

template <int K, int J>
__device__ __forceinline__ void bitonic_round(uint32_t (&w)[4], int lane) {
  if (J >= 64) {  // partner is another register of the same lane
    constexpr int JR = J / 64;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if ((r & JR) == 0) {
        const bool up = (((r * 64) & K) == 0);  // K >= 128 here: direction depends on the register index only
        const uint32_t a = w[r], b = w[r | JR];
        const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
        w[r] = up ? lo : hi;
        w[r | JR] = up ? hi : lo;
      }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t other = lane_xor<J>(w[r]);
      const int e = r * 64 + lane;
      const bool up = (e & K) == 0, lower = (lane & J) == 0;
      const uint32_t lo = w[r] < other ? w[r] : other, hi = w[r] < other ? other : w[r];
      w[r] = (lower == up) ? lo : hi;
    }
  }
}

template <int K, int J>
struct BitonicJ {
  static __device__ __forceinline__ void run(uint32_t (&w)[4], int lane) {
    bitonic_round<K, J>(w, lane);
    BitonicJ<K, J / 2>::run(w, lane);
  }
};
template <int K>
struct BitonicJ<K, 0> {
  static __device__ __forceinline__ void run(uint32_t (&)[4], int) {}
};
template <int K>
struct BitonicK {
  static __device__ __forceinline__ void run(uint32_t (&w)[4], int lane) {
    BitonicK<K / 2>::run(w, lane);
    BitonicJ<K, K / 2>::run(w, lane);
  }
};
template <>
struct BitonicK<1> {
  static __device__ __forceinline__ void run(uint32_t (&)[4], int) {}
};

// QUOT (the quotient form, snerf_kplanes_scatter_quotient): gvec is ONE tensor G[N][row_stride] = gfeat .* feat (feat = the forward's product
// over the six planes), and the gradient vector of plane q at an entry is G / v_q with v_q re-interpolated here from the entry's cell --
// the 4 texels this lane group is about to add into, so the reads follow the sorted order and stay in cache.  lane = (x-corner, channel):
// each half interpolates its x-column, v_permlane32_swap adds the two halves.  v_q == 0 (then G == 0 as well: the information is gone)
// yields 0 here and the exact term is added by quotient_fixup_kernel.
template <int NP, typename GV, bool QUOT = false>
__global__ __launch_bounds__(256) void scatter_grouped_kernel(snerf_kplanes_desc d, int64_t N, const GV* __restrict__ gvec,
                                                             const float4* __restrict__ sorted_rec, float* __restrict__ gplanes,
                                                             int64_t groups_per_seg, int seg_begin, int per_scale,
                                                             const float* __restrict__ planes = nullptr, int row_stride = 0) {
  constexpr int C = 32, CH = 256, UNROLL = 16;
  // per wave, per entry (in WALK order) 8 dwords: {gvec row offset (elements), x0 | y0 << 16, -, -, wx0*wy0, wx0*wy1, wx1*wy0, wx1*wy1}.
  // Everything per-entry is prepared lane-parallel (4 entries per lane) so that the walk -- one entry per wave instruction -- costs
  // ~10 VALU instructions per entry: the first version of this kernel spent ~60 and was VALU-issue bound (0.79 ms with 43 % fewer
  // atomic instructions than the plain run-length kernel's 0.81 ms).
  __shared__ __align__(16) uint32_t s_rec[4][CH * 8];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int half = lane >> 5, ch = lane & 31;
  const int64_t group = (int64_t)blockIdx.x * 4 + wave;
  const int seg = seg_begin + (int)(group / groups_per_seg);
  if (seg >= d.n_scales * NP) return;
  const int64_t i0 = (group - (int64_t)(seg - seg_begin) * groups_per_seg) * CH;
  if (i0 >= N) return;
  const int cnt = (int)((N - i0) < CH ? (N - i0) : CH);
  const int s = seg / NP, q = seg % NP;
  int a, b;
  seg_axes<NP>(q, a, b);
  const int W = d.res[s][a], H = d.res[s][b] > 0 ? d.res[s][b] : 1;
  const float4* rec = sorted_rec + (int64_t)(per_scale ? seg : q) * N + i0;
  // wave-uniform base; rows are addressed with 32-bit element offsets
  const GV* gseg = QUOT ? gvec + (int64_t)s * C : gvec + (int64_t)seg * N * C;
  const uint32_t rstride = QUOT ? (uint32_t)row_stride : (uint32_t)C;
  float* gch = gplanes + d.off[s][q] + ch;
  const float* pch = QUOT ? planes + d.off[s][q] + ch : nullptr;
  uint32_t* R = s_rec[wave];
  const bool sortable = (int64_t)W * H <= (1 << 23);  // key << 8 | index must fit 31 bits (the launcher guarantees N * C < 2^31)

  // 1. sort words: (row * W + x0) << 8 | entry, entry e = r * 64 + lane
  uint32_t word[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = r * 64 + lane;
    word[r] = 0xffffffffu;
    if (e < cnt) {
      const float4 rc = rec[e];
      const int x0 = (int)floorf(axis_pix(rc.y, W)), y0 = (int)floorf(axis_pix(rc.z, H));  // == axis_tap(.).i0
      word[r] = sortable ? (((uint32_t)(y0 * W + x0) << 8) | (uint32_t)e) : (uint32_t)e;
    }
  }
  // 2. in-wave bitonic sort (ascending; 0xffffffff pads go last).  Not sortable (> 2^23 texels): the words are already ascending.
  if (sortable) BitonicK<CH>::run(word, lane);
  // 3. walk records, written at their position in the sorted order; positions >= cnt get a null record (zero weights)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = r * 64 + lane;
    uint4 hd = make_uint4(0u, 0xffffffffu, 0u, 0u);
    float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < cnt) {
      const float4 rc = rec[word[r] & 255u];
      const AxisTap tx = axis_tap(rc.y, W);
      const AxisTap ty = axis_tap(rc.z, H);
      hd.x = (uint32_t)__float_as_int(rc.x) * rstride;
      hd.y = (uint32_t)tx.i0 | ((uint32_t)ty.i0 << 16);
      if (QUOT) {
        hd.z = (uint32_t)(ty.i0 * W + tx.i0) * (uint32_t)C;
        hd.w = (tx.i1 != tx.i0 ? 1u : 0u) | (ty.i1 != ty.i0 ? 2u : 0u);
      }
      const float4 tw = tap_weights(tx, ty);                  // (x0y0, x1y0, x0y1, x1y1): the forward's products, bit for bit
      wt = make_float4(tw.x, tw.z, tw.y, tw.w);               // stored as (x0y0, x0y1 | x1y0, x1y1): one float2 per x-corner; a clamped second tap has weight 0
    }
    *reinterpret_cast<uint4*>(R + e * 8) = hd;
    *reinterpret_cast<float4*>(R + e * 8 + 4) = wt;
  }
  // the records were stored one lane per entry and are read back by every lane of the wave: order the LDS stores before the loads
  // explicitly (same wave, so no s_barrier -- but neither the compiler nor the LDS queue may move a load above these stores)
  wave_lds_publish();
  // 4. the walk.  One pending cell (x0 | y0 << 16 = ppk) with two accumulators per lane: rows y0 and y0 + 1; lane = (x-corner, channel).
  //    A row or column beyond the border only ever accumulates zeros and is never flushed (guards on != 0).
  uint32_t ppk = 0xfffffff0u;  // matches no record, nor record - 1
  float p0 = 0.f, p1 = 0.f;
  const uint32_t* Rl = R + 4 + half * 2;  // this lane's weight pair of an entry
  const int64_t rowC = (int64_t)W * C;
  const uint32_t rowC32 = (uint32_t)W * (uint32_t)C;
  for (int e0 = 0; e0 < cnt; e0 += UNROLL) {
    // UNROLL gvec rows in flight per wave (the pass streams 1 GB of them: with 8 in flight it was latency-bound, 0.61 vs 0.56 ms at 16);
    // the small per-entry fields are re-read from LDS when the entry is processed, so only g[] stays in registers
    float g[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) g[u] = load_g(gseg, (int64_t)(R[(e0 + u) * 8] + ch));  // wave-uniform LDS address: broadcast
    float t0[QUOT ? UNROLL : 1], t1[QUOT ? UNROLL : 1];  // QUOT: this lane's x-column of the entry's cell, rows y0 and y0 + 1
    if constexpr (QUOT) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        // element offset of texel (x0, y0) and the "x0 + 1 / y0 + 1 exist" flags were prepared lane-parallel with the record (a clamped
        // corner has weight 0 and re-reads the unclamped texel, as the forward's taps do; a null record reads texel 0 with zero weights)
        const uint32_t toff = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[(e0 + u) * 8 + 2]);
        const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[(e0 + u) * 8 + 3]);
        const uint32_t o0 = toff + (half ? (fl & 1u) * (uint32_t)C : 0u);
        t0[u] = pch[o0];
        t1[u] = pch[o0 + ((fl & 2u) ? rowC32 : 0u)];
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t packed = (uint32_t)__builtin_amdgcn_readfirstlane((int)R[(e0 + u) * 8 + 1]);
      const float2 wt = *reinterpret_cast<const float2*>(Rl + (e0 + u) * 8);
      float gq = g[u];
      if constexpr (QUOT) {
        // the plane's value at the entry, with the forward's own formula and order (bilerp4): both x-columns in every lane
        const float4 w4 = *reinterpret_cast<const float4*>(R + (e0 + u) * 8 + 4);  // wave-uniform address: broadcast
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t0[u]), __float_as_uint(t0[u]), false, false);  // row y0: [0] = x0 column, [1] = x1
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t1[u]), __float_as_uint(t1[u]), false, false);  // row y0 + 1
        const float vq = bilerp4(__uint_as_float(s0[0]), __uint_as_float(s0[1]), __uint_as_float(s1[0]), __uint_as_float(s1[1]), w4.x, w4.z, w4.y, w4.w);
        gq = fabsf(vq) >= QUOT_TINY ? gq * __builtin_amdgcn_rcpf(vq) : 0.f;  // zero or subnormal (v_rcp_f32 may flush it): left to the fix-up
      }
      const float v0 = gq * wt.x, v1 = gq * wt.y;
      if (packed == ppk) {
        p0 += v0; p1 += v1;
      } else if (packed == ppk + 1u) {
        // x-adjacent cell of the same row pair: texel column pX is complete (flush it, 32 lanes), column pX + 1 carries over from
        // half 1 to half 0
        float* dst = gch + ((int64_t)(ppk >> 16) * W + (ppk & 0xffffu)) * C;
        if (half == 0 && p0 != 0.f) atomicAdd(dst, p0);
        if (half == 0 && p1 != 0.f) atomicAdd(dst + rowC, p1);
        const float m0 = __shfl_xor(p0, 32, 64), m1 = __shfl_xor(p1, 32, 64);
        p0 = (half == 0 ? m0 : 0.f) + v0;
        p1 = (half == 0 ? m1 : 0.f) + v1;
        ppk = packed;
      } else {
        float* dst = gch + ((int64_t)(ppk >> 16) * W + (ppk & 0xffffu) + half) * C;
        if (p0 != 0.f) atomicAdd(dst, p0);
        if (p1 != 0.f) atomicAdd(dst + rowC, p1);
        ppk = packed; p0 = v0; p1 = v1;
      }
    }
  }
  {
    float* dst = gch + ((int64_t)(ppk >> 16) * W + (ppk & 0xffffu) + half) * C;
    if (p0 != 0.f) atomicAdd(dst, p0);
    if (p1 != 0.f) atomicAdd(dst + rowC, p1);
  }
}

// ---- pass B, round 4: lane = channel, each HALF of the wave walks its own 128 of the wave's 256 re-sorted entries ----
// scatter_grouped_kernel above spends a wave instruction per entry on 64 lanes = (x-corner, channel): both halves interpolate the same v_q, the
// per-entry scalars go through v_readfirstlane, the x-carry through ds_bpermute, and every run end costs ~100 issued instructions (64-bit
// texel addresses, four exec-masked atomics) -- ~125 per entry, issue-bound at 0.35 of the HBM roofline (profiles/r03_kernels.md section 1).
// Here a lane owns ONE channel of ALL FOUR texels of the cell: it loads the four texels itself (no lane swaps; v_q is computed once per
// channel), keeps four accumulators, and an x-adjacent next cell moves the x0 + 1 column's accumulators to the x0 column IN THE LANE.  Two
// entries advance per wave instruction (lanes 0-31: entries [0, 128) of the walk order, lanes 32-63: [128, 256)), so everything per-entry is
// either lane-parallel (records, prepared as before) or costs half an instruction; the run-end logic is branch-free selects plus atomics
// under the lanes' own predicates, addressed with 32-bit byte offsets from wave-uniform bases (saddr form).  Same arithmetic per
// contribution as before except that the weighted add is one fma (gq * w + acc).
template <int NP, bool QUOT, int UNROLL = 8>
__global__ __launch_bounds__(256) void scatter_halfwave_kernel(snerf_kplanes_desc d, int64_t N, const float* __restrict__ gvec,
                                                              const float4* __restrict__ sorted_rec, float* __restrict__ gplanes,
                                                              int64_t groups_per_seg, int seg_begin, int per_scale,
                                                              const float* __restrict__ planes, int row_stride) {
  constexpr int C = 32, CH = 256, HALF = CH / 2;
  static_assert(HALF % UNROLL == 0, "a half's records must not run into the other half's");
  // per wave, per entry (in WALK order) 8 dwords: {gvec row BYTE offset, x0 | y0 << 16, texel (x0, y0) BYTE offset in the plane,
  // x1-exists ? 128 : 0 | y1-exists << 31, w(x0y0), w(x0y1), w(x1y0), w(x1y1)}
  __shared__ __align__(16) uint32_t s_rec[4][CH * 8];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int half = lane >> 5, ch = lane & 31;
  const int64_t group = (int64_t)blockIdx.x * 4 + wave;
  const int seg = seg_begin + (int)(group / groups_per_seg);
  if (seg >= d.n_scales * NP) return;
  const int64_t i0 = (group - (int64_t)(seg - seg_begin) * groups_per_seg) * CH;
  if (i0 >= N) return;
  const int cnt = (int)((N - i0) < CH ? (N - i0) : CH);
  const int s = seg / NP, q = seg % NP;
  int a, b;
  seg_axes<NP>(q, a, b);
  const int W = d.res[s][a], H = d.res[s][b] > 0 ? d.res[s][b] : 1;
  const float4* rec = sorted_rec + (int64_t)(per_scale ? seg : q) * N + i0;
  const char* gseg = reinterpret_cast<const char*>(QUOT ? gvec + (int64_t)s * C : gvec + (int64_t)seg * N * C);
  const uint32_t rstrideB = (QUOT ? (uint32_t)row_stride : (uint32_t)C) * 4u;
  char* gbase = reinterpret_cast<char*>(gplanes + d.off[s][q]);
  const char* pbase = QUOT ? reinterpret_cast<const char*>(planes + d.off[s][q]) : nullptr;
  uint32_t* R = s_rec[wave];
  const bool sortable = (int64_t)W * H <= (1 << 23);
  const uint32_t rowB = (uint32_t)W * (uint32_t)(C * 4);

  uint32_t word[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = r * 64 + lane;
    word[r] = 0xffffffffu;
    if (e < cnt) {
      const float4 rc = rec[e];
      const int x0 = (int)floorf(axis_pix(rc.y, W)), y0 = (int)floorf(axis_pix(rc.z, H));
      word[r] = sortable ? (((uint32_t)(y0 * W + x0) << 8) | (uint32_t)e) : (uint32_t)e;
    }
  }
  if (sortable) BitonicK<CH>::run(word, lane);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = r * 64 + lane;
    uint4 hd = make_uint4(0u, 0xffffffffu, 0u, 0u);  // null record: zero weights, a key no cell has
    float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < cnt) {
      const float4 rc = rec[word[r] & 255u];
      const AxisTap tx = axis_tap(rc.y, W);
      const AxisTap ty = axis_tap(rc.z, H);
      hd.x = (uint32_t)__float_as_int(rc.x) * rstrideB;
      hd.y = (uint32_t)tx.i0 | ((uint32_t)ty.i0 << 16);
      hd.z = (uint32_t)(ty.i0 * W + tx.i0) * (uint32_t)(C * 4);
      hd.w = (tx.i1 != tx.i0 ? (uint32_t)(C * 4) : 0u) | (ty.i1 != ty.i0 ? 0x80000000u : 0u);
      const float4 tw = tap_weights(tx, ty);     // (x0y0, x1y0, x0y1, x1y1): the forward's products, bit for bit
      wt = make_float4(tw.x, tw.z, tw.y, tw.w);  // stored (x0y0, x0y1, x1y0, x1y1)
    }
    *reinterpret_cast<uint4*>(R + e * 8) = hd;
    *reinterpret_cast<float4*>(R + e * 8 + 4) = wt;
  }
  // the records were stored one lane per entry and are read back by every lane of the wave: order the LDS stores before the loads
  // explicitly (same wave, so no s_barrier -- but neither the compiler nor the LDS queue may move a load above these stores)
  wave_lds_publish();
  // the walk: this half's entries are R[half * HALF + i]; pending cell = (pk, pa: byte offset of its (x0, y0) texel + this lane's channel,
  // pdx / pdy: byte steps to its x0 + 1 column / y0 + 1 row, 0 where clamped) with accumulators p00 (x0,y0), p01 (x0,y1), p10 (x1,y0), p11 (x1,y1).
  // The pending cell starts as the half's FIRST entry with empty accumulators, so no flush ever sees an invalid cell.  Zero sums (a clamped
  // column / row, a null record of the chunk's tail) are not sent: measured 0.445 -> 0.431 ms, 13 % fewer atomic requests.
  const uint32_t* Rh = R + half * (HALF * 8);
  const uint32_t chB = (uint32_t)ch * 4u;
  uint32_t pk, pa, pdx, pdy;
  {
    const uint4 hd = *reinterpret_cast<const uint4*>(Rh);
    pk = hd.y; pa = hd.z + chB; pdx = hd.w & 0xffu; pdy = (hd.w >> 31) ? rowB : 0u;
  }
  float p00 = 0.f, p01 = 0.f, p10 = 0.f, p11 = 0.f;
  auto add = [&](uint32_t off, float v) {
    if (v != 0.f) atomicAdd(reinterpret_cast<float*>(gbase + off), v);
  };
  const int iters = cnt > HALF ? HALF : cnt;  // half 1 walks null records where the chunk is short
  for (int e0 = 0; e0 < iters; e0 += UNROLL) {
    float g[UNROLL];
    float t00[QUOT ? UNROLL : 1], t10[QUOT ? UNROLL : 1], t01[QUOT ? UNROLL : 1], t11[QUOT ? UNROLL : 1];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint4 hd = *reinterpret_cast<const uint4*>(Rh + (e0 + u) * 8);
      g[u] = *reinterpret_cast<const float*>(gseg + (hd.x + chB));
      if constexpr (QUOT) {
        const uint32_t o00 = hd.z + chB, dx = hd.w & 0xffu, dy = (hd.w >> 31) ? rowB : 0u;
        t00[u] = *reinterpret_cast<const float*>(pbase + o00);
        t10[u] = *reinterpret_cast<const float*>(pbase + (o00 + dx));
        t01[u] = *reinterpret_cast<const float*>(pbase + (o00 + dy));
        t11[u] = *reinterpret_cast<const float*>(pbase + (o00 + dx + dy));
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint4 hd = *reinterpret_cast<const uint4*>(Rh + (e0 + u) * 8);
      const float4 w = *reinterpret_cast<const float4*>(Rh + (e0 + u) * 8 + 4);
      float gq = g[u];
      if constexpr (QUOT) {
        const float vq = bilerp4(t00[u], t10[u], t01[u], t11[u], w.x, w.z, w.y, w.w);  // the forward's own number
        const bool normal = fabsf(vq) >= QUOT_TINY;
        gq = normal ? gq * __builtin_amdgcn_rcpf(vq) : 0.f;  // v_q == 0: G is 0 too, the fix-up supplies the exact term
        // a SUBNORMAL v_q beside a usable G (the other planes' product is large): v_rcp_f32 may flush it, the IEEE division does not
        // (G != 0 implies v_q != 0 for a G formed from this forward's features; the inner test only guards callers' inconsistent inputs)
        if (__builtin_expect(!normal && g[u] != 0.f, 0)) gq = vq != 0.f ? __fdiv_rn(g[u], vq) : 0.f;
      }
      const uint32_t key = hd.y;
      const bool same = key == pk, adj = key == pk + 1u;
      if (!same) {
        add(pa, p00);
        add(pa + pdy, p01);
        if (!adj) {
          add(pa + pdx, p10);
          add(pa + pdx + pdy, p11);
        }
      }
      const float n00 = same ? p00 : (adj ? p10 : 0.f), n01 = same ? p01 : (adj ? p11 : 0.f);
      const float n10 = same ? p10 : 0.f, n11 = same ? p11 : 0.f;
      p00 = __fmaf_rn(gq, w.x, n00);
      p01 = __fmaf_rn(gq, w.y, n01);
      p10 = __fmaf_rn(gq, w.z, n10);
      p11 = __fmaf_rn(gq, w.w, n11);
      pk = key;
      pa = hd.z + chB;
      pdx = hd.w & 0xffu;
      pdy = (hd.w >> 31) ? rowB : 0u;
    }
  }
  add(pa, p00);
  add(pa + pdy, p01);
  add(pa + pdx, p10);
  add(pa + pdx + pdy, p11);
}

// ---- quotient form, the two small kernels around pass B ----
// The fix list holds ELEMENTS (round 4; rows before): {index into the [N, C n_scales] feature tensor, the feature gradient there} for every
// element whose feature vanished (zero, or below the smallest normal float) while its gradient did not: G / v_q cannot give plane q's gradient
// there (v_q == 0 took the other planes' product with it, or the product of six normal values underflowed).  Producers: quotient_prepare_kernel
// below, or the sigma_net backward's epilogue (mlp_lp.hip: snerf_mlp_bwd_x16_quotient), which forms G from the feature tile it holds in LDS.
// G = gfeat .* feat (C / 4 lanes per (sample, scale) row, float4 each); vanished features get G = 0 (pass B then adds exactly nothing for the
// channel) and, where the gradient is not zero, an entry in the fix list.
template <int C>
__global__ __launch_bounds__(256) void quotient_prepare_kernel(int64_t rows, const float* __restrict__ gfeat, const float* __restrict__ feat,
                                                              float* __restrict__ G, int32_t* __restrict__ list, int capacity, int32_t* __restrict__ count,
                                                              int32_t* __restrict__ count_next) {
  // two counters, used alternately: this launch resets the OTHER one (its last reader, the previous step's fix-up, is behind us in stream
  // order), so no memset launch sits between the MLP backward and this kernel
  if (blockIdx.x == 0 && threadIdx.x == 0 && count_next) *count_next = 0;
  constexpr int LPR = C / 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / LPR;
  if (row >= rows) return;
  const float4 g = *reinterpret_cast<const float4*>(gfeat + gid * 4);
  const float4 f = *reinterpret_cast<const float4*>(feat + gid * 4);
  const bool zx = fabsf(f.x) < QUOT_TINY, zy = fabsf(f.y) < QUOT_TINY, zz = fabsf(f.z) < QUOT_TINY, zw = fabsf(f.w) < QUOT_TINY;
  float4 Gv = f4_mul(g, f);
  Gv.x = zx ? 0.f : Gv.x; Gv.y = zy ? 0.f : Gv.y; Gv.z = zz ? 0.f : Gv.z; Gv.w = zw ? 0.f : Gv.w;
  *reinterpret_cast<float4*>(G + gid * 4) = Gv;
  if (zx && g.x != 0.f) fix_append(list, capacity, count, (int32_t)(gid * 4), g.x);
  if (zy && g.y != 0.f) fix_append(list, capacity, count, (int32_t)(gid * 4 + 1), g.y);
  if (zz && g.z != 0.f) fix_append(list, capacity, count, (int32_t)(gid * 4 + 2), g.z);
  if (zw && g.w != 0.f) fix_append(list, capacity, count, (int32_t)(gid * 4 + 3), g.w);
}

// Exact gradient of the listed elements, one lane per element (sample n, scale s, channel ch; g = the feature gradient).  With v_p the six
// planes' values at the sample (the forward's own numbers: bilerp4):
//   exactly ONE |v_z| below QUOT_TINY -> plane z receives g * prod_{p != z} v_p (the others' terms contain v_z: zero);
//   none (the product of six normal values underflowed)  -> EVERY plane q receives g * prod_{p != q} v_p;
//   two or more -> every term contains a vanished factor: nothing to add.
// Rare by construction (a trained texel that is exactly 0.0f, or an underflowing product), so no run-length combining.
template <int NP>
__global__ __launch_bounds__(256) void quotient_fixup_kernel(snerf_kplanes_desc d, const float* __restrict__ planes, snerf_coords c,
                                                            const int32_t* __restrict__ list, const int32_t* __restrict__ count,
                                                            int capacity, float* __restrict__ gplanes, int n_scales_total, int scale_begin, int scale_end,
                                                            int32_t* __restrict__ overflow_peak) {
  constexpr int C = 32;
  int n_list = *count;
  // more entries were appended than the list holds: the ones beyond the capacity are lost.  Record it where the host can see it (sticky
  // maximum of the demanded entry count) instead of dropping gradients silently.
  if (overflow_peak && n_list > capacity && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(overflow_peak, n_list);
  n_list = n_list < capacity ? n_list : capacity;
  const int F = n_scales_total * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_list; i += gridDim.x * blockDim.x) {
    const int elem = list[2 * i];
    const float g = __int_as_float(list[2 * i + 1]);
    const int64_t n = elem / F;
    const int col = elem - (int)n * F;
    const int s = col / C, ch = col - s * C;
    if (s < scale_begin || s >= scale_end) continue;
    float p[4];
    load_coords<NP>(c, n, p);
    AxisTap tap[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tap[k] = axis_tap(p[k], d.res[s][k] > 0 ? d.res[s][k] : 1);
    float v[NP];
    int zeros = 0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const AxisTap& tx = tap[pair_a<NP>(q)];
      const AxisTap& ty = tap[pair_b<NP>(q)];
      const int W = d.res[s][pair_a<NP>(q)];
      const float* base = planes + d.off[s][q] + ch;
      const float4 w = tap_weights(tx, ty);
      v[q] = bilerp4(base[((int64_t)ty.i0 * W + tx.i0) * C], base[((int64_t)ty.i0 * W + tx.i1) * C], base[((int64_t)ty.i1 * W + tx.i0) * C],
                     base[((int64_t)ty.i1 * W + tx.i1) * C], w.x, w.y, w.z, w.w);
      zeros += fabsf(v[q]) < QUOT_TINY;
    }
    if (zeros >= 2) continue;
    float suf[NP + 1];
    suf[NP] = 1.f;
#pragma unroll
    for (int q = NP - 1; q >= 0; --q) suf[q] = suf[q + 1] * v[q];
    float pre = g;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const float term = pre * suf[q + 1];  // g * prod_{p != q} v_p
      pre *= v[q];
      if (zeros == 1 && fabsf(v[q]) >= QUOT_TINY) continue;  // one vanished plane: only IT has a non-zero gradient
      const AxisTap& tx = tap[pair_a<NP>(q)];
      const AxisTap& ty = tap[pair_b<NP>(q)];
      const int W = d.res[s][pair_a<NP>(q)];
      float* gb = gplanes + d.off[s][q] + ch;
      const float a00 = term * tx.w0 * ty.w0, a01 = term * tx.w0 * ty.w1, a10 = term * tx.w1 * ty.w0, a11 = term * tx.w1 * ty.w1;
      if (a00 != 0.f) atomicAdd(gb + ((int64_t)ty.i0 * W + tx.i0) * C, a00);
      if (a01 != 0.f) atomicAdd(gb + ((int64_t)ty.i1 * W + tx.i0) * C, a01);
      if (a10 != 0.f) atomicAdd(gb + ((int64_t)ty.i0 * W + tx.i1) * C, a10);
      if (a11 != 0.f) atomicAdd(gb + ((int64_t)ty.i1 * W + tx.i1) * C, a11);
    }
  }
}

// Measured and dropped (profiles/r01_kernels.md): a per-wave LDS texel cache behind the register stage (64 direct-mapped texel
// rows, tags in lanes) halves the atomic requests (8.4 M -> 4.5 M) but is no faster: with ds_add_f32 each LDS instruction costs
// ~146 cycles (6x slower end to end), with read-add-write the LDS round trip per flush sits on the wave's critical path
// (0.88-0.93 ms vs 0.80-0.86 ms); deferring the add/write to the next flush so that the read has time to land: 1.09 ms.
// Ablation of this kernel: 0.28 ms without gvec loads and atomics, 0.36 ms without atomics, 0.81-0.86 ms complete -- the in-loop
// atomics are what it waits for, and the time follows neither the request count nor the prefetch depth nor the VALU count.

static int check_desc(const snerf_kplanes_desc* d, const snerf_coords* c, int64_t N) {
  SNERF_REQUIRE(d && c, "kplanes_sorted: null descriptor");
  SNERF_REQUIRE(d->n_scales >= 1 && d->n_scales <= SNERF_MAX_SCALES, "kplanes_sorted: n_scales=%d", d->n_scales);
  SNERF_REQUIRE(d->C == 8 || d->C == 16 || d->C == 32, "kplanes_sorted: C=%d unsupported", d->C);
  SNERF_REQUIRE(d->n_coords == 3 || d->n_coords == 4, "kplanes_sorted: n_coords=%d", d->n_coords);
  SNERF_REQUIRE(N >= 0 && N < (1LL << 31), "kplanes_sorted: N=%lld", (long long)N);
  SNERF_REQUIRE(c->mode == 0 || c->mode == 1, "kplanes_sorted: coords.mode=%d", c->mode);
  return 0;
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_kplanes_sort_workspace(const snerf_kplanes_desc* desc, int64_t N, int64_t* hist_cells, int64_t* index_elems) {
  SNERF_REQUIRE(desc && hist_cells && index_elems, "kplanes_sort_workspace: null argument");
  SegTable st;
  int rc = build_segs(desc, st);
  if (rc) return rc;
  // hist: cells + room for the scan's block sums (one per 1024 cells, +1024 slack)
  *hist_cells = (int64_t)st.cell_off[st.n_segs] + ((int64_t)st.cell_off[st.n_segs] + 1023) / 1024 + 1024;
  *index_elems = (int64_t)st.n_segs * N;
  return 0;
}

extern "C" int snerf_kplanes_sort_samples(const snerf_kplanes_desc* desc, const snerf_coords* coords, int64_t N, int32_t* hist, int32_t* rank,
                                          float* sorted_rec, snerf_stream_t stream) {
  int rc = check_desc(desc, coords, N);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(hist && rank && sorted_rec, "kplanes_sort_samples: null workspace");
  float4* sorted_n = reinterpret_cast<float4*>(sorted_rec);
  SegTable st;
  rc = build_segs(desc, st);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t cells = st.cell_off[st.n_segs];
  const int nb = (int)((cells + 1023) / 1024);
  int32_t* block_sums = hist + cells;
  rc = check_hip(hipMemsetAsync(hist, 0, (size_t)cells * sizeof(int32_t), s), "kplanes_sort memset");
  if (rc) return rc;
  const dim3 gs((unsigned)ceil_div(N, 256));
  if (desc->n_coords == 4) hipLaunchKernelGGL((sort_keys_kernel<6, false>), gs, dim3(256), 0, s, *coords, st, N, hist, rank, nullptr, nullptr);
  else hipLaunchKernelGGL((sort_keys_kernel<3, false>), gs, dim3(256), 0, s, *coords, st, N, hist, rank, nullptr, nullptr);
  hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, s, hist, cells, block_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb);
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(256), 0, s, hist, cells, block_sums);
  if (desc->n_coords == 4) hipLaunchKernelGGL((sort_keys_kernel<6, true>), gs, dim3(256), 0, s, *coords, st, N, nullptr, rank, hist, sorted_n);
  else hipLaunchKernelGGL((sort_keys_kernel<3, true>), gs, dim3(256), 0, s, *coords, st, N, nullptr, rank, hist, sorted_n);
  SNERF_LAUNCH_CHECK("kplanes_sort_samples");
  return 0;
}

template <int C, int NP>
static int launch_gradvec(const snerf_kplanes_desc* d, const float* planes, const snerf_coords* c, int64_t N, const float* gout, void* gvec, int gvec_bf16,
                          hipStream_t st) {
  const dim3 grid((unsigned)ceil_div(N * (C / 4), 256));
  if (gvec_bf16) hipLaunchKernelGGL((gradvec_kernel<C, NP, __bf16>), grid, dim3(256), 0, st, *d, planes, *c, N, gout, (__bf16*)gvec);
  else hipLaunchKernelGGL((gradvec_kernel<C, NP, float>), grid, dim3(256), 0, st, *d, planes, *c, N, gout, (float*)gvec);
  SNERF_LAUNCH_CHECK("kplanes_gradvec");
  return 0;
}
// scatter_halfwave_kernel addresses texels and gradient rows with 32-bit BYTE offsets.  SNERF_PASSB_GROUPED=1 (dev A-B) keeps the round-2 kernel.
static bool halfwave_ok(const snerf_kplanes_desc* d, int scale_begin, int scale_end) {
  const char* env = getenv("SNERF_PASSB_GROUPED");  // read per call: tools/bench_passb.py flips it inside one process
  const bool forced_off = env && atoi(env) != 0;
  if (forced_off || d->C != 32) return false;
  for (int s = scale_begin; s < scale_end; ++s) {
    int64_t mx = 1;
    for (int k = 0; k < d->n_coords; ++k) mx = mx > d->res[s][k] ? mx : d->res[s][k];
    if (mx * mx * d->C * 4 >= (1LL << 32) || mx >= 65536) return false;
  }
  return true;
}

template <int C, int NP>
static int launch_scatter_sorted(const snerf_kplanes_desc* d, int64_t N, const void* gvec, int gvec_bf16, const float4* sorted_n, float* gp, int scale_begin,
                                 int scale_end, hipStream_t st) {
  constexpr int run = 256;
  const int64_t groups_per_seg = (N + run - 1) / run;
  // the kernel stops at segment n_scales * NP: hand it a descriptor that ends at scale_end
  snerf_kplanes_desc dd = *d;
  dd.n_scales = scale_end;
  SegTable stb;
  int rc = build_segs(d, stb);
  if (rc) return rc;
  if (C == 32 && N * C < (1LL << 31)) {  // 32-bit gvec row offsets inside the kernel
    const int64_t gps = (N + 255) / 256;
    const dim3 grid((unsigned)ceil_div(gps * (scale_end - scale_begin) * NP, 4));
    if (!gvec_bf16 && N * C * 4 < (1LL << 32) && halfwave_ok(d, scale_begin, scale_end)) {
      hipLaunchKernelGGL((scatter_halfwave_kernel<NP, false>), grid, dim3(256), 0, st, dd, N, (const float*)gvec, sorted_n, gp, gps, scale_begin * NP,
                         stb.per_scale, nullptr, 0);
      SNERF_LAUNCH_CHECK("kplanes_scatter_halfwave");
      return 0;
    }
    if (gvec_bf16) hipLaunchKernelGGL((scatter_grouped_kernel<NP, __bf16>), grid, dim3(256), 0, st, dd, N, (const __bf16*)gvec, sorted_n, gp, gps,
                                      scale_begin * NP, stb.per_scale);
    else hipLaunchKernelGGL((scatter_grouped_kernel<NP, float>), grid, dim3(256), 0, st, dd, N, (const float*)gvec, sorted_n, gp, gps, scale_begin * NP,
                            stb.per_scale);
    SNERF_LAUNCH_CHECK("kplanes_scatter_grouped");
    return 0;
  }
  const int64_t threads = groups_per_seg * (scale_end - scale_begin) * NP * (2 * C);
  if (gvec_bf16) hipLaunchKernelGGL((scatter_sorted_kernel<C, NP, __bf16>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, dd, N,
                                    (const __bf16*)gvec, sorted_n, gp, run, groups_per_seg, scale_begin * NP, stb.per_scale);
  else hipLaunchKernelGGL((scatter_sorted_kernel<C, NP, float>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, dd, N, (const float*)gvec,
                          sorted_n, gp, run, groups_per_seg, scale_begin * NP, stb.per_scale);
  SNERF_LAUNCH_CHECK("kplanes_scatter_sorted");
  return 0;
}

#define DISPATCH2(FN, ...)                                                              \
  do {                                                                                  \
    if (desc->n_coords == 4) {                                                          \
      if (desc->C == 32) return FN<32, 6>(__VA_ARGS__);                                 \
      if (desc->C == 16) return FN<16, 6>(__VA_ARGS__);                                 \
      return FN<8, 6>(__VA_ARGS__);                                                     \
    } else {                                                                            \
      if (desc->C == 32) return FN<32, 3>(__VA_ARGS__);                                 \
      if (desc->C == 16) return FN<16, 3>(__VA_ARGS__);                                 \
      return FN<8, 3>(__VA_ARGS__);                                                     \
    }                                                                                   \
  } while (0)

extern "C" int snerf_kplanes_gradvec(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N, const float* grad_out,
                                     void* gvec, int32_t gvec_bf16, snerf_stream_t stream) {
  int rc = check_desc(desc, coords, N);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && grad_out && gvec, "kplanes_gradvec: null buffer");
  DISPATCH2(launch_gradvec, desc, planes, coords, N, grad_out, gvec, gvec_bf16, (hipStream_t)stream);
}

extern "C" int snerf_kplanes_scatter_sorted_scales(const snerf_kplanes_desc* desc, int64_t N, const void* gvec, int32_t gvec_bf16, const float* sorted_rec,
                                                   float* grad_planes, int32_t scale_begin, int32_t scale_end, snerf_stream_t stream) {
  snerf_coords dummy = {};
  int rc = check_desc(desc, &dummy, N);
  if (rc) return rc;
  SNERF_REQUIRE(scale_begin >= 0 && scale_begin <= scale_end && scale_end <= desc->n_scales, "kplanes_scatter_sorted: scales [%d, %d) of %d",
                scale_begin, scale_end, desc->n_scales);
  if (N == 0 || scale_begin == scale_end) return 0;
  SNERF_REQUIRE(gvec && sorted_rec && grad_planes, "kplanes_scatter_sorted: null buffer");
  DISPATCH2(launch_scatter_sorted, desc, N, gvec, gvec_bf16, reinterpret_cast<const float4*>(sorted_rec), grad_planes, scale_begin, scale_end,
            (hipStream_t)stream);
}

extern "C" int snerf_kplanes_scatter_sorted(const snerf_kplanes_desc* desc, int64_t N, const void* gvec, int32_t gvec_bf16, const float* sorted_rec,
                                            float* grad_planes, snerf_stream_t stream) {
  SNERF_REQUIRE(desc, "kplanes_scatter_sorted: null descriptor");
  return snerf_kplanes_scatter_sorted_scales(desc, N, gvec, gvec_bf16, sorted_rec, grad_planes, 0, desc->n_scales, stream);
}

// ---- quotient form of the sorted scatter (snerf.h) ----
static int quotient_ok(const snerf_kplanes_desc* d, int64_t N) {
  SNERF_REQUIRE(d, "kplanes_quotient: null descriptor");
  SNERF_REQUIRE(d->C == 32 && d->concat == 1 && d->n_scales >= 1 && (int64_t)N * d->C * d->n_scales < (1LL << 31),
                "kplanes_quotient: built for C = 32, concatenated scales and N * 32 * n_scales < 2^31 (C=%d concat=%d N=%lld)", d->C, d->concat, (long long)N);
  return 0;
}

extern "C" int snerf_kplanes_quotient_supported(const snerf_kplanes_desc* desc, int64_t N) {
  return desc && desc->C == 32 && desc->concat == 1 && desc->n_scales >= 1 && N >= 0 && (int64_t)N * desc->C * desc->n_scales < (1LL << 31) &&
         (desc->n_coords == 3 || desc->n_coords == 4);
}

extern "C" int snerf_kplanes_quotient_prepare(const snerf_kplanes_desc* desc, int64_t N, const float* grad_feat, const float* feat, float* G,
                                              int32_t* fix_list, int32_t fix_capacity, int32_t* fix_count, int32_t* fix_count_next, snerf_stream_t stream) {
  int rc = quotient_ok(desc, N);
  if (rc) return rc;
  SNERF_REQUIRE(N >= 0 && fix_capacity >= 0, "kplanes_quotient_prepare: N=%lld capacity=%d", (long long)N, fix_capacity);
  SNERF_REQUIRE(fix_count, "kplanes_quotient_prepare: null counter");
  hipStream_t st = (hipStream_t)stream;
  if (!fix_count_next) {  // single counter: reset it here
    rc = check_hip(hipMemsetAsync(fix_count, 0, sizeof(int32_t), st), "kplanes_quotient_prepare memset");
    if (rc) return rc;
  }
  if (N == 0) {
    if (fix_count_next) return check_hip(hipMemsetAsync(fix_count_next, 0, sizeof(int32_t), st), "kplanes_quotient_prepare memset");
    return 0;
  }
  SNERF_REQUIRE(grad_feat && feat && G && (fix_list || fix_capacity == 0), "kplanes_quotient_prepare: null buffer");
  const int64_t rows = N * desc->n_scales;
  hipLaunchKernelGGL((quotient_prepare_kernel<32>), dim3((unsigned)ceil_div(rows * 8, 256)), dim3(256), 0, st, rows, grad_feat, feat, G, fix_list,
                     fix_capacity, fix_count, fix_count_next);
  SNERF_LAUNCH_CHECK("kplanes_quotient_prepare");
  return 0;
}

extern "C" int snerf_kplanes_scatter_quotient_scales(const snerf_kplanes_desc* desc, const float* planes, int64_t N, const float* G, const float* sorted_rec,
                                                     float* grad_planes, int32_t scale_begin, int32_t scale_end, snerf_stream_t stream) {
  int rc = quotient_ok(desc, N);
  if (rc) return rc;
  SNERF_REQUIRE(scale_begin >= 0 && scale_begin <= scale_end && scale_end <= desc->n_scales, "kplanes_scatter_quotient: scales [%d, %d) of %d",
                scale_begin, scale_end, desc->n_scales);
  if (N == 0 || scale_begin == scale_end) return 0;
  SNERF_REQUIRE(planes && G && sorted_rec && grad_planes, "kplanes_scatter_quotient: null buffer");
  snerf_kplanes_desc dd = *desc;
  dd.n_scales = scale_end;
  SegTable stb;
  rc = build_segs(desc, stb);
  if (rc) return rc;
  const int NP = desc->n_coords == 4 ? 6 : 3;
  for (int s = scale_begin; s < scale_end; ++s) {  // pass B addresses a plane's texels with 32-bit element offsets
    int64_t mx = 1;
    for (int k = 0; k < desc->n_coords; ++k) mx = mx > desc->res[s][k] ? mx : desc->res[s][k];
    SNERF_REQUIRE(mx * mx * desc->C < (1LL << 31), "kplanes_scatter_quotient: plane of %lld^2 texels too large for 32-bit offsets", (long long)mx);
  }
  const int64_t gps = (N + 255) / 256;
  const dim3 grid((unsigned)ceil_div(gps * (scale_end - scale_begin) * NP, 4));
  const float4* rec = reinterpret_cast<const float4*>(sorted_rec);
  const int stride = desc->C * desc->n_scales;
  hipStream_t st = (hipStream_t)stream;
  if ((int64_t)N * stride * 4 < (1LL << 32) && halfwave_ok(desc, scale_begin, scale_end)) {
    if (NP == 6) hipLaunchKernelGGL((scatter_halfwave_kernel<6, true>), grid, dim3(256), 0, st, dd, N, G, rec, grad_planes, gps, scale_begin * NP, stb.per_scale, planes, stride);
    else hipLaunchKernelGGL((scatter_halfwave_kernel<3, true>), grid, dim3(256), 0, st, dd, N, G, rec, grad_planes, gps, scale_begin * NP, stb.per_scale, planes, stride);
    SNERF_LAUNCH_CHECK("kplanes_scatter_quotient");
    return 0;
  }
  if (NP == 6) hipLaunchKernelGGL((scatter_grouped_kernel<6, float, true>), grid, dim3(256), 0, st, dd, N, G, rec, grad_planes, gps, scale_begin * NP, stb.per_scale,
                                  planes, stride);
  else hipLaunchKernelGGL((scatter_grouped_kernel<3, float, true>), grid, dim3(256), 0, st, dd, N, G, rec, grad_planes, gps, scale_begin * NP, stb.per_scale,
                          planes, stride);
  SNERF_LAUNCH_CHECK("kplanes_scatter_quotient");
  return 0;
}

extern "C" int snerf_kplanes_quotient_fixup(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                                            const int32_t* fix_list, const int32_t* fix_count, int32_t fix_capacity, float* grad_planes,
                                            int32_t scale_begin, int32_t scale_end, int32_t* overflow_peak, snerf_stream_t stream) {
  int rc = check_desc(desc, coords, N);
  if (rc) return rc;
  rc = quotient_ok(desc, N);
  if (rc) return rc;
  SNERF_REQUIRE(scale_begin >= 0 && scale_begin <= scale_end && scale_end <= desc->n_scales, "kplanes_quotient_fixup: scales [%d, %d) of %d", scale_begin,
                scale_end, desc->n_scales);
  if (N == 0 || fix_capacity == 0 || scale_begin == scale_end) return 0;
  SNERF_REQUIRE(planes && fix_list && fix_count && grad_planes, "kplanes_quotient_fixup: null buffer");
  // a fixed small grid that strides over the (device-side) count: an empty list costs one launch
  hipStream_t st = (hipStream_t)stream;
  if (desc->n_coords == 4) hipLaunchKernelGGL((quotient_fixup_kernel<6>), dim3(64), dim3(256), 0, st, *desc, planes, *coords, fix_list, fix_count,
                                              fix_capacity, grad_planes, desc->n_scales, scale_begin, scale_end, overflow_peak);
  else hipLaunchKernelGGL((quotient_fixup_kernel<3>), dim3(64), dim3(256), 0, st, *desc, planes, *coords, fix_list, fix_count, fix_capacity,
                          grad_planes, desc->n_scales, scale_begin, scale_end, overflow_peak);
  SNERF_LAUNCH_CHECK("kplanes_quotient_fixup");
  return 0;
}

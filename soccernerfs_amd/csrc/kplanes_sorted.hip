// K-Planes plane-gradient scatter, sorted variant: ~6x fewer atomic requests than the sample-major scatter.
//
// Why: fp32 atomics on gfx950 cost ~one memory-side 64-B request each (MI355X_MICROARCH.md "Global float atomics"), and the
// sample-major scatter (kplanes.hip) can only merge contributions of CONSECUTIVE samples of one ray.  Counted on real
// training batches (tools/scatter_stats.py, 262 144 samples, 30 planes): 6.8 M (plane,row) flushes early in training, 3.7 M
// after 3 k steps -- but only 0.7-0.9 M DISTINCT (plane, row, x0) keys: the three time planes have just n_times x W
// texels, and coarse spatial planes a few thousand.  Sorting the samples of every (scale, plane) segment by texel key
// makes equal keys adjacent, so run-length combining removes ~85 % of the requests.
//
// Pipeline (all segments of a plane set at once; seg = scale * NP + plane, every segment has exactly N entries):
//   K1 rank    : one lane per sample: key = cell(seg) + row0 * W + x0;  rank = atomicAdd(&hist[key], 1)   (int atomics)
//   K2 scan    : exclusive prefix sum of hist (3 small kernels)
//   K3 reorder : sorted_rec[scan[key] + rank] = {n, fx, fy} (sample id + its pixel coordinates on that plane) -- a counting
//                sort; positions depend only on the sample coordinates, so K1-K3 run on a side stream under the forward /
//                MLP backward, and pass B never has to touch the ray buffers again
//   A  gradvec : sample-major, float4 per lane (like the forward gather): g_q = dL/d(interp of plane q) -> gvec[seg][n][C]
//   B  scatter : per segment, lane groups of 2*C lanes (x-corner, channel) walk RUN consecutive SORTED entries, multiply by
//                the bilinear weights, run-length-combine per row and flush with one 256-B atomic instruction per run.
#include <stdlib.h>

#include "kplanes_common.hpp"

namespace snerf {

struct SegTable {
  int n_seg;
  int cell_off[SNERF_MAX_SCALES * 6 + 1];  // first histogram cell of each segment; [n_seg] = total cells
};

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
__global__ __launch_bounds__(256) void sort_keys_kernel(snerf_kplanes_desc d, snerf_coords c, SegTable st, int64_t N, int32_t* __restrict__ hist,
                                                       int32_t* __restrict__ rank, const int32_t* __restrict__ scan, float4* __restrict__ sorted_rec) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float p[4];
  load_coords<NP>(c, n, p);
  {
    const int s = blockIdx.y;  // one lane per (sample, scale): 5x the lanes in flight for the (latency-bound) returning atomics
    int i0[4];
    float px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      px[k] = axis_pix(p[k], d.res[s][k] > 0 ? d.res[s][k] : 1);
      i0[k] = (int)floorf(px[k]);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      constexpr auto& A = PlanePairs<NP>::a;
      constexpr auto& B = PlanePairs<NP>::b;
      const int seg = s * NP + q;
      const int key = st.cell_off[seg] + i0[B[q]] * d.res[s][A[q]] + i0[A[q]];
      if (!REORDER) {
        rank[(int64_t)seg * N + n] = atomicAdd(hist + key, 1);
      } else {
        // sorted record: sample id + its pixel coordinates on this plane, so pass B never touches the ray buffers
        sorted_rec[scan[key] + rank[(int64_t)seg * N + n]] = make_float4(__int_as_float((int)n), px[A[q]], px[B[q]], 0.f);
      }
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

// ---- pass A: gradient w.r.t. each plane's interpolated value, sample-major ----
template <int C, int NP>
__global__ __launch_bounds__(256) void gradvec_kernel(snerf_kplanes_desc d, const float* __restrict__ planes, snerf_coords c, int64_t N,
                                                     const float* __restrict__ gout, float* __restrict__ gvec) {
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
      *reinterpret_cast<float4*>(gvec + ((int64_t)(s * NP + q) * N + n) * C + cg * 4) = gq;
    }
  }
}

// ---- pass B: sorted run-length scatter ----
// One lane group (2*C lanes = (x-corner, channel)) walks RUN consecutive SORTED records {n, fx, fy}.  Records are read
// UNROLL at a time (wave-uniform addresses), the UNROLL gvec rows are fetched together, then combined in order.
template <int C, int NP>
__global__ __launch_bounds__(256) void scatter_sorted_kernel(snerf_kplanes_desc d, int64_t N, const float* __restrict__ gvec,
                                                            const float4* __restrict__ sorted_rec, float* __restrict__ gplanes, int run,
                                                            int64_t groups_per_seg) {
  constexpr int LPS = 2 * C;
  constexpr int UNROLL = 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t group = gid / LPS;
  const int li = (int)(gid % LPS);
  const int half = li / C, ch = li % C;
  const int seg = (int)(group / groups_per_seg);
  if (seg >= d.n_scales * NP) return;
  const int64_t i0 = (group - (int64_t)seg * groups_per_seg) * run;
  if (i0 >= N) return;
  const int cnt = (int)((N - i0) < run ? (N - i0) : run);
  const int s = seg / NP, q = seg % NP;
  int a, b;
  seg_axes<NP>(q, a, b);
  const int W = d.res[s][a], H = d.res[s][b] > 0 ? d.res[s][b] : 1;
  const float4* rec = sorted_rec + (int64_t)seg * N + i0;
  const float* gv = gvec + (int64_t)seg * N * C + ch;
  float* gbase = gplanes + d.off[s][q] + li;
  int pend_key[2] = {-1, -1};
  float pend_val[2] = {0.f, 0.f};
  for (int i = 0; i < cnt; i += UNROLL) {
    float4 r[UNROLL];
    float g[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r[u] = rec[i + u < cnt ? i + u : cnt - 1];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) g[u] = gv[(int64_t)__float_as_int(r[u].x) * C];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (i + u < cnt) {
        const AxisTap tx = tap_from_pix(r[u].y, W);
        const AxisTap ty = tap_from_pix(r[u].z, H);
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

static int build_segs(const snerf_kplanes_desc* d, SegTable& st) {
  const int NP = d->n_coords == 4 ? 6 : 3;
  static const int PA6[6] = {0, 0, 0, 1, 1, 2}, PB6[6] = {1, 2, 3, 2, 3, 3}, PA3[3] = {0, 0, 1}, PB3[3] = {1, 2, 2};
  st.n_seg = d->n_scales * NP;
  int64_t off = 0;
  for (int s = 0; s < d->n_scales; ++s)
    for (int q = 0; q < NP; ++q) {
      const int a = NP == 6 ? PA6[q] : PA3[q], b = NP == 6 ? PB6[q] : PB3[q];
      st.cell_off[s * NP + q] = (int)off;
      off += (int64_t)d->res[s][a] * d->res[s][b];
    }
  SNERF_REQUIRE(off < (1LL << 31), "kplanes_sort: too many plane cells (%lld)", (long long)off);
  st.cell_off[st.n_seg] = (int)off;
  return 0;
}

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
  *hist_cells = (int64_t)st.cell_off[st.n_seg] + ((int64_t)st.cell_off[st.n_seg] + 1023) / 1024 + 1024;
  *index_elems = (int64_t)st.n_seg * N;
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
  const int64_t cells = st.cell_off[st.n_seg];
  const int nb = (int)((cells + 1023) / 1024);
  int32_t* block_sums = hist + cells;
  rc = check_hip(hipMemsetAsync(hist, 0, (size_t)cells * sizeof(int32_t), s), "kplanes_sort memset");
  if (rc) return rc;
  const dim3 gs((unsigned)ceil_div(N, 256), (unsigned)desc->n_scales);
  if (desc->n_coords == 4) hipLaunchKernelGGL((sort_keys_kernel<6, false>), gs, dim3(256), 0, s, *desc, *coords, st, N, hist, rank, nullptr, nullptr);
  else hipLaunchKernelGGL((sort_keys_kernel<3, false>), gs, dim3(256), 0, s, *desc, *coords, st, N, hist, rank, nullptr, nullptr);
  hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, s, hist, cells, block_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb);
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(256), 0, s, hist, cells, block_sums);
  if (desc->n_coords == 4) hipLaunchKernelGGL((sort_keys_kernel<6, true>), gs, dim3(256), 0, s, *desc, *coords, st, N, nullptr, rank, hist, sorted_n);
  else hipLaunchKernelGGL((sort_keys_kernel<3, true>), gs, dim3(256), 0, s, *desc, *coords, st, N, nullptr, rank, hist, sorted_n);
  SNERF_LAUNCH_CHECK("kplanes_sort_samples");
  return 0;
}

template <int C, int NP>
static int launch_gradvec(const snerf_kplanes_desc* d, const float* planes, const snerf_coords* c, int64_t N, const float* gout, float* gvec, hipStream_t st) {
  hipLaunchKernelGGL((gradvec_kernel<C, NP>), dim3((unsigned)ceil_div(N * (C / 4), 256)), dim3(256), 0, st, *d, planes, *c, N, gout, gvec);
  SNERF_LAUNCH_CHECK("kplanes_gradvec");
  return 0;
}
template <int C, int NP>
static int launch_scatter_sorted(const snerf_kplanes_desc* d, int64_t N, const float* gvec, const float4* sorted_n, float* gp, hipStream_t st) {
  static const int run = [] { const char* e = getenv("SNERF_SORTED_RUN"); int v = e ? atoi(e) : 64; return v > 0 ? v : 64; }();
  const int64_t groups_per_seg = (N + run - 1) / run;
  const int64_t threads = groups_per_seg * d->n_scales * NP * (2 * C);
  hipLaunchKernelGGL((scatter_sorted_kernel<C, NP>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, *d, N, gvec, sorted_n, gp, run,
                     groups_per_seg);
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
                                     float* gvec, snerf_stream_t stream) {
  int rc = check_desc(desc, coords, N);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && grad_out && gvec, "kplanes_gradvec: null buffer");
  DISPATCH2(launch_gradvec, desc, planes, coords, N, grad_out, gvec, (hipStream_t)stream);
}

extern "C" int snerf_kplanes_scatter_sorted(const snerf_kplanes_desc* desc, int64_t N, const float* gvec, const float* sorted_rec,
                                            float* grad_planes, snerf_stream_t stream) {
  snerf_coords dummy = {};
  int rc = check_desc(desc, &dummy, N);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(gvec && sorted_rec && grad_planes, "kplanes_scatter_sorted: null buffer");
  DISPATCH2(launch_scatter_sorted, desc, N, gvec, reinterpret_cast<const float4*>(sorted_rec), grad_planes, (hipStream_t)stream);
}

// Fused proposal density: plane gather -> 8 -> 64 -> 1 net -> trunc_exp in ONE kernel per proposal level.
//
// Replaces KPlanesDensityField.get_density (NS/fields/kplanes_field.py:410-460: interpolate_kplanes on the level's six C = 8 planes, sigma_net
// 8 -> 64 -> 1, trunc_exp) as ProposalNetworkSampler.generate_ray_samples calls it per level (NS/model_components/ray_samplers.py:559-600).
// Unfused that is kplanes_gather_fwd_kernel<8,6> -> [N,8] fp32 in HBM -> mlp_lp_fwd_kernel<T,32,64,1,64> -> density: two launches per level
// at the head of every step (1.57 M samples, three times the field's).  Here a tile of 64 samples goes from the texel reads to the density
// inside one workgroup; the feature tile exists only as the MFMA A-operand image in LDS (and, on steps that update the proposal networks,
// as the [N,8] fp32 tensor their backward kernels read).
//
// Arithmetic = the unfused kernels': the gather is kplanes_gather_fwd_kernel's (bilerp4, product over the planes in plane order), the net is
// mlp_lp_fwd_kernel<T,32,64,1,64> phase for phase (same operand images, same MFMA sequence), so densities are bit-identical to gather +
// snerf_mlp_fwd with the same 16-bit operands (tests/test_gpu_proposal_fused.py).
//
// Work decomposition: 256 threads = 4 waves per tile of 64 samples.  Gather: 4 lanes per sample, 2 channels (one 8-byte load) per lane and
// texel -- the four lanes of a sample read one 32-byte texel together; 24 independent loads per lane in flight.  Then the net: wave w owns hidden
// units 16 w .. 16 w + 15 for all 64 samples (one k-step: the 8 real inputs sit in a zero-padded K = 32), output layer by row block.
// ~22 KB of LDS: several workgroups per CU hide the gather's L2 latency; the proposal planes (2.7 / 8.7 MB) live in L2 / Infinity Cache.
#include "kplanes_common.hpp"
#include "mlp_lp_common.hpp"

namespace snerf {

constexpr int PF_TS = 64, PF_C = 8, PF_K0 = 32, PF_H = 64, PF_NW = 4;

struct PlanPF {
  static constexpr int LK0 = ldb(PF_K0), LKH = ldb(PF_H);
  static constexpr int W0T = 0;                      // [H][LK0]
  static constexpr int WOT = W0T + PF_H * LK0;       // [16][LKH]
  static constexpr int XS = WOT + 16 * LKH;          // [TS][LK0]  columns 8..31 stay zero
  static constexpr int A1 = XS + PF_TS * LK0;        // [TS][LKH]
  static constexpr int TOTAL = A1 + PF_TS * LKH;
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
};

struct DensityArgs {
  snerf_kplanes_desc d;
  const float* planes;
  snerf_coords c;
  int64_t N;
  const float* W;      // [8 x 64 | 64 x 1] row-major [in][out]
  int relu;            // hidden activation (the linear-decoder model's proposal fields have none)
  float* dens;         // [N] exp(net(features))
  float* feat;         // optional [N,8] fp32: the features, for the proposal net's / planes' backward kernels
};

template <typename T>
__global__ __launch_bounds__(PF_NW * 64) void density_fwd_kernel(DensityArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  using P = PlanPF;
  constexpr int MT = PF_TS / 16, HT = PF_H / 16;
  static_assert(HT == PF_NW && MT == PF_NW, "one hidden-unit block and one row block per wave");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stage_w<T>(a.W, PF_C, PF_H, PF_K0, PF_H, nullptr, 0, smem + P::W0T, P::LK0);
  stage_w<T>(a.W + PF_C * PF_H, PF_H, 1, PF_H, 16, nullptr, 0, smem + P::WOT, P::LKH);
  T* Xs = smem + P::XS;
  for (int idx = threadIdx.x; idx < PF_TS * (PF_K0 - PF_C); idx += blockDim.x) Xs[(idx / (PF_K0 - PF_C)) * P::LK0 + PF_C + idx % (PF_K0 - PF_C)] = (T)0.f;
  const snerf_kplanes_desc& d = a.d;
  const int srow = threadIdx.x >> 2, cg = threadIdx.x & 3;  // sample of the tile, channel pair
  int res[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) res[k] = d.res[0][k] > 0 ? d.res[0][k] : 1;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * PF_TS;
    // ---- gather: this lane's two channels of sample n0 + srow ----
    float2 prod = make_float2(0.f, 0.f);
    const int64_t n = n0 + srow;
    if (n < a.N) {
      float p[4];
      if (a.c.mode == 0) {
        const float4 v = *reinterpret_cast<const float4*>(a.c.pts + n * 4);
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
      } else {
        const uint32_t r = (uint32_t)n / (uint32_t)a.c.S;  // N < 2^31 (checked by the launcher)
        load_coords_ray(a.c, (int64_t)r, (int)((uint32_t)n - r * (uint32_t)a.c.S), p);
      }
      AxisTap tap[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) tap[k] = axis_tap(p[k], res[k]);
      float2 t[6][4];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const AxisTap& tx = tap[pair_a<6>(q)];
        const AxisTap& ty = tap[pair_b<6>(q)];
        const int W = d.res[0][pair_a<6>(q)];
        const float* base = a.planes + d.off[0][q] + cg * 2;
        const float* r0 = base + ((int64_t)ty.i0 * W) * PF_C;
        const float* r1 = base + ((int64_t)ty.i1 * W) * PF_C;
        t[q][0] = *reinterpret_cast<const float2*>(r0 + (int64_t)tx.i0 * PF_C);
        t[q][1] = *reinterpret_cast<const float2*>(r0 + (int64_t)tx.i1 * PF_C);
        t[q][2] = *reinterpret_cast<const float2*>(r1 + (int64_t)tx.i0 * PF_C);
        t[q][3] = *reinterpret_cast<const float2*>(r1 + (int64_t)tx.i1 * PF_C);
      }
      float2 pr = make_float2(1.f, 1.f);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const float4 w = tap_weights(tap[pair_a<6>(q)], tap[pair_b<6>(q)]);
        pr.x *= bilerp4(t[q][0].x, t[q][1].x, t[q][2].x, t[q][3].x, w.x, w.y, w.z, w.w);
        pr.y *= bilerp4(t[q][0].y, t[q][1].y, t[q][2].y, t[q][3].y, w.x, w.y, w.z, w.w);
      }
      prod = make_float2(0.f + pr.x, 0.f + pr.y);  // the unfused gather sums over its one scale from 0 (-0 -> +0)
      if (a.feat) *reinterpret_cast<float2*>(a.feat + n * PF_C + cg * 2) = prod;
    }
    __syncthreads();  // the previous tile's output phase has read A1; nobody reads Xs any more
    {
      typedef T v2 __attribute__((ext_vector_type(2)));
      const v2 b = {Ops<T>::cvt(prod.x), Ops<T>::cvt(prod.y)};
      *reinterpret_cast<v2*>(Xs + srow * P::LK0 + cg * 2) = b;
    }
    __syncthreads();
    // ---- hidden layer (mlp_lp_fwd_kernel<T,32,64,1,64>: wave = column block) ----
    {
      f32x4 acc[MT] = {};
      mma_rr<MT, PF_K0>(Xs, P::LK0, smem + P::W0T, P::LK0, wave, acc, lane);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4 v = acc[m];
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        store_rt<T>(smem + P::A1, P::LKH, nullptr, 0, m, wave, v, lane);
      }
    }
    __syncthreads();
    // ---- output layer + trunc_exp (wave = row block) ----
    {
      f32x4 acc[1] = {};
      mma_rr<1, PF_H>(smem + P::A1 + wave * 16 * P::LKH, P::LKH, smem + P::WOT, P::LKH, 0, acc, lane);
      if ((lane & 15) == 0) {
        const int64_t row0 = n0 + wave * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (row0 + r < a.N) a.dens[row0 + r] = expf(acc[0][r]);  // trunc_exp forward (activations.py:32)
      }
    }
  }
}

template <typename T>
static int launch_density(const DensityArgs& a, hipStream_t st) {
  const int64_t n_tiles = (a.N + PF_TS - 1) / PF_TS;
  int64_t grid = 256 * 6;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL((density_fwd_kernel<T>), dim3((unsigned)grid), dim3(PF_NW * 64), PlanPF::BYTES, st, a, n_tiles);
  SNERF_LAUNCH_CHECK("kplanes_density_fwd");
  return 0;
}

static bool density_shape_ok(const snerf_kplanes_desc* d, const snerf_mlp_desc* m) {
  return d && m && d->C == PF_C && d->n_coords == 4 && d->n_scales == 1 && m->d_in == PF_C && m->hidden == PF_H && m->n_hidden == 1 && m->d_out == 1 &&
         (m->operands == 1 || m->operands == 2) && m->out_act == 0 && (m->hidden_act == 0 || m->hidden_act == 1);
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_kplanes_density_fwd_supported(const snerf_kplanes_desc* desc, const snerf_mlp_desc* net) { return density_shape_ok(desc, net) ? 1 : 0; }

extern "C" int snerf_kplanes_density_fwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N, const snerf_mlp_desc* net,
                                         const float* W, float* density, float* feat, snerf_stream_t stream) {
  SNERF_REQUIRE(desc && coords && net, "kplanes_density_fwd: null descriptor");
  SNERF_REQUIRE(density_shape_ok(desc, net), "kplanes_density_fwd: built for one scale of six C = 8 planes and the 8 -> 64 -> 1 net with 16-bit operands "
                "(C=%d n_coords=%d n_scales=%d; net %d -> %d x %d -> %d, operands %d)", desc->C, desc->n_coords, desc->n_scales, net->d_in, net->hidden,
                net->n_hidden, net->d_out, net->operands);
  SNERF_REQUIRE(N >= 0 && N < (1LL << 31), "kplanes_density_fwd: N=%lld", (long long)N);
  SNERF_REQUIRE(coords->mode == 0 || coords->mode == 1, "kplanes_density_fwd: coords.mode=%d", coords->mode);
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && W && density, "kplanes_density_fwd: null buffer");
  SNERF_REQUIRE(coords->mode == 0 ? coords->pts != nullptr : (coords->origins && coords->dirs && coords->ebins && coords->times && coords->S > 0),
                "kplanes_density_fwd: incomplete coordinates");
  DensityArgs a = {};
  a.d = *desc; a.planes = planes; a.c = *coords; a.N = N; a.W = W; a.relu = net->hidden_act == 1; a.dens = density; a.feat = feat;
  return net->operands == 2 ? launch_density<fp16>(a, (hipStream_t)stream) : launch_density<bf16>(a, (hipStream_t)stream);
}

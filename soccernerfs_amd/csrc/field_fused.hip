// Fused K-Planes field forward: plane gather -> sigma_net -> trunc_exp density, colour net -> rgb in ONE kernel.
// (A fused backward -- recomputed forward -> both nets' backward -> per-plane gradient vectors -- was built and parity-tested in round 2 and
// removed in round 3: 256 VGPRs + spills, 2.5 ms against 0.85 ms for the three unfused kernels, and with the quotient form of the scatter
// it would have saved only ~0.4 GB of the backward's traffic.  DESIGN.md section 4.2.)
//
// Replaces KPlanesField.get_density + get_outputs (NS/fields/kplanes_field.py:275-358): interpolate_kplanes (:77-126), sigma_net
// (:249-261: features 32 n_scales -> 128 -> 16), trunc_exp on the last output (:308-311), color_net on the 15 geometry features
// (:263-273 with disable_viewing_dependent: 15 -> 64 -> 64 -> 3, Sigmoid).  In the unfused path feat[N,160], h[N,16], gh[N,16] and
// gfeat[N,160] cross HBM between five launches; here a tile of 32 samples lives in LDS from the texel reads to density / rgb.  Only what
// other kernels need touches HBM: density [N], rgb [N,3] (compositing) and, for a training step, what the unfused backward kernels read:
// the operand-typed feature tile, the 16 sigma_net outputs, and the fp32 features (the quotient scatter's numerator).
//
// 16-bit MFMA operands with fp32 accumulation (v_mfma_f32_16x16x32_bf16 / _f16), the arithmetic of mlp_lp.hip: the fused kernels give
// bit-identical density / rgb / weight gradients to the unfused 16-bit kernels fed the same planes (tests/test_gpu_field_fused.py).
// The exact-fp32 parity path stays unfused (mlp.hip).
//
// Work decomposition (512 threads = 8 waves, tile = 32 samples, persistent grid; two workgroups per CU so that one workgroup's
// gather phase -- HBM / L2 latency bound -- runs under the other's MFMA phases):
//   gather   16 lanes per sample: (4-channel group) x (scale parity, by wave); float4 texel reads as kplanes_gather_fwd_kernel, the Hadamard
//            product of the six planes in registers, rounded once into the LDS A-operand image X [32][32 n_scales + 8]
//   sigma 0  wave w owns hidden units 16w..16w+15; ITS B operand of W0 (32 n_scales / 32 k-steps x 8 values) lives in registers for the
//            whole persistent loop, so the 160 x 128 matrix never enters LDS
//   sigma 1, colour 0/1/out: small products from LDS-resident weights (21 KB).
#include "kplanes_common.hpp"
#include "mlp_lp_common.hpp"

namespace snerf {

constexpr int FF_TS = 32;     // samples per tile
constexpr int FF_H = 128;     // sigma_net hidden width
constexpr int FF_HC = 64;     // color_net hidden width
constexpr int FF_GEO = 15;    // geometry features = colour-net inputs; output column 15 of sigma_net is the density pre-activation
constexpr int FF_NW = 8;      // waves per workgroup

struct FieldArgs {
  snerf_kplanes_desc d;
  const float* planes;
  snerf_coords c;
  int64_t N;
  const float* Wsig;  // [K0 x 128 | 128 x 16] row-major [in][out]
  const float* Wcol;  // [15 x 64 | 64 x 64 | 64 x 3]
  float* dens;        // [N]   exp(sigma_net(.)[15])
  float* rgb;         // [N,3] sigmoid(color_net(.))
  void* feat16;       // optional [N, 32 n_scales] in the operand type: the rounded feature tile, for an UNFUSED backward (snerf_mlp_bwd_x16)
  float* h;           // optional [N,16]: the raw sigma_net outputs (color_net's input, column 15 = log density)
  float* feat32;      // optional [N, 32 n_scales] fp32: the features before rounding, for the quotient form of the plane scatter
};

template <int NS>
struct PlanFF {
  static constexpr int K0 = 32 * NS, LK0 = ldb(K0), LKH = ldb(FF_H), LKC = ldb(FF_HC), LKX = ldb(32);
  static constexpr int SWOT = 0;                          // sigma out  [16][LKH]
  static constexpr int CW0T = SWOT + 16 * LKH;            // colour L0  [64][LKX]
  static constexpr int CW1T = CW0T + FF_HC * LKX;         // colour L1  [64][LKC]
  static constexpr int CWOT = CW1T + FF_HC * LKC;         // colour out [16][LKC]
  static constexpr int CX = CWOT + 16 * LKC;              // colour input tile [TS][LKX] (columns 15..31 stay zero)
  static constexpr int A1 = CX + FF_TS * LKX;             // sigma hidden tile [TS][LKH]
  static constexpr int XS = A1 + FF_TS * LKH;             // feature tile [TS][LK0]; the colour hidden tiles reuse it once sigma layer 0 is done
  static constexpr int XS_LEN = FF_TS * LK0 > 2 * FF_TS * LKC ? FF_TS * LK0 : 2 * FF_TS * LKC;
  static constexpr int TOTAL = XS + XS_LEN;
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
};

// features of one sample for this lane's 4 channels and scale s: product over the six planes (interpolate_kplanes, kplanes_field.py:77-126)
__device__ __forceinline__ float4 scale_features(const snerf_kplanes_desc& d, const float* __restrict__ planes, const float p[4], int s, int cg,
                                                 float4 (*v_out)[6] = nullptr) {
  AxisTap tap[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) tap[k] = axis_tap(p[k], d.res[s][k] > 0 ? d.res[s][k] : 1);
  float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const float4 v = plane_sample<32>(planes + d.off[s][q], d.res[s][pair_a<6>(q)], tap[pair_a<6>(q)], tap[pair_b<6>(q)], cg);
    if (v_out) (*v_out)[q] = v;
    prod = f4_mul(prod, v);
  }
  return prod;
}

template <typename T>
__device__ __forceinline__ void relu4(f32x4& v) {
  v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
}

// this wave's B operand of sigma_net layer 0: hidden units 16 wave .. +15, k = 32 ks + 8 (lane >> 4) .. +8
template <typename T, int K0>
__device__ __forceinline__ void load_breg(const float* __restrict__ W0, int wave, int lane, typename Ops<T>::v8 (&breg)[K0 / 32]) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < K0 / 32; ++ks) {
    typename Ops<T>::v8 b;
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = Ops<T>::cvt(W0[(int64_t)(ks * 32 + lk * 8 + e) * FF_H + wave * 16 + lr]);
    breg[ks] = b;
  }
}

// gather phase: 16 lanes per sample = (4-channel group cg) x (scale parity sg); rounds the features into the A-operand image XS
template <typename T, int NS, bool F32OUT = false>
__device__ __forceinline__ void gather_tile(const FieldArgs& a, int64_t n0, T* XS) {
  using P = PlanFF<NS>;
  // waves 0-3 take the even scales, waves 4-7 the odd ones: the scale index is wave-uniform, so the descriptor reads stay scalar
  const int sample = (threadIdx.x & 255) >> 3, cg = threadIdx.x & 7, sg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int64_t n = n0 + sample;
  float p[4];
  const bool live = n < a.N;
  if (live) load_coords<6>(a.c, n, p);
#pragma unroll 1  // one scale's 24 texel reads in flight at a time: unrolled, the three scales' loads cost ~160 VGPRs and the occupancy the kernel lives on
  for (int s = sg; s < NS; s += 2) {
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) f = scale_features(a.d, a.planes, p, s, cg);
    const typename Ops<T>::v4 t = {Ops<T>::cvt(f.x), Ops<T>::cvt(f.y), Ops<T>::cvt(f.z), Ops<T>::cvt(f.w)};
    *reinterpret_cast<typename Ops<T>::v4*>(XS + sample * P::LK0 + s * 32 + cg * 4) = t;
    if constexpr (F32OUT)
      if (live) *reinterpret_cast<float4*>(a.feat32 + n * (32 * NS) + s * 32 + cg * 4) = f;  // 8 lanes x 16 B = one 128-B row segment
  }
}

// sigma_net layer 0 from the register-resident B operand: A1 = relu(X W0), column block = wave
template <typename T, int NS>
__device__ __forceinline__ void sigma_layer0(const T* XS, const typename Ops<T>::v8 (&breg)[NS], T* A1, T* A1t, int ldt, int wave, int lane) {
  using P = PlanFF<NS>;
  constexpr int MT = FF_TS / 16;
  const int lr = lane & 15, lk = lane >> 4;
  f32x4 acc[MT] = {};
#pragma unroll
  for (int ks = 0; ks < NS; ++ks)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = Ops<T>::mfma(ld8(XS + (m * 16 + lr) * P::LK0 + ks * 32 + lk * 8), breg[ks], acc[m]);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    relu4<T>(acc[m]);
    store_rt<T>(A1, P::LKH, A1t, ldt, m, wave, acc[m], lane);
  }
}

// KEEP: what a training step leaves behind for its backward -- 0 nothing (eval), 1 the 16-bit feature tile + the sigma_net outputs,
// 2 those + the fp32 features (quotient scatter).  Compile-time: as run-time pointer tests these cost every variant registers.
template <typename T, int NS, int KEEP = 0>
__global__ __launch_bounds__(FF_NW * 64, 4) void field_fwd_kernel(FieldArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  using P = PlanFF<NS>;
  constexpr int K0 = P::K0, MT = FF_TS / 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T *XS = smem + P::XS, *A1 = smem + P::A1, *CX = smem + P::CX, *CA1 = smem + P::XS, *CA2 = smem + P::XS + FF_TS * P::LKC;
  typename Ops<T>::v8 breg[NS];
  load_breg<T, K0>(a.Wsig, wave, lane, breg);
  stage_w<T>(a.Wsig + K0 * FF_H, FF_H, 16, FF_H, 16, nullptr, 0, smem + P::SWOT, P::LKH);
  stage_w<T>(a.Wcol, FF_GEO, FF_HC, 32, FF_HC, nullptr, 0, smem + P::CW0T, P::LKX);
  stage_w<T>(a.Wcol + FF_GEO * FF_HC, FF_HC, FF_HC, FF_HC, FF_HC, nullptr, 0, smem + P::CW1T, P::LKC);
  stage_w<T>(a.Wcol + FF_GEO * FF_HC + FF_HC * FF_HC, FF_HC, 3, FF_HC, 16, nullptr, 0, smem + P::CWOT, P::LKC);
  for (int idx = threadIdx.x; idx < FF_TS * P::LKX; idx += blockDim.x) CX[idx] = (T)0.f;  // columns 15..31 stay zero for the whole kernel
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * FF_TS;
    __syncthreads();  // weights staged / the previous tile's colour layers have read CA1, CA2 (= XS)
    gather_tile<T, NS, KEEP == 2>(a, n0, XS);
    __syncthreads();
    if constexpr (KEEP >= 1) {  // the tile's rows are contiguous in feat16: one coalesced 16-B store per 8 features
      T* F16 = reinterpret_cast<T*>(a.feat16);
      for (int vi = threadIdx.x; vi < FF_TS * (K0 / 8); vi += FF_NW * 64) {
        const int r = vi / (K0 / 8), c8 = vi - r * (K0 / 8);
        if (n0 + r < a.N) *reinterpret_cast<typename Ops<T>::v8*>(F16 + (n0 + r) * K0 + c8 * 8) = ld8(XS + r * P::LK0 + c8 * 8);
      }
    }
    sigma_layer0<T, NS>(XS, breg, A1, nullptr, 0, wave, lane);
    __syncthreads();
    if (wave < MT) {  // sigma_net output layer: 16 columns; column 15 -> density, columns 0..14 -> colour-net input
      f32x4 acc[1] = {};
      mma_rr<1, FF_H>(A1 + wave * 16 * P::LKH, P::LKH, smem + P::SWOT, P::LKH, 0, acc, lane);
      const int col = lane & 15, row0 = wave * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = acc[0][r];
        const int64_t n = n0 + row0 + r;
        if (col == FF_GEO && n < a.N) a.dens[n] = expf(y);  // trunc_exp forward (activations.py:32)
        if constexpr (KEEP >= 1)
          if (n < a.N) a.h[n * 16 + col] = y;
        CX[(row0 + r) * P::LKX + col] = col < FF_GEO ? Ops<T>::cvt(y) : (T)0.f;
      }
    }
    __syncthreads();
    {  // colour layer 0: K = 32 (15 used), 4 column blocks x MT row blocks = 8 blocks, one per wave
      f32x4 acc[1] = {};
      const int nt = wave & 3, mt = wave >> 2;
      mma_rr<1, 32>(CX + mt * 16 * P::LKX, P::LKX, smem + P::CW0T, P::LKX, nt, acc, lane);
      relu4<T>(acc[0]);
      store_rt<T>(CA1, P::LKC, nullptr, 0, mt, nt, acc[0], lane);
    }
    __syncthreads();
    {
      f32x4 acc[1] = {};
      const int nt = wave & 3, mt = wave >> 2;
      mma_rr<1, FF_HC>(CA1 + mt * 16 * P::LKC, P::LKC, smem + P::CW1T, P::LKC, nt, acc, lane);
      relu4<T>(acc[0]);
      store_rt<T>(CA2, P::LKC, nullptr, 0, mt, nt, acc[0], lane);
    }
    __syncthreads();
    if (wave < MT) {
      f32x4 acc[1] = {};
      mma_rr<1, FF_HC>(CA2 + wave * 16 * P::LKC, P::LKC, smem + P::CWOT, P::LKC, 0, acc, lane);
      const int col = lane & 15;
      const int64_t row0 = n0 + wave * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col < 3 && row0 + r < a.N) a.rgb[(row0 + r) * 3 + col] = 1.f / (1.f + expf(-acc[0][r]));
    }
  }
}

static int validate_field(const snerf_kplanes_desc* d, const snerf_coords* c, int64_t N, const snerf_mlp_desc* sd, const snerf_mlp_desc* cd,
                          int max_scales = 6) {
  SNERF_REQUIRE(d && c && sd && cd, "kplanes_field: null descriptor");
  SNERF_REQUIRE(d->C == 32 && d->n_coords == 4 && d->concat == 1 && d->n_scales >= 1 && d->n_scales <= max_scales,
                "kplanes_field: the fused kernels are built for 4-D planes, C = 32, concatenated scales (<= %d); got C=%d coords=%d concat=%d scales=%d",
                max_scales, d->C, d->n_coords, d->concat, d->n_scales);
  SNERF_REQUIRE(sd->d_in == 32 * d->n_scales && sd->hidden == FF_H && sd->n_hidden == 1 && sd->d_out == 16 && sd->hidden_act == 1 && sd->out_act == 0,
                "kplanes_field: sigma_net must be %d -> 128 (ReLU) -> 16", 32 * d->n_scales);
  SNERF_REQUIRE(cd->d_in == FF_GEO && cd->hidden == FF_HC && cd->n_hidden == 2 && cd->d_out == 3 && cd->hidden_act == 1 && cd->out_act == 1,
                "kplanes_field: color_net must be 15 -> 64 -> 64 (ReLU) -> 3 (Sigmoid)");
  SNERF_REQUIRE((sd->operands == 1 || sd->operands == 2) && cd->operands == sd->operands,
                "kplanes_field: the fused kernels compute with bf16 / fp16 MFMA operands (operands = 1 / 2, both nets alike); fp32 runs unfused");
  SNERF_REQUIRE(N >= 0 && N < (1LL << 31), "kplanes_field: N=%lld", (long long)N);
  SNERF_REQUIRE(c->mode == 0 || c->mode == 1, "kplanes_field: coords.mode=%d", c->mode);
  if (c->mode == 1) SNERF_REQUIRE(c->S >= 1 && N % c->S == 0, "kplanes_field: N=%lld not a multiple of S=%d", (long long)N, c->S);
  return 0;
}

template <typename T, int NS, int KEEP>
static int launch_field_fwd_k(const FieldArgs& a, hipStream_t st) {
  using P = PlanFF<NS>;
  const int64_t n_tiles = (a.N + FF_TS - 1) / FF_TS;
  int per_cu = (int)(LDS_LIMIT_B / P::BYTES);
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);  // 125 VGPRs: two 8-wave workgroups per CU
  int64_t grid = 256 * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  auto k = field_fwd_kernel<T, NS, KEEP>;
  SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(FF_NW * 64), P::BYTES, st, a, n_tiles);
  SNERF_LAUNCH_CHECK("kplanes_field_fwd");
  return 0;
}
template <typename T, int NS>
static int launch_field_fwd(const FieldArgs& a, hipStream_t st) {
  if (a.feat16 && a.feat32) return launch_field_fwd_k<T, NS, 2>(a, st);
  if (a.feat16) return launch_field_fwd_k<T, NS, 1>(a, st);
  return launch_field_fwd_k<T, NS, 0>(a, st);
}


#define FF_DISPATCH(FN, operands, ns, ...)                                                   \
  do {                                                                                       \
    if ((operands) == 2) {                                                                   \
      switch (ns) {                                                                          \
        case 1: return FN<fp16, 1>(__VA_ARGS__); case 2: return FN<fp16, 2>(__VA_ARGS__);    \
        case 3: return FN<fp16, 3>(__VA_ARGS__); case 4: return FN<fp16, 4>(__VA_ARGS__);    \
        default: return FN<fp16, 5>(__VA_ARGS__);                                            \
      }                                                                                      \
    }                                                                                        \
    switch (ns) {                                                                            \
      case 1: return FN<bf16, 1>(__VA_ARGS__); case 2: return FN<bf16, 2>(__VA_ARGS__);      \
      case 3: return FN<bf16, 3>(__VA_ARGS__); case 4: return FN<bf16, 4>(__VA_ARGS__);      \
      default: return FN<bf16, 5>(__VA_ARGS__);                                              \
    }                                                                                        \
  } while (0)

// six scales as well (BASELINE config 3: K0 = 192)
#define FF_DISPATCH_FWD(FN, operands, ns, ...)                                               \
  do {                                                                                       \
    if ((ns) == 6) {                                                                         \
      if ((operands) == 2) return FN<fp16, 6>(__VA_ARGS__);                                  \
      return FN<bf16, 6>(__VA_ARGS__);                                                       \
    }                                                                                        \
    FF_DISPATCH(FN, operands, ns, __VA_ARGS__);                                              \
  } while (0)

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_kplanes_field_fwd_supported(const snerf_kplanes_desc* desc, const snerf_mlp_desc* sigma, const snerf_mlp_desc* color) {
  snerf_coords c = {};
  return desc && sigma && color && validate_field(desc, &c, 0, sigma, color, 6) == 0 ? 1 : 0;
}

extern "C" int snerf_kplanes_field_fwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                                       const snerf_mlp_desc* sigma, const float* W_sigma, const snerf_mlp_desc* color, const float* W_color,
                                       float* density, float* rgb, void* feat16, float* h, float* feat32, snerf_stream_t stream) {
  int rc = validate_field(desc, coords, N, sigma, color, 6);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && W_sigma && W_color && density && rgb, "kplanes_field_fwd: null buffer");
  FieldArgs a = {};
  a.d = *desc; a.planes = planes; a.c = *coords; a.N = N; a.Wsig = W_sigma; a.Wcol = W_color; a.dens = density; a.rgb = rgb;
  SNERF_REQUIRE((feat16 != nullptr) == (h != nullptr) && (!feat32 || feat16),
                "kplanes_field_fwd: the training outputs come as a set: feat16 and h together, feat32 only with them");
  a.feat16 = feat16; a.h = h; a.feat32 = feat32;
  FF_DISPATCH_FWD(launch_field_fwd, sigma->operands, desc->n_scales, a, (hipStream_t)stream);
}

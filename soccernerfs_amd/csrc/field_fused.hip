// Fused K-Planes field: plane gather -> sigma_net -> trunc_exp density, colour net -> rgb in ONE kernel, and its backward
// (recomputed forward -> colour/sigma net backward -> per-plane gradient vectors) in one kernel.
//
// Replaces KPlanesField.get_density + get_outputs (NS/fields/kplanes_field.py:275-358): interpolate_kplanes (:77-126), sigma_net
// (:249-261: features 32 n_scales -> 128 -> 16), trunc_exp on the last output (:308-311), color_net on the 15 geometry features
// (:263-273 with disable_viewing_dependent: 15 -> 64 -> 64 -> 3, Sigmoid).  In the unfused path feat[N,160], h[N,16], gh[N,16] and
// gfeat[N,160] cross HBM between five launches; here a tile of 32 samples lives in LDS from the texel reads to density / rgb (forward) and
// from the recomputed forward to the gradient vectors of the 30 planes (backward).  Only what other kernels need touches HBM:
// density [N], rgb [N,3] (compositing), and gvec [30][N][32] (the sorted scatter walks it in plane order, kplanes_sorted.hip).
//
// 16-bit MFMA operands with fp32 accumulation (v_mfma_f32_16x16x32_bf16 / _f16), the arithmetic of mlp_lp.hip: the fused kernels give
// bit-identical density / rgb / weight gradients to the unfused 16-bit kernels fed the same planes (tests/test_gpu_field_fused.py).
// The exact-fp32 parity path stays unfused (mlp.hip).
//
// Work decomposition (512 threads = 8 waves, tile = 32 samples, persistent grid; forward: two workgroups per CU so that one workgroup's
// gather phase -- HBM / L2 latency bound -- runs under the other's MFMA phases; backward: one, its weights + tiles fill the LDS):
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
  // backward
  const float* gdens;   // [N]   dL/d density
  const float* grgb;    // [N,3] dL/d rgb
  float* gWsig; float* gWcol;          // weight gradients, accumulated (float atomics) ...
  long long* gWsig_fx; long long* gWcol_fx;  // ... or as fixed point (deterministic mode)
  void* gvec; int gvec_bf16;           // [n_scales * 6][N][32] per-plane gradient vectors (pass A of the sorted scatter)
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

// accumulator block -> row-major and transposed images with OPERAND rounding (activations; store_rt rounds gradient tiles)
template <typename T>
__device__ __forceinline__ void store_rt_op(T* R, int ldr, T* Tr, int ldt, int mt, int nt, const f32x4& v, int lane) {
  const int col = nt * 16 + (lane & 15);
  const int row0 = mt * 16 + (lane >> 4) * 4;
  const typename Ops<T>::v4 t = {Ops<T>::cvt(v[0]), Ops<T>::cvt(v[1]), Ops<T>::cvt(v[2]), Ops<T>::cvt(v[3])};
#pragma unroll
  for (int r = 0; r < 4; ++r) R[(row0 + r) * ldr + col] = t[r];
  if (Tr) *reinterpret_cast<typename Ops<T>::v4*>(Tr + col * ldt + row0) = t;
}

// sigma layer 0 for the backward's plan (row-major + transposed hidden tile)
template <typename T, int NS, typename P>
__device__ __forceinline__ void sigma_layer0_b(const T* XS, const typename Ops<T>::v8 (&breg)[NS], T* A1, T* A1t, int wave, int lane) {
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
    store_rt_op<T>(A1, P::LKH, A1t, P::LKT, m, wave, acc[m], lane);
  }
}

typedef __bf16 ff_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_gq(float* gv, int64_t idx, float4 g) { *reinterpret_cast<float4*>(gv + idx) = g; }
__device__ __forceinline__ void store_gq(__bf16* gv, int64_t idx, float4 g) {
  const ff_bf16x4 b = {(__bf16)g.x, (__bf16)g.y, (__bf16)g.z, (__bf16)g.w};
  *reinterpret_cast<ff_bf16x4*>(gv + idx) = b;
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
                          int max_scales = 5) {
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
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_B); attr_set = true; }
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


// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int NS>
struct PlanFB {
  static constexpr int K0 = 32 * NS, LK0 = ldb(K0), LKH = ldb(FF_H), LKC = ldb(FF_HC), LKX = ldb(32), LKO = ldb(32), LKT = ldb(FF_TS);
  // weights (elements of T)
  static constexpr int SWOT = 0;                        // sigma out, forward     [16][LKH]
  static constexpr int SWOR = SWOT + 16 * LKH;          // sigma out as stored    [H][LKO]     dZ1 = gzo WO^T
  static constexpr int CW0T = SWOR + FF_H * LKO;        // colour L0, forward     [64][LKX]
  static constexpr int CW0R = CW0T + FF_HC * LKX;       // colour L0 as stored    [32][LKC]    gCX = gz1c W0^T
  static constexpr int CW1T = CW0R + 32 * LKC;          // colour L1, forward     [64][LKC]
  static constexpr int CW1R = CW1T + FF_HC * LKC;       // colour L1 as stored    [64][LKC]    gz1c = gz2c W1^T
  static constexpr int CWOT = CW1R + FF_HC * LKC;       // colour out, forward    [16][LKC]
  static constexpr int CWOR = CWOT + 16 * LKC;          // colour out as stored   [64][LKO]    gz2c = gzo WO^T
  // sigma tiles
  static constexpr int XS = CWOR + FF_HC * LKO;         // features               [TS][LK0]
  static constexpr int XT = XS + FF_TS * LK0;           // features transposed    [K0][LKT]    dW0 = X^T gz1
  static constexpr int A1 = XT + K0 * LKT;              // hidden                 [TS][LKH]
  static constexpr int A1T = A1 + FF_TS * LKH;          // hidden transposed      [H][LKT]     dWO = A1^T gzo; then gz1 transposed
  static constexpr int GZ1 = A1T + FF_H * LKT;          // gradient of Z1         [TS][LKH]
  static constexpr int SGZO = GZ1 + FF_TS * LKH;        // gradient of the 16 sigma outputs [TS][LKO] (columns 16..31 zero)
  static constexpr int SGZOT = SGZO + FF_TS * LKO;      //   transposed           [16][LKT]
  // colour tiles; the fp32 feature-gradient tile GF [TS][K0 + 4] reuses this region once the colour backward is done
  static constexpr int CX = SGZOT + 16 * LKT;           // colour input           [TS][LKX] (columns 15..31 zero)
  static constexpr int CXT = CX + FF_TS * LKX;          //   transposed           [32][LKT]
  static constexpr int CA1 = CXT + 32 * LKT;            // hidden 1               [TS][LKC]
  static constexpr int CA1T = CA1 + FF_TS * LKC;        //   transposed           [64][LKT]; then gz1c transposed
  static constexpr int CA2 = CA1T + FF_HC * LKT;        // hidden 2               [TS][LKC]
  static constexpr int CA2T = CA2 + FF_TS * LKC;        //   transposed           [64][LKT]; then gz2c transposed
  static constexpr int GZC2 = CA2T + FF_HC * LKT;       // gradient of Z2         [TS][LKC]
  static constexpr int GZC1 = GZC2 + FF_TS * LKC;       // gradient of Z1         [TS][LKC]
  static constexpr int CGZO = GZC1 + FF_TS * LKC;       // gradient of the 3 outputs [TS][LKO] (columns 3..31 zero)
  static constexpr int CGZOT = CGZO + FF_TS * LKO;      //   transposed           [16][LKT]
  static constexpr int TOTAL = CGZOT + 16 * LKT;
  static constexpr int LGF = K0 + 4;                    // GF row stride (floats)
  static_assert((TOTAL - CX) * 2 >= FF_TS * LGF * 4, "the feature-gradient tile must fit the colour region");
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
};

__device__ __forceinline__ void gw_add2(float* g, long long* gfx, int64_t idx, float v) {
  if (gfx) fx_atomic_add(gfx + idx, v); else atomicAdd(g + idx, v);
}

template <typename T, int NS, typename GV>
__global__ __launch_bounds__(FF_NW * 64, 2) void field_bwd_kernel(FieldArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  using P = PlanFB<NS>;
  constexpr float GS = Ops<T>::GS;
  constexpr int K0 = P::K0, MT = FF_TS / 16, K0T = K0 / 16, HT = FF_H / 16, HCT = FF_HC / 16;
  constexpr int NB0 = (K0T * HT + FF_NW - 1) / FF_NW;   // dW0 blocks of sigma_net per wave (10 at 5 scales)
  constexpr int NBX = (K0T + FF_NW - 1) / FF_NW;        // gX column blocks per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  T *XS = smem + P::XS, *XT = smem + P::XT, *A1 = smem + P::A1, *A1T = smem + P::A1T, *GZ1 = smem + P::GZ1, *GZ1T = smem + P::A1T;
  T *SGZO = smem + P::SGZO, *SGZOT = smem + P::SGZOT;
  T *CX = smem + P::CX, *CXT = smem + P::CXT, *CA1 = smem + P::CA1, *CA1T = smem + P::CA1T, *CA2 = smem + P::CA2, *CA2T = smem + P::CA2T;
  T *GZC2 = smem + P::GZC2, *GZC2T = smem + P::CA2T, *GZC1 = smem + P::GZC1, *GZC1T = smem + P::CA1T, *CGZO = smem + P::CGZO, *CGZOT = smem + P::CGZOT;
  float* GF = reinterpret_cast<float*>(smem + P::CX);
  const float* Wc0 = a.Wcol;
  const float* Wc1 = a.Wcol + FF_GEO * FF_HC;
  const float* Wco = Wc1 + FF_HC * FF_HC;
  const float* Wso = a.Wsig + K0 * FF_H;
  // ---- weights: LDS images + the two register-resident B operands of the 160 x 128 matrix ----
  stage_w<T>(Wso, FF_H, 16, FF_H, 16, nullptr, 0, smem + P::SWOT, P::LKH);
  stage_w<T>(Wso, FF_H, 16, FF_H, 32, smem + P::SWOR, P::LKO, nullptr, 0);
  stage_w<T>(Wc0, FF_GEO, FF_HC, 32, FF_HC, smem + P::CW0R, P::LKC, smem + P::CW0T, P::LKX);
  stage_w<T>(Wc1, FF_HC, FF_HC, FF_HC, FF_HC, smem + P::CW1R, P::LKC, smem + P::CW1T, P::LKC);
  stage_w<T>(Wco, FF_HC, 3, FF_HC, 16, nullptr, 0, smem + P::CWOT, P::LKC);
  stage_w<T>(Wco, FF_HC, 3, FF_HC, 32, smem + P::CWOR, P::LKO, nullptr, 0);
  // The 160 x 128 matrix W0 of sigma_net does not fit LDS beside everything else.  Both of its B-operand images -- forward (this wave's 16
  // hidden units, k over the features) and dX = gz1 W0^T (this wave's feature columns, k over the hidden units) -- are re-read from
  // global memory (80 KB, L2-resident) and rounded right before the product that uses them, once per tile: kept in registers across the
  // persistent loop they cost 52 VGPRs that the gather phases need (105-138 spilled registers otherwise).
  // zero the padded columns that stay zero for the whole kernel: sigma gzo 16..31, colour gzo 3..31 (all of it once), CX 15..31
  for (int idx = threadIdx.x; idx < FF_TS * P::LKO; idx += blockDim.x) { SGZO[idx] = (T)0.f; CGZO[idx] = (T)0.f; }
  for (int idx = threadIdx.x; idx < 16 * P::LKT; idx += blockDim.x) { SGZOT[idx] = (T)0.f; CGZOT[idx] = (T)0.f; }
  // weight-gradient accumulators, alive across the persistent loop
  f32x4 dW0s[NB0] = {};
  f32x4 dWOs = {};          // block it = wave
  f32x4 dW0c = {};          // [32 x 64] = 2 x 4 blocks: it = wave >> 2, nt = wave & 3
  f32x4 dW1c[2] = {};       // [64 x 64] = 16 blocks: t = wave + 8 j
  f32x4 dWOc = {};          // [64 x 16]: it = wave (waves 0..3)

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * FF_TS;
    __syncthreads();  // weights staged / the previous tile's gradvec phase has read GF (= the colour region) and XS is free again
    // the colour region was overwritten by GF: re-zero the padded columns of CX (15..31) and CGZO / CGZOT
    for (int idx = threadIdx.x; idx < FF_TS * P::LKX; idx += blockDim.x) CX[idx] = (T)0.f;
    for (int idx = threadIdx.x; idx < FF_TS * P::LKO; idx += blockDim.x) CGZO[idx] = (T)0.f;
    for (int idx = threadIdx.x; idx < 16 * P::LKT; idx += blockDim.x) CGZOT[idx] = (T)0.f;
    for (int idx = threadIdx.x; idx < 32 * P::LKT; idx += blockDim.x) CXT[idx] = (T)0.f;
    // ---- P1 gather: X and X^T ----
    {
      const int sample = (threadIdx.x & 255) >> 3, cg = threadIdx.x & 7, sg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
      const int64_t n = n0 + sample;
      float p[4];
      const bool live = n < a.N;
      if (live) load_coords<6>(a.c, n, p);
#pragma unroll 1
      for (int s = sg; s < NS; s += 2) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) f = scale_features(a.d, a.planes, p, s, cg);
        const T t0 = Ops<T>::cvt(f.x), t1 = Ops<T>::cvt(f.y), t2 = Ops<T>::cvt(f.z), t3 = Ops<T>::cvt(f.w);
        const typename Ops<T>::v4 t = {t0, t1, t2, t3};
        const int c0 = s * 32 + cg * 4;
        *reinterpret_cast<typename Ops<T>::v4*>(XS + sample * P::LK0 + c0) = t;
        XT[(c0 + 0) * P::LKT + sample] = t0; XT[(c0 + 1) * P::LKT + sample] = t1;
        XT[(c0 + 2) * P::LKT + sample] = t2; XT[(c0 + 3) * P::LKT + sample] = t3;
      }
    }
    __syncthreads();
    // ---- P2 sigma layer 0: A1, A1^T ----
    {
      typename Ops<T>::v8 breg[NS];
      load_breg<T, K0>(a.Wsig, wave, lane, breg);
      sigma_layer0_b<T, NS, P>(XS, breg, A1, A1T, wave, lane);
    }
    __syncthreads();
    // ---- P3 sigma output layer: colour input CX / CX^T; gradient of the density column ----
    if (wave < MT) {
      f32x4 acc[1] = {};
      mma_rr<1, FF_H>(A1 + wave * 16 * P::LKH, P::LKH, smem + P::SWOT, P::LKH, 0, acc, lane);
      const int col = lane & 15, row0 = wave * 16 + (lane >> 4) * 4;
      f32x4 cx, gd = {};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = acc[0][r];
        const int64_t n = n0 + row0 + r;
        cx[r] = col < FF_GEO ? y : 0.f;
        if (col == FF_GEO && n < a.N) gd[r] = a.gdens[n] * expf(fminf(fmaxf(y, -15.f), 15.f)) * GS;  // trunc_exp backward (activations.py:38-39)
      }
      // colour input: operand rounding (cvt), not the gradient-tile scaling
      const typename Ops<T>::v4 t = {Ops<T>::cvt(cx[0]), Ops<T>::cvt(cx[1]), Ops<T>::cvt(cx[2]), Ops<T>::cvt(cx[3])};
#pragma unroll
      for (int r = 0; r < 4; ++r) CX[(row0 + r) * P::LKX + col] = t[r];
      *reinterpret_cast<typename Ops<T>::v4*>(CXT + col * P::LKT + row0) = t;
      if (col == FF_GEO) store_rt<T>(SGZO, P::LKO, SGZOT, P::LKT, wave, 0, gd, lane);  // lanes of column 15 only
    }
    __syncthreads();
    // ---- P4 / P5 colour hidden layers ----
    {
      f32x4 acc[1] = {};
      const int nt = wave & 3, mt = wave >> 2;
      mma_rr<1, 32>(CX + mt * 16 * P::LKX, P::LKX, smem + P::CW0T, P::LKX, nt, acc, lane);
      relu4<T>(acc[0]);
      store_rt_op<T>(CA1, P::LKC, CA1T, P::LKT, mt, nt, acc[0], lane);
    }
    __syncthreads();
    {
      f32x4 acc[1] = {};
      const int nt = wave & 3, mt = wave >> 2;
      mma_rr<1, FF_HC>(CA1 + mt * 16 * P::LKC, P::LKC, smem + P::CW1T, P::LKC, nt, acc, lane);
      relu4<T>(acc[0]);
      store_rt_op<T>(CA2, P::LKC, CA2T, P::LKT, mt, nt, acc[0], lane);
    }
    __syncthreads();
    // ---- P6 colour output: gradient w.r.t. its pre-activation ----
    if (wave < MT) {
      f32x4 acc[1] = {};
      mma_rr<1, FF_HC>(CA2 + wave * 16 * P::LKC, P::LKC, smem + P::CWOT, P::LKC, 0, acc, lane);
      const int col = lane & 15, row0 = wave * 16 + (lane >> 4) * 4;
      f32x4 gv = {};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + row0 + r;
        if (col < 3 && n < a.N) {
          const float sg = 1.f / (1.f + expf(-acc[0][r]));
          gv[r] = a.grgb[n * 3 + col] * sg * (1.f - sg) * GS;
        }
      }
      store_rt<T>(CGZO, P::LKO, CGZOT, P::LKT, wave, 0, gv, lane);
    }
    __syncthreads();
    // ---- P7 dWO(colour) += A2^T gzo ; then gz2 = (gzo WO^T) .* relu'(A2) ----
    if (wave < HCT) {
      f32x4 acc[1] = {dWOc};
      mma_rr<1, FF_TS>(CA2T + wave * 16 * P::LKT, P::LKT, CGZOT, P::LKT, 0, acc, lane);
      dWOc = acc[0];
    }
    __syncthreads();  // CA2T is overwritten by GZC2T
    {
      const int nt = wave & 3, mt = wave >> 2;
      f32x4 acc[1] = {};
      mma_rr<1, 32>(CGZO + mt * 16 * P::LKO, P::LKO, smem + P::CWOR, P::LKO, nt, acc, lane);
      const int col = nt * 16 + (lane & 15), row0 = mt * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!((float)CA2[(row0 + r) * P::LKC + col] > 0.f)) acc[0][r] = 0.f;
      store_rt<T>(GZC2, P::LKC, GZC2T, P::LKT, mt, nt, acc[0], lane);
    }
    __syncthreads();
    // ---- P8 dW1(colour) += A1^T gz2 ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = wave + FF_NW * j;
      f32x4 acc[1] = {dW1c[j]};
      mma_rr<1, FF_TS>(CA1T + (t / HCT) * 16 * P::LKT, P::LKT, GZC2T, P::LKT, t % HCT, acc, lane);
      dW1c[j] = acc[0];
    }
    __syncthreads();  // CA1T is overwritten by GZC1T
    // ---- P9 gz1 = (gz2 W1^T) .* relu'(A1) ----
    {
      const int nt = wave & 3, mt = wave >> 2;
      f32x4 acc[1] = {};
      mma_rr<1, FF_HC>(GZC2 + mt * 16 * P::LKC, P::LKC, smem + P::CW1R, P::LKC, nt, acc, lane);
      const int col = nt * 16 + (lane & 15), row0 = mt * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!((float)CA1[(row0 + r) * P::LKC + col] > 0.f)) acc[0][r] = 0.f;
      store_rt<T>(GZC1, P::LKC, GZC1T, P::LKT, mt, nt, acc[0], lane);
    }
    __syncthreads();
    // ---- P10 dW0(colour) += CX^T gz1 ; gCX = gz1 W0^T -> columns 0..14 of the sigma output gradient ----
    {
      f32x4 acc[1] = {dW0c};
      mma_rr<1, FF_TS>(CXT + (wave >> 2) * 16 * P::LKT, P::LKT, GZC1T, P::LKT, wave & 3, acc, lane);
      dW0c = acc[0];
    }
    if (wave < MT) {  // only the first 16 of the 32 padded inputs carry anything (15 geometry features)
      f32x4 acc[1] = {};
      mma_rr<1, FF_HC>(GZC1 + wave * 16 * P::LKC, P::LKC, smem + P::CW0R, P::LKC, 0, acc, lane);
      const int col = lane & 15;
      if (col < FF_GEO) store_rt<T>(SGZO, P::LKO, SGZOT, P::LKT, wave, 0, acc[0], lane);  // already carries the factor GS; column 15 was written in P3
    }
    __syncthreads();
    // ---- P11 dWO(sigma) += A1^T gzo ----
    {
      f32x4 acc[1] = {dWOs};
      mma_rr<1, FF_TS>(A1T + wave * 16 * P::LKT, P::LKT, SGZOT, P::LKT, 0, acc, lane);
      dWOs = acc[0];
    }
    __syncthreads();  // A1T is overwritten by GZ1T
    // ---- P12 gz1 = (gzo WO^T) .* relu'(A1) ----
    {
      f32x4 acc[MT] = {};
      mma_rr<MT, 32>(SGZO, P::LKO, smem + P::SWOR, P::LKO, wave, acc, lane);
      const int col = wave * 16 + (lane & 15);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!((float)A1[(row0 + r) * P::LKH + col] > 0.f)) acc[m][r] = 0.f;
        store_rt<T>(GZ1, P::LKH, GZ1T, P::LKT, m, wave, acc[m], lane);
      }
    }
    __syncthreads();
    // ---- P13 dW0(sigma) += X^T gz1 ; gX = gz1 W0^T -> GF (fp32) ----
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + FF_NW * j;
      if (t < K0T * HT) {
        f32x4 acc[1] = {dW0s[j]};
        mma_rr<1, FF_TS>(XT + (t / HT) * 16 * P::LKT, P::LKT, GZ1T, P::LKT, t % HT, acc, lane);
        dW0s[j] = acc[0];
      }
    }
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      const int nb = wave + FF_NW * j;
      if (nb < K0T) {
        f32x4 acc[MT] = {};
        typename Ops<T>::v8 bT[4];  // feature columns nb * 16 + lr, k = hidden units 32 ks + 8 lk .. +8: 32 contiguous bytes of W0 per k-step
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float4* src = reinterpret_cast<const float4*>(a.Wsig + (int64_t)(nb * 16 + lr) * FF_H + ks * 32 + lk * 8);
          const float4 w0 = src[0], w1 = src[1];
          bT[ks] = typename Ops<T>::v8{Ops<T>::cvt(w0.x), Ops<T>::cvt(w0.y), Ops<T>::cvt(w0.z), Ops<T>::cvt(w0.w),
                                       Ops<T>::cvt(w1.x), Ops<T>::cvt(w1.y), Ops<T>::cvt(w1.z), Ops<T>::cvt(w1.w)};
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = Ops<T>::mfma(ld8(GZ1 + (m * 16 + lr) * P::LKH + ks * 32 + lk * 8), bT[ks], acc[m]);
        const int col = nb * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) GF[(row0 + r) * P::LGF + col] = acc[m][r] * (1.f / GS);
        }
      }
    }
    __syncthreads();
    // ---- P14 per-plane gradient vectors (pass A of the sorted scatter): g_q = dL/dfeat .* prod_{p != q} v_p ----
    {
      const int sample = (threadIdx.x & 255) >> 3, cg = threadIdx.x & 7, sg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
      const int64_t n = n0 + sample;
      if (n < a.N) {
        float p[4];
        load_coords<6>(a.c, n, p);
        GV* gvec = reinterpret_cast<GV*>(a.gvec);
#pragma unroll 1
        for (int s = sg; s < NS; s += 2) {
          float4 v[6];
          scale_features(a.d, a.planes, p, s, cg, &v);
          const float4 g = *reinterpret_cast<const float4*>(GF + sample * P::LGF + s * 32 + cg * 4);
          float4 suf[7];
          suf[6] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
          for (int q = 5; q >= 0; --q) suf[q] = f4_mul(suf[q + 1], v[q]);
          float4 pre = g;
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            const float4 gq = f4_mul(pre, suf[q + 1]);
            pre = f4_mul(pre, v[q]);
            store_gq(gvec, ((int64_t)(s * 6 + q) * a.N + n) * 32 + cg * 4, gq);
          }
        }
      }
    }
  }

  // ---- flush the weight gradients (each 16-lane group adds 64 contiguous bytes) ----
  const int cl = lane & 15, r0 = (lane >> 4) * 4;
  const float inv = 1.f / GS;
#pragma unroll
  for (int j = 0; j < NB0; ++j) {
    const int t = wave + FF_NW * j;
    if (t < K0T * HT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gw_add2(a.gWsig, a.gWsig_fx, (int64_t)((t / HT) * 16 + r0 + r) * FF_H + (t % HT) * 16 + cl, dW0s[j][r] * inv);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) gw_add2(a.gWsig, a.gWsig_fx, (int64_t)K0 * FF_H + (int64_t)(wave * 16 + r0 + r) * 16 + cl, dWOs[r] * inv);
  {
    const int it = wave >> 2, nt = wave & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = it * 16 + r0 + r;
      if (row < FF_GEO) gw_add2(a.gWcol, a.gWcol_fx, (int64_t)row * FF_HC + nt * 16 + cl, dW0c[r] * inv);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = wave + FF_NW * j;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      gw_add2(a.gWcol, a.gWcol_fx, (int64_t)FF_GEO * FF_HC + (int64_t)((t / HCT) * 16 + r0 + r) * FF_HC + (t % HCT) * 16 + cl, dW1c[j][r] * inv);
  }
  if (wave < HCT && cl < 3) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      gw_add2(a.gWcol, a.gWcol_fx, (int64_t)FF_GEO * FF_HC + FF_HC * FF_HC + (int64_t)(wave * 16 + r0 + r) * 3 + cl, dWOc[r] * inv);
  }
}

template <typename T, int NS>
static int launch_field_bwd(const FieldArgs& a, hipStream_t st) {
  using P = PlanFB<NS>;
  static_assert(P::BYTES <= LDS_LIMIT_B, "fused backward tile does not fit LDS");
  const int64_t n_tiles = (a.N + FF_TS - 1) / FF_TS;
  int64_t grid = 256;  // one 8-wave workgroup per CU (LDS)
  if (grid > n_tiles) grid = n_tiles;
  if (a.gvec_bf16) {
    auto k = field_bwd_kernel<T, NS, __bf16>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_B); attr_set = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(FF_NW * 64), P::BYTES, st, a, n_tiles);
  } else {
    auto k = field_bwd_kernel<T, NS, float>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT_B); attr_set = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(FF_NW * 64), P::BYTES, st, a, n_tiles);
  }
  SNERF_LAUNCH_CHECK("kplanes_field_bwd");
  return 0;
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

// the forward alone also fits six scales (BASELINE config 3: K0 = 192); the backward's tiles do not
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

extern "C" int snerf_kplanes_field_supported(const snerf_kplanes_desc* desc, const snerf_mlp_desc* sigma, const snerf_mlp_desc* color) {
  snerf_coords c = {};
  return desc && sigma && color && validate_field(desc, &c, 0, sigma, color) == 0 ? 1 : 0;
}

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

extern "C" int snerf_kplanes_field_bwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                                       const snerf_mlp_desc* sigma, const float* W_sigma, const snerf_mlp_desc* color, const float* W_color,
                                       const float* g_density, const float* g_rgb, float* gW_sigma, float* gW_color, int64_t* gW_sigma_fx,
                                       int64_t* gW_color_fx, void* gvec, int32_t gvec_bf16, snerf_stream_t stream) {
  int rc = validate_field(desc, coords, N, sigma, color);
  if (rc) return rc;
  if (N == 0) return 0;
  SNERF_REQUIRE(planes && W_sigma && W_color && g_density && g_rgb && gvec, "kplanes_field_bwd: null buffer");
  SNERF_REQUIRE((gW_sigma && gW_color) || (gW_sigma_fx && gW_color_fx), "kplanes_field_bwd: weight-gradient buffers (float or fixed point) are null");
  FieldArgs a = {};
  a.d = *desc; a.planes = planes; a.c = *coords; a.N = N; a.Wsig = W_sigma; a.Wcol = W_color; a.gdens = g_density; a.grgb = g_rgb;
  a.gWsig = gW_sigma; a.gWcol = gW_color; a.gWsig_fx = reinterpret_cast<long long*>(gW_sigma_fx); a.gWcol_fx = reinterpret_cast<long long*>(gW_color_fx);
  a.gvec = gvec; a.gvec_bf16 = gvec_bf16;
  FF_DISPATCH(launch_field_bwd, sigma->operands, desc->n_scales, a, (hipStream_t)stream);
}

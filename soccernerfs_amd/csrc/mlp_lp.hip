// Tiny MLPs with 16-bit MFMA operands and fp32 accumulation (snerf_mlp_desc.operands = 1: bf16, 2: fp16): the precision class of the
// reference's own MLPs (tcnn FullyFusedMLP computes in fp16 with fp32 accumulation; BASELINE config 2 names bf16) at 16x the matrix rate
// of the exact fp32 path in mlp.hip.  One hidden layer (sigma_net d_in -> 128 -> 16, proposal nets 8 -> 64 -> 1) or two (color_net
// 15 -> 64 -> 64 -> 3): NS/fields/kplanes_field.py:249-273,397-407.  Master weights, inputs, outputs, gradients and the weight-gradient accumulators stay
// fp32; only MFMA operands are rounded (round-to-nearest-even) when they are staged into LDS.  fp16 operands carry the backward's
// gradient tiles multiplied by 2^13 (a power of two: exact), as tcnn's loss scale does, and saturate instead of overflowing; bf16
// needs neither.  OPT-IN: the exact fp32 kernels stay the default and the parity path (measured trade-off: profiles/r01_kernels.md).
//
// v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k = 8(l>>4) .. +8] and B[k = 8(l>>4) .. +8][col l&15] -- 8 CONSECUTIVE k per lane.
// Every product below is therefore arranged as "both operands row-major along the contraction index" (one ds_read_b128 per operand per
// MFMA), which means each matrix is kept in LDS in the orientation(s) its products need:
//   forward   Z1 = X W0        A = X  [TS][K0]      B^T = W0t [H][K0]     (W0 transposed while staging)
//             Y  = A1 WO       A = A1 [TS][H]       B^T = WOt [16][H]
//   backward  dZ1 = dY WO^T    A = gzo [TS][32]     B^T = WOr [H][32]     (WO as stored, outputs padded 16 -> 32)
//             dX  = dZ1 W0^T   A = gz  [TS][H]      B^T = W0r [K0][H]     (W0 as stored)
//             dW0 = X^T dZ1    A = Xt  [K0][TS]     B^T = gzt [H][TS]     (contraction over the tile's samples: TS = 32 = one MFMA)
//             dWO = A1^T dY    A = A1t [H][TS]      B^T = gzot [16][TS]
// The transposed copies cost nothing extra to write from an accumulator: a lane of the C layout holds 4 consecutive ROWS of one column,
// i.e. 8 contiguous bytes of the transposed image.  Row stride = K + 8 elements (16-B aligned rows, conflict-free b128 reads).
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "mlp_lp_common.hpp"

namespace snerf {

// X tile: global fp32 -> registers one tile ahead -> LDS (row-major and, for the backward, transposed)
template <int TS, int K0, int NT>
struct XTileB {
  static constexpr int PER = (TS * K0 + NT - 1) / NT;
  float v[PER];
  template <typename T>
  __device__ __forceinline__ void fetch(const MlpArgs& a, int64_t n0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / K0, c = idx - r * K0;
      const int64_t n = n0 + r;
      v[i] = (idx < TS * K0 && n < a.N && c < a.d0) ? a.X[n * a.ldx + c] : 0.f;
    }
  }
  template <typename T>
  __device__ __forceinline__ void store(T* Xs, int ldx, T* Xt, int ldt) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / K0, c = idx - r * K0;
      if (idx < TS * K0) {
        const T b = Ops<T>::cvt(v[i]);
        Xs[r * ldx + c] = b;
        if (Xt) Xt[c * ldt + r] = b;
      }
    }
  }
};

// X tile that arrives in the operand type already (MlpArgs.x16: the feature tile snerf_kplanes_field_fwd wrote): 16-B loads of 8 elements,
// no conversion.  Needs d_in == K0, ldx % 8 == 0 and a 16-B aligned base (checked by the launcher).
template <int TS, int K0, int NT, typename T>
struct XTile16 {
  static constexpr int NV = TS * K0 / 8, PER = (NV + NT - 1) / NT, VR = K0 / 8;
  typename Ops<T>::v8 v[PER];
  template <typename>
  __device__ __forceinline__ void fetch(const MlpArgs& a, int64_t n0) {
    const T* X16 = reinterpret_cast<const T*>(a.X);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int vi = threadIdx.x + i * NT;
      const int c8 = vi / TS, r = vi - c8 * TS;  // lanes along the rows: the transposed image's stores land in consecutive LDS addresses
      const int64_t n = n0 + r;
      typename Ops<T>::v8 z = {};
      v[i] = (vi < NV && n < a.N) ? *reinterpret_cast<const typename Ops<T>::v8*>(X16 + n * a.ldx + c8 * 8) : z;
    }
  }
  __device__ __forceinline__ void store(T* Xs, int ldx, T* Xt, int ldt) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int vi = threadIdx.x + i * NT;
      const int c8 = vi / TS, r = vi - c8 * TS;  // lanes along the rows: the transposed image's stores land in consecutive LDS addresses
      if (vi < NV) {
        *reinterpret_cast<typename Ops<T>::v8*>(Xs + r * ldx + c8 * 8) = v[i];
        if (Xt) {
#pragma unroll
          for (int e = 0; e < 8; ++e) Xt[(c8 * 8 + e) * ldt + r] = v[i][e];
        }
      }
    }
  }
};

template <int H>
constexpr int waves_b() { return H >= 128 ? 8 : 4; }

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int K0, int H, int NH, int TS>
struct PlanF {
  static constexpr int LK0 = ldb(K0), LKH = ldb(H);
  static constexpr int W0T = 0;                    // [H][LK0]
  static constexpr int W1T = W0T + H * LK0;        // [H][LKH]   (NH == 2)
  static constexpr int WOT = W1T + (NH == 2 ? H * LKH : 0);  // [16][LKH]
  static constexpr int XS = WOT + 16 * LKH;        // [TS][LK0]
  static constexpr int A1 = XS + TS * LK0;         // [TS][LKH]
  static constexpr int A2 = A1 + TS * LKH;         // [TS][LKH]  (NH == 2)
  static constexpr int TOTAL = A2 + (NH == 2 ? TS * LKH : 0);
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
};

template <typename T, int K0, int H, int NH, int TS>
__global__ __launch_bounds__(waves_b<H>() * 64) void mlp_lp_fwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  using P = PlanF<K0, H, NH, TS>;
  constexpr int MT = TS / 16, NW = waves_b<H>(), HT = H / 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stage_w<T>(a.W + a.woff[0], a.d0, H, K0, H, nullptr, 0, smem + P::W0T, P::LK0);
  if (NH == 2) stage_w<T>(a.W + a.woff[1], H, H, H, H, nullptr, 0, smem + P::W1T, P::LKH);
  stage_w<T>(a.W + a.woff[NH], H, a.dout, H, 16, nullptr, 0, smem + P::WOT, P::LKH);
  const T* Alast = smem + (NH == 2 ? P::A2 : P::A1);
  const bool relu = a.hidden_act == 1;
  XTileB<TS, K0, NW * 64> xt;
  xt.template fetch<T>(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    xt.store(smem + P::XS, P::LK0, (T*)nullptr, 0);
    if (tile + gridDim.x < n_tiles) xt.template fetch<T>(a, (tile + gridDim.x) * TS);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
      const int nt = wave + NW * j;
      if (nt < HT) {
        f32x4 acc[MT] = {};
        mma_rr<MT, K0>(smem + P::XS, P::LK0, smem + P::W0T, P::LK0, nt, acc, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 v = acc[m];
          if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
          store_rt<T>(smem + P::A1, P::LKH, nullptr, 0, m, nt, v, lane);
        }
      }
    }
    __syncthreads();
    if (NH == 2) {
#pragma unroll
      for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
        const int nt = wave + NW * j;
        if (nt < HT) {
          f32x4 acc[MT] = {};
          mma_rr<MT, H>(smem + P::A1, P::LKH, smem + P::W1T, P::LKH, nt, acc, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            f32x4 v = acc[m];
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            store_rt<T>(smem + P::A2, P::LKH, nullptr, 0, m, nt, v, lane);
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < (MT + NW - 1) / NW; ++j) {
      const int mt = wave + NW * j;
      if (mt < MT) {
        f32x4 acc[1] = {};
        mma_rr<1, H>(Alast + mt * 16 * P::LKH, P::LKH, smem + P::WOT, P::LKH, 0, acc, lane);
        const int col = lane & 15;
        const int64_t row0 = n0 + mt * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t n = row0 + r;
          if (n < a.N && col < a.dout) {
            float y = acc[0][r];
            if (a.aux_out && col == a.aux_col) a.aux_out[n] = expf(y);  // trunc_exp forward (activations.py:32)
            if (a.out_act == 1) y = 1.f / (1.f + expf(-y));
            a.Y[n * a.ldy + col] = y;
          }
        }
      }
    }
  }
}

// ---- transposed LDS reads (gfx950 ds_read_b64_tr_b16): operand fragments from a [k][row] image ----
typedef short tr_v4 __attribute__((ext_vector_type(4)));

// A-operand fragment from a k-major image Tm[k][row] (row stride ld elements): element j of lane (r = lane & 15, g = lane >> 4) =
// Tm[kbase + 8 g + j][r0 + r].  Lane 4 q + p of a 16-lane group supplies the address of row q, columns 4 p .. 4 p + 3 of its 4 x 16 block
// (cdna_hip_programming.md T10).  EXEC must be all ones: call from wave-uniform control flow only.
template <typename T>
__device__ __forceinline__ typename Ops<T>::v8 ld8_tr(const T* Tm, int ld, int kbase, int r0, int lane) {
  const int q = (lane & 15) >> 2, p = lane & 3, g = lane >> 4;
  const T* a0 = Tm + (kbase + 8 * g + q) * ld + r0 + 4 * p;
  typedef __attribute__((address_space(3))) tr_v4 lds_v4;
  const tr_v4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
  const tr_v4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * ld));
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename Ops<T>::v8, w);
}

// acc[m] (row block row_base / 16 + m, column block nt) += A * Bt^T with A given k-major (At[k][row], read transposed) and Bt row-major along k
template <int MT, int K, typename T>
__device__ __forceinline__ void mma_tr(const T* At, int ldat, int row_base, const T* Bt, int ldbt, int nt, f32x4 (&acc)[MT], int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const T* bp = Bt + (nt * 16 + lr) * ldbt + lk * 8;
#pragma unroll
  for (int ks = 0; ks < K / 32; ++ks) {
    const typename Ops<T>::v8 b = ld8(bp + ks * 32);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = Ops<T>::mfma(ld8_tr<T>(At, ldat, ks * 32, row_base + m * 16, lane), b, acc[m]);
  }
}

// ---------------------------------------------------------------------------------------------
// backward (recomputes the forward per tile, as the fp32 kernel)
// ---------------------------------------------------------------------------------------------
// 128-wide one-hidden-layer nets (sigma_net): 8 waves x 16 hidden units -- each wave keeps ITS layer-0 B operand (K0 / 32 k-steps x 8
// values per lane) in registers for the whole persistent loop, so the transposed image W0T (H x K0: 43-51 KB) never enters LDS.  That is what
// lets the 6-scale field (K0 = 192, BASELINE config 3) fit the backward at all (177 KB with W0T, 126 KB without).
template <int H, int NH>
constexpr bool wreg_b() { return H == 128 && NH == 1; }

template <int K0, int H, int NH, int TS>
struct PlanB {
  static constexpr int LK0 = ldb(K0), LKH = ldb(H), LKO = ldb(32), LKT = ldb(TS);
  static constexpr int W0T = 0;                    // [H][LK0]   forward (absent when register-resident)
  static constexpr int W0R = W0T + (wreg_b<H, NH>() ? 0 : H * LK0);  // [K0][LKH]  dX
  static constexpr int W1T = W0R + K0 * LKH;       // [H][LKH]   forward, layer 1 (NH == 2)
  static constexpr int W1R = W1T + (NH == 2 ? H * LKH : 0);  // [H][LKH]  dA1 (NH == 2)
  static constexpr int WOT = W1R + (NH == 2 ? H * LKH : 0);  // [16][LKH]  forward
  static constexpr int WOR = WOT + 16 * LKH;       // [H][LKO]   dZ_last
  static constexpr int XS = WOR + H * LKO;         // [TS][LK0]
  static constexpr int XT = XS + TS * LK0;         // [K0][LKT]
  static constexpr int A1 = XT + K0 * LKT;         // [TS][LKH]
  static constexpr int A1T = A1 + (NH == 1 ? 0 : TS * LKH);  // [H][LKT]; reused for the transposed gradient of Z1 once its weight-gradient product is done.
                                                   // (one hidden layer: no [sample][unit] copy of A1 -- it is read from this image, transposed)
  static constexpr int A2 = A1T + H * LKT;         // [TS][LKH]  (NH == 2)
  static constexpr int A2T = A2 + (NH == 2 ? TS * LKH : 0);  // [H][LKT]; reused for the transposed gradient of Z2
  // one hidden layer, input at least as wide as the hidden layer: the row-major X tile is dead once the hidden layer is computed (the weight
  // gradient reads the transposed image), so the row-major gradient of Z_last takes its place -- that is what lets 64-sample tiles of the
  // 160 -> 128 net fit (146 KB instead of 163)
  static constexpr bool GZ_IN_XS = NH == 1 && K0 >= H;
  static constexpr int GZ_OWN = A2T + (NH == 2 ? H * LKT : 0);
  static constexpr int GZ = GZ_IN_XS ? XS : GZ_OWN;          // [TS][LKH]  gradient of Z_last
  static constexpr int GZ1 = GZ_OWN + ((GZ_IN_XS || NH == 1) ? 0 : TS * LKH);  // [TS][LKH]  gradient of Z1 (NH == 2; one hidden layer: no row-major gradient image at all)
  static constexpr int GZO = GZ1 + (NH == 2 ? TS * LKH : 0);  // [TS][LKO]
  static constexpr int GZOT = GZO + TS * LKO;      // [16][LKT]
  static constexpr int TOTAL = GZOT + 16 * LKT;
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
};

// (Round 2 also formed the quotient scatter's G = gX .* feat in this kernel's gX epilogue: that instantiation needed 256 VGPRs + spills and the
// step got slower -- profiles/r02_kernels.md section 10 -- so the separate quotient_prepare pass stayed and the variant was removed.  A lesson
// kept: an optional epilogue must be a template parameter; as a run-time `if` EVERY instantiation paid its registers.)
// QG (round 4): the quotient epilogue again, this time from the LDS-resident 16-bit X tile (Xt stays alive to the end of the tile for the
// layer-0 weight gradient): G = gX .* X costs one 8-byte LDS read per accumulator block instead of round 2's 160 dependent global loads per
// sample.  X here is the operand-typed feature (bf16 / fp16 of the forward's fp32 product): G carries that rounding, which is of the size of
// the MFMA operand roundings gX went through already.  A template parameter: the other instantiations do not pay for it.
template <typename T, int K0, int H, int NH, int TS, bool X16 = false, bool QG = false>
__global__ __launch_bounds__(waves_b<H>() * 64) void mlp_lp_bwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr float GS = Ops<T>::GS;
  using P = PlanB<K0, H, NH, TS>;
  static_assert(TS % 32 == 0, "the weight-gradient products contract over the tile's samples in steps of 32");
  constexpr int MT = TS / 16, NW = waves_b<H>(), HT = H / 16, K0T = K0 / 16;
  constexpr int NB0 = (K0T * HT + NW - 1) / NW, NBO = (HT + NW - 1) / NW, NBH = (HT * HT + NW - 1) / NW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // "last" = the hidden layer in front of the output layer (layer 1 when NH == 1)
  T *Xs = smem + P::XS, *Xt = smem + P::XT, *A1 = smem + P::A1, *A1t = smem + P::A1T, *gzo = smem + P::GZO, *gzot = smem + P::GZOT;
  T *Al = NH == 2 ? smem + P::A2 : A1, *Alt = NH == 2 ? smem + P::A2T : A1t;  // last hidden activations (row-major / transposed)
  T *gz = smem + P::GZ, *gzt = Alt;                                          // gradient of Z_last; its transposed image reuses Alt
  T *gz1 = NH == 2 ? smem + P::GZ1 : gz, *gz1t = NH == 2 ? A1t : gzt;        // gradient of Z1
  constexpr bool WREG = wreg_b<H, NH>();
  stage_w<T>(a.W + a.woff[0], a.d0, H, K0, H, smem + P::W0R, P::LKH, WREG ? (T*)nullptr : smem + P::W0T, P::LK0);
  typename Ops<T>::v8 breg[WREG ? K0 / 32 : 1];
  if (WREG) {  // this wave's hidden units 16 wave .. +15, k = 32 ks + 8 (lane >> 4) .. +8
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < K0 / 32; ++ks) {
      typename Ops<T>::v8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ks * 32 + lk * 8 + e;
        const float wv = a.W[a.woff[0] + (int64_t)(k < a.d0 ? k : a.d0 - 1) * H + wave * 16 + lr];  // unconditional (clamped) load: see stage_w
        b[e] = Ops<T>::cvt(k < a.d0 ? wv : 0.f);
      }
      breg[ks] = b;
    }
  }
  if (NH == 2) stage_w<T>(a.W + a.woff[1], H, H, H, H, smem + P::W1R, P::LKH, smem + P::W1T, P::LKH);
  stage_w<T>(a.W + a.woff[NH], H, a.dout, H, 32, smem + P::WOR, P::LKO, nullptr, 0);
  stage_w<T>(a.W + a.woff[NH], H, a.dout, H, 16, nullptr, 0, smem + P::WOT, P::LKH);
  // columns 16..31 of gzo (the padded half of the K = 32 contraction) stay zero for the whole kernel
  for (int idx = threadIdx.x; idx < TS * 16; idx += blockDim.x) gzo[(idx / 16) * P::LKO + 16 + (idx % 16)] = (T)0.f;
  const bool relu = a.hidden_act == 1;
  if constexpr (QG) {  // the caller alternates between two list counters: reset the one the NEXT step will use (see quotient_prepare_kernel)
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.fix_count_next) *a.fix_count_next = 0;
  }
  f32x4 dW0[NB0] = {};
  f32x4 dWh[NH == 2 ? NBH : 1] = {};
  f32x4 dWo[NBO] = {};
  std::conditional_t<X16, XTile16<TS, K0, NW * 64, T>, XTileB<TS, K0, NW * 64>> xt;
  xt.template fetch<T>(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    xt.store(Xs, P::LK0, Xt, P::LKT);
    if (tile + gridDim.x < n_tiles) xt.template fetch<T>(a, (tile + gridDim.x) * TS);
    // this tile's incoming gradients, issued now and consumed by the output phase two barriers later: read there, their latency was a fifth
    // of the tile's time (per-phase clocks, profiles/r03_kernels.md section 8)
    constexpr int JO = (MT + NW - 1) / NW;
    float gyp[JO][4], gap[JO][4];
#pragma unroll
    for (int j = 0; j < JO; ++j) {
      const int mt = wave + NW * j, col = lane & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + mt * 16 + (lane >> 4) * 4 + r;
        const bool live = mt < MT && n < a.N && col < a.dout;
        gyp[j][r] = (live && a.gY) ? a.gY[n * a.ldgy + col] : 0.f;
        gap[j][r] = (live && a.gaux && col == a.aux_col) ? a.gaux[n] : 0.f;
      }
    }
    __syncthreads();
    // ---- hidden layer: A1 (row-major) and A1t ----
#pragma unroll
    for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
      const int nt = wave + NW * j;
      if (nt < HT) {
        f32x4 acc[MT] = {};
        if (WREG) {
          const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
          for (int ks = 0; ks < K0 / 32; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = Ops<T>::mfma(ld8(Xs + (m * 16 + lr) * P::LK0 + ks * 32 + lk * 8), breg[WREG ? ks : 0], acc[m]);
        } else {
          mma_rr<MT, K0>(Xs, P::LK0, smem + P::W0T, P::LK0, nt, acc, lane);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 v = acc[m];
          if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
          // one hidden layer: only the unit-major image; the output layer and the relu mask read it back (transposed / 8 bytes per block)
          store_rt(NH == 1 ? (T*)nullptr : A1, P::LKH, A1t, P::LKT, m, nt, v, lane);
        }
      }
    }
    __syncthreads();
    if (NH == 2) {  // ---- second hidden layer: A2 and A2t ----
#pragma unroll
      for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
        const int nt = wave + NW * j;
        if (nt < HT) {
          f32x4 acc[MT] = {};
          mma_rr<MT, H>(A1, P::LKH, smem + P::W1T, P::LKH, nt, acc, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            f32x4 v = acc[m];
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            store_rt(Al, P::LKH, Alt, P::LKT, m, nt, v, lane);
          }
        }
      }
      __syncthreads();
    }
    // ---- output layer forward + gradient w.r.t. its pre-activation: gzo (row-major, cols 0..15) and gzot ----
#pragma unroll
    for (int j = 0; j < (MT + NW - 1) / NW; ++j) {
      const int mt = wave + NW * j;
      if (mt < MT) {
        f32x4 acc[1] = {};
        if constexpr (NH == 1) mma_tr<1, H, T>(Alt, P::LKT, mt * 16, smem + P::WOT, P::LKH, 0, acc, lane);
        else mma_rr<1, H>(Al + mt * 16 * P::LKH, P::LKH, smem + P::WOT, P::LKH, 0, acc, lane);
        const int col = lane & 15;
        const int rl0 = mt * 16 + (lane >> 4) * 4;
        f32x4 gv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t n = n0 + rl0 + r;
          float g = 0.f;
          if (n < a.N && col < a.dout) {
            const float y = acc[0][r];
            g = gyp[j][r];
            if (a.out_act == 1) {
              const float sg = 1.f / (1.f + expf(-y));
              g = g * sg * (1.f - sg);
            }
            if (a.gaux && col == a.aux_col) g += gap[j][r] * expf(fminf(fmaxf(y, -15.f), 15.f));  // trunc_exp backward (activations.py:38-39)
          }
          gv[r] = g * GS;
        }
        store_rt(gzo, P::LKO, gzot, P::LKT, mt, 0, gv, lane);
      }
    }
    __syncthreads();
    // ---- dWO += A_last^T gzo (contraction over the tile's samples) ----
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + NW * j;
      if (it < HT) {
        f32x4 acc[1] = {dWo[j]};
        mma_rr<1, TS>(Alt + it * 16 * P::LKT, P::LKT, gzot, P::LKT, 0, acc, lane);
        dWo[j] = acc[0];
      }
    }
    __syncthreads();  // Alt is overwritten by gzt below
    // ---- gz = (gzo WO^T) .* relu'(A_last): row-major and transposed ----
#pragma unroll
    for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
      const int nt = wave + NW * j;
      if (nt < HT) {
        f32x4 acc[MT] = {};
        mma_rr<MT, 32>(gzo, P::LKO, smem + P::WOR, P::LKO, nt, acc, lane);
        const int col = nt * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row0 = m * 16 + (lane >> 4) * 4;
          f32x4 v = acc[m];
          if (relu) {
            if constexpr (NH == 1) {  // the activations of this block: the 8 bytes of Alt that gzt overwrites just below (same lane)
              const typename Ops<T>::v4 act = *reinterpret_cast<const typename Ops<T>::v4*>(Alt + col * P::LKT + row0);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (!((float)act[r] > 0.f)) v[r] = 0.f;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (!((float)Al[(row0 + r) * P::LKH + col] > 0.f)) v[r] = 0.f;
            }
          }
          store_rt(NH == 1 ? (T*)nullptr : gz, P::LKH, gzt, P::LKT, m, nt, v, lane);  // one hidden layer: the input gradient reads gzt transposed
        }
      }
    }
    __syncthreads();
    if (NH == 2) {
      // ---- dW1 += A1^T gz (contraction over the tile's samples) ----
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + NW * j;
        if (t < HT * HT) {
          f32x4 acc[1] = {dWh[j]};
          mma_rr<1, TS>(A1t + (t / HT) * 16 * P::LKT, P::LKT, gzt, P::LKT, t % HT, acc, lane);
          dWh[j] = acc[0];
        }
      }
      __syncthreads();  // A1t is overwritten by gz1t below
      // ---- gz1 = (gz W1^T) .* relu'(A1): row-major and transposed ----
#pragma unroll
      for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
        const int nt = wave + NW * j;
        if (nt < HT) {
          f32x4 acc[MT] = {};
          mma_rr<MT, H>(gz, P::LKH, smem + P::W1R, P::LKH, nt, acc, lane);
          const int col = nt * 16 + (lane & 15);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int row0 = m * 16 + (lane >> 4) * 4;
            f32x4 v = acc[m];
            if (relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (!((float)A1[(row0 + r) * P::LKH + col] > 0.f)) v[r] = 0.f;
            }
            store_rt(gz1, P::LKH, gz1t, P::LKT, m, nt, v, lane);
          }
        }
      }
      __syncthreads();
    }
    // ---- dW0 += X^T gz1 ----
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + NW * j;
      if (t < K0T * HT) {
        f32x4 acc[1] = {dW0[j]};
        mma_rr<1, TS>(Xt + (t / HT) * 16 * P::LKT, P::LKT, gz1t, P::LKT, t % HT, acc, lane);
        dW0[j] = acc[0];
      }
    }
    // ---- gX = gz1 W0^T.  One hidden layer: gz1 lives unit-major only (gz1t) and is read transposed; a wave keeps ONE row block's fragments
    //      (all of k) in registers and walks the column blocks with them.  Two hidden layers: (column block, row block) units dealt
    //      round-robin to the waves, operands from the row-major image ----
    if (a.gX || QG) {
      if constexpr (NH == 1) {
        static_assert(NW % MT == 0, "waves per workgroup must be a multiple of the tile's row blocks");
        constexpr int NG = NW / MT;
        const int m = wave % MT, grp = wave / MT, lr = lane & 15, lk = lane >> 4;
        typename Ops<T>::v8 af[H / 32];
#pragma unroll
        for (int ks = 0; ks < H / 32; ++ks) af[ks] = ld8_tr<T>(gz1t, P::LKT, ks * 32, m * 16, lane);
#pragma unroll
        for (int jn = 0; jn < (K0T + NG - 1) / NG; ++jn) {
          const int nt = grp + NG * jn;
          if (nt < K0T) {
            f32x4 acc = {};
            const T* bp = smem + P::W0R + (nt * 16 + lr) * P::LKH + lk * 8;
#pragma unroll
            for (int ks = 0; ks < H / 32; ++ks) acc = Ops<T>::mfma(af[ks], ld8(bp + ks * 32), acc);
            const int col = nt * 16 + lr;
            const int64_t row0 = n0 + m * 16 + lk * 4;
            if constexpr (QG) {
              // X[row0 .. row0 + 3][col] = 8 contiguous bytes of the unit-major image
              const typename Ops<T>::v4 xv = *reinterpret_cast<const typename Ops<T>::v4*>(Xt + col * P::LKT + m * 16 + lk * 4);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (row0 + r < a.N && col < a.d0) {
                  const float x = (float)xv[r], gx = acc[r] * (1.f / GS);
                  const bool vanished = fabsf(x) < QUOT_TINY;
                  const int64_t e = (row0 + r) * a.ldg + col;
                  a.G[e] = vanished ? 0.f : gx * x;
                  if (vanished && gx != 0.f) fix_append(a.fix_list, a.fix_capacity, a.fix_count, (int32_t)e, gx);
                }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (row0 + r < a.N && col < a.d0) a.gX[(row0 + r) * a.ldgx + col] = acc[r] * (1.f / GS);
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < (K0T * MT + NW - 1) / NW; ++j) {
          const int u = wave + NW * j;
          if (u < K0T * MT) {
            const int nt = u / MT, m = u - nt * MT;
            f32x4 acc[1] = {};
            mma_rr<1, H>(gz1 + m * 16 * P::LKH, P::LKH, smem + P::W0R, P::LKH, nt, acc, lane);
            const int col = nt * 16 + (lane & 15);
            const int64_t row0 = n0 + m * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (row0 + r < a.N && col < a.d0) a.gX[(row0 + r) * a.ldgx + col] = acc[0][r] * (1.f / GS);
          }
        }
      }
    }
  }
  // ---- flush weight gradients ----
  if (a.gW || a.gWfx || a.ws) {
    const int cl = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + NW * j;
      if (t < K0T * HT) {
        const int it = t / HT, nt = t % HT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = it * 16 + r0 + r;
          if (row < a.d0) gw_add(a, a.woff[0] + (int64_t)row * H + nt * 16 + cl, dW0[j][r] * (1.f / GS));
        }
      }
    }
    if (NH == 2) {
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + NW * j;
        if (t < HT * HT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) gw_add(a, a.woff[1] + (int64_t)((t / HT) * 16 + r0 + r) * H + (t % HT) * 16 + cl, dWh[j][r] * (1.f / GS));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + NW * j;
      if (it < HT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cl < a.dout) gw_add(a, a.woff[NH] + (int64_t)(it * 16 + r0 + r) * a.dout + cl, dWo[j][r] * (1.f / GS));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward, transposed-read form (gfx950 ds_read_b64_tr_b16): ONE LDS image per activation / gradient tile
// ---------------------------------------------------------------------------------------------
// The kernel above keeps every tile twice -- [sample][unit] for the products that contract over units, [unit][sample] for those that contract
// over samples -- and the accumulator layout (a lane holds 4 consecutive SAMPLES of one unit) makes the [sample][unit] copy 16 scattered 2-byte
// LDS stores per 16 x 16 block: ~150 ds_write_b16 per lane and 64-sample tile against ~80 MFMAs.  Here only the [unit][sample] image is written
// (one 8-byte store per block: the 4 samples are contiguous there) and the products that need the other orientation read it with
// ds_read_b64_tr_b16, which hands each lane a COLUMN of a 4 x 16 block: two of them make the 8-element operand fragment of a 16x16x32 MFMA.
// X stays row-major [sample][feature] (written with 16-byte stores) and is read transposed for the layer-0 weight gradient.
template <int K0, int H, int NH, int TS>
struct PlanT {
  static constexpr int LK0 = ldb(K0), LKH = ldb(H), LKO = ldb(32), LKT = ldb(TS);
  static constexpr int W0T = 0;                                       // [H][LK0]   forward (absent when register-resident)
  static constexpr int W0R = W0T + (wreg_b<H, NH>() ? 0 : H * LK0);   // [K0][LKH]  dX
  static constexpr int W1T = W0R + K0 * LKH;                          // [H][LKH]   forward, layer 1 (NH == 2)
  static constexpr int W1R = W1T + (NH == 2 ? H * LKH : 0);           // [H][LKH]   dA1 (NH == 2)
  static constexpr int WOT = W1R + (NH == 2 ? H * LKH : 0);           // [16][LKH]  forward
  static constexpr int WOR = WOT + 16 * LKH;                          // [H][LKO]   dZ_last
  static constexpr int XS = WOR + H * LKO;                            // [TS][LK0]  X, row-major
  static constexpr int A1T = XS + TS * LK0;                           // [H][LKT]   A1 (unit-major); later, in place, the gradient of Z1
  static constexpr int A2T = A1T + H * LKT;                           // [H][LKT]   A2 (NH == 2); later, in place, the gradient of Z2
  static constexpr int GZOT = A2T + (NH == 2 ? H * LKT : 0);          // [32][LKT]  gradient of the output pre-activation; rows 16..31 stay zero
  static constexpr int TOTAL = GZOT + 32 * LKT;
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
};

template <typename T, int K0, int H, int NH, int TS, bool X16 = false>
__global__ __launch_bounds__(waves_b<H>() * 64) void mlp_lp_bwd_tr_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr float GS = Ops<T>::GS;
  using P = PlanT<K0, H, NH, TS>;
  static_assert(TS % 32 == 0, "the weight-gradient products contract over the tile's samples in steps of 32");
  constexpr int MT = TS / 16, NW = waves_b<H>(), HT = H / 16, K0T = K0 / 16;
  constexpr int NB0 = (K0T * HT + NW - 1) / NW, NBO = (HT + NW - 1) / NW, NBH = (HT * HT + NW - 1) / NW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T *Xs = smem + P::XS, *A1t = smem + P::A1T, *gzot = smem + P::GZOT;
  T* Alt = NH == 2 ? smem + P::A2T : A1t;  // last hidden activations, unit-major; the gradient of Z_last replaces them in place
  constexpr bool WREG = wreg_b<H, NH>();
  stage_w<T>(a.W + a.woff[0], a.d0, H, K0, H, smem + P::W0R, P::LKH, WREG ? (T*)nullptr : smem + P::W0T, P::LK0);
  typename Ops<T>::v8 breg[WREG ? K0 / 32 : 1];
  if (WREG) {  // this wave's hidden units 16 wave .. +15, k = 32 ks + 8 (lane >> 4) .. +8
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < K0 / 32; ++ks) {
      typename Ops<T>::v8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ks * 32 + lk * 8 + e;
        const float wv = a.W[a.woff[0] + (int64_t)(k < a.d0 ? k : a.d0 - 1) * H + wave * 16 + lr];  // unconditional (clamped) load: see stage_w
        b[e] = Ops<T>::cvt(k < a.d0 ? wv : 0.f);
      }
      breg[ks] = b;
    }
  }
  if (NH == 2) stage_w<T>(a.W + a.woff[1], H, H, H, H, smem + P::W1R, P::LKH, smem + P::W1T, P::LKH);
  stage_w<T>(a.W + a.woff[NH], H, a.dout, H, 32, smem + P::WOR, P::LKO, nullptr, 0);
  stage_w<T>(a.W + a.woff[NH], H, a.dout, H, 16, nullptr, 0, smem + P::WOT, P::LKH);
  // rows 16..31 of gzot (the padded half of the K = 32 contraction over the outputs) stay zero for the whole kernel
  for (int idx = threadIdx.x; idx < 16 * P::LKT; idx += blockDim.x) gzot[16 * P::LKT + idx] = (T)0.f;
  const bool relu = a.hidden_act == 1;
  f32x4 dW0[NB0] = {};
  f32x4 dWh[NH == 2 ? NBH : 1] = {};
  f32x4 dWo[NBO] = {};
  typedef typename Ops<T>::v4 v4t;
  // (gradient block .* relu'(activation block)) written over the activation block: same lane, same 8 bytes
  auto mask_store = [&](T* At, int mt, int nt, f32x4 v) {
    const int col = nt * 16 + (lane & 15), row0 = mt * 16 + (lane >> 4) * 4;
    v4t* cell = reinterpret_cast<v4t*>(At + col * P::LKT + row0);
    if (relu) {
      const v4t act = *cell;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!((float)act[r] > 0.f)) v[r] = 0.f;
    }
    const v4t t = {Ops<T>::cvtg(v[0]), Ops<T>::cvtg(v[1]), Ops<T>::cvtg(v[2]), Ops<T>::cvtg(v[3])};
    *cell = t;
  };
  std::conditional_t<X16, XTile16<TS, K0, NW * 64, T>, XTileB<TS, K0, NW * 64>> xt;
  xt.template fetch<T>(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    xt.store(Xs, P::LK0, (T*)nullptr, 0);
    if (tile + gridDim.x < n_tiles) xt.template fetch<T>(a, (tile + gridDim.x) * TS);
    // this tile's incoming gradients, issued now and consumed by the output phase two barriers later
    constexpr int JO = (MT + NW - 1) / NW;
    float gyp[JO][4], gap[JO][4];
#pragma unroll
    for (int j = 0; j < JO; ++j) {
      const int mt = wave + NW * j, col = lane & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + mt * 16 + (lane >> 4) * 4 + r;
        const bool live = mt < MT && n < a.N && col < a.dout;
        gyp[j][r] = (live && a.gY) ? a.gY[n * a.ldgy + col] : 0.f;
        gap[j][r] = (live && a.gaux && col == a.aux_col) ? a.gaux[n] : 0.f;
      }
    }
    __syncthreads();
    // ---- hidden layer: A1t ----
#pragma unroll
    for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
      const int nt = wave + NW * j;
      if (nt < HT) {
        f32x4 acc[MT] = {};
        if (WREG) {
          const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
          for (int ks = 0; ks < K0 / 32; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = Ops<T>::mfma(ld8(Xs + (m * 16 + lr) * P::LK0 + ks * 32 + lk * 8), breg[WREG ? ks : 0], acc[m]);
        } else {
          mma_rr<MT, K0>(Xs, P::LK0, smem + P::W0T, P::LK0, nt, acc, lane);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 v = acc[m];
          if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
          store_rt<T>(nullptr, 0, A1t, P::LKT, m, nt, v, lane);
        }
      }
    }
    __syncthreads();
    if (NH == 2) {  // ---- second hidden layer: A2t = relu(A1 W1), A1 read transposed ----
#pragma unroll
      for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
        const int nt = wave + NW * j;
        if (nt < HT) {
          f32x4 acc[MT] = {};
          mma_tr<MT, H, T>(A1t, P::LKT, 0, smem + P::W1T, P::LKH, nt, acc, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            f32x4 v = acc[m];
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            store_rt<T>(nullptr, 0, Alt, P::LKT, m, nt, v, lane);
          }
        }
      }
      __syncthreads();
    }
    // ---- output layer forward + gradient w.r.t. its pre-activation: gzot rows 0..15 ----
#pragma unroll
    for (int j = 0; j < JO; ++j) {
      const int mt = wave + NW * j;
      if (mt < MT) {
        f32x4 acc[1] = {};
        mma_tr<1, H, T>(Alt, P::LKT, mt * 16, smem + P::WOT, P::LKH, 0, acc, lane);
        const int col = lane & 15;
        const int rl0 = mt * 16 + (lane >> 4) * 4;
        f32x4 gv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t n = n0 + rl0 + r;
          float g = 0.f;
          if (n < a.N && col < a.dout) {
            const float y = acc[0][r];
            g = gyp[j][r];
            if (a.out_act == 1) {
              const float sg = 1.f / (1.f + expf(-y));
              g = g * sg * (1.f - sg);
            }
            if (a.gaux && col == a.aux_col) g += gap[j][r] * expf(fminf(fmaxf(y, -15.f), 15.f));  // trunc_exp backward (activations.py:38-39)
          }
          gv[r] = g * GS;
        }
        store_rt<T>(nullptr, 0, gzot, P::LKT, mt, 0, gv, lane);
      }
    }
    __syncthreads();
    // ---- dWO += A_last^T gzo (contraction over the tile's samples) ----
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + NW * j;
      if (it < HT) {
        f32x4 acc[1] = {dWo[j]};
        mma_rr<1, TS>(Alt + it * 16 * P::LKT, P::LKT, gzot, P::LKT, 0, acc, lane);
        dWo[j] = acc[0];
      }
    }
    __syncthreads();  // Alt is overwritten below
    // ---- gradient of Z_last = (gzo WO^T) .* relu'(A_last), over A_last; gzo read transposed ----
#pragma unroll
    for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
      const int nt = wave + NW * j;
      if (nt < HT) {
        f32x4 acc[MT] = {};
        mma_tr<MT, 32, T>(gzot, P::LKT, 0, smem + P::WOR, P::LKO, nt, acc, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) mask_store(Alt, m, nt, acc[m]);
      }
    }
    __syncthreads();
    if (NH == 2) {
      // ---- dW1 += A1^T gz ----
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + NW * j;
        if (t < HT * HT) {
          f32x4 acc[1] = {dWh[j]};
          mma_rr<1, TS>(A1t + (t / HT) * 16 * P::LKT, P::LKT, Alt, P::LKT, t % HT, acc, lane);
          dWh[j] = acc[0];
        }
      }
      __syncthreads();  // A1t is overwritten below
      // ---- gradient of Z1 = (gz W1^T) .* relu'(A1), over A1; gz read transposed ----
#pragma unroll
      for (int j = 0; j < (HT + NW - 1) / NW; ++j) {
        const int nt = wave + NW * j;
        if (nt < HT) {
          f32x4 acc[MT] = {};
          mma_tr<MT, H, T>(Alt, P::LKT, 0, smem + P::W1R, P::LKH, nt, acc, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) mask_store(A1t, m, nt, acc[m]);
        }
      }
      __syncthreads();
    }
    // ---- dW0 += X^T gz1: X read transposed ([sample][feature] is the k-major image of X^T).  t = wave + NW j: with NW a multiple of HT the
    //      wave's hidden block (t % HT) is the same for every j, so its gz1 fragments are read once ----
    {
      static_assert(NW % HT == 0 || HT % NW == 0, "block-to-wave map of the layer-0 weight gradient");
      const int lr = lane & 15, lk = lane >> 4;
      if constexpr (NW % HT == 0) {
        const int nt = wave % HT;
        typename Ops<T>::v8 bf[TS / 32];
#pragma unroll
        for (int ks = 0; ks < TS / 32; ++ks) bf[ks] = ld8(A1t + (nt * 16 + lr) * P::LKT + ks * 32 + lk * 8);
#pragma unroll
        for (int j = 0; j < NB0; ++j) {
          const int t = wave + NW * j;
          if (t < K0T * HT) {
#pragma unroll
            for (int ks = 0; ks < TS / 32; ++ks) dW0[j] = Ops<T>::mfma(ld8_tr<T>(Xs, P::LK0, ks * 32, (t / HT) * 16, lane), bf[ks], dW0[j]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NB0; ++j) {
          const int t = wave + NW * j;
          if (t < K0T * HT) {
            f32x4 acc[1] = {dW0[j]};
            mma_tr<1, TS, T>(Xs, P::LK0, (t / HT) * 16, A1t, P::LKT, t % HT, acc, lane);
            dW0[j] = acc[0];
          }
        }
      }
    }
    // ---- gX = gz1 W0^T, gz1 read transposed: a wave keeps ONE row block's operand fragments (all of k) in registers and walks the
    //      column blocks with them -- the transposed reads are paid once per row block, not once per 16 x 16 output block ----
    if (a.gX) {
      static_assert(NW % MT == 0, "waves per workgroup must be a multiple of the tile's row blocks");
      constexpr int NG = NW / MT;
      const int m = wave % MT, grp = wave / MT, lr = lane & 15, lk = lane >> 4;
      typename Ops<T>::v8 af[H / 32];
#pragma unroll
      for (int ks = 0; ks < H / 32; ++ks) af[ks] = ld8_tr<T>(A1t, P::LKT, ks * 32, m * 16, lane);
#pragma unroll
      for (int jn = 0; jn < (K0T + NG - 1) / NG; ++jn) {
        const int nt = grp + NG * jn;
        if (nt < K0T) {
          f32x4 acc = {};
          const T* bp = smem + P::W0R + (nt * 16 + lr) * P::LKH + lk * 8;
#pragma unroll
          for (int ks = 0; ks < H / 32; ++ks) acc = Ops<T>::mfma(af[ks], ld8(bp + ks * 32), acc);
          const int col = nt * 16 + lr;
          const int64_t row0 = n0 + m * 16 + lk * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < a.N && col < a.d0) a.gX[(row0 + r) * a.ldgx + col] = acc[r] * (1.f / GS);
        }
      }
    }
  }
  // ---- flush weight gradients ----
  if (a.gW || a.gWfx || a.ws) {
    const int cl = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + NW * j;
      if (t < K0T * HT) {
        const int it = t / HT, nt = t % HT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = it * 16 + r0 + r;
          if (row < a.d0) gw_add(a, a.woff[0] + (int64_t)row * H + nt * 16 + cl, dW0[j][r] * (1.f / GS));
        }
      }
    }
    if (NH == 2) {
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + NW * j;
        if (t < HT * HT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) gw_add(a, a.woff[1] + (int64_t)((t / HT) * 16 + r0 + r) * H + (t % HT) * 16 + cl, dWh[j][r] * (1.f / GS));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + NW * j;
      if (it < HT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cl < a.dout) gw_add(a, a.woff[NH] + (int64_t)(it * 16 + r0 + r) * a.dout + cl, dWo[j][r] * (1.f / GS));
      }
    }
  }
}

template <typename T, int K0, int H, int NH, bool X16>
static int launch_b_tr(const MlpArgs& a, hipStream_t st) {
  constexpr int TS = 64;
  using P = PlanT<K0, H, NH, TS>;
  static_assert(P::BYTES <= LDS_LIMIT_B, "transposed-read backward tile does not fit LDS");
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  int per_cu = (int)(LDS_LIMIT_B / P::BYTES);
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  if (waves_b<H>() * per_cu > 8) per_cu = 8 / waves_b<H>();  // the kernels hold 130-250 VGPRs: two waves per SIMD
  int64_t grid = 256 * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  auto k = mlp_lp_bwd_tr_kernel<T, K0, H, NH, TS, X16>;
  SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves_b<H>() * 64), P::BYTES, st, a, n_tiles);
  SNERF_LAUNCH_CHECK("mlp_bwd (16-bit operands, transposed reads)");
  return 0;
}

template <typename T, int K0, int H, int NH>
static int launch_b(const MlpArgs& a, bool bwd, hipStream_t st) {
  // 64-wide nets (color_net, the proposal nets, NeRFPlayer's heads): the transposed-read kernel (color_net 0.149 -> 0.123 ms, proposal nets
  // 0.140 -> 0.124).  128-wide nets (sigma_net) stay on the two-image kernel: their time goes into the layer-0 weight gradient and the input
  // gradient over a 160-wide input, which read more and write less -- 0.142 ms against 0.147 (64-sample tiles) / 0.152 (128) transposed.
  if constexpr (H <= 64) {
    if (bwd) {
      if (a.x16) {
        set_error("mlp_bwd_x16: 16-bit inputs are built for the d_in -> 128 -> d_out one-hidden-layer shapes (sigma_net), got hidden=%d n_hidden=%d", H, NH);
        return SNERF_ERR_UNSUPPORTED;
      }
      return launch_b_tr<T, K0, H, NH, false>(a, st);
    }
  }
  if constexpr (H > 64) if (bwd) {
    constexpr int TS = 32;
    using P = PlanB<K0, H, NH, TS>;
    static_assert(P::BYTES <= LDS_LIMIT_B, "bf16 backward tile does not fit LDS");
    int64_t n_tiles = (a.N + TS - 1) / TS;
    int per_cu = (int)(LDS_LIMIT_B / P::BYTES);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (a.x16) {
      // built for the sigma_net shapes (d_in = 32 n_scales -> 128 -> 16): what the fused field forward hands over
      if constexpr (H == 128 && NH == 1) {
        SNERF_REQUIRE(a.d0 == K0 && a.ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0,
                      "mlp_bwd_x16: needs d_in a multiple of 32 (%d), ldx a multiple of 8 (%d) and a 16-byte aligned X", a.d0, a.ldx);
        // 64-sample tiles where they fit (the 160 -> 128 net of the preset: 146 KB; the 16-bit X prefetch is small enough for the registers,
        // the fp32 one below is not: 93 spilled VGPRs)
        constexpr int TS16 = PlanB<K0, H, NH, 64>::BYTES <= LDS_LIMIT_B ? 64 : 32;
        using P16 = PlanB<K0, H, NH, TS16>;
        n_tiles = (a.N + TS16 - 1) / TS16;
        grid = n_tiles < 256 ? n_tiles : 256;
        if (a.G) {
          auto k = mlp_lp_bwd_kernel<T, K0, H, NH, TS16, true, true>;
          SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
          hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves_b<H>() * 64), P16::BYTES, st, a, n_tiles);
        } else {
          auto k = mlp_lp_bwd_kernel<T, K0, H, NH, TS16, true>;
          SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
          hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves_b<H>() * 64), P16::BYTES, st, a, n_tiles);
        }
      } else {
        set_error("mlp_bwd_x16: 16-bit inputs are built for the d_in -> 128 -> d_out one-hidden-layer shapes (sigma_net), got hidden=%d n_hidden=%d", H, NH);
        return SNERF_ERR_UNSUPPORTED;
      }
    } else {
      auto k = mlp_lp_bwd_kernel<T, K0, H, NH, TS>;
      SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
      hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves_b<H>() * 64), P::BYTES, st, a, n_tiles);
    }
    SNERF_LAUNCH_CHECK("mlp_bwd (16-bit operands)");
    return 0;
  }
  {
    constexpr int TS = (H <= 64 && NH == 1) ? 64 : 32;
    using P = PlanF<K0, H, NH, TS>;
    const int64_t n_tiles = (a.N + TS - 1) / TS;
    int per_cu = (int)(LDS_LIMIT_B / P::BYTES);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    auto k = mlp_lp_fwd_kernel<T, K0, H, NH, TS>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves_b<H>() * 64), P::BYTES, st, a, n_tiles);
  }
  SNERF_LAUNCH_CHECK("mlp_fwd (16-bit operands)");
  return 0;
}

// (input width padded to 32, hidden width, hidden layers)
#define SNERF_MLP_BF16_SHAPES(X) X(32, 64, 1) X(32, 128, 1) X(64, 128, 1) X(96, 128, 1) X(128, 128, 1) X(160, 128, 1) X(192, 128, 1) /* 6 scales: BASELINE config 3 */ \
  X(32, 64, 2) /* K-Planes color_net 15->64->64->3 */ X(64, 64, 2) /* nerfplayer mlp_head 63->64->64->3 */

bool mlp_bf16_supported(const snerf_mlp_desc* d) {
  if (!d || d->d_out > 16 || d->d_in < 1) return false;
  const int k0 = (d->d_in + 31) / 32 * 32;
#define CASE(K0, H, NH) \
  if (k0 == K0 && d->hidden == H && d->n_hidden == NH) return true;
  SNERF_MLP_BF16_SHAPES(CASE)
#undef CASE
  return false;
}

// wave-owns-rows backward of the 64-wide nets (mlp_rows.hip)
bool mlp_rows_supported(const snerf_mlp_desc* d, const void* args);
int mlp_rows_dispatch(const snerf_mlp_desc* d, const void* args, hipStream_t st);

// wave-owns-rows backward of sigma_net (mlp_rows128.hip)
bool mlp_rows128_supported(const snerf_mlp_desc* d, const void* args);
int mlp_rows128_dispatch(const snerf_mlp_desc* d, const void* args, hipStream_t st);
// SNERF_MLP_SIGMA_ROWS=0: dev A-B switch back to the workgroup-tile kernel for sigma_net's 16-bit-input backward (read per call)
static bool sigma_rows_off() {
  const char* e = getenv("SNERF_MLP_SIGMA_ROWS");
  return e && atoi(e) == 0;
}

int mlp_bf16_dispatch(const snerf_mlp_desc* d, const void* args, bool bwd, hipStream_t st) {
  const MlpArgs& a = *static_cast<const MlpArgs*>(args);
  if (bwd && a.variant == 0 && mlp_rows_supported(d, args)) return mlp_rows_dispatch(d, args, st);
  if (bwd && a.variant == 0 && !a.gWfx && mlp_rows128_supported(d, args) && !sigma_rows_off()) return mlp_rows128_dispatch(d, args, st);
  const int k0 = (d->d_in + 31) / 32 * 32;
#define CASE(K0, H, NH)                                                                   \
  if (k0 == K0 && d->hidden == H && d->n_hidden == NH)                                    \
    return d->operands == 2 ? launch_b<fp16, K0, H, NH>(a, bwd, st) : launch_b<bf16, K0, H, NH>(a, bwd, st);
  SNERF_MLP_BF16_SHAPES(CASE)
#undef CASE
  set_error("mlp (16-bit operands): unsupported shape d_in=%d hidden=%d n_hidden=%d", d->d_in, d->hidden, d->n_hidden);
  return SNERF_ERR_UNSUPPORTED;
}

}  // namespace snerf

// Single dense layers with 16-bit MFMA operands (bf16, fp32 accumulation): Y = act(X W) and its backward, the 16-bit twin of mlp.hip's
// dense_fwd_kernel / dense_bwd_kernel.  The nets of the full NeRFPlayer field that lie outside the fused kernels' shape table (deformation net
// 3 -> 128 x 3 -> 3, colour head 15 -> 64 x 3 -> 3, the 33 -> 64 -> 32 MLP: NS/fields/nerfplayer_field.py:231-316) are chained from single layers; in
// fp32 (v_mfma_f32_16x16x4_f32) their backward was 2.7 ms of a 6.95-ms step (profiles/r05_nerfplayer_full_timeline.txt), the one kernel family of this
// library still on the fp32 matrix rate while the reference runs these nets in tcnn's fp16.
//
// Every product is written "both operands row-major along the contraction index" (mlp_lp_common.hpp: mma_rr, v_mfma_f32_16x16x32_bf16 fed by two
// ds_read_b128 per lane and instruction), which fixes the LDS images:
//   forward   Y  = X W        A = Xs  [TS][K]  (the tile as loaded)           B^T = WT  [M][K]  (W transposed, resident)
//   backward  gX = dZ W^T     A = Gs  [TS][M]  (dZ = dY .* act'(Y))           B^T = WR  [K][M]  (W as stored, resident)
//             dW = X^T dZ     A = XsT [K][TS]  (the tile written transposed)  B^T = GsT [M][TS] (dZ written transposed)
// The tile of the next iteration (X; dY and Y in the backward) is in registers while the current one is multiplied: unconditional loads from clamped
// addresses, uniform tile base + 32-bit offsets.  Weight gradients stay in registers for the whole persistent loop (wave w owns column blocks w, w + NW,
// ...) and are flushed once per workgroup through gw_add (float atomics or fixed-point cells).  K, M <= 128; TS = 64 rows per iteration.
#include "mlp_lp_common.hpp"

namespace snerf {
namespace dlp {

constexpr int TS = 64;
// 8 waves only where the weight-gradient blocks need them (128 x 128: 64 accumulator blocks); everywhere else 4 -- an 8-wave workgroup with 60+ KB of LDS
// starves beside the optimiser sweeps that run on the side stream (the 128 -> 3 layer's backward took 1.15 ms per launch with 8 waves, round 5)
template <int KP, int MP>
constexpr int waves() { return (KP >= 128 && MP >= 128) ? 8 : 4; }

// TS x C floats of a row-major global matrix, one tile ahead in registers.  base = first row of the tile; rmax = last valid row inside the tile.
template <int C, int NT>
struct Tile {
  static_assert((TS * C) % NT == 0, "tile must divide over the workgroup");
  static constexpr int PER = TS * C / NT;
  float v[PER];
  __device__ __forceinline__ void fetch(const float* __restrict__ base, int ld, int rmax, int c_act) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / C, c = idx - r * C;
      v[i] = base[(r < rmax ? r : rmax) * ld + (c < c_act ? c : c_act - 1)];
    }
  }
};

template <typename T, int KP, int MP>
struct FwdPlan {
  static constexpr int WT = 0;                         // [MP][ldb(KP)]
  static constexpr int XS = WT + MP * ldb(KP);         // [TS][ldb(KP)]
  static constexpr int TOTAL = XS + TS * ldb(KP);
  static constexpr size_t BYTES = (size_t)TOTAL * sizeof(T);
};
template <typename T, int KP, int MP>
struct BwdPlan {
  static constexpr int WR = 0;                         // [KP][ldb(MP)]
  static constexpr int XST = WR + KP * ldb(MP);        // [KP][ldb(TS)]
  static constexpr int GS = XST + KP * ldb(TS);        // [TS][ldb(MP)]
  static constexpr int GST = GS + TS * ldb(MP);        // [MP][ldb(TS)]
  static constexpr int TOTAL = GST + MP * ldb(TS);
  static constexpr size_t BYTES = (size_t)TOTAL * sizeof(T);
};

// MlpArgs reuse (as mlp.hip's dense kernels): d0 = K, dout = M, W = the [K][M] matrix, hidden_act = 1 -> ReLU, out_act = 1 -> Sigmoid (at most one set)
template <typename T, int KP, int MP>
__global__ __launch_bounds__((waves<KP, MP>() * 64)) void dense_lp_fwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  using P = FwdPlan<T, KP, MP>;
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr int NW = waves<KP, MP>(), NT = NW * 64, NTB = MP / 16, MT = TS / 16, LX = ldb(KP);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T* Xs = smem + P::XS;
  stage_w<T>(a.W, a.d0, a.dout, KP, MP, nullptr, 0, smem + P::WT, LX);
  Tile<KP, NT> xt;
  auto last_row = [&](int64_t n0) { return (int)((a.N - 1 - n0) < (int64_t)(TS - 1) ? (a.N - 1 - n0) : (int64_t)(TS - 1)); };
  xt.fetch(a.X + (int64_t)blockIdx.x * TS * a.ldx, a.ldx, last_row((int64_t)blockIdx.x * TS), a.d0);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    const int rmax = last_row(n0);
    __syncthreads();  // the previous tile's products have read Xs (first iteration: nothing to wait for but the staging of W below)
#pragma unroll
    for (int i = 0; i < Tile<KP, NT>::PER; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / KP, c = idx - r * KP;
      Xs[r * LX + c] = Ops<T>::cvt((r <= rmax && c < a.d0) ? xt.v[i] : 0.f);
    }
    if (tile + gridDim.x < n_tiles) {
      const int64_t n1 = (tile + gridDim.x) * TS;
      xt.fetch(a.X + n1 * a.ldx, a.ldx, last_row(n1), a.d0);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (NTB + NW - 1) / NW; ++j) {
      const int nt = wave + NW * j;
      if (nt < NTB) {
        f32x4 acc[MT] = {};
        mma_rr<MT, KP, T>(Xs, LX, smem + P::WT, LX, nt, acc, lane);
        const int col = nt * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int64_t row0 = n0 + m * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (row0 + r < a.N && col < a.dout) {
              float y = acc[m][r];
              if (a.hidden_act == 1) y = fmaxf(y, 0.f);
              if (a.out_act == 1) y = 1.f / (1.f + expf(-y));
              a.Y[(row0 + r) * a.ldy + col] = y;
            }
          }
        }
      }
    }
  }
}

// a.Y is the layer's stored OUTPUT (post-activation), read-only: dZ = dY .* act'(Y)
template <typename T, int KP, int MP>
__global__ __launch_bounds__((waves<KP, MP>() * 64)) void dense_lp_bwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  using P = BwdPlan<T, KP, MP>;
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr int NW = waves<KP, MP>(), NT = NW * 64, KT = KP / 16, NTB = MP / 16, MT = TS / 16;
  constexpr int NJ = (NTB + NW - 1) / NW, LM = ldb(MP), LT = ldb(TS);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T *WR = smem + P::WR, *XsT = smem + P::XST, *Gs = smem + P::GS, *GsT = smem + P::GST;
  stage_w<T>(a.W, a.d0, a.dout, KP, MP, WR, LM, nullptr, 0);
  f32x4 dW[NJ][KT] = {};
  Tile<KP, NT> xt;
  Tile<MP, NT> gt, yt;
  const bool want_dw = a.gW || a.gWfx || a.ws;
  auto last_row = [&](int64_t n0) { return (int)((a.N - 1 - n0) < (int64_t)(TS - 1) ? (a.N - 1 - n0) : (int64_t)(TS - 1)); };
  auto fetch = [&](int64_t n0) {
    const int rmax = last_row(n0);
    if (want_dw) xt.fetch(a.X + n0 * a.ldx, a.ldx, rmax, a.d0);
    gt.fetch(a.gY + n0 * a.ldgy, a.ldgy, rmax, a.dout);
    yt.fetch(a.Y + n0 * a.ldy, a.ldy, rmax, a.dout);
  };
  fetch((int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    const int rmax = last_row(n0);
    __syncthreads();
    if (want_dw) {
#pragma unroll
      for (int i = 0; i < Tile<KP, NT>::PER; ++i) {
        const int idx = threadIdx.x + i * NT;
        const int r = idx / KP, c = idx - r * KP;
        XsT[c * LT + r] = Ops<T>::cvt((r <= rmax && c < a.d0) ? xt.v[i] : 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < Tile<MP, NT>::PER; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / MP, c = idx - r * MP;
      float g = gt.v[i];
      const float y = yt.v[i];
      if (a.hidden_act == 1) g = y > 0.f ? g : 0.f;
      if (a.out_act == 1) g = g * y * (1.f - y);
      const T b = Ops<T>::cvt((r <= rmax && c < a.dout) ? g : 0.f);
      Gs[r * LM + c] = b;
      GsT[c * LT + r] = b;
    }
    if (tile + gridDim.x < n_tiles) fetch((tile + gridDim.x) * TS);
    __syncthreads();
    if (want_dw) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nt = wave + NW * j;
        if (nt < NTB) mma_rr<KT, TS, T>(XsT, LT, GsT, LT, nt, dW[j], lane);  // dW[kc block][m block nt] += X^T dZ over the tile's rows
      }
    }
    if (a.gX) {
#pragma unroll
      for (int j = 0; j < (KT + NW - 1) / NW; ++j) {
        const int kt = wave + NW * j;
        if (kt < KT) {
          f32x4 acc[MT] = {};
          mma_rr<MT, MP, T>(Gs, LM, WR, LM, kt, acc, lane);
          const int col = kt * 16 + (lane & 15);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int64_t row0 = n0 + m * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (row0 + r < a.N && col < a.d0) a.gX[(row0 + r) * a.ldgx + col] = acc[m][r];
          }
        }
      }
    }
  }
  if (want_dw) {
    const int cl = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nt = wave + NW * j;
      if (nt < NTB) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = kt * 16 + r0 + r, col = nt * 16 + cl;
            if (row < a.d0 && col < a.dout) gw_add(a, (int64_t)row * a.dout + col, dW[j][kt][r]);
          }
        }
      }
    }
  }
}

template <typename T, int KP, int MP>
static int launch(const MlpArgs& a, bool bwd, hipStream_t st) {
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  const size_t bytes = bwd ? BwdPlan<T, KP, MP>::BYTES : FwdPlan<T, KP, MP>::BYTES;
  static_assert(BwdPlan<T, KP, MP>::BYTES <= LDS_LIMIT_B, "dense_lp backward does not fit LDS");
  int per_cu = (int)(LDS_LIMIT_B / bytes);
  const int by_waves = 16 / waves<KP, MP>();  // at most 4 waves per SIMD worth of workgroups; registers decide the rest
  per_cu = per_cu < 1 ? 1 : (per_cu > by_waves ? by_waves : per_cu);
  if (per_cu > 2) per_cu = 2;
  int64_t grid = 256 * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  if (bwd) {
    auto k = dense_lp_bwd_kernel<T, KP, MP>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves<KP, MP>() * 64), bytes, st, a, n_tiles);
  } else {
    auto k = dense_lp_fwd_kernel<T, KP, MP>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves<KP, MP>() * 64), bytes, st, a, n_tiles);
  }
  SNERF_LAUNCH_CHECK(bwd ? "dense_bwd_lp" : "dense_fwd_lp");
  return 0;
}

static int dispatch(int K, int M, const MlpArgs& a, bool bwd, hipStream_t st) {
  const int kp = K <= 32 ? 32 : (K <= 64 ? 64 : 128);
  const int mp = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
#define CASE(KP, MP) \
  if (kp == KP && mp == MP) return launch<bf16, KP, MP>(a, bwd, st);
  CASE(32, 32) CASE(32, 64) CASE(32, 128) CASE(64, 32) CASE(64, 64) CASE(64, 128) CASE(128, 32) CASE(128, 64) CASE(128, 128)
#undef CASE
  set_error("dense_lp: no kernel for K=%d M=%d", K, M);
  return 1;
}

}  // namespace dlp
}  // namespace snerf

using namespace snerf;

// snerf.h (ABI 13): 1 iff the 16-bit dense-layer kernels are built for this shape and operand type (bf16 = 1; K, M <= 128)
extern "C" int snerf_dense_lp_supported(int32_t K, int32_t M, int32_t operands) {
  return operands == 1 && K >= 1 && K <= 128 && M >= 1 && M <= 128;
}

extern "C" int snerf_dense_fwd_lp(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, float* Y, int32_t ldy,
                                  int32_t operands, snerf_stream_t stream) {
  SNERF_REQUIRE(snerf_dense_lp_supported(K, M, operands), "dense_fwd_lp: K=%d M=%d operands=%d (bf16 = 1; K, M <= 128)", K, M, operands);
  SNERF_REQUIRE(act >= 0 && act <= 2 && N >= 0 && ldx >= K && ldy >= M, "dense_fwd_lp: act=%d N=%lld ldx=%d ldy=%d", act, (long long)N, ldx, ldy);
  if (N == 0) return 0;
  SNERF_REQUIRE(W && X && Y, "dense_fwd_lp: null buffer");
  MlpArgs a = {};
  a.X = X; a.N = N; a.ldx = ldx; a.d0 = K; a.W = W; a.dout = M; a.Y = Y; a.ldy = ldy; a.hidden_act = act == 1; a.out_act = act == 2;
  return dlp::dispatch(K, M, a, false, (hipStream_t)stream);
}

extern "C" int snerf_dense_bwd_lp(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, const float* Y, int32_t ldy,
                                  const float* gY, int32_t ldgy, float* gX, int32_t ldgx, float* gW, int64_t* gW_fx, int32_t operands,
                                  snerf_stream_t stream) {
  SNERF_REQUIRE(snerf_dense_lp_supported(K, M, operands), "dense_bwd_lp: K=%d M=%d operands=%d (bf16 = 1; K, M <= 128)", K, M, operands);
  SNERF_REQUIRE(act >= 0 && act <= 2 && N >= 0 && ldx >= K && ldy >= M && ldgy >= M && (!gX || ldgx >= K), "dense_bwd_lp: act=%d N=%lld ldx=%d ldy=%d ldgy=%d ldgx=%d",
                act, (long long)N, ldx, ldy, ldgy, ldgx);
  SNERF_REQUIRE(!(gW && gW_fx), "dense_bwd_lp: give gW or gW_fx, not both");
  if (N == 0) return 0;
  SNERF_REQUIRE(W && X && Y && gY && (gX || gW || gW_fx), "dense_bwd_lp: null buffer");
  MlpArgs a = {};
  a.X = X; a.N = N; a.ldx = ldx; a.d0 = K; a.W = W; a.dout = M; a.Y = const_cast<float*>(Y); a.ldy = ldy; a.hidden_act = act == 1; a.out_act = act == 2;
  a.gY = gY; a.ldgy = ldgy; a.gX = gX; a.ldgx = ldgx; a.gW = gW; a.gWfx = reinterpret_cast<long long*>(gW_fx);
  return dlp::dispatch(K, M, a, true, (hipStream_t)stream);
}

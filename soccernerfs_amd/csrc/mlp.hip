// Tiny bias-free MLPs (the reference's tcnn.Network / FullyFusedMLP call sites) on the matrix cores.
//
// Replaces tcnn.Network(n_input_dims, n_output_dims, {FullyFusedMLP, ReLU, None|Sigmoid, n_neurons, n_hidden_layers})
// at NS/fields/kplanes_field.py:249-273,397-407 and NS/fields/nerfplayer_nerfacto_field.py:94-104,238-248,301-311.
//
// Numerics: exact fp32 -- v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md §3),
// so the fp32 parity bar (rtol 1e-5) against the CPU oracle holds; tcnn itself computes in fp16 (>= reference precision).
// One 256-thread workgroup (4 waves) walks 64-sample tiles persistently: activations live in LDS (row stride = 2 mod 32
// floats => conflict-free A-operand reads), weights (<= 112 KB per net, L2-resident) stream from global as B operands,
// one load per k-step reused across the tile's 4 row blocks.  The backward recomputes the forward per tile (nothing but
// X and dY ever touches HBM), keeps the weight-gradient accumulators in registers across the whole persistent loop and
// flushes them once per workgroup with 64-B-contiguous atomics.
#include "common.hpp"

namespace snerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TS = 64;   // samples per tile
constexpr int MT = 4;    // 16-row blocks per tile
constexpr int OUTP = 16; // padded output width (all nets here have <= 16 outputs)

struct MlpArgs {
  const float* X; int64_t N; int ldx; int d0;
  const float* W; int woff[4];
  int dout;
  float* Y; int ldy;
  int hidden_act, out_act;
  int aux_col; float* aux_out;
  const float* gY; int ldgy;
  const float* gaux;
  float* gX; int ldgx;
  float* gW;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[m] (m = 0..3 row blocks) += A[64 x K] (LDS, stride lda) * Wg[K x ldw] column block nt.
// Wg row-major with `ldw` columns; rows >= Kact / cols >= Nact read as zero.
__device__ __forceinline__ void mma_cols(const float* As, int lda, int K, const float* __restrict__ Wg, int ldw, int Kact, int Nact, int nt,
                                         f32x4 (&acc)[MT], int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const int col = nt * 16 + lr;
  const bool colok = col < Nact;
#pragma unroll 4
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + lk;
    float b = (colok && k < Kact) ? Wg[(int64_t)k * ldw + col] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float a = As[(m * 16 + lr) * lda + k];
      acc[m] = mfma4(a, b, acc[m]);
    }
  }
}

// acc (one 16x16 block: rows mt*16.., cols nt*16..) += A[.. x K] * Wg
__device__ __forceinline__ void mma_one(const float* As, int lda, int K, const float* __restrict__ Wg, int ldw, int Kact, int Nact, int mt, int nt,
                                        f32x4& acc, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const int col = nt * 16 + lr;
  const bool colok = col < Nact;
#pragma unroll 8
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + lk;
    float b = (colok && k < Kact) ? Wg[(int64_t)k * ldw + col] : 0.f;
    float a = As[(mt * 16 + lr) * lda + k];
    acc = mfma4(a, b, acc);
  }
}

// acc[m] += G[64 x K] (LDS) * Wg^T, i.e. B[k][n] = Wg[n][k]; output column block nt indexes Wg ROWS. (dX = dZ * W^T)
__device__ __forceinline__ void mma_cols_T(const float* Gs, int ldg, int K, const float* __restrict__ Wg, int ldw, int Kact, int Nact, int nt,
                                           f32x4 (&acc)[MT], int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const int row = nt * 16 + lr;  // row of Wg = output column
  const bool rowok = row < Nact;
#pragma unroll 4
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + lk;
    float b = (rowok && k < Kact) ? Wg[(int64_t)row * ldw + k] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float a = Gs[(m * 16 + lr) * ldg + k];
      acc[m] = mfma4(a, b, acc[m]);
    }
  }
}

// acc (16x16 block it,nt of dW) += A^T[K-block it][64 samples] * G[64][N-block nt]
__device__ __forceinline__ void mma_outer(const float* As, int lda, const float* Gs, int ldg, int it, int nt, f32x4& acc, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s0 = 0; s0 < TS; s0 += 4) {
    float a = As[(s0 + lk) * lda + it * 16 + lr];
    float b = Gs[(s0 + lk) * ldg + nt * 16 + lr];
    acc = mfma4(a, b, acc);
  }
}

// store a 16x16 accumulator block into an LDS activation tile (rows mt*16.., cols nt*16..), optional ReLU
__device__ __forceinline__ void store_block(float* Ys, int ldy, int mt, int nt, const f32x4& acc, bool relu, int lane) {
  const int col = nt * 16 + (lane & 15);
  const int row0 = mt * 16 + (lane >> 4) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = acc[r];
    if (relu) v = fmaxf(v, 0.f);
    Ys[(row0 + r) * ldy + col] = v;
  }
}

__device__ __forceinline__ int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ constexpr int ld_of(int width) { return ((width + 31) / 32) * 32 + 2; }  // = 2 mod 32

// Loads the X tile (rows n0.., d0 columns, zero padded to K0 columns and TS rows) into LDS.
__device__ __forceinline__ void load_x_tile(const MlpArgs& a, int64_t n0, float* Xs, int ldxs, int K0) {
  for (int idx = threadIdx.x; idx < TS * K0; idx += blockDim.x) {
    int r = idx / K0, c = idx - r * K0;
    int64_t n = n0 + r;
    Xs[r * ldxs + c] = (n < a.N && c < a.d0) ? a.X[n * a.ldx + c] : 0.f;
  }
}

// Forward through the hidden layers for one tile; leaves A_l (post-activation) in act[l] (l = 1..NH).
template <int D0P, int H, int NH>
__device__ __forceinline__ void forward_hidden(const MlpArgs& a, float* const* act, const int* lds, int wave, int lane) {
  constexpr int HT = H / 16;
  const bool relu = a.hidden_act == 1;
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    const float* in = act[l];
    float* out = act[l + 1];
    const int K = l == 0 ? D0P : H;
    const int Kact = l == 0 ? a.d0 : H;
    const float* Wl = a.W + a.woff[l];
    if (HT >= 4) {
#pragma unroll
      for (int j = 0; j < (HT >= 4 ? HT / 4 : 1); ++j) {
        const int nt = wave + 4 * j;
        f32x4 acc[MT] = {};
        mma_cols(in, lds[l], K, Wl, H, Kact, H, nt, acc, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) store_block(out, lds[l + 1], m, nt, acc[m], relu, lane);
      }
    } else {  // H == 16: one column block, waves split the row blocks
      f32x4 acc = {};
      mma_one(in, lds[l], K, Wl, H, Kact, H, wave, 0, acc, lane);
      store_block(out, lds[l + 1], wave, 0, acc, relu, lane);
    }
    __syncthreads();
  }
}

template <int D0P, int H, int NH>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) float smem[];
  constexpr int LD0 = ld_of(D0P), LDH = ld_of(H);
  float* act[NH + 1];
  int lds[NH + 1];
  act[0] = smem; lds[0] = LD0;
  {
    float* p = smem + TS * LD0;
#pragma unroll
    for (int l = 1; l <= NH; ++l) { act[l] = p; lds[l] = LDH; p += TS * LDH; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* Wout = a.W + a.woff[NH];
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();  // previous tile's readers are done with LDS
    load_x_tile(a, n0, act[0], LD0, D0P);
    __syncthreads();
    forward_hidden<D0P, H, NH>(a, act, lds, wave, lane);
    // output layer: wave w -> row block w
    f32x4 acc = {};
    mma_one(act[NH], lds[NH], H, Wout, a.dout, H, a.dout, wave, 0, acc, lane);
    const int col = lane & 15;
    const int64_t row0 = n0 + wave * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = row0 + r;
      if (n < a.N && col < a.dout) {
        float y = acc[r];
        if (a.aux_out && col == a.aux_col) a.aux_out[n] = expf(y);  // trunc_exp forward (activations.py:32)
        if (a.out_act == 1) y = 1.f / (1.f + expf(-y));
        a.Y[n * a.ldy + col] = y;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int D0P, int H, int NH>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) float smem[];
  constexpr int LD0 = ld_of(D0P), LDH = ld_of(H), LDO = ld_of(OUTP);
  constexpr int D0T = D0P / 16, HT = H / 16;
  static_assert(D0P % 16 == 0 && H % 16 == 0, "dims are padded to 16");
  static_assert(NH == 1 || HT >= 4, "two hidden layers need hidden >= 64");
  float* act[NH + 1];
  int lds[NH + 1];
  act[0] = smem; lds[0] = LD0;
  float* p = smem + TS * LD0;
#pragma unroll
  for (int l = 1; l <= NH; ++l) { act[l] = p; lds[l] = LDH; p += TS * LDH; }
  float* gz = p;            // [TS][LDH] grad wrt the current hidden pre-activation
  p += TS * LDH;
  float* gzo = p;           // [TS][LDO] grad wrt the output pre-activation
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool relu = a.hidden_act == 1;

  // weight-gradient accumulators, alive across the persistent loop; block t of a layer is owned by wave t % 4
  constexpr int NB0 = (D0T * HT + 3) / 4;   // layer 0: [D0P x H]
  constexpr int NBH = (HT * HT + 3) / 4;    // hidden->hidden (NH == 2)
  constexpr int NBO = (HT + 3) / 4;         // output: [H x 16]
  f32x4 dW0[NB0] = {};
  f32x4 dWh[NH == 2 ? NBH : 1] = {};
  f32x4 dWo[NBO] = {};

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    load_x_tile(a, n0, act[0], LD0, D0P);
    __syncthreads();
    forward_hidden<D0P, H, NH>(a, act, lds, wave, lane);
    // ---- output layer forward (needed for sigmoid' / trunc_exp') and grad wrt its pre-activation ----
    {
      f32x4 acc = {};
      mma_one(act[NH], lds[NH], H, a.W + a.woff[NH], a.dout, H, a.dout, wave, 0, acc, lane);
      const int col = lane & 15;
      const int rl0 = wave * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + rl0 + r;
        float g = 0.f;
        if (n < a.N && col < a.dout) {
          float y = acc[r];
          if (a.gY) g = a.gY[n * a.ldgy + col];
          if (a.out_act == 1) {
            float sg = 1.f / (1.f + expf(-y));
            g = g * sg * (1.f - sg);
          }
          if (a.gaux && col == a.aux_col) g += a.gaux[n] * expf(fminf(fmaxf(y, -15.f), 15.f));  // trunc_exp backward (activations.py:38-39)
        }
        gzo[(rl0 + r) * LDO + col] = g;
      }
    }
    __syncthreads();
    // ---- dW_out += A_NH^T * gzo ;  gA_NH = gzo * W_out^T ----
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + 4 * j;
      if (it < HT) mma_outer(act[NH], lds[NH], gzo, LDO, it, 0, dWo[j], lane);
    }
    // gz = (gzo * W_out^T) .* relu'(A_NH): output width H, K = 16
    {
      const float* Wo = a.W + a.woff[NH];
      if (HT >= 4) {
#pragma unroll
        for (int j = 0; j < (HT >= 4 ? HT / 4 : 1); ++j) {
          const int nt = wave + 4 * j;
          f32x4 acc[MT] = {};
          mma_cols_T(gzo, LDO, OUTP, Wo, a.dout, a.dout, H, nt, acc, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int col = nt * 16 + (lane & 15);
            const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[m][r];
              if (relu && !(act[NH][(row0 + r) * LDH + col] > 0.f)) v = 0.f;
              gz[(row0 + r) * LDH + col] = v;
            }
          }
        }
      } else {
        f32x4 acc[MT] = {};
        if (wave == 0) {
          mma_cols_T(gzo, LDO, OUTP, Wo, a.dout, a.dout, H, 0, acc, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int col = lane & 15;
            const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = acc[m][r];
              if (relu && !(act[NH][(row0 + r) * LDH + col] > 0.f)) v = 0.f;
              gz[(row0 + r) * LDH + col] = v;
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- hidden->hidden layer (NH == 2): dW_h += A_1^T * gz ; gz' = (gz * W_h^T) .* relu'(A_1) ----
    if (NH == 2) {
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + 4 * j;
        if (t < HT * HT) mma_outer(act[1], LDH, gz, LDH, t / HT, t % HT, dWh[j], lane);
      }
      const float* Wh = a.W + a.woff[1];
      f32x4 acc2[(HT >= 4 ? HT / 4 : 1)][MT] = {};
#pragma unroll
      for (int j = 0; j < (HT >= 4 ? HT / 4 : 1); ++j) mma_cols_T(gz, LDH, H, Wh, H, H, H, wave + 4 * j, acc2[j], lane);
      __syncthreads();  // everyone finished reading gz
#pragma unroll
      for (int j = 0; j < (HT >= 4 ? HT / 4 : 1); ++j) {
        const int nt = wave + 4 * j;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int col = nt * 16 + (lane & 15);
          const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc2[j][m][r];
            if (relu && !(act[1][(row0 + r) * LDH + col] > 0.f)) v = 0.f;
            gz[(row0 + r) * LDH + col] = v;
          }
        }
      }
      __syncthreads();
    }
    // ---- first layer: dW_0 += X^T * gz ; gX = gz * W_0^T ----
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + 4 * j;
      if (t < D0T * HT) mma_outer(act[0], LD0, gz, LDH, t / HT, t % HT, dW0[j], lane);
    }
    if (a.gX) {
      const float* W0 = a.W + a.woff[0];
#pragma unroll
      for (int j = 0; j < (D0T + 3) / 4; ++j) {
        const int nt = wave + 4 * j;
        if (nt < D0T) {
          f32x4 acc[MT] = {};
          mma_cols_T(gz, LDH, H, W0, H, H, a.d0, nt, acc, lane);
          const int col = nt * 16 + (lane & 15);
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

  // ---- flush weight gradients (each 16-lane group adds 64 contiguous bytes) ----
  if (a.gW) {
    const int cl = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + 4 * j;
      if (t < D0T * HT) {
        const int it = t / HT, nt = t % HT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = it * 16 + r0 + r, col = nt * 16 + cl;
          if (row < a.d0) atomicAdd(a.gW + a.woff[0] + (int64_t)row * H + col, dW0[j][r]);
        }
      }
    }
    if (NH == 2) {
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + 4 * j;
        if (t < HT * HT) {
          const int it = t / HT, nt = t % HT;
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(a.gW + a.woff[1] + (int64_t)(it * 16 + r0 + r) * H + nt * 16 + cl, dWh[j][r]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + 4 * j;
      if (it < HT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cl < a.dout) atomicAdd(a.gW + a.woff[NH] + (int64_t)(it * 16 + r0 + r) * a.dout + cl, dWo[j][r]);
      }
    }
  }
}

template <int D0P, int H, int NH>
static int launch(const MlpArgs& a, bool bwd, hipStream_t st) {
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  size_t lds = (size_t)TS * (ld_of(D0P) + NH * ld_of(H)) * sizeof(float);
  if (bwd) lds += (size_t)TS * (ld_of(H) + ld_of(OUTP)) * sizeof(float);
  int blocks_per_cu = (int)(160 * 1024 / lds);
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  if (blocks_per_cu > 4) blocks_per_cu = 4;
  int64_t grid = 256 * blocks_per_cu;
  if (grid > n_tiles) grid = n_tiles;
  if (bwd) {
    auto k = mlp_bwd_kernel<D0P, H, NH>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, a, n_tiles);
  } else {
    auto k = mlp_fwd_kernel<D0P, H, NH>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, a, n_tiles);
  }
  SNERF_LAUNCH_CHECK(bwd ? "mlp_bwd" : "mlp_fwd");
  return 0;
}

static int dispatch(const snerf_mlp_desc* d, const MlpArgs& a, bool bwd, hipStream_t st) {
  const int d0p = (d->d_in + 15) / 16 * 16;
#define CASE(D0P, H, NH) \
  if (d0p == D0P && d->hidden == H && d->n_hidden == NH) return launch<D0P, H, NH>(a, bwd, st);
  CASE(16, 64, 1)   // K-Planes proposal sigma_net 8->64->1; nerfplayer-nerfacto proposal 10->16... (see below)
  CASE(16, 64, 2)   // K-Planes color_net 15->64->64->3
  CASE(32, 128, 1)  // sigma_net, 1 scale
  CASE(64, 128, 1)  // 2 scales
  CASE(96, 128, 1)
  CASE(128, 128, 1)
  CASE(160, 128, 1)  // k-planes preset: 5 scales
  CASE(192, 128, 1)  // 6 scales (config 3)
  CASE(32, 64, 1)    // default sigma hidden 64; nerfplayer mlp_base 32->64->16
  CASE(64, 64, 1)
  CASE(128, 64, 1)
  CASE(160, 64, 1)
  CASE(16, 16, 1)    // nerfplayer-nerfacto proposal 10->16->1
  CASE(32, 64, 2)
  CASE(48, 64, 2)
  CASE(64, 64, 2)    // nerfplayer mlp_head 63->64->64->3
#undef CASE
  set_error("mlp: unsupported shape d_in=%d hidden=%d n_hidden=%d", d->d_in, d->hidden, d->n_hidden);
  return SNERF_ERR_UNSUPPORTED;
}

static int fill(const snerf_mlp_desc* d, MlpArgs& a) {
  SNERF_REQUIRE(d, "mlp: null descriptor");
  SNERF_REQUIRE(d->d_in >= 1 && d->d_in <= 192, "mlp: d_in=%d", d->d_in);
  SNERF_REQUIRE(d->d_out >= 1 && d->d_out <= OUTP, "mlp: d_out=%d (<= %d)", d->d_out, OUTP);
  SNERF_REQUIRE(d->n_hidden == 1 || d->n_hidden == 2, "mlp: n_hidden=%d", d->n_hidden);
  SNERF_REQUIRE(d->hidden_act == 0 || d->hidden_act == 1, "mlp: hidden_act=%d", d->hidden_act);
  SNERF_REQUIRE(d->out_act == 0 || d->out_act == 1, "mlp: out_act=%d", d->out_act);
  a.d0 = d->d_in; a.dout = d->d_out; a.hidden_act = d->hidden_act; a.out_act = d->out_act;
  int off = 0, prev = d->d_in;
  for (int l = 0; l < d->n_hidden; ++l) { a.woff[l] = off; off += prev * d->hidden; prev = d->hidden; }
  a.woff[d->n_hidden] = off;
  return 0;
}

}  // namespace snerf

using namespace snerf;

extern "C" int64_t snerf_mlp_param_count(const snerf_mlp_desc* d) {
  if (!d) return -1;
  int64_t n = 0, prev = d->d_in;
  for (int l = 0; l < d->n_hidden; ++l) { n += prev * d->hidden; prev = d->hidden; }
  return n + prev * d->d_out;
}

extern "C" int snerf_mlp_fwd(const snerf_mlp_desc* d, const float* W, const float* X, int32_t ldx, int64_t N, float* Y, int32_t ldy,
                             int32_t aux_col, float* aux_out, snerf_stream_t stream) {
  MlpArgs a = {};
  int rc = fill(d, a);
  if (rc) return rc;
  SNERF_REQUIRE(N >= 0 && ldx >= d->d_in && ldy >= d->d_out, "mlp_fwd: N=%lld ldx=%d ldy=%d", (long long)N, ldx, ldy);
  if (N == 0) return 0;
  SNERF_REQUIRE(W && X && Y, "mlp_fwd: null buffer");
  SNERF_REQUIRE(!aux_out || (aux_col >= 0 && aux_col < d->d_out), "mlp_fwd: aux_col=%d", aux_col);
  a.X = X; a.N = N; a.ldx = ldx; a.W = W; a.Y = Y; a.ldy = ldy; a.aux_col = aux_col; a.aux_out = aux_out;
  return dispatch(d, a, false, (hipStream_t)stream);
}

extern "C" int snerf_mlp_bwd(const snerf_mlp_desc* d, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                             int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream) {
  MlpArgs a = {};
  int rc = fill(d, a);
  if (rc) return rc;
  SNERF_REQUIRE(N >= 0 && ldx >= d->d_in, "mlp_bwd: N=%lld ldx=%d", (long long)N, ldx);
  if (N == 0) return 0;
  SNERF_REQUIRE(W && X, "mlp_bwd: null buffer");
  SNERF_REQUIRE(gY || gaux, "mlp_bwd: no incoming gradient");
  SNERF_REQUIRE(!gY || ldgy >= d->d_out, "mlp_bwd: ldgy=%d", ldgy);
  SNERF_REQUIRE(!gaux || (aux_col >= 0 && aux_col < d->d_out), "mlp_bwd: aux_col=%d", aux_col);
  SNERF_REQUIRE(!gX || ldgx >= d->d_in, "mlp_bwd: ldgx=%d", ldgx);
  a.X = X; a.N = N; a.ldx = ldx; a.W = W; a.gY = gY; a.ldgy = ldgy; a.aux_col = aux_col; a.gaux = gaux; a.gX = gX; a.ldgx = ldgx; a.gW = gW;
  return dispatch(d, a, true, (hipStream_t)stream);
}

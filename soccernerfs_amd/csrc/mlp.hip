// Tiny bias-free MLPs (the reference's tcnn.Network / FullyFusedMLP call sites) on the matrix cores.
//
// Replaces tcnn.Network(n_input_dims, n_output_dims, {FullyFusedMLP, ReLU, None|Sigmoid, n_neurons, n_hidden_layers})
// at NS/fields/kplanes_field.py:249-273,397-407 and NS/fields/nerfplayer_nerfacto_field.py:94-104,238-248,301-311.
//
// Numerics: exact fp32 -- v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md §3),
// so the fp32 parity bar (rtol 1e-5) against the CPU oracle holds; tcnn itself computes in fp16 (>= reference precision).
//
// Structure: persistent workgroups (4 waves for the 64-wide nets, 8 for the 128-wide ones) walk TS-sample tiles; activations
// live in LDS (row stride = 2 mod 32 floats => conflict-free A-operand reads), every MFMA operand is read in groups of 8 k-steps
// one group ahead of the MFMAs that consume it, nothing but X and dY ever touches HBM.
//   forward, one hidden layer (mlp_fwd_wreg_kernel): each wave owns 16 hidden units, its layer-0 B operand stays in registers for
//     the whole loop; only the 16-wide output weights are in LDS; two workgroups per CU.
//   forward with two hidden layers / backward: all weights staged ONCE per workgroup into LDS (zero-padded, row stride = 2 mod 32
//     so both the plain and the transposed B-operand reads spread over the banks).  The backward recomputes the forward per tile,
//     keeps the weight-gradient accumulators in registers across the whole persistent loop and flushes them once per workgroup
//     with 64-B-contiguous atomics.  TS (64/32/16) is chosen per shape so that weights + tiles fit and, for 4-wave workgroups,
//     two of them share a CU.
#include "common.hpp"

namespace snerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int OUTP = 16;  // padded output width (all nets here have <= 16 outputs)
constexpr int LDS_LIMIT = 160 * 1024;

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
  long long* gWfx;  // deterministic mode: weight gradients accumulate here as fixed point instead (common.hpp)
  int x16;          // X holds the 16-bit operand type (what snerf_kplanes_field_fwd wrote), not fp32: 16-bit kernels only
  // quotient epilogue of the backward (snerf_mlp_bwd_x16_quotient; 16-bit kernels, one hidden layer of 128): instead of gX the kernel writes
  // G = gX .* X (X = the 16-bit tile it holds in LDS) and lists the elements whose X vanished while gX did not (common.hpp: fix_append)
  float* G; int ldg;
  int32_t* fix_list; int fix_capacity;
  int32_t* fix_count; int32_t* fix_count_next;
  int variant;      // backward only: 0 = the default kernel for the shape, 1 = the workgroup-tile kernels of mlp_lp.hip (snerf_mlp_bwd_tile)
  float* ws; int ws_rep; int64_t ws_stride;  // weight-gradient workspace (mlp_lp_common.hpp; honoured by the 16-bit kernels only)
  // dense layers wider than one 128 x 128 block (snerf_dense_fwd / _bwd tile them): row stride of W in global memory (0 = dout), and
  // "add to what is there" for the forward's output (later K blocks of a linear layer) / the backward's input gradient (later column blocks)
  int ldw_g, acc_y, acc_gx;
};

__device__ __forceinline__ void gw_add(const MlpArgs& a, int64_t idx, float v) {
  if (a.gWfx) fx_atomic_add(a.gWfx + idx, v); else atomicAdd(a.gW + idx, v);
}

__host__ __device__ constexpr int ld_of(int width) { return ((width + 31) / 32) * 32 + 2; }  // = 2 mod 32
__host__ __device__ constexpr int ldw_of(int n) { return n == OUTP ? 18 : ld_of(n); }

// LDS plan (floats)
template <int D0P, int H, int NH, int TS, bool BWD>
struct Plan {
  static constexpr int LD0 = ld_of(D0P), LDH = ld_of(H), LDO = ld_of(OUTP);
  static constexpr int LW0 = ldw_of(H), LWO = ldw_of(OUTP);
  static constexpr int W0 = 0;                             // [D0P][LW0]
  static constexpr int W1 = W0 + D0P * LW0;                // [H][LW0] (NH == 2)
  static constexpr int WO = W1 + (NH == 2 ? H * LW0 : 0);  // [H][LWO]
  static constexpr int ACT0 = WO + H * LWO;                // X tile [TS][LD0]
  static constexpr int ACT1 = ACT0 + TS * LD0;             // A1 [TS][LDH]
  static constexpr int ACT2 = ACT1 + TS * LDH;             // A2 [TS][LDH] (NH == 2)
  static constexpr int GZ = ACT2 + (NH == 2 ? TS * LDH : 0);  // bwd: [TS][LDH]
  static constexpr int GZO = GZ + (BWD ? TS * LDH : 0);       // bwd: [TS][LDO]
  static constexpr int TOTAL = GZO + (BWD ? TS * LDO : 0);
  static constexpr size_t BYTES = (size_t)TOTAL * sizeof(float);
};

// waves per workgroup: the 128-wide nets leave room for only ONE workgroup per CU (weights + tiles ~150 KB of LDS), so they run
// 8 waves (2 per SIMD) to overlap one wave's LDS / barrier waits with the other's MFMAs (measured: 55 % of wave cycles were waits
// at 1 wave per SIMD, MFMA pipe 29 % busy -- profiles/r01_kernels.md); the 64-wide nets fit 2+ workgroups per CU with 4 waves.
template <int H>
constexpr int waves_of() { return H >= 128 ? 8 : 4; }

template <int D0P, int H, int NH, bool BWD>
constexpr int pick_ts() {
  // 4-wave workgroups (H < 128): a tile size that lets at least two workgroups share a CU -- one wave per SIMD cannot hide its own
  // LDS / barrier waits (color_net backward: 90 KB at TS = 64 -> one workgroup per CU; 58 KB at TS = 32 -> two)
  if (H < 128 && Plan<D0P, H, NH, 64, BWD>::BYTES > LDS_LIMIT / 2 && Plan<D0P, H, NH, 32, BWD>::BYTES <= LDS_LIMIT / 2) return 32;
  return Plan<D0P, H, NH, 64, BWD>::BYTES <= LDS_LIMIT ? 64 : (Plan<D0P, H, NH, 32, BWD>::BYTES <= LDS_LIMIT ? 32 : 16);
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// The MFMA helpers below read their operands in GROUPS of G k-steps, the next group's LDS reads issued before the current
// group's MFMAs (register double buffer).  Written the obvious way -- read a, read b, mfma -- the compiler emits
// ds_read / s_waitcnt lgkmcnt(0) / v_mfma per k-step: every MFMA waits a full LDS round trip and the matrix pipe sat 23 % busy
// (rocprofv3: SQ_VALU_MFMA_BUSY_CYCLES vs kernel time, profiles/r01_kernels.md).
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
#pragma unroll
  for (int i = 0; i < N; ++i) f(i);
}

// acc[m] (m = row blocks) += A[TS x K] (LDS, stride lda) * W[K x ..] (LDS, stride ldw), column block nt
template <int MT, int K>
__device__ __forceinline__ void mma_cols(const float* As, int lda, const float* Ws, int ldw, int nt, f32x4 (&acc)[MT], int lane) {
  constexpr int KS = K / 4, G = KS % 8 == 0 ? 8 : 4, NG = KS / G;
  static_assert(KS % G == 0, "K must be a multiple of 16");
  const int lr = lane & 15, lk = lane >> 4;
  const float* wp = Ws + lk * ldw + nt * 16 + lr;
  const float* ap = As + lr * lda + lk;
  float bq[2][G], aq[2][G][MT];
  auto load = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int k0 = (g * G + i) * 4;
      bq[buf][i] = wp[k0 * ldw];
#pragma unroll
      for (int m = 0; m < MT; ++m) aq[buf][i][m] = ap[m * 16 * lda + k0];
    }
  };
  load(0, 0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) load(g + 1, (g + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the next group's reads ABOVE this group's MFMAs (the scheduler sinks them otherwise)
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma4(aq[g & 1][i][m], bq[g & 1][i], acc[m]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one 16x16 block: rows mt*16.., cols nt*16..
template <int K>
__device__ __forceinline__ void mma_one(const float* As, int lda, const float* Ws, int ldw, int mt, int nt, f32x4& acc, int lane) {
  f32x4 a1[1] = {acc};
  mma_cols<1, K>(As + mt * 16 * lda, lda, Ws, ldw, nt, a1, lane);
  acc = a1[0];
}

// acc[m] += G[TS x K] (LDS) * W^T, i.e. B[k][n] = W[n][k]; output column block nt indexes W ROWS (dX = dZ * W^T)
template <int MT, int K>
__device__ __forceinline__ void mma_cols_T(const float* Gs, int ldg, const float* Ws, int ldw, int nt, f32x4 (&acc)[MT], int lane) {
  constexpr int KS = K / 4, G = KS % 8 == 0 ? 8 : 4, NG = KS / G;
  static_assert(KS % G == 0, "K must be a multiple of 16");
  const int lr = lane & 15, lk = lane >> 4;
  const float* wp = Ws + (nt * 16 + lr) * ldw + lk;
  const float* gp = Gs + lr * ldg + lk;
  float bq[2][G], aq[2][G][MT];
  auto load = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int k0 = (g * G + i) * 4;
      bq[buf][i] = wp[k0];
#pragma unroll
      for (int m = 0; m < MT; ++m) aq[buf][i][m] = gp[m * 16 * ldg + k0];
    }
  };
  load(0, 0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) load(g + 1, (g + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the next group's reads ABOVE this group's MFMAs (the scheduler sinks them otherwise)
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma4(aq[g & 1][i][m], bq[g & 1][i], acc[m]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc (16x16 block it,nt of dW) += A^T[K-block it][TS samples] * G[TS][N-block nt]
template <int TS>
__device__ __forceinline__ void mma_outer(const float* As, int lda, const float* Gs, int ldg, int it, int nt, f32x4& acc, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const float* ap = As + lk * lda + it * 16 + lr;
  const float* gp = Gs + lk * ldg + nt * 16 + lr;
  float aq[TS / 4], gq[TS / 4];
#pragma unroll
  for (int i = 0; i < TS / 4; ++i) { aq[i] = ap[i * 4 * lda]; gq[i] = gp[i * 4 * ldg]; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < TS / 4; ++i) acc = mfma4(aq[i], gq[i], acc);
}

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

// stage one weight matrix [rows_act x cols_act] (global, row-major) into LDS [rows_pad][ldw], zero padded
__device__ __forceinline__ void stage_weights(const float* __restrict__ Wg, int rows_act, int cols_act, float* Ws, int rows_pad, int cols_pad, int ldw,
                                              int gld = 0) {
  const int64_t g = gld > 0 ? gld : cols_act;  // row stride in global memory (a column block of a wider matrix: gld = its full width)
  for (int idx = threadIdx.x; idx < rows_pad * cols_pad; idx += blockDim.x) {
    const int r = idx / cols_pad, c = idx - r * cols_pad;
    Ws[r * ldw + c] = (r < rows_act && c < cols_act) ? Wg[(int64_t)r * g + c] : 0.f;
  }
}

// X tile (rows n0.., d0 columns, zero padded to K0 columns / TS rows): global -> registers (issued one tile AHEAD so the
// loads fly under the current tile's MFMAs), registers -> LDS after the tile's last reader has passed the barrier.
template <int TS, int K0, int NT>
struct XTile {
  static constexpr int PER = (TS * K0 + NT - 1) / NT;
  float v[PER];
  __device__ __forceinline__ void fetch(const MlpArgs& a, int64_t n0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / K0, c = idx - r * K0;
      const int64_t n = n0 + r;
      v[i] = (idx < TS * K0 && n < a.N && c < a.d0) ? a.X[n * a.ldx + c] : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* Xs, int ldxs) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / K0, c = idx - r * K0;
      if (idx < TS * K0) Xs[r * ldxs + c] = v[i];
    }
  }
};

// hidden layers of one tile; leaves A_l (post-activation) in act[l]
template <int D0P, int H, int NH, int TS, bool BWD>
__device__ __forceinline__ void forward_hidden(float* smem, bool relu, int wave, int lane) {
  using P = Plan<D0P, H, NH, TS, BWD>;
  constexpr int HT = H / 16, MT = TS / 16, NW = waves_of<H>();
#pragma unroll
  for (int l = 0; l < NH; ++l) {
    const float* in = smem + (l == 0 ? P::ACT0 : P::ACT1);
    float* out = smem + (l == 0 ? P::ACT1 : P::ACT2);
    const int lda = l == 0 ? P::LD0 : P::LDH;
    const int K = l == 0 ? D0P : H;
    const float* Wl = smem + (l == 0 ? P::W0 : P::W1);
    if (HT >= 4) {
#pragma unroll
      for (int j = 0; j < (HT >= NW ? HT / NW : 1); ++j) {
        const int nt = wave + NW * j;
        f32x4 acc[MT] = {};
        if (l == 0) mma_cols<MT, D0P>(in, lda, Wl, P::LW0, nt, acc, lane); else mma_cols<MT, H>(in, lda, Wl, P::LW0, nt, acc, lane);
#pragma unroll
        for (int m = 0; m < MT; ++m) store_block(out, P::LDH, m, nt, acc[m], relu, lane);
      }
    } else {  // H == 16: one column block, waves split the row blocks
      if (wave < MT) {
        f32x4 acc = {};
        if (l == 0) mma_one<D0P>(in, lda, Wl, P::LW0, wave, 0, acc, lane); else mma_one<H>(in, lda, Wl, P::LW0, wave, 0, acc, lane);
        store_block(out, P::LDH, wave, 0, acc, relu, lane);
      }
    }
    __syncthreads();
  }
}

template <int D0P, int H, int NH, int TS, bool BWD>
__device__ __forceinline__ void stage_all(const MlpArgs& a, float* smem) {
  using P = Plan<D0P, H, NH, TS, BWD>;
  stage_weights(a.W + a.woff[0], a.d0, H, smem + P::W0, D0P, H, P::LW0);
  if (NH == 2) stage_weights(a.W + a.woff[1], H, H, smem + P::W1, H, H, P::LW0);
  stage_weights(a.W + a.woff[NH], H, a.dout, smem + P::WO, H, OUTP, P::LWO);
}

template <int D0P, int H, int NH, int TS>
__global__ __launch_bounds__(waves_of<H>() * 64) void mlp_fwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) float smem[];
  using P = Plan<D0P, H, NH, TS, false>;
  constexpr int MT = TS / 16, NW = waves_of<H>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stage_all<D0P, H, NH, TS, false>(a, smem);
  const float* act_last = smem + (NH == 2 ? P::ACT2 : P::ACT1);
  XTile<TS, D0P, NW * 64> xt;
  xt.fetch(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();  // weights staged / previous tile's readers done
    xt.store(smem + P::ACT0, P::LD0);
    if (tile + gridDim.x < n_tiles) xt.fetch(a, (tile + gridDim.x) * TS);  // prefetch the next tile
    __syncthreads();
    forward_hidden<D0P, H, NH, TS, false>(smem, a.hidden_act == 1, wave, lane);
    // output layer: row block mt handled by wave mt % NW
#pragma unroll
    for (int j = 0; j < (MT + NW - 1) / NW; ++j) {
      const int mt = wave + NW * j;
      if (mt < MT) {
        f32x4 acc = {};
        mma_one<H>(act_last, P::LDH, smem + P::WO, P::LWO, mt, 0, acc, lane);
        const int col = lane & 15;
        const int64_t row0 = n0 + mt * 16 + (lane >> 4) * 4;
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
  }
}

// Forward, one hidden layer, H / 16 == number of waves: wave w always computes hidden units 16w .. 16w+15, so ITS B operand of
// layer 0 (D0P / 4 k-steps x one float per lane) is loop-invariant -- it lives in registers for the whole persistent loop and W0
// never enters LDS.  That removes a third of the LDS reads and 83 KB of the 130 KB the sigma net needed, so two workgroups
// (16 waves) share a CU instead of one and barrier stalls of one overlap MFMAs of the other.
template <int D0P, int H, int TS>
struct PlanWreg {
  static constexpr int LD0 = ld_of(D0P), LDH = ld_of(H), LWO = ldw_of(OUTP);
  static constexpr int WO = 0;                   // [H][LWO]
  static constexpr int ACT0 = WO + H * LWO;      // X tile [TS][LD0]
  static constexpr int ACT1 = ACT0 + TS * LD0;   // A1 [TS][LDH]
  static constexpr int TOTAL = ACT1 + TS * LDH;
  static constexpr size_t BYTES = (size_t)TOTAL * sizeof(float);
};

template <int MT, int K>
__device__ __forceinline__ void mma_cols_breg(const float* As, int lda, const float (&breg)[K / 4], f32x4 (&acc)[MT], int lane) {
  constexpr int KS = K / 4, G = KS % 8 == 0 ? 8 : 4, NG = KS / G;
  const int lr = lane & 15, lk = lane >> 4;
  const float* ap = As + lr * lda + lk;
  float aq[2][G][MT];
  auto load = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int m = 0; m < MT; ++m) aq[buf][i][m] = ap[m * 16 * lda + (g * G + i) * 4];
  };
  load(0, 0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) load(g + 1, (g + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma4(aq[g & 1][i][m], breg[g * G + i], acc[m]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int D0P, int H, int TS>
__global__ __launch_bounds__(H / 16 * 64) void mlp_fwd_wreg_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) float smem[];
  using P = PlanWreg<D0P, H, TS>;
  constexpr int MT = TS / 16, NW = H / 16, KS = D0P / 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  float breg[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int k = 4 * i + lk;
    breg[i] = k < a.d0 ? a.W[a.woff[0] + (int64_t)k * H + wave * 16 + lr] : 0.f;
  }
  stage_weights(a.W + a.woff[1], H, a.dout, smem + P::WO, H, OUTP, P::LWO);
  const bool relu = a.hidden_act == 1;
  XTile<TS, D0P, NW * 64> xt;
  xt.fetch(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();  // WO staged / previous tile's readers done
    xt.store(smem + P::ACT0, P::LD0);
    if (tile + gridDim.x < n_tiles) xt.fetch(a, (tile + gridDim.x) * TS);  // prefetch the next tile
    __syncthreads();
    {
      f32x4 acc[MT] = {};
      mma_cols_breg<MT, D0P>(smem + P::ACT0, P::LD0, breg, acc, lane);
#pragma unroll
      for (int m = 0; m < MT; ++m) store_block(smem + P::ACT1, P::LDH, m, wave, acc[m], relu, lane);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (MT + NW - 1) / NW; ++j) {
      const int mt = wave + NW * j;
      if (mt < MT) {
        f32x4 acc = {};
        mma_one<H>(smem + P::ACT1, P::LDH, smem + P::WO, P::LWO, mt, 0, acc, lane);
        const int col = lane & 15;
        const int64_t row0 = n0 + mt * 16 + (lane >> 4) * 4;
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
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int D0P, int H, int NH, int TS>
__global__ __launch_bounds__(waves_of<H>() * 64) void mlp_bwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) float smem[];
  using P = Plan<D0P, H, NH, TS, true>;
  constexpr int D0T = D0P / 16, HT = H / 16, MT = TS / 16, NW = waves_of<H>();
  constexpr int NJ = HT >= NW ? HT / NW : 1;  // column blocks of a hidden layer per wave
  static_assert(D0P % 16 == 0 && H % 16 == 0, "dims are padded to 16");
  static_assert(NH == 1 || HT >= 4, "two hidden layers need hidden >= 64");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool relu = a.hidden_act == 1;
  float* X = smem + P::ACT0;
  float* A1 = smem + P::ACT1;
  float* Alast = smem + (NH == 2 ? P::ACT2 : P::ACT1);
  float* gz = smem + P::GZ;
  float* gzo = smem + P::GZO;
  stage_all<D0P, H, NH, TS, true>(a, smem);

  // weight-gradient accumulators, alive across the persistent loop; block t of a layer is owned by wave t % NW
  constexpr int NB0 = (D0T * HT + NW - 1) / NW;   // layer 0: [D0P x H]
  constexpr int NBH = (HT * HT + NW - 1) / NW;    // hidden->hidden (NH == 2)
  constexpr int NBO = (HT + NW - 1) / NW;         // output: [H x 16]
  f32x4 dW0[NB0] = {};
  f32x4 dWh[NH == 2 ? NBH : 1] = {};
  f32x4 dWo[NBO] = {};

  XTile<TS, D0P, NW * 64> xt;
  xt.fetch(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    xt.store(X, P::LD0);
    if (tile + gridDim.x < n_tiles) xt.fetch(a, (tile + gridDim.x) * TS);  // prefetch the next tile
    __syncthreads();
    forward_hidden<D0P, H, NH, TS, true>(smem, relu, wave, lane);
    // ---- output layer forward (needed for sigmoid' / trunc_exp') and grad wrt its pre-activation ----
#pragma unroll
    for (int j = 0; j < (MT + NW - 1) / NW; ++j) {
      const int mt = wave + NW * j;
      if (mt < MT) {
        f32x4 acc = {};
        mma_one<H>(Alast, P::LDH, smem + P::WO, P::LWO, mt, 0, acc, lane);
        const int col = lane & 15;
        const int rl0 = mt * 16 + (lane >> 4) * 4;
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
          gzo[(rl0 + r) * P::LDO + col] = g;
        }
      }
    }
    __syncthreads();
    // ---- dW_out += A_last^T * gzo ----
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + NW * j;
      if (it < HT) mma_outer<TS>(Alast, P::LDH, gzo, P::LDO, it, 0, dWo[j], lane);
    }
    // ---- gz = (gzo * W_out^T) .* relu'(A_last): output width H, K = 16 ----
    if (HT >= 4) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nt = wave + NW * j;
        f32x4 acc[MT] = {};
        mma_cols_T<MT, OUTP>(gzo, P::LDO, smem + P::WO, P::LWO, nt, acc, lane);
        const int col = nt * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[m][r];
            if (relu && !(Alast[(row0 + r) * P::LDH + col] > 0.f)) v = 0.f;
            gz[(row0 + r) * P::LDH + col] = v;
          }
        }
      }
    } else if (wave == 0) {
      f32x4 acc[MT] = {};
      mma_cols_T<MT, OUTP>(gzo, P::LDO, smem + P::WO, P::LWO, 0, acc, lane);
      const int col = lane & 15;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[m][r];
          if (relu && !(Alast[(row0 + r) * P::LDH + col] > 0.f)) v = 0.f;
          gz[(row0 + r) * P::LDH + col] = v;
        }
      }
    }
    __syncthreads();
    // ---- hidden->hidden layer (NH == 2): dW_h += A_1^T * gz ; gz' = (gz * W_h^T) .* relu'(A_1) ----
    if (NH == 2) {
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + NW * j;
        if (t < HT * HT) mma_outer<TS>(A1, P::LDH, gz, P::LDH, t / HT, t % HT, dWh[j], lane);
      }
      f32x4 acc2[NJ][MT] = {};
#pragma unroll
      for (int j = 0; j < NJ; ++j) mma_cols_T<MT, H>(gz, P::LDH, smem + P::W1, P::LW0, wave + NW * j, acc2[j], lane);
      __syncthreads();  // everyone finished reading gz
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nt = wave + NW * j;
        const int col = nt * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row0 = m * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc2[j][m][r];
            if (relu && !(A1[(row0 + r) * P::LDH + col] > 0.f)) v = 0.f;
            gz[(row0 + r) * P::LDH + col] = v;
          }
        }
      }
      __syncthreads();
    }
    // ---- first layer: dW_0 += X^T * gz ; gX = gz * W_0^T ----
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + NW * j;
      if (t < D0T * HT) mma_outer<TS>(X, P::LD0, gz, P::LDH, t / HT, t % HT, dW0[j], lane);
    }
    if (a.gX) {
#pragma unroll
      for (int j = 0; j < (D0T + NW - 1) / NW; ++j) {
        const int nt = wave + NW * j;
        if (nt < D0T) {
          f32x4 acc[MT] = {};
          mma_cols_T<MT, H>(gz, P::LDH, smem + P::W0, P::LW0, nt, acc, lane);
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
  if (a.gW || a.gWfx) {
    const int cl = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
      const int t = wave + NW * j;
      if (t < D0T * HT) {
        const int it = t / HT, nt = t % HT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = it * 16 + r0 + r, col = nt * 16 + cl;
          if (row < a.d0) gw_add(a, a.woff[0] + (int64_t)row * H + col, dW0[j][r]);
        }
      }
    }
    if (NH == 2) {
#pragma unroll
      for (int j = 0; j < NBH; ++j) {
        const int t = wave + NW * j;
        if (t < HT * HT) {
          const int it = t / HT, nt = t % HT;
#pragma unroll
          for (int r = 0; r < 4; ++r) gw_add(a, a.woff[1] + (int64_t)(it * 16 + r0 + r) * H + nt * 16 + cl, dWh[j][r]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int it = wave + NW * j;
      if (it < HT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cl < a.dout) gw_add(a, a.woff[NH] + (int64_t)(it * 16 + r0 + r) * a.dout + cl, dWo[j][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// single dense layers: Y = act(X W), and its backward.  Nets outside the fused kernels' shape table (the full NeRFPlayer's
// three-hidden-layer / 32-output nets, NS/fields/nerfplayer_field.py:231-316) are chained from these: the weight matrix lives in
// LDS for the whole persistent loop, a TS-row tile of X (and of dZ in the backward) is staged per iteration, activations travel
// through HBM between layers (N x 128 floats = 100 MB at 196 k samples: ~0.03 ms each way, against ~0.1 ms of MFMA work).
// ---------------------------------------------------------------------------------------------
template <int KP, int MP, int TS, bool BWD>
struct DensePlan {
  static constexpr int LDX = ld_of(KP), LDM = ld_of(MP), LW = ldw_of(MP);
  static constexpr int W0 = 0;                       // [KP][LW]
  static constexpr int XT = W0 + KP * LW;            // [TS][LDX]
  static constexpr int GZ = XT + TS * LDX;           // bwd: [TS][LDM]
  static constexpr int TOTAL = GZ + (BWD ? TS * LDM : 0);
  static constexpr size_t BYTES = (size_t)TOTAL * sizeof(float);
};
template <int MP>
constexpr int dense_waves() { return MP >= 128 ? 8 : 4; }

// MlpArgs reuse: d0 = K, dout = M, W = the [K][M] matrix, hidden_act = 1 -> ReLU, out_act = 1 -> Sigmoid (at most one set)
template <int KP, int MP, int TS>
__global__ __launch_bounds__(dense_waves<MP>() * 64) void dense_fwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) float smem[];
  using P = DensePlan<KP, MP, TS, false>;
  constexpr int MT = TS / 16, NW = dense_waves<MP>(), NTB = MP / 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  stage_weights(a.W, a.d0, a.dout, smem + P::W0, KP, MP, P::LW, a.ldw_g);
  XTile<TS, KP, NW * 64> xt;
  xt.fetch(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    xt.store(smem + P::XT, P::LDX);
    if (tile + gridDim.x < n_tiles) xt.fetch(a, (tile + gridDim.x) * TS);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (NTB + NW - 1) / NW; ++j) {
      const int nt = wave + NW * j;
      if (nt < NTB) {
        f32x4 acc[MT] = {};
        mma_cols<MT, KP>(smem + P::XT, P::LDX, smem + P::W0, P::LW, nt, acc, lane);
        const int col = nt * 16 + (lane & 15);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int64_t row0 = n0 + m * 16 + (lane >> 4) * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (row0 + r < a.N && col < a.dout) {
              float y = acc[m][r];
              if (a.acc_y) y += a.Y[(row0 + r) * a.ldy + col];  // a later K block (the launcher sets the activation on the last one only)
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

// a.Y here is the layer's stored OUTPUT (post-activation), read-only: dZ = dY .* act'(Y)
template <int KP, int MP, int TS>
__global__ __launch_bounds__(dense_waves<MP>() * 64) void dense_bwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) float smem[];
  using P = DensePlan<KP, MP, TS, true>;
  constexpr int MT = TS / 16, NW = dense_waves<MP>(), KT = KP / 16, NTB = MP / 16, NT = NW * 64;
  constexpr int NB = (KT * NTB + NW - 1) / NW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* Xs = smem + P::XT;
  float* Gs = smem + P::GZ;
  stage_weights(a.W, a.d0, a.dout, smem + P::W0, KP, MP, P::LW, a.ldw_g);
  f32x4 dW[NB] = {};
  XTile<TS, KP, NT> xt;
  xt.fetch(a, (int64_t)blockIdx.x * TS);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    __syncthreads();
    xt.store(Xs, P::LDX);
    for (int idx = threadIdx.x; idx < TS * MP; idx += NT) {
      const int r = idx / MP, c = idx - r * MP;
      const int64_t n = n0 + r;
      float g = 0.f;
      if (n < a.N && c < a.dout) {
        g = a.gY[n * a.ldgy + c];
        const float y = a.Y[n * a.ldy + c];
        if (a.hidden_act == 1) g = y > 0.f ? g : 0.f;
        if (a.out_act == 1) g = g * y * (1.f - y);
      }
      Gs[r * P::LDM + c] = g;
    }
    if (tile + gridDim.x < n_tiles) xt.fetch(a, (tile + gridDim.x) * TS);
    __syncthreads();
    if (a.gW || a.gWfx) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int t = wave + NW * j;
        if (t < KT * NTB) mma_outer<TS>(Xs, P::LDX, Gs, P::LDM, t / NTB, t % NTB, dW[j], lane);
      }
    }
    if (a.gX) {
#pragma unroll
      for (int j = 0; j < (KT + NW - 1) / NW; ++j) {
        const int nt = wave + NW * j;
        if (nt < KT) {
          f32x4 acc[MT] = {};
          mma_cols_T<MT, MP>(Gs, P::LDM, smem + P::W0, P::LW, nt, acc, lane);
          const int col = nt * 16 + (lane & 15);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int64_t row0 = n0 + m * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (row0 + r < a.N && col < a.d0) {
                float* gx = a.gX + (row0 + r) * a.ldgx + col;
                *gx = a.acc_gx ? *gx + acc[m][r] : acc[m][r];  // a later column block adds its share of dZ W^T
              }
          }
        }
      }
    }
  }
  if (a.gW || a.gWfx) {
    const int cl = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = wave + NW * j;
      if (t < KT * NTB) {
        const int it = t / NTB, nt = t % NTB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = it * 16 + r0 + r, col = nt * 16 + cl;
          if (row < a.d0 && col < a.dout) gw_add(a, (int64_t)row * (a.ldw_g > 0 ? a.ldw_g : a.dout) + col, dW[j][r]);
        }
      }
    }
  }
}

template <int KP, int MP>
static int launch_dense(const MlpArgs& a, bool bwd, hipStream_t st) {
  constexpr int TS = 64;
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  if (bwd) {
    using P = DensePlan<KP, MP, TS, true>;
    static_assert(P::BYTES <= LDS_LIMIT, "dense backward tile does not fit LDS");
    int per_cu = (int)(LDS_LIMIT / P::BYTES);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    int64_t grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    auto k = dense_bwd_kernel<KP, MP, TS>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(dense_waves<MP>() * 64), P::BYTES, st, a, n_tiles);
  } else {
    using P = DensePlan<KP, MP, TS, false>;
    int per_cu = (int)(LDS_LIMIT / P::BYTES);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    int64_t grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    auto k = dense_fwd_kernel<KP, MP, TS>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(dense_waves<MP>() * 64), P::BYTES, st, a, n_tiles);
  }
  SNERF_LAUNCH_CHECK(bwd ? "dense_bwd" : "dense_fwd");
  return 0;
}

static int dispatch_dense(int K, int M, const MlpArgs& a, bool bwd, hipStream_t st) {
  const int kp = K <= 16 ? 16 : (K <= 32 ? 32 : (K <= 48 ? 48 : (K <= 64 ? 64 : 128)));
  const int mp = M <= 16 ? 16 : (M <= 32 ? 32 : (M <= 64 ? 64 : 128));
#define CASE(KP, MP) \
  if (kp == KP && mp == MP) return launch_dense<KP, MP>(a, bwd, st);
#define ROW(KP) CASE(KP, 16) CASE(KP, 32) CASE(KP, 64) CASE(KP, 128)
  ROW(16) ROW(32) ROW(48) ROW(64) ROW(128)
#undef ROW
#undef CASE
  set_error("dense: unsupported shape K=%d M=%d", K, M);
  return SNERF_ERR_UNSUPPORTED;
}

template <int D0P, int H, int NH>
static int launch(const MlpArgs& a, bool bwd, hipStream_t st) {
  if (bwd) {
    constexpr int TS = pick_ts<D0P, H, NH, true>();
    using P = Plan<D0P, H, NH, TS, true>;
    static_assert(P::BYTES <= LDS_LIMIT, "backward tile does not fit LDS");
    const int64_t n_tiles = (a.N + TS - 1) / TS;
    int per_cu = (int)(LDS_LIMIT / P::BYTES);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    auto k = mlp_bwd_kernel<D0P, H, NH, TS>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves_of<H>() * 64), P::BYTES, st, a, n_tiles);
  } else if (NH == 1 && H / 16 == waves_of<H>()) {
    constexpr int TS = H >= 128 ? 16 : 64;
    using P = PlanWreg<D0P, H, TS>;
    const int64_t n_tiles = (a.N + TS - 1) / TS;
    int per_cu = (int)(LDS_LIMIT / P::BYTES);
    const int by_regs = H >= 128 ? 2 : 4;  // 8-wave workgroups holding D0P/4 weight registers per lane: two per CU
    per_cu = per_cu < 1 ? 1 : (per_cu > by_regs ? by_regs : per_cu);
    int64_t grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    auto k = mlp_fwd_wreg_kernel<D0P, H, TS>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(H / 16 * 64), P::BYTES, st, a, n_tiles);
  } else {
    constexpr int TS = pick_ts<D0P, H, NH, false>();
    using P = Plan<D0P, H, NH, TS, false>;
    static_assert(P::BYTES <= LDS_LIMIT, "forward tile does not fit LDS");
    const int64_t n_tiles = (a.N + TS - 1) / TS;
    int per_cu = (int)(LDS_LIMIT / P::BYTES);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = 256 * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    auto k = mlp_fwd_kernel<D0P, H, NH, TS>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(waves_of<H>() * 64), P::BYTES, st, a, n_tiles);
  }
  SNERF_LAUNCH_CHECK(bwd ? "mlp_bwd" : "mlp_fwd");
  return 0;
}

// (padded input width, hidden width, hidden layers) the fused kernels are instantiated for
#define SNERF_MLP_SHAPES(X)                                                   \
  X(16, 64, 1)   /* K-Planes proposal sigma_net 8->64->1 */                   \
  X(16, 64, 2)   /* K-Planes color_net 15->64->64->3 */                       \
  X(32, 128, 1)  /* sigma_net, 1 scale */                                     \
  X(64, 128, 1)  /* 2 scales */                                               \
  X(96, 128, 1)                                                               \
  X(128, 128, 1)                                                              \
  X(160, 128, 1) /* k-planes preset: 5 scales */                              \
  X(192, 128, 1) /* 6 scales (config 3) */                                    \
  X(32, 64, 1)   /* default sigma hidden 64; nerfplayer mlp_base 32->64->16 */ \
  X(64, 64, 1)                                                                \
  X(128, 64, 1)                                                               \
  X(160, 64, 1)                                                               \
  X(16, 16, 1)   /* nerfplayer-nerfacto proposal 10->16->1 */                 \
  X(32, 64, 2)                                                                \
  X(48, 64, 2)                                                                \
  X(64, 64, 2)   /* nerfplayer mlp_head 63->64->64->3 */

// bf16-operand kernels (mlp_lp.hip)
bool mlp_bf16_supported(const snerf_mlp_desc* d);
int mlp_bf16_dispatch(const snerf_mlp_desc* d, const void* args, bool bwd, hipStream_t st);

static int dispatch(const snerf_mlp_desc* d, const MlpArgs& a, bool bwd, hipStream_t st) {
  if (d->operands != 0) return mlp_bf16_dispatch(d, &a, bwd, st);
  const int d0p = (d->d_in + 15) / 16 * 16;
#define CASE(D0P, H, NH) \
  if (d0p == D0P && d->hidden == H && d->n_hidden == NH) return launch<D0P, H, NH>(a, bwd, st);
  SNERF_MLP_SHAPES(CASE)
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
  SNERF_REQUIRE(d->operands >= 0 && d->operands <= 2, "mlp: operands=%d (0 fp32, 1 bf16, 2 fp16)", d->operands);
  a.d0 = d->d_in; a.dout = d->d_out; a.hidden_act = d->hidden_act; a.out_act = d->out_act;
  int off = 0, prev = d->d_in;
  for (int l = 0; l < d->n_hidden; ++l) { a.woff[l] = off; off += prev * d->hidden; prev = d->hidden; }
  a.woff[d->n_hidden] = off;
  return 0;
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_mlp_supported(const snerf_mlp_desc* d) {
  if (!d || d->d_in < 1 || d->d_in > 192 || d->d_out < 1 || d->d_out > OUTP || d->n_hidden < 1 || d->n_hidden > 2) return 0;
  if (d->operands == 1 || d->operands == 2) return mlp_bf16_supported(d) ? 1 : 0;
  if (d->operands != 0) return 0;
  const int d0p = (d->d_in + 15) / 16 * 16;
#define CASE(D0P, H, NH) \
  if (d0p == D0P && d->hidden == H && d->n_hidden == NH) return 1;
  SNERF_MLP_SHAPES(CASE)
#undef CASE
  return 0;
}

// weight-gradient workspace of snerf_mlp_bwd_ws: GW_REPLICAS_H replicas of the flat gradient, each padded to 64 floats
constexpr int GW_REPLICAS_H = 16;  // = mlp_lp_common.hpp's GW_REPLICAS
static int64_t gw_ws_stride(const snerf_mlp_desc* d) {
  int64_t n = 0, prev = d->d_in;
  for (int l = 0; l < d->n_hidden; ++l) { n += prev * d->hidden; prev = d->hidden; }
  n += prev * d->d_out;
  return (n + 63) / 64 * 64;
}

__global__ __launch_bounds__(256) void gw_reduce_kernel(float* __restrict__ ws, int64_t stride, int reps, float* __restrict__ gW, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < reps; ++k) {  // fixed order; the replicas themselves were filled by float atomics
    s += ws[k * stride + i];
    ws[k * stride + i] = 0.f;
  }
  if (s != 0.f) atomicAdd(gW + i, s);
}

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

static int mlp_bwd_impl(const snerf_mlp_desc* d, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                        int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, long long* gWfx, snerf_stream_t stream,
                        int x16 = 0, int variant = 0, float* ws = nullptr) {
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
  a.gWfx = gWfx;
  a.x16 = x16;
  a.variant = variant;
  if (ws && d->operands != 0) { a.ws = ws; a.ws_rep = GW_REPLICAS_H; a.ws_stride = gw_ws_stride(d); }
  else if (ws) { a.gW = ws; }  // exact-fp32 kernels: no replica routing -- everything lands in replica 0, which the reduce folds in like the others
  SNERF_REQUIRE(!x16 || d->operands == 1 || d->operands == 2, "mlp_bwd_x16: a 16-bit input needs 16-bit operands (desc.operands = 1 / 2), got %d", d->operands);
  return dispatch(d, a, true, (hipStream_t)stream);
}

extern "C" int snerf_mlp_bwd(const snerf_mlp_desc* d, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                             int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream) {
  return mlp_bwd_impl(d, W, X, ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx, gW, nullptr, stream);
}

extern "C" int64_t snerf_mlp_gw_workspace_floats(const snerf_mlp_desc* d) {
  if (!d || d->d_in < 1 || d->n_hidden < 1 || d->n_hidden > 2) return -1;
  return GW_REPLICAS_H * gw_ws_stride(d);
}

extern "C" int snerf_mlp_bwd_ws(const snerf_mlp_desc* d, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* workspace, snerf_stream_t stream) {
  SNERF_REQUIRE(workspace, "mlp_bwd_ws: null workspace");
  return mlp_bwd_impl(d, W, X, ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx, nullptr, nullptr, stream, 0, 0, workspace);
}

extern "C" int snerf_mlp_gw_reduce(const snerf_mlp_desc* d, float* workspace, float* gW, snerf_stream_t stream) {
  SNERF_REQUIRE(d && workspace && gW, "mlp_gw_reduce: null argument");
  const int64_t n = snerf_mlp_param_count(d);
  SNERF_REQUIRE(n > 0, "mlp_gw_reduce: bad descriptor");
  hipLaunchKernelGGL(gw_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, gw_ws_stride(d), GW_REPLICAS_H, gW, n);
  SNERF_LAUNCH_CHECK("mlp_gw_reduce");
  return 0;
}

// the workgroup-tile backward of mlp_lp.hip whatever the default kernel for the shape is (A-B runs; cross-check of the wave-owns-rows kernel)
extern "C" int snerf_mlp_bwd_tile(const snerf_mlp_desc* d, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                  int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream) {
  return mlp_bwd_impl(d, W, X, ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx, gW, nullptr, stream, 0, 1);
}

extern "C" int snerf_mlp_bwd_fx(const snerf_mlp_desc* d, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, int64_t* gW_fx, snerf_stream_t stream) {
  return mlp_bwd_impl(d, W, X, ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx, nullptr, reinterpret_cast<long long*>(gW_fx), stream);
}

extern "C" int snerf_mlp_bwd_x16(const snerf_mlp_desc* d, const float* W, const void* X16, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                 int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream) {
  return mlp_bwd_impl(d, W, reinterpret_cast<const float*>(X16), ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx, gW, nullptr, stream, 1);
}

// snerf.h: the sigma_net backward with the quotient epilogue (G = gX .* X16 + the fix list) instead of gX
static int mlp_bwd_x16_quotient_impl(const snerf_mlp_desc* d, const float* W, const void* X16, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                      int32_t aux_col, const float* gaux, float* G, int32_t ldg, int32_t* fix_list, int32_t fix_capacity,
                                      int32_t* fix_count, int32_t* fix_count_next, float* gW, float* ws, snerf_stream_t stream) {
  MlpArgs a = {};
  int rc = fill(d, a);
  if (rc) return rc;
  SNERF_REQUIRE(d->operands == 1 || d->operands == 2, "mlp_bwd_x16_quotient: needs 16-bit operands (desc.operands = 1 / 2), got %d", d->operands);
  SNERF_REQUIRE(d->hidden == 128 && d->n_hidden == 1 && d->d_in % 32 == 0, "mlp_bwd_x16_quotient: built for the sigma_net shapes 32 k -> 128 -> d_out (d_in=%d hidden=%d x %d)",
                d->d_in, d->hidden, d->n_hidden);
  SNERF_REQUIRE(N >= 0 && ldx >= d->d_in && ldg >= d->d_in && fix_capacity >= 0, "mlp_bwd_x16_quotient: N=%lld ldx=%d ldg=%d capacity=%d", (long long)N, ldx, ldg,
                fix_capacity);
  SNERF_REQUIRE((int64_t)N * ldg < (1LL << 31), "mlp_bwd_x16_quotient: N * ldg = %lld does not fit the 32-bit element index of the fix list", (long long)N * ldg);
  SNERF_REQUIRE(fix_count, "mlp_bwd_x16_quotient: null counter");
  hipStream_t st = (hipStream_t)stream;
  if (!fix_count_next) {
    rc = check_hip(hipMemsetAsync(fix_count, 0, sizeof(int32_t), st), "mlp_bwd_x16_quotient memset");
    if (rc) return rc;
  }
  if (N == 0) {
    if (fix_count_next) return check_hip(hipMemsetAsync(fix_count_next, 0, sizeof(int32_t), st), "mlp_bwd_x16_quotient memset");
    return 0;
  }
  SNERF_REQUIRE(W && X16 && G && (fix_list || fix_capacity == 0), "mlp_bwd_x16_quotient: null buffer");
  SNERF_REQUIRE(gY || gaux, "mlp_bwd_x16_quotient: no incoming gradient");
  SNERF_REQUIRE(!gY || ldgy >= d->d_out, "mlp_bwd_x16_quotient: ldgy=%d", ldgy);
  SNERF_REQUIRE(!gaux || (aux_col >= 0 && aux_col < d->d_out), "mlp_bwd_x16_quotient: aux_col=%d", aux_col);
  a.X = reinterpret_cast<const float*>(X16); a.N = N; a.ldx = ldx; a.W = W; a.gY = gY; a.ldgy = ldgy; a.aux_col = aux_col; a.gaux = gaux; a.gW = gW;
  a.x16 = 1;
  a.G = G; a.ldg = ldg; a.fix_list = fix_list; a.fix_capacity = fix_capacity; a.fix_count = fix_count; a.fix_count_next = fix_count_next;
  if (ws) { a.ws = ws; a.ws_rep = GW_REPLICAS_H; a.ws_stride = gw_ws_stride(d); }
  return dispatch(d, a, true, st);
}

extern "C" int snerf_mlp_bwd_x16_quotient(const snerf_mlp_desc* d, const float* W, const void* X16, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                          int32_t aux_col, const float* gaux, float* G, int32_t ldg, int32_t* fix_list, int32_t fix_capacity,
                                          int32_t* fix_count, int32_t* fix_count_next, float* gW, snerf_stream_t stream) {
  return mlp_bwd_x16_quotient_impl(d, W, X16, ldx, N, gY, ldgy, aux_col, gaux, G, ldg, fix_list, fix_capacity, fix_count, fix_count_next, gW, nullptr, stream);
}

extern "C" int snerf_mlp_bwd_x16_quotient_ws(const snerf_mlp_desc* d, const float* W, const void* X16, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                             int32_t aux_col, const float* gaux, float* G, int32_t ldg, int32_t* fix_list, int32_t fix_capacity,
                                             int32_t* fix_count, int32_t* fix_count_next, float* workspace, snerf_stream_t stream) {
  SNERF_REQUIRE(workspace, "mlp_bwd_x16_quotient_ws: null workspace");
  return mlp_bwd_x16_quotient_impl(d, W, X16, ldx, N, gY, ldgy, aux_col, gaux, G, ldg, fix_list, fix_capacity, fix_count, fix_count_next, nullptr, workspace, stream);
}

// One bias-free dense layer Y[N,M] = act(X[N,K] W[K,M]) (act: 0 none, 1 ReLU, 2 Sigmoid).  The kernels hold one 128 x 128 block of W in LDS;
// wider layers are tiled here: column blocks are independent launches, row (K) blocks accumulate into Y and the last one applies the activation.
extern "C" int snerf_dense_fwd(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, float* Y, int32_t ldy,
                               snerf_stream_t stream) {
  SNERF_REQUIRE(K >= 1 && K <= 4096 && M >= 1 && M <= 4096 && act >= 0 && act <= 2, "dense_fwd: K=%d M=%d act=%d (K, M <= 4096)", K, M, act);
  SNERF_REQUIRE(N >= 0 && ldx >= K && ldy >= M, "dense_fwd: N=%lld ldx=%d ldy=%d", (long long)N, ldx, ldy);
  if (N == 0) return 0;
  SNERF_REQUIRE(W && X && Y, "dense_fwd: null buffer");
  for (int k0 = 0; k0 < K; k0 += 128)
    for (int j0 = 0; j0 < M; j0 += 128) {
      const int kb = K - k0 < 128 ? K - k0 : 128, mb = M - j0 < 128 ? M - j0 : 128;
      MlpArgs a = {};
      a.X = X + k0; a.N = N; a.ldx = ldx; a.d0 = kb; a.W = W + (int64_t)k0 * M + j0; a.ldw_g = M; a.dout = mb; a.Y = Y + j0; a.ldy = ldy;
      const bool last_k = k0 + 128 >= K;  // the activation belongs to the complete sum
      a.hidden_act = act == 1 && last_k; a.out_act = act == 2 && last_k; a.acc_y = k0 > 0;
      int rc = dispatch_dense(kb, mb, a, false, (hipStream_t)stream);
      if (rc) return rc;
    }
  return 0;
}

// Backward of snerf_dense_fwd from the layer's stored output Y: gX[N,K] = dZ W^T (written; may be NULL), gW[K,M] += X^T dZ (atomic; may
// be NULL), dZ = gY .* act'(Y).  Tiled like the forward: a K block writes its own columns of gX, later column blocks add to them.
static int dense_bwd_impl(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, const float* Y, int32_t ldy,
                          const float* gY, int32_t ldgy, float* gX, int32_t ldgx, float* gW, long long* gWfx, snerf_stream_t stream) {
  SNERF_REQUIRE(K >= 1 && K <= 4096 && M >= 1 && M <= 4096 && act >= 0 && act <= 2, "dense_bwd: K=%d M=%d act=%d (K, M <= 4096)", K, M, act);
  SNERF_REQUIRE(N >= 0 && ldx >= K && ldy >= M && ldgy >= M && (!gX || ldgx >= K), "dense_bwd: N=%lld ldx=%d ldy=%d ldgy=%d ldgx=%d", (long long)N, ldx,
                ldy, ldgy, ldgx);
  if (N == 0) return 0;
  SNERF_REQUIRE(W && X && Y && gY && (gX || gW || gWfx), "dense_bwd: null buffer");
  for (int k0 = 0; k0 < K; k0 += 128)
    for (int j0 = 0; j0 < M; j0 += 128) {
      const int kb = K - k0 < 128 ? K - k0 : 128, mb = M - j0 < 128 ? M - j0 : 128;
      MlpArgs a = {};
      a.X = X + k0; a.N = N; a.ldx = ldx; a.d0 = kb; a.W = W + (int64_t)k0 * M + j0; a.ldw_g = M; a.dout = mb;
      a.Y = const_cast<float*>(Y) + j0; a.ldy = ldy; a.hidden_act = act == 1; a.out_act = act == 2;
      a.gY = gY + j0; a.ldgy = ldgy; a.gX = gX ? gX + k0 : nullptr; a.ldgx = ldgx; a.acc_gx = j0 > 0;
      a.gW = gW ? gW + (int64_t)k0 * M + j0 : nullptr;
      a.gWfx = gWfx ? gWfx + (int64_t)k0 * M + j0 : nullptr;
      int rc = dispatch_dense(kb, mb, a, true, (hipStream_t)stream);
      if (rc) return rc;
    }
  return 0;
}

extern "C" int snerf_dense_bwd(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, const float* Y, int32_t ldy,
                               const float* gY, int32_t ldgy, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream) {
  return dense_bwd_impl(W, K, M, act, X, ldx, N, Y, ldy, gY, ldgy, gX, ldgx, gW, nullptr, stream);
}

// snerf.h (ABI 13): the weight gradient accumulated into fixed-point cells (any arrival order of the workgroups gives the same bits)
extern "C" int snerf_dense_bwd_fx(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, const float* Y, int32_t ldy,
                                  const float* gY, int32_t ldgy, float* gX, int32_t ldgx, int64_t* gW_fx, snerf_stream_t stream) {
  return dense_bwd_impl(W, K, M, act, X, ldx, N, Y, ldy, gY, ldgy, gX, ldgx, nullptr, reinterpret_cast<long long*>(gW_fx), stream);
}

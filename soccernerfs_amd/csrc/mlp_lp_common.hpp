// Building blocks shared by the 16-bit-operand MLP kernels (mlp_lp.hip) and the fused K-Planes field kernels (field_fused.hip):
// operand types, the "both operands row-major along the contraction index" MFMA helper, accumulator -> LDS stores, weight staging.
#pragma once
#include "common.hpp"

namespace snerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16;
typedef _Float16 fp16;

template <typename T>
struct Ops;
template <>
struct Ops<bf16> {
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  typedef __bf16 v4 __attribute__((ext_vector_type(4)));
  static constexpr float GS = 1.f;  // gradient tile scale
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ bf16 cvt(float x) { return (bf16)x; }
  static __device__ __forceinline__ bf16 cvtg(float x) { return (bf16)x; }
};
template <>
struct Ops<fp16> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  typedef _Float16 v4 __attribute__((ext_vector_type(4)));
  static constexpr float GS = 8192.f;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ fp16 cvt(float x) { return (fp16)fminf(fmaxf(x, -65504.f), 65504.f); }
  static __device__ __forceinline__ fp16 cvtg(float x) { return (fp16)fminf(fmaxf(x, -65504.f), 65504.f); }
};

struct MlpArgs {  // same fields as mlp.hip's (filled there)
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
  int variant;      // backward only: 0 = the default kernel for the shape, 1 = the workgroup-tile kernels of mlp_lp.hip (snerf_mlp_bwd_tile: A-B, cross-check)
  // weight-gradient workspace (snerf_mlp_bwd_ws; 16-bit kernels): ws_rep replicas of the flat gradient, ws_stride floats apart.  Workgroup b adds
  // into replica b % ws_rep instead of gW, so an address collects grid / ws_rep same-address atomics instead of one per workgroup (256 of
  // them took ~25 us at the end of every launch, whatever the element count); snerf_mlp_gw_reduce folds the replicas into gW later.
  float* ws; int ws_rep; int64_t ws_stride;
};

__device__ __forceinline__ void gw_add(const MlpArgs& a, int64_t idx, float v) {
  if (a.gWfx) fx_atomic_add(a.gWfx + idx, v);
  else if (a.ws) atomicAdd(a.ws + (int64_t)(blockIdx.x % (unsigned)a.ws_rep) * a.ws_stride + idx, v);
  else atomicAdd(a.gW + idx, v);
}
constexpr int GW_REPLICAS = 16;  // replicas of a snerf_mlp_bwd_ws workspace

constexpr int LDS_LIMIT_B = 160 * 1024;
__host__ __device__ constexpr int ldb(int k) { return k + 8; }

template <typename T>
__device__ __forceinline__ typename Ops<T>::v8 ld8(const T* p) { return *reinterpret_cast<const typename Ops<T>::v8*>(p); }

// acc[m] (row block m of A, column block nt of the result) += A[., K] * Bt[nt*16 .., K]^T ; both row-major along K
template <int MT, int K, typename T>
__device__ __forceinline__ void mma_rr(const T* A, int lda, const T* Bt, int ldbt, int nt, f32x4 (&acc)[MT], int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const T* bp = Bt + (nt * 16 + lr) * ldbt + lk * 8;
  const T* ap = A + lr * lda + lk * 8;
#pragma unroll
  for (int ks = 0; ks < K / 32; ++ks) {
    const typename Ops<T>::v8 b = ld8(bp + ks * 32);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = Ops<T>::mfma(ld8(ap + m * 16 * lda + ks * 32), b, acc[m]);
  }
}

// accumulator block (rows mt*16.., cols nt*16..) -> row-major image R[row][col] and/or transposed image T[col][row]
template <typename T>
__device__ __forceinline__ void store_rt(T* R, int ldr, T* Tr, int ldt, int mt, int nt, const f32x4& v, int lane) {
  const int col = nt * 16 + (lane & 15);
  const int row0 = mt * 16 + (lane >> 4) * 4;
  const typename Ops<T>::v4 t = {Ops<T>::cvtg(v[0]), Ops<T>::cvtg(v[1]), Ops<T>::cvtg(v[2]), Ops<T>::cvtg(v[3])};
  if (R) {
#pragma unroll
    for (int r = 0; r < 4; ++r) R[(row0 + r) * ldr + col] = t[r];
  }
  if (Tr) *reinterpret_cast<typename Ops<T>::v4*>(Tr + col * ldt + row0) = t;
}

// W [rows_act][cols_act] fp32 row-major (global) -> LDS as stored (R [rows_pad][ldr], zero padded) and/or transposed (T [cols_pad][ldt]).
// Eight UNCONDITIONAL loads (clamped, always valid addresses) are in flight per thread before the first LDS store waits for one: written as
// `cond ? Wg[i] : 0` the load sits in a branch with its own `s_waitcnt vmcnt(0)`, one L2 round trip per element and thread -- 40 in a row for
// sigma_net's 160 x 128 layer, ~25 us in front of every launch (round 5, seen in the ISA).
template <typename T>
__device__ __forceinline__ void stage_w(const float* __restrict__ Wg, int rows_act, int cols_act, int rows_pad, int cols_pad, T* R, int ldr, T* Tr, int ldt) {
  constexpr int U = 8;
  const int total = rows_pad * cols_pad, nt = blockDim.x;
  for (int base = threadIdx.x; base < total; base += nt * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nt < total ? base + u * nt : total - 1;
      const int r = idx / cols_pad, c = idx - r * cols_pad;
      v[u] = Wg[(int64_t)(r < rows_act ? r : rows_act - 1) * cols_act + (c < cols_act ? c : cols_act - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nt;
      if (idx < total) {
        const int r = idx / cols_pad, c = idx - r * cols_pad;
        const T b = Ops<T>::cvt((r < rows_act && c < cols_act) ? v[u] : 0.f);
        if (R) R[r * ldr + c] = b;
        if (Tr) Tr[c * ldt + r] = b;
      }
    }
  }
}

}  // namespace snerf

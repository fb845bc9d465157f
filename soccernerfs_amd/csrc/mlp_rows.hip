// Barrier-free backward of the 64-wide tiny MLPs (K-Planes color_net 15 -> 64 -> 64 -> 3, proposal nets 8 -> 64 -> 1, NeRFPlayer heads;
// NS/fields/kplanes_field.py:249-273,397-407): a WAVE owns 32 samples and carries them through forward recompute, data gradients and the
// weight-gradient products alone -- no __syncthreads inside the persistent loop (mlp_lp.hip's workgroup tile crosses nine per 64 samples
// with a handful of MFMAs between them: MFMA pipes 4 % busy, 60 % of wave cycles waiting, profiles/r04_mfma_pmc.csv).
//
// Everything is computed TRANSPOSED, so that an accumulator tile is the next product's operand with no lane movement and no LDS:
//     Z^T [units x samples] = W^T [units x k] * A_prev^T [k x samples]        A operand = weights (LDS, read-only), B operand = activations
// v_mfma_f32_16x16x32: D lane (g = lane >> 4, c = lane & 15) holds rows 4g .. 4g+3 of column c, i.e. 4 consecutive UNITS of sample c; the B
// operand wants k = 8g .. 8g+7 of column c.  Two D blocks (units 32s + 4g + r and 32s + 16 + 4g + r) converted to 16 bit are exactly one
// B fragment of k-step s if the contraction index is PERMUTED: slot (g, j) <-> unit pi(32s + 8g + j) = 32s + 16 (j >> 2) + 4g + (j & 3).
// The weights are staged into LDS once per workgroup with that permutation of their contraction index, so activations and gradients never
// leave the registers between layers.  The 16-wide contractions (layer 0 of a <= 16-wide input, the output layer's gradient) use
// v_mfma_f32_16x16x16, whose B operand (k = 4g .. 4g+3) IS the D layout.
// Only the weight gradients contract over SAMPLES, which sit on the lanes: for those the wave writes the two operands as [sample][unit]
// images into its private LDS scratch (8-byte stores straight from the packed fragments) and reads them back with ds_read_b64_tr_b16
// (a lane receives a COLUMN of a 4 x 16 block).  The sample slot order of those fragments is (g, j) <-> sample 4g + j (j < 4), 16 + 4g + j - 4:
// the same for both operands, so the contraction is consistent.  Weight-gradient accumulators (dW0, dW1, dWO: 48-144 registers) stay in the
// wave's registers for the whole launch; at the end the workgroup's waves are summed through LDS in a fixed order and flushed with one
// atomic per element and workgroup.
#include <stdint.h>

#include "mlp_lp_common.hpp"

namespace snerf {

typedef short rows_s4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Ops16;
template <>
struct Ops16<bf16> {
  static __device__ __forceinline__ f32x4 mfma(Ops<bf16>::v4 a, Ops<bf16>::v4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(rows_s4, a), __builtin_bit_cast(rows_s4, b), c, 0, 0, 0);
  }
};
template <>
struct Ops16<fp16> {
  static __device__ __forceinline__ f32x4 mfma(Ops<fp16>::v4 a, Ops<fp16>::v4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};

// hidden unit held by contraction slot p of a permuted-k image (see the header)
__host__ __device__ constexpr int rows_pi(int p) { return (p & ~31) + 16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3); }

// row stride (elements) of a [sample][width] scratch image: (stride in dwords) mod 64 is an odd multiple of 8, so the 8 rows a 32-lane half
// touches in one ds_read_b64_tr_b16 (4 rows x 32 B per 16-lane group, two groups) fall on 8 disjoint sets of 8 banks
__host__ __device__ constexpr int rows_img_ld(int width) { return width <= 16 ? 16 : (width <= 32 ? 48 : (width <= 64 ? 80 : 144)); }

// Waves per workgroup: one 8-wave workgroup per CU (two waves per SIMD at <= 256 registers).  A 4-wave variant (two workgroups per CU, grid 512) was
// measured in round 5 on the suspicion that the big workgroup starves beside kernels of small workgroups: alone 0.042 ms against 0.038 (colour net),
// and in the untraced step the kernels were stretched exactly as before (proposal level 1: 0.24-0.27 ms beside the optimiser sweep, 0.03 alone;
// profiles/r05_step_offsets_nw4.txt) -- the stretch is memory latency under an HBM-saturating neighbour, not workgroup slots.
constexpr int ROWS_NW = 8;

// local arrays picked at compile time without taking their address (keeps them in registers)
template <bool SECOND, typename X, typename Y>
__device__ __forceinline__ auto& rows_pick(X& x, Y& y) {
  if constexpr (SECOND) return y; else return x;
}

template <int K0P, int H, int NH>
struct PlanR {
  static constexpr int NW = ROWS_NW;
  static constexpr int L0 = K0P + (K0P == 16 ? 4 : 8), LH = H + 8, LO = 16 + 4;
  static constexpr int W0T = 0;                      // [H][L0]    forward layer 0: natural k (features)
  static constexpr int W0R = W0T + H * L0;           // [K0P][LH]  gX: k = hidden units, permuted
  static constexpr int W1T = W0R + K0P * LH;         // [H][LH]    forward layer 1 (NH == 2): row = out unit, k = in units, permuted
  static constexpr int W1R = W1T + (NH == 2 ? H * LH : 0);  // [H][LH]  gZ1: row = in unit, k = out units, permuted
  static constexpr int WOT = W1R + (NH == 2 ? H * LH : 0);  // [16][LH] output layer: row = output, k = hidden units, permuted
  static constexpr int WOR = WOT + 16 * LH;          // [H][LO]    gZ_last: row = hidden unit, k = outputs, natural
  static constexpr int WEND = (WOR + H * LO + 7) / 8 * 8;
  // per wave: two [32 samples][.] images.  One hidden layer: the wide operand is A for dWO (A_last | gZo) and B for dW0 (X | gZ1), so the images
  // are (wide, narrow) and dW0 swaps their roles; two hidden layers: dW1 (A1 | gZ2) needs two wide ones.
  static constexpr int WIDE = H > K0P ? H : K0P, NARROW = NH == 2 ? H : (K0P > 16 ? K0P : 16);
  static constexpr int LIA = rows_img_ld(WIDE), LIB = rows_img_ld(NARROW);
  static constexpr int IMG_A = 32 * LIA, IMG_B = 32 * LIB;
  static constexpr int SCR = IMG_A + IMG_B;
  static constexpr int TOTAL = WEND + NW * SCR;
  static constexpr int NGW = K0P * H + (NH == 2 ? H * H : 0) + H * 16;  // floats of the workgroup's weight-gradient reduction ([block][r][lane])
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
  static_assert((size_t)NGW * 4 <= (size_t)NW * SCR * 2, "the weight-gradient reduction reuses the waves' scratch");
  static_assert(BYTES <= 160 * 1024, "weights + scratch exceed the CU's LDS");
};

template <typename T, int K0P, int H, int NH>
__global__ __launch_bounds__(ROWS_NW * 64) void mlp_rows_bwd_kernel(MlpArgs a, int64_t n_pairs) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  using P = PlanR<K0P, H, NH>;
  typedef typename Ops<T>::v8 v8t;
  typedef typename Ops<T>::v4 v4t;
  constexpr float GS = Ops<T>::GS;
  constexpr int NW = P::NW, HB = H / 16, HK = H / 32, KB0 = K0P / 16, KS0 = K0P >= 32 ? K0P / 32 : 1;
  constexpr bool X16K = K0P == 16;  // layer 0 contracts over <= 16 features: one 16x16x16 MFMA per block
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;

  // ---- weights -> LDS, once per workgroup (the only barrier in front of the loop).  Constant trip counts: the loops unroll fully, so a
  //      matrix's global loads are all in flight before the first LDS store waits for one ----
  {
    constexpr int NT = NW * 64;
    const float* W0 = a.W + a.woff[0];
#pragma unroll
    for (int i = 0; i < (H * K0P + NT - 1) / NT; ++i) {  // W0T[u][f] = W0[f][u];  W0R[f][p] = W0[f][pi(p)]
      const int idx = threadIdx.x + i * NT;
      if (idx < H * K0P) {
        const int u = idx / K0P, f = idx - u * K0P;
        const int f2 = idx / H, p2 = idx - f2 * H;
        const float v1 = W0[(int64_t)(f < a.d0 ? f : 0) * H + u], v2 = W0[(int64_t)(f2 < a.d0 ? f2 : 0) * H + rows_pi(p2)];
        smem[P::W0T + u * P::L0 + f] = Ops<T>::cvt(f < a.d0 ? v1 : 0.f);
        smem[P::W0R + f2 * P::LH + p2] = Ops<T>::cvt(f2 < a.d0 ? v2 : 0.f);
      }
    }
    if constexpr (NH == 2) {
      const float* W1 = a.W + a.woff[1];
#pragma unroll
      for (int i = 0; i < (H * H + NT - 1) / NT; ++i) {
        const int idx = threadIdx.x + i * NT;
        if (idx < H * H) {
          const int r = idx / H, p = idx - r * H;
          const float v1 = W1[(int64_t)rows_pi(p) * H + r], v2 = W1[(int64_t)r * H + rows_pi(p)];
          smem[P::W1T + r * P::LH + p] = Ops<T>::cvt(v1);  // row = out unit, slot p = in unit pi(p)
          smem[P::W1R + r * P::LH + p] = Ops<T>::cvt(v2);  // row = in unit, slot p = out unit pi(p)
        }
      }
    }
    const float* WO = a.W + a.woff[NH];
    const int dl = a.dout - 1;
#pragma unroll
    for (int i = 0; i < (16 * H + NT - 1) / NT; ++i) {  // WOT[o][p] = WO[pi(p)][o];  WOR[u][o] = WO[u][o]
      const int idx = threadIdx.x + i * NT;
      if (idx < 16 * H) {
        const int o = idx / H, p = idx - o * H;
        const int u2 = idx / 16, o2 = idx - u2 * 16;
        const float v1 = WO[(int64_t)rows_pi(p) * a.dout + (o < a.dout ? o : dl)], v2 = WO[(int64_t)u2 * a.dout + (o2 < a.dout ? o2 : dl)];
        smem[P::WOT + o * P::LH + p] = Ops<T>::cvt(o < a.dout ? v1 : 0.f);
        smem[P::WOR + u2 * P::LO + o2] = Ops<T>::cvt(o2 < a.dout ? v2 : 0.f);
      }
    }
  }
  __syncthreads();

  T* imgA = smem + P::WEND + wave * P::SCR;
  T* imgB = imgA + P::IMG_A;
  const bool relu = a.hidden_act == 1;
  // The kernel is VALU-bound (~1400 instructions per pair against 80 MFMAs for color_net), so the element-wise work is written for
  // instruction count: relu = ONE v_max_i32 on the float's bits (negative floats are negative integers; fmaxf costs a second, canonicalising
  // v_max), relu' = packed 16-bit mask arithmetic on the already converted gradient words, sigmoid / exp by v_exp_f32 + v_rcp_f32 (their
  // 1-2 ulp sit 15 bits below the 16-bit rounding the values meet next).
  const int rlo = relu ? 0 : (int)0x80000000;  // identity: max with INT_MIN
  const uint32_t nomask = relu ? 0u : 0xffffffffu;
  auto hact = [&](f32x4& v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int b = __float_as_int(v[e]);
      v[e] = __int_as_float(b > rlo ? b : rlo);
    }
  };
  // gradient fragment .* relu'(activation fragment), both packed 16-bit: the activations are >= 0 after relu, so "positive" is "bits != 0";
  // per 32-bit word min(a, 1) * 0xffff in packed u16 arithmetic gives the keep-mask of its two elements
  typedef uint32_t rows_u4 __attribute__((ext_vector_type(4)));
  const uint32_t c_one2 = 0x00010001u, c_all2 = 0xffffffffu;
  auto mask_by = [&](v8t gq, v8t act) -> v8t {
    rows_u4 gw = __builtin_bit_cast(rows_u4, gq);
    const rows_u4 aw = __builtin_bit_cast(rows_u4, act);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      uint32_t m;  // VOP3P, both halves at once (the compiler's own lowering of the vector form went through v_cmp / v_cndmask / v_perm)
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(aw[w]), "v"(c_one2));
      asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(c_all2));
      gw[w] &= m | nomask;
    }
    return __builtin_bit_cast(v8t, gw);
  };

  // operand fragments of a weight-gradient product: block `blk` (16 columns) of a [32 samples][ld] image, contraction slot (g, j) <-> sample
  // 4g + j (j < 4) / 16 + 4g + j - 4.  Lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3 of its 4 x 16 block.
  auto tr8 = [&](const T* img, int ld, int blk) -> v8t {
    const int q = c >> 2, p = c & 3;
    const T* a0 = img + (4 * g + q) * ld + blk * 16 + 4 * p;
    typedef __attribute__((address_space(3))) rows_s4 lds_v4;
    const rows_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const rows_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 16 * ld));
    typedef short s8 __attribute__((ext_vector_type(8)));
    const s8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(v8t, w);
  };
  // a packed fragment set (k-step s holds unit blocks 2s and 2s + 1 of sample c) -> image rows 16 sl + c
  auto put_packed = [&](T* img, int ld, int sl, const v8t (&pk)[HK]) {
#pragma unroll
    for (int s = 0; s < HK; ++s) {
      const v4t lo = {pk[s][0], pk[s][1], pk[s][2], pk[s][3]}, hi = {pk[s][4], pk[s][5], pk[s][6], pk[s][7]};
      T* row = img + (16 * sl + c) * ld + 32 * s + 4 * g;
      *reinterpret_cast<v4t*>(row) = lo;
      *reinterpret_cast<v4t*>(row + 16) = hi;
    }
  };
  auto pack2 = [&](const f32x4& b0, const f32x4& b1, bool grad) -> v8t {
    v8t r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[e] = grad ? Ops<T>::cvtg(b0[e]) : Ops<T>::cvt(b0[e]);
      r[4 + e] = grad ? Ops<T>::cvtg(b1[e]) : Ops<T>::cvt(b1[e]);
    }
    return r;
  };

  // weight-gradient accumulators: D lane (g, c) of block (ba, bb) = dW[16 ba + 4g + r][16 bb + c]
  f32x4 dW0[KB0][HB] = {};
  f32x4 dW1[NH == 2 ? HB : 1][NH == 2 ? HB : 1] = {};
  f32x4 dWo[HB] = {};

  // ---- this wave's inputs, one pair ahead ----
  // RAW loaded values only: masking and conversion happen when the pair is consumed (`prep`), so nothing touches the prefetched registers
  // -- and no s_waitcnt is due -- until a whole pair of compute later.  Every load is UNCONDITIONAL at a clamped (always valid) address: a
  // guarded load compiles to a branch with `s_waitcnt vmcnt(0)` inside, which serialised the ~16 loads of a pair in the first version of
  // this kernel (10 k cycles per pair).  The launcher guarantees whole-float4 rows of X (ldx % 4 == 0, 16-byte aligned base).
  constexpr int PERL = X16K ? 4 : 8;  // features per lane and k-step
  struct In {
    float4 xq[2][KS0][PERL / 4];
    float gy[2][4], ga[2];
  };
  const int64_t nlast = a.N - 1;
  const float* gyp = a.gY ? a.gY : a.X;      // absent inputs: element 0 of X stands in (always valid), the value is masked in `prep`
  const float* gap = a.gaux ? a.gaux : a.X;
  const int64_t gy_ld = a.gY ? a.ldgy : 0, ga_on = a.gaux ? 1 : 0;
  const bool gvec = a.gX && (a.ldgx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gX) & 15) == 0;
  auto fetch = [&](In& in, int64_t pair) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int64_t n = pair * 32 + 16 * sl + c;
      const int64_t nn = n < a.N ? n : nlast;
      const float* xr = a.X + nn * a.ldx;
#pragma unroll
      for (int s = 0; s < KS0; ++s)
#pragma unroll
        for (int q = 0; q < PERL / 4; ++q) {
          const int f0 = 32 * s + PERL * g + 4 * q;
          in.xq[sl][s][q] = *reinterpret_cast<const float4*>(xr + (f0 <= a.ldx - 4 ? f0 : a.ldx - 4));  // features beyond the row: its last float4 again
        }
      in.ga[sl] = gap[nn * ga_on];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        in.gy[sl][r] = gyp[nn * gy_ld + (a.gY && o < a.dout ? o : 0)];
      }
    }
  };
  // converted operands of the current pair
  v4t x4[2];
  v8t x8[2][KS0];
  float gyc[2][4], gac[2];
  auto prep = [&](const In& in, int64_t pair) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const bool live = pair * 32 + 16 * sl + c < a.N;
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        float v[PERL];
#pragma unroll
        for (int q = 0; q < PERL / 4; ++q) {
          const float4 t = in.xq[sl][s][q];
          v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int e = 0; e < PERL; ++e) v[e] = (live && 32 * s + PERL * g + e < a.d0) ? v[e] : 0.f;
        if constexpr (X16K) {
          x4[sl] = v4t{Ops<T>::cvt(v[0]), Ops<T>::cvt(v[1]), Ops<T>::cvt(v[2]), Ops<T>::cvt(v[3])};
        } else {
          v8t bq;
#pragma unroll
          for (int e = 0; e < 8; ++e) bq[e] = Ops<T>::cvt(v[e]);
          x8[sl][s] = bq;
        }
      }
      gac[sl] = (live && a.gaux) ? in.ga[sl] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) gyc[sl][r] = (live && a.gY && 4 * g + r < a.dout) ? in.gy[sl][r] : 0.f;
    }
  };

  const int64_t stride = (int64_t)gridDim.x * NW;
  int64_t pair = (int64_t)blockIdx.x * NW + wave;
  In nxt;
  if (pair < n_pairs) fetch(nxt, pair);
  for (; pair < n_pairs; pair += stride) {
    prep(nxt, pair);
    // the next pair's loads (the last iteration re-reads its own pair: valid addresses, nothing consumes them)
    fetch(nxt, pair + stride < n_pairs ? pair + stride : pair);

    // ---- forward, layer 0: Z1^T = W0^T X^T ----
    v8t A1[2][HK];
    {
      f32x4 d[2][HB];
#pragma unroll
      for (int hb = 0; hb < HB; ++hb) {
        const T* wrow = smem + P::W0T + (16 * hb + c) * P::L0;
        f32x4 acc[2] = {};
        if constexpr (X16K) {
          const v4t w = *reinterpret_cast<const v4t*>(wrow + 4 * g);
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) acc[sl] = Ops16<T>::mfma(w, x4[sl], acc[sl]);
        } else {
#pragma unroll
          for (int s = 0; s < KS0; ++s) {
            const v8t w = ld8(wrow + 32 * s + 8 * g);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) acc[sl] = Ops<T>::mfma(w, x8[sl][s], acc[sl]);
          }
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          hact(acc[sl]);
          d[sl][hb] = acc[sl];
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int s = 0; s < HK; ++s) A1[sl][s] = pack2(d[sl][2 * s], d[sl][2 * s + 1], false);
    }
    // ---- forward, layer 1 (NH == 2): Z2^T = W1^T A1^T ----
    v8t A2[NH == 2 ? 2 : 1][HK];
    if constexpr (NH == 2) {
      f32x4 d[2][HB];
#pragma unroll
      for (int hb = 0; hb < HB; ++hb) {
        const T* wrow = smem + P::W1T + (16 * hb + c) * P::LH + 8 * g;
        f32x4 acc[2] = {};
#pragma unroll
        for (int s = 0; s < HK; ++s) {
          const v8t w = ld8(wrow + 32 * s);
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) acc[sl] = Ops<T>::mfma(w, A1[sl][s], acc[sl]);
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          hact(acc[sl]);
          d[sl][hb] = acc[sl];
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int s = 0; s < HK; ++s) A2[sl][s] = pack2(d[sl][2 * s], d[sl][2 * s + 1], false);
    }
    auto& AL = rows_pick<NH == 2>(A1, A2);  // last hidden activations

    // ---- output layer + gradient w.r.t. its pre-activation (lane: outputs 4g .. 4g+3 of sample c) ----
    v4t G0[2];
    {
      f32x4 y[2] = {};
      const T* wrow = smem + P::WOT + c * P::LH + 8 * g;
#pragma unroll
      for (int s = 0; s < HK; ++s) {
        const v8t w = ld8(wrow + 32 * s);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) y[sl] = Ops<T>::mfma(w, AL[sl][s], y[sl]);
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        float gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 4 * g + r;
          float gg = gyc[sl][r];
          if (a.out_act == 1) {
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-y[sl][r]));
            gg = gg * sg * (1.f - sg);
          }
          // trunc_exp backward (activations.py:38-39); ga is 0 without gaux.  A select, not a branch: the clamp keeps exp finite for every column
          gg += o == a.aux_col ? gac[sl] * __expf(fminf(fmaxf(y[sl][r], -15.f), 15.f)) : 0.f;
          gv[r] = o < a.dout ? gg * GS : 0.f;
        }
        G0[sl] = v4t{Ops<T>::cvtg(gv[0]), Ops<T>::cvtg(gv[1]), Ops<T>::cvtg(gv[2]), Ops<T>::cvtg(gv[3])};
      }
    }
    // ---- dWO += A_last^T gZo (contraction over the pair's 32 samples) ----
    put_packed(imgA, P::LIA, 0, AL[0]);
    put_packed(imgA, P::LIA, 1, AL[1]);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) *reinterpret_cast<v4t*>(imgB + (16 * sl + c) * P::LIB + 4 * g) = G0[sl];
    wave_lds_publish();
    {
      const v8t b = tr8(imgB, P::LIB, 0);
#pragma unroll
      for (int ba = 0; ba < HB; ++ba) dWo[ba] = Ops<T>::mfma(tr8(imgA, P::LIA, ba), b, dWo[ba]);
    }
    // ---- gZ_last^T = (WO gZo^T) .* relu'(A_last) ----
    v8t GL[2][HK];
    {
      f32x4 d[2][HB];
#pragma unroll
      for (int hb = 0; hb < HB; ++hb) {
        const v4t w = *reinterpret_cast<const v4t*>(smem + P::WOR + (16 * hb + c) * P::LO + 4 * g);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          f32x4 acc = {};
          d[sl][hb] = Ops16<T>::mfma(w, G0[sl], acc);
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int s = 0; s < HK; ++s) GL[sl][s] = mask_by(pack2(d[sl][2 * s], d[sl][2 * s + 1], true), AL[sl][s]);
    }
    v8t G1[NH == 2 ? 2 : 1][HK];
    if constexpr (NH == 2) {
      // ---- dW1 += A1^T gZ2 ----
      wave_lds_publish();  // the dWO reads of both images are done before they are overwritten
      put_packed(imgA, P::LIA, 0, A1[0]);
      put_packed(imgA, P::LIA, 1, A1[1]);
      put_packed(imgB, P::LIB, 0, GL[0]);
      put_packed(imgB, P::LIB, 1, GL[1]);
      wave_lds_publish();
      {
        v8t b[HB];
#pragma unroll
        for (int bb = 0; bb < HB; ++bb) b[bb] = tr8(imgB, P::LIB, bb);
#pragma unroll
        for (int ba = 0; ba < HB; ++ba) {
          const v8t av = tr8(imgA, P::LIA, ba);
#pragma unroll
          for (int bb = 0; bb < HB; ++bb) dW1[ba][bb] = Ops<T>::mfma(av, b[bb], dW1[ba][bb]);
        }
      }
      // ---- gZ1^T = (W1 gZ2^T) .* relu'(A1) ----
      f32x4 d[2][HB];
#pragma unroll
      for (int hb = 0; hb < HB; ++hb) {
        const T* wrow = smem + P::W1R + (16 * hb + c) * P::LH + 8 * g;
        f32x4 acc[2] = {};
#pragma unroll
        for (int s = 0; s < HK; ++s) {
          const v8t w = ld8(wrow + 32 * s);
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) acc[sl] = Ops<T>::mfma(w, GL[sl][s], acc[sl]);
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          d[sl][hb] = acc[sl];
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int s = 0; s < HK; ++s) G1[sl][s] = mask_by(pack2(d[sl][2 * s], d[sl][2 * s + 1], true), A1[sl][s]);
    }
    auto& GZ1 = rows_pick<NH == 2>(GL, G1);  // gradient of Z1

    // ---- gX^T = W0 gZ1^T -> global (lane: features 16 kb + 4g .. +3 of sample c) ----
    if (a.gX) {
#pragma unroll
      for (int kb = 0; kb < KB0; ++kb) {
        const T* wrow = smem + P::W0R + (16 * kb + c) * P::LH + 8 * g;
        f32x4 acc[2] = {};
#pragma unroll
        for (int s = 0; s < HK; ++s) {
          const v8t w = ld8(wrow + 32 * s);
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) acc[sl] = Ops<T>::mfma(w, GZ1[sl][s], acc[sl]);
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int64_t n = pair * 32 + 16 * sl + c;
          const int f0 = 16 * kb + 4 * g;
          float* dst = a.gX + n * a.ldgx + f0;
          const f32x4 o4 = {acc[sl][0] * (1.f / GS), acc[sl][1] * (1.f / GS), acc[sl][2] * (1.f / GS), acc[sl][3] * (1.f / GS)};
          if (n < a.N) {
            if (gvec && f0 + 3 < a.d0) {
              *reinterpret_cast<f32x4*>(dst) = o4;  // 16 lanes x 4 groups: whole 64-byte rows when ldgx = 16
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (f0 + r < a.d0) dst[r] = o4[r];
            }
          }
        }
      }
    }
    // ---- dW0 += X^T gZ1 ----  (one hidden layer: X goes to the narrow image, gZ1 to the wide one)
    wave_lds_publish();
    T* const xi = NH == 2 ? imgA : imgB;
    T* const gi = NH == 2 ? imgB : imgA;
    constexpr int LXI = NH == 2 ? P::LIA : P::LIB, LGI = NH == 2 ? P::LIB : P::LIA;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      T* row = xi + (16 * sl + c) * LXI;
      if constexpr (X16K) {
        *reinterpret_cast<v4t*>(row + 4 * g) = x4[sl];
      } else {
#pragma unroll
        for (int s = 0; s < KS0; ++s) *reinterpret_cast<v8t*>(row + 32 * s + 8 * g) = x8[sl][s];
      }
    }
    put_packed(gi, LGI, 0, GZ1[0]);
    put_packed(gi, LGI, 1, GZ1[1]);
    wave_lds_publish();
    {
      v8t b[HB];
#pragma unroll
      for (int bb = 0; bb < HB; ++bb) b[bb] = tr8(gi, LGI, bb);
#pragma unroll
      for (int ba = 0; ba < KB0; ++ba) {
        const v8t av = tr8(xi, LXI, ba);
#pragma unroll
        for (int bb = 0; bb < HB; ++bb) dW0[ba][bb] = Ops<T>::mfma(av, b[bb], dW0[ba][bb]);
      }
    }
    wave_lds_publish();  // before the next pair's dWO images
  }

  // ---- weight gradients: sum the workgroup's waves through LDS, then one atomic per element and workgroup ----
  // LDS image: [block][r][lane] (a wave's accumulator register r of block `blk` is 64 consecutive floats: conflict-free).
  if (a.gW || a.gWfx || a.ws) {
    float* red = reinterpret_cast<float*>(smem + P::WEND);
    constexpr int NB0 = KB0 * HB, NB1 = NH == 2 ? HB * HB : 0, NBLK = NB0 + NB1 + HB;
    __syncthreads();  // every wave is done with its scratch
    auto each_block = [&](auto&& f) {
#pragma unroll
      for (int ba = 0; ba < KB0; ++ba)
#pragma unroll
        for (int bb = 0; bb < HB; ++bb) f(ba * HB + bb, dW0[ba][bb]);
      if constexpr (NH == 2) {
#pragma unroll
        for (int ba = 0; ba < HB; ++ba)
#pragma unroll
          for (int bb = 0; bb < HB; ++bb) f(NB0 + ba * HB + bb, dW1[ba][bb]);
      }
#pragma unroll
      for (int ba = 0; ba < HB; ++ba) f(NB0 + NB1 + ba, dWo[ba]);
    };
    // as many private copies as the waves' scratch holds (proposal nets: one per wave, color_net: three): the waves of a round
    // store (round 0) or add (later rounds) side by side, one barrier per round; wave order and copy order are fixed, so the workgroup's
    // fp32 sum has one association order (deterministic mode relies on it).  (ds_add_f32 from all waves at once was 3x slower than this.)
    constexpr int CAP = (int)(((size_t)NW * P::SCR * 2) / ((size_t)P::NGW * 4));  // inside the waves' scratch: the launch asks for no LDS beyond the loop's
    constexpr int COPIES = CAP >= NW ? NW : CAP;
    static_assert(COPIES >= 1, "no room for the weight-gradient reduction");
    for (int w0 = 0; w0 < NW; w0 += COPIES) {
      if (wave >= w0 && wave < w0 + COPIES) {
        float* mine = red + (wave - w0) * P::NGW;
        each_block([&](int blk, const f32x4& v) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* cell = mine + (blk * 4 + r) * 64 + lane;
            *cell = (w0 == 0 ? 0.f : *cell) + v[r];
          }
        });
      }
      __syncthreads();
    }
    // block `blk`, register r, lane (g, c) = dW[16 ba + 4g + r][16 bb + c].  The workgroups start at different blocks, so that 256 of them do
    // not queue on the same cache lines at the same moment.
    for (int i = wave; i < NBLK * 4; i += NW) {
      const int q = (i + 4 * (int)blockIdx.x) % (NBLK * 4);
      const int blk = q >> 2, r = q & 3;
      float v = red[q * 64 + lane];
#pragma unroll
      for (int k = 1; k < COPIES; ++k) v += red[k * P::NGW + q * 64 + lane];
      v *= 1.f / GS;
      if (blk < NB0) {
        const int f = 16 * (blk / HB) + 4 * g + r, u = 16 * (blk % HB) + c;
        if (f < a.d0) gw_add(a, a.woff[0] + (int64_t)f * H + u, v);
      } else if (blk < NB0 + NB1) {
        const int b1 = blk - NB0;
        gw_add(a, a.woff[1] + (int64_t)(16 * (b1 / HB) + 4 * g + r) * H + 16 * (b1 % HB) + c, v);
      } else {
        const int u = 16 * (blk - NB0 - NB1) + 4 * g + r;
        if (c < a.dout) gw_add(a, a.woff[NH] + (int64_t)u * a.dout + c, v);
      }
    }
  }
}

template <typename T, int K0P, int H, int NH>
static int launch_rows(const MlpArgs& a, hipStream_t st) {
  using P = PlanR<K0P, H, NH>;
  static_assert(P::BYTES <= LDS_LIMIT_B, "rows backward does not fit LDS");
  const int64_t n_pairs = (a.N + 31) / 32;
  int64_t grid = (n_pairs + P::NW - 1) / P::NW;
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  auto k = mlp_rows_bwd_kernel<T, K0P, H, NH>;
  SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(P::NW * 64), P::BYTES, st, a, n_pairs);
  SNERF_LAUNCH_CHECK("mlp_bwd (16-bit operands, wave-owns-rows)");
  return 0;
}

// shapes served: 64 hidden units, fp32 X in float4-granular rows; one hidden layer with inputs up to 64 wide, two hidden layers with inputs up to 16 wide (the
// 64 x 64 weight-gradient accumulator of the second hidden layer leaves no registers for a wider layer 0: 100-220 spilled VGPRs measured)
bool mlp_rows_supported(const snerf_mlp_desc* d, const void* args) {
  const MlpArgs& a = *static_cast<const MlpArgs*>(args);
  if (d->hidden != 64 || d->d_out > 16 || a.x16 || a.G || !(d->operands == 1 || d->operands == 2)) return false;
  // X is fetched as whole float4s, unconditionally (the kernel's prefetch has no guarded loads): rows must be float4-granular and aligned;
  // anything else takes the workgroup-tile kernel
  if ((a.ldx & 3) != 0 || a.ldx < 4 || (reinterpret_cast<uintptr_t>(a.X) & 15) != 0) return false;
  return (d->n_hidden == 1 && d->d_in <= 64) || (d->n_hidden == 2 && d->d_in <= 16);
}

int mlp_rows_dispatch(const snerf_mlp_desc* d, const void* args, hipStream_t st) {
  const MlpArgs& a = *static_cast<const MlpArgs*>(args);
  const int k0 = d->d_in <= 16 ? 16 : (d->d_in <= 32 ? 32 : 64);
#define CASE(K0P, NH)                              \
  if (k0 == K0P && d->n_hidden == NH)              \
    return d->operands == 2 ? launch_rows<fp16, K0P, 64, NH>(a, st) : launch_rows<bf16, K0P, 64, NH>(a, st);
  CASE(16, 1) CASE(16, 2) CASE(32, 1) CASE(64, 1)
#undef CASE
  set_error("mlp rows backward: unsupported shape d_in=%d hidden=%d n_hidden=%d", d->d_in, d->hidden, d->n_hidden);
  return SNERF_ERR_UNSUPPORTED;
}

}  // namespace snerf

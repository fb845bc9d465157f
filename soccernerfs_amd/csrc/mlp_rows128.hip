// Backward of sigma_net (d_in = 32 n_scales -> 128 -> 16, one hidden layer; NS/fields/kplanes_field.py:249-261) with 16-bit MFMA operands, the
// wave-owns-rows way of mlp_rows.hip scaled to a 128-wide hidden layer and a 160-wide input.
//
// The weight-gradient accumulator of layer 0 alone is 160 x 128 fp32 = 320 registers per lane if ONE wave held it, so the kernel splits the work in
// two kinds of phases over a 128-sample tile of an 8-wave workgroup:
//   * chain phases, barrier-free: wave w carries ITS 16 samples through the forward recompute, the output gradient, the hidden gradient and the input
//     gradient (+ the quotient epilogue G = gX .* X), everything transposed ([units x samples]) so that an accumulator tile is the next product's B
//     operand with no LDS round trip (see mlp_rows.hip's header for the idiom);
//   * weight-gradient phases, cooperative: the waves leave A1 / gZo / gZ1 as [sample][unit] images in LDS (the X tile is there already) and wave w
//     accumulates the blocks of hidden units 16w .. 16w+15 over all 128 samples, operands read with ds_read_b64_tr_b16.
// Five barriers per 128 samples (the workgroup-tile kernel of mlp_lp.hip: nine per 64), each with hundreds of MFMAs' worth of work in front of it.
//
// ONE LDS image of W0 serves both of its uses, W0T [unit][feature]: the forward Z1^T = W0^T X^T reads its rows (16-byte reads, natural feature
// order), the input gradient gX^T = W0 gZ1^T reads it TRANSPOSED with ds_read_b64_tr_b16 -- 4 rows (units) x 16 contiguous columns (features) per
// 16-lane group, conflict-free with the row stride chosen below -- in the contraction-slot order the packed gradient fragments have anyway:
// slot (g, j) of k-step s = unit pi(32s + 8g + j) = 32s + 16 (j >> 2) + 4g + (j & 3)  (mlp_rows.hip), i.e. rows 32s + 4g .. +3 and 32s + 16 + 4g .. +3.
// (A first version kept W0 [feature][unit] and read the forward's operand transposed in four 4-unit chunks 8 apart: 4-way bank conflicts on 80 reads
// per wave and tile made the kernel LDS-bound -- 0.112 ms against the tile kernel's 0.123.)
#include <stdint.h>

#include "mlp_lp_common.hpp"

namespace snerf {
namespace r128 {

typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <typename T>
struct M16;
template <>
struct M16<bf16> {
  static __device__ __forceinline__ f32x4 mfma(Ops<bf16>::v4 a, Ops<bf16>::v4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
  }
};
template <>
struct M16<fp16> {
  static __device__ __forceinline__ f32x4 mfma(Ops<fp16>::v4 a, Ops<fp16>::v4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};

constexpr int NW = 8, TS = 128, H = 128, HB = 8, HK = 4;

// hidden unit held by contraction slot p of a packed fragment / a permuted-k weight image (mlp_rows.hip)
__host__ __device__ constexpr int pi_unit(int p) { return (p & ~31) + 16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3); }

template <int K0>
struct Plan {
  // row strides (elements): (stride / 2) mod 64 is an odd multiple of 8 for every image read with ds_read_b64_tr_b16 (W0T, XI, AI, GO): the 8 rows a
  // 32-lane half touches fall on 8 disjoint sets of 8 banks
  static constexpr int LWT = K0 + 16, LW = H + 8, LO = 20, LX = K0 + 16, LA = H + 16, LG = 16;
  static constexpr int W0T = 0;                    // [H][LWT]   W0T[u][f] = W0[f][u]
  static constexpr int WOT = W0T + H * LWT;        // [16][LW]   row = output, slot p = unit pi(p)
  static constexpr int WOR = WOT + 16 * LW;        // [H][LO]    WO[u][.], natural rows
  static constexpr int XI = (WOR + H * LO + 7) / 8 * 8;  // [TS][LX]   X tile, [sample][feature]
  static constexpr int AI = XI + TS * LX;          // [TS][LA]   A1, later gZ1: [sample][unit]
  static constexpr int GO = AI + TS * LA;          // [TS][LG]   gZo: [sample][output]
  static constexpr int TOTAL = GO + TS * LG;
  static constexpr size_t BYTES = (size_t)TOTAL * 2;
};

// two ds_read_b64_tr_b16 = one operand fragment: element j of lane (g, i) = img[row0 + 4g + j (j < 4) | row0 + 16 + 4g + j - 4][col0 + i]
template <typename T>
__device__ __forceinline__ typename Ops<T>::v8 tr8(const T* img, int ld, int row0, int col0, int g, int c) {
  const int q = c >> 2, p = c & 3;
  const T* a0 = img + (row0 + 4 * g + q) * ld + col0 + 4 * p;
  typedef __attribute__((address_space(3))) s4 lds_v4;
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 16 * ld));
  const s8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename Ops<T>::v8, w);
}

template <typename T, int K0, bool QG>
__global__ __launch_bounds__(NW * 64) void sigma_bwd_kernel(MlpArgs a, int64_t n_tiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  using P = Plan<K0>;
  typedef typename Ops<T>::v8 v8t;
  typedef typename Ops<T>::v4 v4t;
  constexpr float GS = Ops<T>::GS;
  constexpr int KS0 = K0 / 32, KB0 = K0 / 16, NT = NW * 64;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int g = lane >> 4, c = lane & 15;
  T *W0T = smem + P::W0T, *WOT = smem + P::WOT, *WOR = smem + P::WOR, *XI = smem + P::XI, *AI = smem + P::AI, *GO = smem + P::GO;

  // ---- weights -> LDS (unconditional batched loads: mlp_lp_common.hpp::stage_w) ----
  stage_w<T>(a.W + a.woff[0], a.d0, H, K0, H, nullptr, 0, W0T, P::LWT);
  {
    const float* WO = a.W + a.woff[1];
    const int dl = a.dout - 1;
#pragma unroll
    for (int i = 0; i < (H * 16 + NT - 1) / NT; ++i) {  // WOT[o][p] = WO[pi(p)][o];  WOR[u][o] = WO[u][o]
      const int idx = threadIdx.x + i * NT;
      const int o = (idx / H) & 15, p = idx & (H - 1);
      const int u2 = (idx >> 4) & (H - 1), o2 = idx & 15;
      const float v1 = WO[(int64_t)pi_unit(p) * a.dout + (o < a.dout ? o : dl)], v2 = WO[(int64_t)u2 * a.dout + (o2 < a.dout ? o2 : dl)];
      if (idx < H * 16) {
        WOT[o * P::LW + p] = Ops<T>::cvt(o < a.dout ? v1 : 0.f);
        WOR[u2 * P::LO + o2] = Ops<T>::cvt(o2 < a.dout ? v2 : 0.f);
      }
    }
  }
  if constexpr (QG) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.fix_count_next) *a.fix_count_next = 0;
  }
  const bool relu = a.hidden_act == 1;
  const int rlo = relu ? 0 : (int)0x80000000;
  const uint32_t nomask = relu ? 0u : 0xffffffffu, c_one2 = 0x00010001u, c_all2 = 0xffffffffu;
  auto hact = [&](f32x4& v) {  // relu = one v_max_i32 on the float's bits (mlp_rows.hip)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int b = __float_as_int(v[e]);
      v[e] = __int_as_float(b > rlo ? b : rlo);
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
  auto mask_by = [&](v8t gq, v8t act) -> v8t {  // gradient .* relu'(activation), packed: activations are >= 0, so positive <=> bits != 0
    u4 gw = __builtin_bit_cast(u4, gq);
    const u4 aw = __builtin_bit_cast(u4, act);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      uint32_t m;
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(aw[w]), "v"(c_one2));
      asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(c_all2));
      gw[w] &= m | nomask;
    }
    return __builtin_bit_cast(v8t, gw);
  };

  // weight-gradient accumulators of THIS wave's hidden units 16 wave .. +15: dW0 block fb = [features 16 fb + 4g + r][unit 16 wave + c]
  f32x4 dW0[KB0] = {};
  f32x4 dWo = {};  // [unit 16 wave + 4g + r][output c]

  // ---- the X tile one tile ahead in registers (16-byte chunks, row-major: coalesced), the wave's own gY / gaux likewise ----
  constexpr int NV = TS * (K0 / 8), PERX = (NV + NT - 1) / NT;
  const T* X16 = reinterpret_cast<const T*>(a.X);
  const int64_t nlast = a.N - 1;
  v8t xq[PERX];
  float4 gyq;
  float gaq;
  const float* gyp = a.gY ? a.gY : reinterpret_cast<const float*>(a.X);  // absent inputs: a valid address, masked when consumed
  const float* gap = a.gaux ? a.gaux : reinterpret_cast<const float*>(a.X);
  const int64_t gy_ld = a.gY ? a.ldgy : 0, gy_off = a.gY ? 4 * g : 0, ga_on = a.gaux ? 1 : 0;
  auto fetch = [&](int64_t tile) {
    const int64_t n0 = tile * TS;
#pragma unroll
    for (int i = 0; i < PERX; ++i) {
      const int vi = threadIdx.x + i * NT;
      const int vc = vi < NV ? vi : NV - 1;
      const int r = vc / (K0 / 8), c8 = vc - r * (K0 / 8);
      const int64_t n = n0 + r < a.N ? n0 + r : nlast;
      xq[i] = *reinterpret_cast<const v8t*>(X16 + n * a.ldx + c8 * 8);
    }
    const int64_t n = n0 + 16 * wave + c;
    const int64_t nn = n < a.N ? n : nlast;
    gyq = *reinterpret_cast<const float4*>(gyp + nn * gy_ld + gy_off);
    gaq = gap[nn * ga_on];
  };

  const bool gvec = a.gX && (a.ldgx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gX) & 15) == 0;
  int64_t tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TS;
    lds_barrier();  // B0: the previous tile's weight-gradient reads of XI / AI are done (first tile: the weights are staged).  LDS-only barriers:
    //               __syncthreads() would wait for the prefetch loads and the G stores in flight as well
    // ---- X tile -> LDS (rows beyond N: zeros); this wave's incoming gradients -> registers ----
#pragma unroll
    for (int i = 0; i < PERX; ++i) {
      const int vi = threadIdx.x + i * NT;
      if (vi < NV) {
        const int r = vi / (K0 / 8), c8 = vi - r * (K0 / 8);
        v8t z = {};
        *reinterpret_cast<v8t*>(XI + r * P::LX + c8 * 8) = n0 + r < a.N ? xq[i] : z;
      }
    }
    const bool live = n0 + 16 * wave + c < a.N;
    float gy[4] = {gyq.x, gyq.y, gyq.z, gyq.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) gy[r] = (live && a.gY && 4 * g + r < a.dout) ? gy[r] : 0.f;
    const float ga = (live && a.gaux) ? gaq : 0.f;
    fetch(tile + gridDim.x < n_tiles ? tile + gridDim.x : tile);  // the last tile re-reads itself: valid addresses, nothing consumes them
    lds_barrier();  // B1: XI complete

    // ================= chain, part 1: forward recompute + output gradient, this wave's 16 samples =================
    const T* xrow = XI + (16 * wave + c) * P::LX + 8 * g;
    v8t A1p[HK];
    {
      v8t xb[KS0];
#pragma unroll
      for (int s = 0; s < KS0; ++s) xb[s] = ld8(xrow + 32 * s);
      f32x4 d[HB];
#pragma unroll
      for (int b = 0; b < HB; ++b) {
        f32x4 acc = {};
        const T* wrow = W0T + (16 * b + c) * P::LWT + 8 * g;  // A operand = W0^T: row = unit 16b + c, k = features 32s + 8g + j
#pragma unroll
        for (int s = 0; s < KS0; ++s) acc = Ops<T>::mfma(ld8(wrow + 32 * s), xb[s], acc);
        hact(acc);
        d[b] = acc;
      }
#pragma unroll
      for (int s = 0; s < HK; ++s) A1p[s] = pack2(d[2 * s], d[2 * s + 1], false);  // slot (g, j) of k-step s = unit pi(32s + 8g + j)
    }
    v4t G0;
    {
      f32x4 y = {};
      const T* wrow = WOT + c * P::LW + 8 * g;
#pragma unroll
      for (int s = 0; s < HK; ++s) y = Ops<T>::mfma(ld8(wrow + 32 * s), A1p[s], y);
      float gv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 4 * g + r;
        float gg = gy[r];
        if (a.out_act == 1) {
          const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-y[r]));
          gg = gg * sg * (1.f - sg);
        }
        gg += o == a.aux_col ? ga * __expf(fminf(fmaxf(y[r], -15.f), 15.f)) : 0.f;  // trunc_exp backward (activations.py:38-39)
        gv[r] = o < a.dout ? gg * GS : 0.f;
      }
      G0 = v4t{Ops<T>::cvtg(gv[0]), Ops<T>::cvtg(gv[1]), Ops<T>::cvtg(gv[2]), Ops<T>::cvtg(gv[3])};
    }
    // images for dWO: A1 [sample][unit] (a packed fragment holds units 32s + 4g .. +3 and 32s + 16 + 4g .. +3: two 8-byte stores) and gZo [sample][output]
    auto put_packed = [&](const v8t (&pk)[HK]) {
      T* arow = AI + (16 * wave + c) * P::LA + 4 * g;
#pragma unroll
      for (int s = 0; s < HK; ++s) {
        const v4t lo = {pk[s][0], pk[s][1], pk[s][2], pk[s][3]}, hi = {pk[s][4], pk[s][5], pk[s][6], pk[s][7]};
        *reinterpret_cast<v4t*>(arow + 32 * s) = lo;
        *reinterpret_cast<v4t*>(arow + 32 * s + 16) = hi;
      }
    };
    put_packed(A1p);
    *reinterpret_cast<v4t*>(GO + (16 * wave + c) * P::LG + 4 * g) = G0;
    lds_barrier();  // B2: A1 / gZo images complete
    // ---- dWO[units 16 wave ..][outputs] += A1^T gZo over the tile's 128 samples ----
#pragma unroll
    for (int ks = 0; ks < TS / 32; ++ks)
      dWo = Ops<T>::mfma(tr8<T>(AI, P::LA, 32 * ks, 16 * wave, g, c), tr8<T>(GO, P::LG, 32 * ks, 0, g, c), dWo);

    // ================= chain, part 2: hidden gradient, input gradient (+ quotient epilogue) =================
    v8t G1p[HK];
    {
      f32x4 d[HB];
#pragma unroll
      for (int b = 0; b < HB; ++b) {
        const v4t w = *reinterpret_cast<const v4t*>(WOR + (16 * b + c) * P::LO + 4 * g);
        f32x4 acc = {};
        d[b] = M16<T>::mfma(w, G0, acc);
      }
#pragma unroll
      for (int s = 0; s < HK; ++s) G1p[s] = mask_by(pack2(d[2 * s], d[2 * s + 1], true), A1p[s]);
    }
    if (a.gX || QG) {
#pragma unroll
      for (int kb = 0; kb < KB0; ++kb) {
        f32x4 acc = {};
#pragma unroll
        for (int s = 0; s < HK; ++s) acc = Ops<T>::mfma(tr8<T>(W0T, P::LWT, 32 * s, 16 * kb, g, c), G1p[s], acc);  // A = W0: row = feature 16 kb + c, k = units (pi order)
        // lane (g, c): features 16 kb + 4g .. +3 of sample 16 wave + c
        const int64_t n = n0 + 16 * wave + c;
        const int f0 = 16 * kb + 4 * g;
        // (d_in == K0 for every shape this kernel serves -- the dispatcher checks it -- so every feature column is live: no per-column guards)
        if constexpr (QG) {
          const v4t xv = *reinterpret_cast<const v4t*>(XI + (16 * wave + c) * P::LX + f0);
          f32x4 o4, gx4;
          bool anyfix = false;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = (float)xv[r];
            gx4[r] = acc[r] * (1.f / GS);
            const bool vanished = fabsf(x) < QUOT_TINY;
            o4[r] = vanished ? 0.f : gx4[r] * x;
            anyfix |= vanished && gx4[r] != 0.f;
          }
          if (n < a.N) {
            *reinterpret_cast<f32x4*>(a.G + n * a.ldg + f0) = o4;
            if (__builtin_expect(anyfix, 0)) {  // ONE branch per block: a vanished feature with a live gradient is rare (never, in training so far)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (fabsf((float)xv[r]) < QUOT_TINY && gx4[r] != 0.f) fix_append(a.fix_list, a.fix_capacity, a.fix_count, (int32_t)(n * a.ldg + f0 + r), gx4[r]);
            }
          }
        } else {
          if (n < a.N) {
            float* dst = a.gX + n * a.ldgx + f0;
            const f32x4 o4 = {acc[0] * (1.f / GS), acc[1] * (1.f / GS), acc[2] * (1.f / GS), acc[3] * (1.f / GS)};
            if (gvec) *reinterpret_cast<f32x4*>(dst) = o4;
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r) dst[r] = o4[r];
            }
          }
        }
      }
    }
    lds_barrier();  // B3: every wave has read the A1 image (dWO); it now becomes the gZ1 image
    put_packed(G1p);
    lds_barrier();  // B4: gZ1 image complete
    // ---- dW0[features][units 16 wave ..] += X^T gZ1 over the tile's 128 samples ----
    {
      v8t bq[TS / 32];
#pragma unroll
      for (int ks = 0; ks < TS / 32; ++ks) bq[ks] = tr8<T>(AI, P::LA, 32 * ks, 16 * wave, g, c);
#pragma unroll
      for (int fb = 0; fb < KB0; ++fb)
#pragma unroll
        for (int ks = 0; ks < TS / 32; ++ks) dW0[fb] = Ops<T>::mfma(tr8<T>(XI, P::LX, 32 * ks, 16 * fb, g, c), bq[ks], dW0[fb]);
    }
  }

  // ---- weight gradients: every wave owns its hidden units outright -- straight to gW / the replica workspace ----
  if (a.gW || a.gWfx || a.ws) {
#pragma unroll
    for (int fb = 0; fb < KB0; ++fb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * fb + 4 * g + r;
        if (f < a.d0) gw_add(a, a.woff[0] + (int64_t)f * H + 16 * wave + c, dW0[fb][r] * (1.f / GS));
      }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (c < a.dout) gw_add(a, a.woff[1] + (int64_t)(16 * wave + 4 * g + r) * a.dout + c, dWo[r] * (1.f / GS));
  }
}

template <typename T, int K0>
static int launch(const MlpArgs& a, hipStream_t st) {
  using P = Plan<K0>;
  static_assert(P::BYTES <= LDS_LIMIT_B, "sigma backward (rows) does not fit LDS");
  const int64_t n_tiles = (a.N + TS - 1) / TS;
  int64_t grid = n_tiles < 256 ? n_tiles : 256;
  if (a.G) {
    auto k = sigma_bwd_kernel<T, K0, true>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NW * 64), P::BYTES, st, a, n_tiles);
  } else {
    auto k = sigma_bwd_kernel<T, K0, false>;
    SNERF_ALLOW_LDS(k, LDS_LIMIT_B);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NW * 64), P::BYTES, st, a, n_tiles);
  }
  SNERF_LAUNCH_CHECK("mlp_bwd (sigma_net, 16-bit operands, wave-owns-rows)");
  return 0;
}

}  // namespace r128

// shapes served: 32 k -> 128 -> d_out <= 16, one hidden layer, X in the operand type (what snerf_kplanes_field_fwd writes) in 16-byte-granular rows,
// gY rows of whole float4s
bool mlp_rows128_supported(const snerf_mlp_desc* d, const void* args) {
  const MlpArgs& a = *static_cast<const MlpArgs*>(args);
  if (d->hidden != 128 || d->n_hidden != 1 || d->d_in % 32 != 0 || d->d_in > 192 || d->d_out > 16 || !(d->operands == 1 || d->operands == 2)) return false;
  if (!a.x16 || (a.ldx & 7) != 0 || (reinterpret_cast<uintptr_t>(a.X) & 15) != 0) return false;
  if (a.gY && ((a.ldgy & 3) != 0 || a.ldgy < 16 || (reinterpret_cast<uintptr_t>(a.gY) & 15) != 0)) return false;
  if (a.G && ((a.ldg & 3) != 0 || (reinterpret_cast<uintptr_t>(a.G) & 15) != 0)) return false;
  return true;
}

int mlp_rows128_dispatch(const snerf_mlp_desc* d, const void* args, hipStream_t st) {
  const MlpArgs& a = *static_cast<const MlpArgs*>(args);
#define CASE(K0) \
  if (d->d_in == K0) return d->operands == 2 ? r128::launch<fp16, K0>(a, st) : r128::launch<bf16, K0>(a, st);
  CASE(32) CASE(64) CASE(96) CASE(128) CASE(160) CASE(192)
#undef CASE
  set_error("mlp rows128 backward: unsupported d_in=%d", d->d_in);
  return SNERF_ERR_UNSUPPORTED;
}

}  // namespace snerf

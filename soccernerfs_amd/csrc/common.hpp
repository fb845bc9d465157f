// Shared host/device helpers for libsnerf (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/snerf.h"

namespace snerf {

// ---- error plumbing: integer codes across the C ABI, never exceptions ----
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define SNERF_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::snerf::set_error(__VA_ARGS__);           \
      return SNERF_ERR_ARG;                      \
    }                                            \
  } while (0)

#define SNERF_LAUNCH_CHECK(name)                                  \
  do {                                                            \
    int _rc = ::snerf::check_hip(hipGetLastError(), name);        \
    if (_rc) return _rc;                                          \
  } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Dynamic LDS above 64 KB has to be allowed per kernel function AND per device (hipFuncSetAttribute acts on the current device).  Which
// (kernel, device) pairs have been told already is a write-once cache of a device property, not library state: one atomic bit per device
// ordinal, set after the call, so any thread on any device gets the same launch behaviour whichever arrives first (core.hip).
void allow_dynamic_lds(const void* kernel, int bytes, unsigned long long* done_bits);
#define SNERF_ALLOW_LDS(kernel, bytes)                                             \
  do {                                                                             \
    static unsigned long long _snerf_lds_done = 0;                                 \
    ::snerf::allow_dynamic_lds((const void*)(kernel), (bytes), &_snerf_lds_done);  \
  } while (0)

constexpr int WAVE = 64;  // CDNA wavefront

// ---- wave-level primitives (64 lanes) ----
// LDS written by some lanes of a wave and read by other lanes of the SAME wave: release the stores, hold the wave's instruction stream at a
// scheduling barrier (no instruction emitted), acquire for the loads.  Costs one s_waitcnt lgkmcnt(0).
__device__ __forceinline__ void wave_lds_publish() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// inclusive prefix sum across the 64 lanes (Hillis-Steele; association order differs from sequential)
__device__ __forceinline__ float wave_inclusive_scan(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    float t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// Workgroup barrier for data exchanged through LDS only: `__syncthreads()` is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`, i.e. it also waits
// for the wave's outstanding GLOBAL loads and stores -- a register prefetch of the next tile issued in front of it is waited for on the spot.
// This one orders LDS accesses only (ISA: s_waitcnt lgkmcnt(0); s_barrier).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- deterministic gradient accumulation (KPlanesTrainConfig.deterministic) ----
// Float atomics make a sum depend on the order in which wavefronts arrive.  In deterministic mode gradients are accumulated as 2^50-scaled
// 64-bit integers instead (integer addition is associative: any order gives the same bits) and converted back once per step
// (snerf_fx_to_float).  Range +-8192, resolution 8.9e-16; a non-finite contribution adds nothing.
constexpr double FX_SCALE = 1125899906842624.0;  // 2^50
__device__ __forceinline__ void fx_atomic_add(long long* p, float v) {
  if (!(fabsf(v) <= 3.402823466e+38f)) return;
  double x = (double)v * FX_SCALE;
  x = fmin(fmax(x, -4.0e18), 4.0e18);
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn(x));
}
// one accumulator that is either a float (atomic fp32 add) or a fixed-point cell
template <bool FX>
__device__ __forceinline__ void grad_add(float* g, long long* gfx, int64_t idx, float v) {
  if (FX) fx_atomic_add(gfx + idx, v); else atomicAdd(g + idx, v);
}

// ---- per-ray prefix sums by the whole wavefront, accumulated in double ----
// The reference's cumsum runs on the CPU in ATen: a sequential double accumulator rounded to fp32 per element (cumsum_cpu_kernel,
// acc_type<float, false> = double); the sample INDICES of the PDF sampler and the median-depth index depend on those roundings
// (bit-exact `inds`).  Round 1 reproduced it with one lane walking the ray (63 lanes idle, ~2.5 us per ray and scan).  Here
// lane l owns the contiguous block [l * PER, (l + 1) * PER) of the ray's n <= 64 * WSCAN_PER elements: a sequential double prefix inside
// the block, a Hillis-Steele scan of the 64 block totals in double, one add.  Double sums of <= 320 fp32 values differ between the two
// association orders by ~1e-16 relative, so the value rounded to fp32 is the same unless a sum sits within that distance of an fp32
// rounding boundary (~1e-9 per element) -- indices stay bit-exact on every fixture and random test of the suite.
//   in[0..n) (LDS) -> out[i] = (float) sum_{j <= i} in[j]   (EXCLUSIVE: sum_{j < i});  REVERSE: the same from the far end (suffix sums).
// Returns the total in every lane.  in and out may alias (a lane reads and writes only its own block).
constexpr int WSCAN_PER = 5;  // 64 * 5 = 320 = the per-ray kernels' MAX_S
template <bool EXCLUSIVE, bool REVERSE>
__device__ __forceinline__ double wave_scan_f64(const float* in, float* out, int n, int lane) {
  const int per = (n + 63) >> 6;
  const int b0 = lane * per;
  float v[WSCAN_PER];
  double local = 0.0;
#pragma unroll
  for (int k = 0; k < WSCAN_PER; ++k) {
    const int i = b0 + k;
    v[k] = (k < per && i < n) ? in[REVERSE ? n - 1 - i : i] : 0.f;
    local = local + (double)v[k];
  }
  double incl = local;  // inclusive scan of the block totals over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(incl, off, 64);
    if (lane >= off) incl = incl + t;
  }
  // sum of all earlier blocks = the previous lane's inclusive value.  (NOT incl - local: an overflowed density makes both +inf, and
  // inf - inf = NaN would poison the elements in FRONT of the overflow inside the same block, which the sequential cumsum leaves finite.)
  double run = __shfl_up(incl, 1, 64);
  if (lane == 0) run = 0.0;
  const double total = __shfl(incl, 63, 64);
#pragma unroll
  for (int k = 0; k < WSCAN_PER; ++k) {
    const int i = b0 + k;
    if (k < per && i < n) {
      if (EXCLUSIVE) out[REVERSE ? n - 1 - i : i] = (float)run;
      run = run + (double)v[k];
      if (!EXCLUSIVE) out[REVERSE ? n - 1 - i : i] = (float)run;
    }
  }
  return total;
}

// ---- quotient form of the K-Planes plane scatter (kplanes_sorted.hip; produced also by the sigma_net backward's epilogue, mlp_lp.hip) ----
// a value below the smallest normal float counts as "vanished": pass B adds nothing for it and the fix-up supplies the exact term
// (v_rcp_f32 may flush a subnormal operand, which would turn 0 * inf into a NaN gradient)
constexpr float QUOT_TINY = 1.17549435e-38f;
// fix list entry = {element index into the [N, C n_scales] feature tensor, the feature gradient there}
__device__ __forceinline__ void fix_append(int32_t* __restrict__ list, int capacity, int32_t* __restrict__ count, int32_t elem, float g) {
  const int slot = atomicAdd(count, 1);
  if (slot < capacity) {
    list[2 * slot] = elem;
    list[2 * slot + 1] = __float_as_int(g);
  }
}

// nan_to_num with torch defaults (nan->0, +inf->FLT_MAX, -inf->-FLT_MAX)
__device__ __forceinline__ float nan_to_num(float v) {
  if (v != v) return 0.f;
  if (v == INFINITY) return 3.4028234663852886e38f;
  if (v == -INFINITY) return -3.4028234663852886e38f;
  return v;
}

}  // namespace snerf

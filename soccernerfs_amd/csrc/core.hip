// Library identity + error plumbing for the C ABI (include/snerf.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace snerf {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

void allow_dynamic_lds(const void* kernel, int bytes, unsigned long long* done_bits) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(done_bits, __ATOMIC_ACQUIRE) & bit) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);  // a failure surfaces at the launch check
  __atomic_fetch_or(done_bits, bit, __ATOMIC_RELEASE);
}

// fixed-point gradient cells -> float gradients (deterministic mode, common.hpp): out (+)= fx * 2^-50; the cells are cleared
__global__ __launch_bounds__(256) void fx_to_float_kernel(long long* __restrict__ fx, float* __restrict__ out, int64_t n, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long q = fx[i];
  if (q != 0) {
    const float v = (float)((double)q * (1.0 / FX_SCALE));
    out[i] = accumulate ? out[i] + v : v;
    fx[i] = 0;
  } else if (!accumulate) {
    out[i] = 0.f;
  }
}
}  // namespace snerf

extern "C" int snerf_fx_to_float(int64_t* fx, float* out, int64_t n, int32_t accumulate, snerf_stream_t stream) {
  SNERF_REQUIRE(n >= 0, "fx_to_float: n=%lld", (long long)n);
  if (n == 0) return 0;
  SNERF_REQUIRE(fx && out, "fx_to_float: null buffer");
  const int64_t blocks = (n + 255) / 256;
  SNERF_REQUIRE(blocks < (1LL << 31), "fx_to_float: n too large for one launch");
  hipLaunchKernelGGL(snerf::fx_to_float_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<long long*>(fx), out, n,
                     accumulate);
  int rc = snerf::check_hip(hipGetLastError(), "fx_to_float");
  return rc;
}

extern "C" int snerf_abi_version(void) { return SNERF_ABI_VERSION; }
extern "C" const char* snerf_last_error(void) { return snerf::g_err; }
extern "C" const char* snerf_target_arch(void) { return "gfx950"; }

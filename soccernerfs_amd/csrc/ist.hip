// Ray importance sampling: IST (temporal-difference) weight maps and the device-side weighted pixel draw.
//
// Reference: DynamicDataset.compute_ist (NS/data/datasets/dynamic_dataset.py:328-470) -- a Python double loop over images on
// the host -- and DynamicBasedPixelSampler.sample_method (NS/data/pixel_samplers.py:340-426) -- ~62 host iterations per
// step of torch.multinomial + torch.nonzero (device syncs) over 518 400-element maps.  Here the maps come from one kernel
// (one lane per pixel, neighbour list in CSR form) and the per-step draw is one kernel: inverse-CDF sampling (binary
// search in a per-image prefix sum built once per cache refresh).
#include "common.hpp"

#include <hip/hip_fp16.h>

namespace snerf {

template <typename T> __device__ __forceinline__ float to_unit(T v);
template <> __device__ __forceinline__ float to_unit<uint8_t>(uint8_t v) { return (float)v / 255.f; }  // base_dataset.py:82
template <> __device__ __forceinline__ float to_unit<float>(float v) { return v; }

// images [M,H,W,3]; nbr_off [M+1], nbr_idx [nnz]: same-camera images with 0.01 < |dt| <= ist_range (:426-429)
template <typename T>
__global__ void ist_kernel(const T* __restrict__ images, const int32_t* __restrict__ nbr_off, const int32_t* __restrict__ nbr_idx, int64_t HW,
                           float alpha, __half* __restrict__ out) {
  const int i = blockIdx.y;
  const int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= HW) return;
  const int b = nbr_off[i], e = nbr_off[i + 1];
  float w;
  if (b == e) {
    w = 1.f;  // no neighbour: uniform map (:432-434)
  } else {
    const T* cur = images + ((int64_t)i * HW + px) * 3;
    const float c0 = to_unit<T>(cur[0]), c1 = to_unit<T>(cur[1]), c2 = to_unit<T>(cur[2]);
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (int k = b; k < e; ++k) {
      const T* o = images + ((int64_t)nbr_idx[k] * HW + px) * 3;
      m0 = fmaxf(m0, fabsf(c0 - to_unit<T>(o[0])));
      m1 = fmaxf(m1, fabsf(c1 - to_unit<T>(o[1])));
      m2 = fmaxf(m2, fabsf(c2 - to_unit<T>(o[2])));
    }
    const float mean = ((m0 + m1) + m2) / 3.f;  // max_diff.mean(dim=2) (:444)
    w = mean > alpha ? mean : 0.f;                // (:446)
  }
  out[(int64_t)i * HW + px] = __float2half(w);    // (:462)
}

// ---- ISG (global-median) maps: DynamicDataset.compute_isg (dynamic_dataset.py:215-326) ----
// Step 1, per camera: median over the camera's images of every pixel channel.  torch.median(dim=0) returns the LOWER median, an
// element of the input (dynamic_dataset.py:292): rank (F - 1) / 2 in sorted order.  One lane per (camera, pixel, channel) selects
// it by rank counting over the <= MAXF values it holds in a private array.
constexpr int ISG_MAXF = 128;

template <typename T>
__global__ void isg_median_kernel(const T* __restrict__ images, const int32_t* __restrict__ cam_off, const int32_t* __restrict__ cam_img, int64_t HW3,
                                  T* __restrict__ medians) {
  const int c = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // pixel-channel
  if (e >= HW3) return;
  const int b = cam_off[c], F = cam_off[c + 1] - b;
  T vals[ISG_MAXF];
  for (int k = 0; k < F; ++k) vals[k] = images[(int64_t)cam_img[b + k] * HW3 + e];
  const int target = (F - 1) / 2;
  T med = vals[0];
  for (int k = 0; k < F; ++k) {
    const T v = vals[k];
    int lt = 0, le = 0;
    for (int j = 0; j < F; ++j) { lt += vals[j] < v; le += vals[j] <= v; }
    if (lt <= target && target < le) { med = v; break; }
  }
  medians[(int64_t)c * HW3 + e] = med;
}

// Step 2, per image: psi = mean_c r^2 / (r^2 + gamma^2), r = image - median of its camera (:299-301), cast to fp16 (:317)
template <typename T>
__global__ void isg_weights_kernel(const T* __restrict__ images, const T* __restrict__ medians, const int32_t* __restrict__ img_cam, int64_t HW,
                                   float gamma2, __half* __restrict__ out) {
  const int i = blockIdx.y;
  const int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= HW) return;
  const T* cur = images + ((int64_t)i * HW + px) * 3;
  const T* med = medians + ((int64_t)img_cam[i] * HW + px) * 3;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float r = to_unit<T>(cur[k]) - to_unit<T>(med[k]);
    const float sq = r * r;
    acc += sq / (sq + gamma2);
  }
  out[(int64_t)i * HW + px] = __float2half((1.0f / 3) * acc);
}

// Weighted pixel draws of DynamicBasedPixelSampler.sample_method (NS/data/pixel_samplers.py:369-411).  Slot j of the chosen images
// receives draws [j * per_image, min((j + 1) * per_image, n)); a draw picks a pixel with probability proportional to its weight by
// inverting the image's inclusive prefix sums cdf[image][0..HW) at u * (remaining mass).
// torch.multinomial's replacement flag, as the reference sets it (:400-402): WITHOUT replacement when the map has at least as many
// non-zero pixels as the slot draws -- a drawn pixel's weight is then removed from the distribution for the slot's later draws
// (sequential removal: the distribution torch.multinomial(replacement=False) samples from; which uniform lands where is RNG-specific,
// so the draws u are explicit inputs and the oracle restates this loop) -- otherwise with replacement.
// One wavefront per slot (~62 slots x 10 draws per step); the removed pixels of a slot sit in LDS and every probe of the search
// subtracts their mass.  The order-sensitive arithmetic is in double, where it is EXACT (weights are fp32 differences of the
// prefix sums: sums of <= 1024 of them fit 53 bits), so kernel and oracle agree bit for bit whatever the summation order.
// The removed list lives in dynamic LDS, sized by the launcher for the largest slot (min(per_image, n) entries x 12 B): the preset draws
// 10 per slot; few images with a large batch (per_image = 10 * ceil(0.15 R / M), pixel_samplers.py:369) need thousands -- up to
// IST_MAX_DRAWS_PER_SLOT fit a CU's 160 KB.  The probe loop is O(draws^2) per slot: beyond ~1000 draws per slot it costs milliseconds.
constexpr int IST_MAX_DRAWS_PER_SLOT = 13000;
__global__ __launch_bounds__(64) void ist_sample_kernel(const float* __restrict__ cdf, int64_t HW, int W, const int64_t* __restrict__ chosen,
                                                        const int32_t* __restrict__ nnz, int per_image, const float* __restrict__ u, int n,
                                                        int64_t* __restrict__ indices, int list_cap) {
  extern __shared__ __align__(16) unsigned char ist_lds[];
  double* rem_w = reinterpret_cast<double*>(ist_lds);                  // [list_cap]
  int32_t* rem_idx = reinterpret_cast<int32_t*>(rem_w + list_cap);     // [list_cap]
  const int slot = blockIdx.x, lane = threadIdx.x;
  const int d0 = slot * per_image;
  if (d0 >= n) return;
  const int cnt = (n - d0) < per_image ? (n - d0) : per_image;
  const int64_t img = chosen[slot];
  const float* c = cdf + img * HW;
  const bool without = nnz != nullptr && nnz[img] >= cnt;
  double removed = 0.0;
  const double total = (double)c[HW - 1];
  for (int k = 0; k < cnt; ++k) {
    const double target = (double)u[d0 + k] * (total - removed);
    // first index whose (adjusted) prefix sum exceeds target: a 64-ary search, lane l probes the end of the l-th slice of [lo, hi] -- four
    // rounds of independent loads for a 960 x 540 map instead of twenty dependent ones (a binary search by one lane took 0.3 ms per step:
    // every probe is a full HBM round trip)
    int64_t lo = 0, hi = HW - 1;
    while (lo < hi) {  // wave-uniform
      const int64_t span = hi - lo + 1;
      const int64_t step = (span + 63) >> 6;
      int64_t probe = lo + (int64_t)(lane + 1) * step - 1;
      if (probe > hi) probe = hi;
      double v = (double)c[probe];
      if (without)
        for (int r = 0; r < k; ++r)
          if (rem_idx[r] <= probe) v -= rem_w[r];
      const unsigned long long hit = __ballot(v > target);  // monotone in the lane index; the last slice always ends at hi
      if (hit == 0ull) { lo = hi; break; }                  // target >= every prefix (rounding): the last pixel
      const int first = __ffsll((long long)hit) - 1;
      const int64_t new_hi = lo + (int64_t)(first + 1) * step - 1;
      const int64_t new_lo = first == 0 ? lo : lo + (int64_t)first * step;
      hi = new_hi < hi ? new_hi : hi;
      lo = new_lo;
    }
    if (without) {
      const double w = (double)c[lo] - (lo > 0 ? (double)c[lo - 1] : 0.0);
      if (lane == 0) { rem_idx[k] = (int32_t)lo; rem_w[k] = w; }
      removed += w;
      __syncthreads();  // one wave per workgroup: orders the LDS write before the next draw's reads
    }
    if (lane == 0) {
      indices[(int64_t)(d0 + k) * 3 + 0] = img;
      indices[(int64_t)(d0 + k) * 3 + 1] = lo / W;
      indices[(int64_t)(d0 + k) * 3 + 2] = lo % W;
    }
  }
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_ist_maps(const void* images, int32_t image_dtype, int32_t M, int32_t H, int32_t W, const int32_t* nbr_off,
                              const int32_t* nbr_idx, float alpha, void* out_f16, snerf_stream_t stream) {
  SNERF_REQUIRE(M >= 0 && H >= 1 && W >= 1, "ist_maps: M=%d H=%d W=%d", M, H, W);
  SNERF_REQUIRE(image_dtype == 0 || image_dtype == 1, "ist_maps: image_dtype=%d (0 = uint8, 1 = float32)", image_dtype);
  if (M == 0) return 0;
  SNERF_REQUIRE(images && nbr_off && out_f16, "ist_maps: null buffer");
  const int64_t HW = (int64_t)H * W;
  dim3 grid((unsigned)ceil_div(HW, 256), (unsigned)M);
  if (image_dtype == 0)
    hipLaunchKernelGGL(ist_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)images, nbr_off, nbr_idx, HW, alpha, (__half*)out_f16);
  else
    hipLaunchKernelGGL(ist_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)images, nbr_off, nbr_idx, HW, alpha, (__half*)out_f16);
  SNERF_LAUNCH_CHECK("ist_maps");
  return 0;
}

extern "C" int snerf_isg_maps(const void* images, int32_t image_dtype, int32_t M, int32_t H, int32_t W, int32_t n_cams, const int32_t* cam_off,
                              const int32_t* cam_img, const int32_t* img_cam, int32_t max_frames, float gamma, void* medians, void* out_f16,
                              snerf_stream_t stream) {
  SNERF_REQUIRE(M >= 0 && H >= 1 && W >= 1 && n_cams >= 0, "isg_maps: M=%d H=%d W=%d n_cams=%d", M, H, W, n_cams);
  SNERF_REQUIRE(image_dtype == 0 || image_dtype == 1, "isg_maps: image_dtype=%d (0 = uint8, 1 = float32)", image_dtype);
  SNERF_REQUIRE(max_frames >= 1 && max_frames <= ISG_MAXF, "isg_maps: %d images of one camera (at most %d)", max_frames, ISG_MAXF);
  if (M == 0 || n_cams == 0) return 0;
  SNERF_REQUIRE(images && cam_off && cam_img && img_cam && medians && out_f16, "isg_maps: null buffer");
  const int64_t HW = (int64_t)H * W;
  hipStream_t st = (hipStream_t)stream;
  dim3 g1((unsigned)ceil_div(HW * 3, 256), (unsigned)n_cams), g2((unsigned)ceil_div(HW, 256), (unsigned)M);
  if (image_dtype == 0) {
    hipLaunchKernelGGL(isg_median_kernel<uint8_t>, g1, dim3(256), 0, st, (const uint8_t*)images, cam_off, cam_img, HW * 3, (uint8_t*)medians);
    hipLaunchKernelGGL(isg_weights_kernel<uint8_t>, g2, dim3(256), 0, st, (const uint8_t*)images, (const uint8_t*)medians, img_cam, HW, gamma * gamma,
                       (__half*)out_f16);
  } else {
    hipLaunchKernelGGL(isg_median_kernel<float>, g1, dim3(256), 0, st, (const float*)images, cam_off, cam_img, HW * 3, (float*)medians);
    hipLaunchKernelGGL(isg_weights_kernel<float>, g2, dim3(256), 0, st, (const float*)images, (const float*)medians, img_cam, HW, gamma * gamma,
                       (__half*)out_f16);
  }
  SNERF_LAUNCH_CHECK("isg_maps");
  return 0;
}

extern "C" int snerf_ist_sample(const float* cdf, int32_t H, int32_t W, const int64_t* chosen_images, const int32_t* nonzero_counts, int32_t per_image,
                                const float* u, int32_t n, int64_t* indices, snerf_stream_t stream) {
  SNERF_REQUIRE(n >= 0 && per_image >= 1 && H >= 1 && W >= 1, "ist_sample: n=%d per_image=%d", n, per_image);
  if (n == 0) return 0;
  const int list_cap = per_image < n ? per_image : n;  // the largest slot's draws: what the without-replacement list must hold
  SNERF_REQUIRE(list_cap <= IST_MAX_DRAWS_PER_SLOT, "ist_sample: %d draws in one image slot (at most %d: the removed-pixel list lives in LDS)", list_cap,
                IST_MAX_DRAWS_PER_SLOT);
  SNERF_REQUIRE(cdf && chosen_images && u && indices, "ist_sample: null buffer");
  SNERF_REQUIRE((int64_t)H * W < (1LL << 31), "ist_sample: image too large");
  const int slots = ceil_div(n, per_image);
  const size_t lds = (size_t)list_cap * (sizeof(double) + sizeof(int32_t));
  if (lds > 48 * 1024) {
    int rc = check_hip(hipFuncSetAttribute((const void*)ist_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "ist_sample LDS size");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ist_sample_kernel, dim3(slots), dim3(64), lds, (hipStream_t)stream, cdf, (int64_t)H * W, W, chosen_images, nonzero_counts,
                     per_image, u, n, indices, list_cap);
  SNERF_LAUNCH_CHECK("ist_sample");
  return 0;
}

// Ray generation (pinhole, no distortion, camera optimiser off) fused with the AABB collider.
//
// Reference: RayGenerator.forward (NS/model_components/ray_generators.py:41-59) ->
// Cameras._generate_rays_from_coords (NS/cameras/cameras.py:505-741; the perspective slice :596-633,:663-670,
// :704-741 -- ~40 small ATen kernels incl. boolean-mask scatters) and AABBBoxCollider._intersect_with_aabb
// (NS/model_components/scene_colliders.py:59-95).  One lane per ray; the per-camera table (fx,fy,cx,cy,c2w,time)
// is a few KB and stays in L1/L2.
#include "common.hpp"

#pragma clang fp contract(off)

namespace snerf {

struct RaygenArgs {
  const int64_t* indices;  // [R,3] (camera, row, col)
  const float* fx; const float* fy; const float* cx; const float* cy;  // [M]
  const float* c2w;        // [M,3,4]
  const float* cam_times;  // [M] or null
  int R;
  float* origins; float* dirs; float* pixel_area; float* dir_norm; float* times;  // [R,3],[R,3],[R],[R],[R]
  // collider
  int collide; int training; float near_plane;
  float aabb_min[3], aabb_max[3];
  float* nears; float* fars;  // [R]
};

__device__ __forceinline__ void cam_to_world(const float* rot /*3x4 row-major*/, float x, float y, float z, float out[3], float& norm) {
  float v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = (x * rot[i * 4 + 0] + y * rot[i * 4 + 1]) + z * rot[i * 4 + 2];  // sum over the last axis (cameras.py:712-714)
  norm = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  // normalize_with_norm (NS/cameras/camera_utils.py:240-252): norm = max(|v|, 4*eps_f64); returns x / norm and norm
  norm = fmaxf(norm, 8.8817841970012523e-16f);
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = v[i] / norm;
}

__global__ void raygen_kernel(RaygenArgs a) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.R) return;
  const int64_t c = a.indices[(int64_t)r * 3], yi = a.indices[(int64_t)r * 3 + 1], xi = a.indices[(int64_t)r * 3 + 2];
  const float y = (float)yi + 0.5f, x = (float)xi + 0.5f;  // image_coords = pixel index + 0.5 (cameras.py:318-319)
  const float fx = a.fx[c], fy = a.fy[c], cx = a.cx[c], cy = a.cy[c];
  const float* m = a.c2w + c * 12;
  float d0[3], dx[3], dy[3], n0, nx, ny;
  cam_to_world(m, (x - cx) / fx, -(y - cy) / fy, -1.f, d0, n0);
  cam_to_world(m, ((x + 1.f) - cx) / fx, -(y - cy) / fy, -1.f, dx, nx);
  cam_to_world(m, (x - cx) / fx, -((y + 1.f) - cy) / fy, -1.f, dy, ny);
  float ax = sqrtf(((d0[0] - dx[0]) * (d0[0] - dx[0]) + (d0[1] - dx[1]) * (d0[1] - dx[1])) + (d0[2] - dx[2]) * (d0[2] - dx[2]));
  float ay = sqrtf(((d0[0] - dy[0]) * (d0[0] - dy[0]) + (d0[1] - dy[1]) * (d0[1] - dy[1])) + (d0[2] - dy[2]) * (d0[2] - dy[2]));
  float o[3] = {m[3], m[7], m[11]};
#pragma unroll
  for (int k = 0; k < 3; ++k) { a.origins[(int64_t)r * 3 + k] = o[k]; a.dirs[(int64_t)r * 3 + k] = d0[k]; }
  a.pixel_area[r] = ax * ay;
  a.dir_norm[r] = n0;
  if (a.times) a.times[r] = a.cam_times ? a.cam_times[c] : 0.f;
  if (a.collide) {
    float tn = -INFINITY, tf = INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float inv = 1.f / (d0[k] + 1e-6f);  // scene_colliders.py:71
      float t1 = (a.aabb_min[k] - o[k]) * inv, t2 = (a.aabb_max[k] - o[k]) * inv;
      tn = fmaxf(tn, fminf(t1, t2));
      tf = fminf(tf, fmaxf(t1, t2));
    }
    float np = a.training ? a.near_plane : 0.f;
    tn = fmaxf(tn, np);
    tf = fmaxf(tf, tn + 1e-6f);
    a.nears[r] = tn; a.fars[r] = tf;
  }
}

// PixelSampler.sample_method (NS/data/pixel_samplers.py:74-77: floor(rand(R,3) * [M,H,W]).long()) fused with the image gather of
// collate_image_dataset_batch (:111-123): one lane per ray instead of seven elementwise / index launches
__global__ void sample_pixels_kernel(const float* __restrict__ u, int R, int M, int H, int W, const uint8_t* __restrict__ images,
                                     int64_t* __restrict__ indices, float* __restrict__ target) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  // rand < 1 so the products stay below M, H, W except for the fp32 rounding of u * n at u -> 1: clamp like an index would fault otherwise
  int64_t c = (int64_t)floorf(u[(int64_t)r * 3] * (float)M), y = (int64_t)floorf(u[(int64_t)r * 3 + 1] * (float)H),
          x = (int64_t)floorf(u[(int64_t)r * 3 + 2] * (float)W);
  c = c < M ? c : M - 1; y = y < H ? y : H - 1; x = x < W ? x : W - 1;
  indices[(int64_t)r * 3] = c; indices[(int64_t)r * 3 + 1] = y; indices[(int64_t)r * 3 + 2] = x;
  if (images) {
    const uint8_t* px = images + (((int64_t)c * H + y) * W + x) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) target[(int64_t)r * 3 + k] = (float)px[k] / 255.0f;  // uint8 -> float32 / 255 (NS/data/datasets/base_dataset.py:82)
  }
}

// The ray batch in order of a per-image key (the frame time): a batch is a set -- losses are means over it, the per-ray draws are i.i.d. -- so
// its order is free, and with equal-time rays next to each other every gather of a time plane (3 of the 6 planes of each scale; the
// temporal hash grid's rows likewise) finds the same two texel rows of its plane for the whole group instead of rows all over the plane:
// fused field forward 0.378 -> 0.309 ms at the k-planes preset (profiles/r03_kernels.md section 11).  One workgroup, bitonic sort in LDS of
// the words key << 14 | ray: unique words, so the order is a pure function of the batch (stable for equal keys).  256 threads on purpose: the
// kernel starts while the previous step's optimiser sweep fills the GPU, and a 1024-thread workgroup waited for sixteen free wave slots on one CU
// until the sweep had drained -- the whole head of the step behind it (measured: +0.1 ms per step).
__global__ __launch_bounds__(256) void sort_rays_kernel(const int64_t* __restrict__ idx_in, const int32_t* __restrict__ image_key, int R, int n_pow2,
                                                        const float* __restrict__ aux_in, int aux_cols, int64_t* __restrict__ idx_out,
                                                        float* __restrict__ aux_out) {
  extern __shared__ uint32_t s_w[];
  for (int i = threadIdx.x; i < n_pow2; i += blockDim.x)
    s_w[i] = i < R ? (((uint32_t)image_key[idx_in[(int64_t)i * 3]] << 14) | (uint32_t)i) : 0xffffffffu;
  __syncthreads();
  for (int k = 2; k <= n_pow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < n_pow2 / 2; t += blockDim.x) {
        const int i = ((t / j) * 2 * j) + (t % j), l = i + j;  // j is a power of two: shifts and masks
        const bool up = (i & k) == 0;
        const uint32_t a = s_w[i], b = s_w[l];
        if ((a > b) == up) { s_w[i] = b; s_w[l] = a; }
      }
      __syncthreads();
    }
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const int64_t src = s_w[r] & 0x3fffu;
#pragma unroll
    for (int c = 0; c < 3; ++c) idx_out[(int64_t)r * 3 + c] = idx_in[src * 3 + c];
    for (int c = 0; c < aux_cols; ++c) aux_out[(int64_t)r * aux_cols + c] = aux_in[src * aux_cols + c];
  }
}

__global__ void aabb_kernel(const float* __restrict__ o, const float* __restrict__ d, int R, float near_plane, int training, const float* amin3,
                            float* __restrict__ nears, float* __restrict__ fars, float a0, float a1, float a2, float b0, float b1, float b2) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float mn[3] = {a0, a1, a2}, mx[3] = {b0, b1, b2};
  float tn = -INFINITY, tf = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float inv = 1.f / (d[(int64_t)r * 3 + k] + 1e-6f);
    float t1 = (mn[k] - o[(int64_t)r * 3 + k]) * inv, t2 = (mx[k] - o[(int64_t)r * 3 + k]) * inv;
    tn = fmaxf(tn, fminf(t1, t2));
    tf = fminf(tf, fmaxf(t1, t2));
  }
  tn = fmaxf(tn, training ? near_plane : 0.f);
  tf = fmaxf(tf, tn + 1e-6f);
  nears[r] = tn; fars[r] = tf;
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_raygen(const snerf_raygen_args* p, snerf_stream_t stream) {
  SNERF_REQUIRE(p, "raygen: null args");
  SNERF_REQUIRE(p->R >= 0, "raygen: R=%d", p->R);
  if (p->R == 0) return 0;
  SNERF_REQUIRE(p->indices && p->fx && p->fy && p->cx && p->cy && p->c2w, "raygen: null camera/index buffer");
  SNERF_REQUIRE(p->origins && p->dirs && p->pixel_area && p->dir_norm, "raygen: null output buffer");
  SNERF_REQUIRE(!p->collide || (p->nears && p->fars), "raygen: collide set but nears/fars null");
  RaygenArgs a;
  a.indices = p->indices; a.fx = p->fx; a.fy = p->fy; a.cx = p->cx; a.cy = p->cy; a.c2w = p->c2w; a.cam_times = p->cam_times; a.R = p->R;
  a.origins = p->origins; a.dirs = p->dirs; a.pixel_area = p->pixel_area; a.dir_norm = p->dir_norm; a.times = p->times;
  a.collide = p->collide; a.training = p->training; a.near_plane = p->near_plane;
  for (int k = 0; k < 3; ++k) { a.aabb_min[k] = p->aabb_min[k]; a.aabb_max[k] = p->aabb_max[k]; }
  a.nears = p->nears; a.fars = p->fars;
  hipLaunchKernelGGL(raygen_kernel, dim3(ceil_div(p->R, 256)), dim3(256), 0, (hipStream_t)stream, a);
  SNERF_LAUNCH_CHECK("raygen");
  return 0;
}

extern "C" int snerf_sample_pixels_uniform(const float* u, int32_t R, int32_t M, int32_t H, int32_t W, const uint8_t* images, int64_t* indices,
                                           float* target, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && M >= 1 && H >= 1 && W >= 1, "sample_pixels_uniform: R=%d M=%d H=%d W=%d", R, M, H, W);
  if (R == 0) return 0;
  SNERF_REQUIRE(u && indices && (!images || target), "sample_pixels_uniform: null buffer");
  hipLaunchKernelGGL(sample_pixels_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, u, R, M, H, W, images, indices, target);
  SNERF_LAUNCH_CHECK("sample_pixels_uniform");
  return 0;
}

extern "C" int snerf_sort_rays_by_key(const int64_t* indices_in, const int32_t* image_key, int32_t n_keys, int32_t R, const float* aux_in,
                                      int32_t aux_cols, int64_t* indices_out, float* aux_out, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && R <= 16384 && n_keys >= 1 && n_keys <= (1 << 18) && aux_cols >= 0,
                "sort_rays_by_key: R=%d (<= 16384) n_keys=%d (<= 262144) aux_cols=%d", R, n_keys, aux_cols);
  if (R == 0) return 0;
  SNERF_REQUIRE(indices_in && image_key && indices_out && indices_in != indices_out && (aux_cols == 0 || (aux_in && aux_out && aux_in != aux_out)),
                "sort_rays_by_key: null or aliased buffer (the sort is out of place)");
  int n = 2;
  while (n < R) n <<= 1;
  hipLaunchKernelGGL(sort_rays_kernel, dim3(1), dim3(256), (size_t)n * sizeof(uint32_t), (hipStream_t)stream, indices_in, image_key, R, n, aux_in, aux_cols,
                     indices_out, aux_out);
  SNERF_LAUNCH_CHECK("sort_rays_by_key");
  return 0;
}

extern "C" int snerf_aabb_collide(const float* origins, const float* dirs, int32_t R, const float* aabb6, float near_plane, int32_t training,
                                  float* nears, float* fars, snerf_stream_t stream) {
  SNERF_REQUIRE(R >= 0 && aabb6, "aabb_collide: bad arguments");
  if (R == 0) return 0;
  SNERF_REQUIRE(origins && dirs && nears && fars, "aabb_collide: null buffer");
  hipLaunchKernelGGL(aabb_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, origins, dirs, R, near_plane, training, nullptr, nears,
                     fars, aabb6[0], aabb6[1], aabb6[2], aabb6[3], aabb6[4], aabb6[5]);
  SNERF_LAUNCH_CHECK("aabb_collide");
  return 0;
}

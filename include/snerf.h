/*
 * libsnerf -- C ABI of the MI355X-native dynamic-NeRF hot path (K-Planes / NeRFPlayer-nerfacto).
 *
 * Drop-in boundary.  Conventions, all entry points:
 *   - plain C: raw DEVICE pointers + sizes, no torch types; the CALLER owns and allocates every
 *     buffer, outputs and gradient accumulators included (as the reference's own native FFI does:
 *     NS/field_components/temporal_grid.py:82-87,126-131); kernels never allocate;
 *   - tensors are contiguous, fp32 unless stated, 16-byte aligned;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls are asynchronous, re-entrant
 *     and keep no global mutable state;
 *   - return 0 on success, <0 for an argument error, >0 = hipError_t; `snerf_last_error()` returns
 *     a thread-local message.  Nothing throws across the ABI (the reference throws C++ exceptions
 *     through pybind: NS/field_components/cuda/csrc/temporal_gridencoder.cu:435,473,596-608).
 *
 * Reference paths are abbreviated NS/ = nerfstudio/nerfstudio/ inside the iSach/SoccerNeRFs tree.
 */
#ifndef SNERF_H
#define SNERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* snerf_stream_t; /* hipStream_t */

#define SNERF_OK 0
#define SNERF_ERR_ARG (-1)
#define SNERF_ERR_UNSUPPORTED (-2)

#define SNERF_MAX_SCALES 8
#define SNERF_ABI_VERSION 1

/* Library identity / diagnostics. */
int snerf_abi_version(void);
const char* snerf_last_error(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* snerf_target_arch(void);

/* ------------------------------------------------------------------------------------------------
 * K-Planes plane set.  Planes are stored CHANNEL-LAST [H][W][C] (one texel = C contiguous floats),
 * all planes of a field in ONE flat buffer; plane p of scale s starts at float offset off[s][p].
 * Plane p pairs coordinates (a,b) in the order XY XZ XT YZ YT ZT (NS/fields/kplanes_field.py:61-65)
 * and has W = res[s][a], H = res[s][b] (reference shape [1,C,reso[b],reso[a]], :67).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t n_scales;                  /* 1..SNERF_MAX_SCALES */
  int32_t C;                         /* features per texel: 8, 16 or 32 */
  int32_t concat;                    /* 1: concatenate scales (width C*n_scales); 0: sum (width C) */
  int32_t n_coords;                  /* 4 (x,y,z,t); 3 = static scene (planes XY XZ YZ only) */
  int32_t res[SNERF_MAX_SCALES][4];  /* per-scale resolution of axes x,y,z,t */
  int64_t off[SNERF_MAX_SCALES][6];  /* float offsets into the flat plane buffer */
} snerf_kplanes_desc;

/* Where sample coordinates come from.
 * mode 0: explicit points pts[N,4] (already in grid_sample's [-1,1] convention) -- the signature of
 *         interpolate_kplanes(pts, ...) (NS/fields/kplanes_field.py:77-126).
 * mode 1: derived per sample from rays, fusing Frustums.get_positions (NS/cameras/rays.py:54),
 *         SceneBox.get_normalized_positions (NS/data/scene_box.py:55-65) and the time rescale
 *         (kplanes_field.py:283-291): pos = o + d*(e[s]+e[s+1])/2; p = (pos-aabb_min)/(aabb_max-aabb_min);
 *         if rescale: p = 2p-1 (main field) else p stays in [0,1] (proposal field quirk, :440);
 *         t = 2*time-1.  N = R*S, sample n = ray n/S, bin n%S. */
typedef struct {
  int32_t mode;
  int32_t S;             /* samples per ray (mode 1) */
  int32_t rescale;       /* mode 1: 1 = map [0,1] -> [-1,1] */
  int32_t _pad;
  const float* pts;      /* mode 0: [N,4] */
  const float* origins;  /* mode 1: [R,3] */
  const float* dirs;     /* mode 1: [R,3] */
  const float* times;    /* mode 1: [R] */
  const float* ebins;    /* mode 1: [R,S+1] euclidean bin edges */
  float aabb_min[3];
  float aabb_max[3];
} snerf_coords;

/* Replaces interpolate_kplanes + grid_sample_wrapper (NS/fields/kplanes_field.py:77-126,
 * NS/utils/interpolation.py:5-33): bilinear, align_corners=True, border padding, Hadamard product over
 * the 6 planes, concat/sum over scales.  out: [N, C*n_scales] (concat) or [N, C]. */
int snerf_kplanes_gather_fwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords,
                             int64_t N, float* out, snerf_stream_t stream);

/* Backward of the above w.r.t. the planes (coordinates carry no gradient on this path: camera
 * optimiser off).  ACCUMULATES (atomic fp32 adds) into grad_planes, which has the
 * layout of `planes`; the caller zeroes it when needed.  Replaces ATen grid_sampler_2d_backward x 6
 * per scale + the product rule. */
int snerf_kplanes_gather_bwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords,
                             int64_t N, const float* grad_out, float* grad_planes, snerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SNERF_H */

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

/* ------------------------------------------------------------------------------------------------
 * Per-ray sampling ops.  One wavefront per ray; S <= 320 samples per ray.
 * "sbins" = bin edges in the normalised spacing domain [0,1]; "ebins" = the same edges in euclidean
 * distance along the ray (what RaySamples.frustums.starts/ends hold).  kind: 0 = uniform spacing
 * (UniformSampler), 1 = uniform/linear-disparity piecewise (UniformLinDispPiecewiseSampler).
 * ------------------------------------------------------------------------------------------------ */

/* SpacedSampler.generate_ray_samples (NS/model_components/ray_samplers.py:79-126).
 * t_rand: NULL (eval, no jitter) or uniform draws [R, rand_cols], rand_cols = S+1, or 1 (single_jitter).
 * Outputs sbins, ebins: [R, S+1]. */
int snerf_spaced_bins(const float* nears, const float* fars, const float* t_rand, int32_t rand_cols, int32_t R, int32_t S,
                      int32_t kind, float* sbins, float* ebins, snerf_stream_t stream);

/* RaySamples.get_weights (NS/cameras/rays.py:127-149): density [R,S], ebins [R,S+1] -> weights [R,S].
 * deltas = ebins[:,1:] - ebins[:,:-1]. */
int snerf_weights_fwd(const float* density, const float* ebins, int32_t R, int32_t S, float* weights, snerf_stream_t stream);

/* Backward of get_weights w.r.t. density.  accumulate != 0: grad_density += ...; else overwritten. */
int snerf_weights_bwd(const float* density, const float* ebins, const float* grad_weights, int32_t R, int32_t S,
                      float* grad_density, int32_t accumulate, snerf_stream_t stream);

/* PDFSampler.generate_ray_samples with include_original=False (ray_samplers.py:274-369) preceded by the
 * weight annealing of ProposalNetworkSampler (:584), optionally fused behind get_weights.
 *   weights source: `density`+`ebins_prev` (weights are computed in-kernel and, if weights_out != NULL,
 *                   stored) or `weights_in` [R,S_prev].
 *   u_mode 0: `u_or_rand` is u [R,S+1] itself (parity mode: every random draw is an input);
 *          1: `u_or_rand` holds uniform draws [R,rand_cols] (rand_cols = S+1 or 1); u = linspace + rand/(S+1);
 *          2: eval, u = bin centres.
 * The CDF is accumulated strictly left to right in fp32, so `inds_out` (int64, searchsorted(cdf,u,"right"))
 * is bit-exact against the CPU oracle.  Outputs: sbins_out, ebins_out [R,S+1]; inds_out optional. */
typedef struct {
  const float* density;     /* [R,S_prev] or NULL */
  const float* weights_in;  /* [R,S_prev] or NULL */
  const float* ebins_prev;  /* [R,S_prev+1] (with density) */
  float* weights_out;       /* [R,S_prev] or NULL */
  const float* sbins_prev;  /* [R,S_prev+1] */
  const float* u_or_rand;
  const float* nears;       /* [R] */
  const float* fars;        /* [R] */
  float* sbins_out;         /* [R,S+1] */
  float* ebins_out;         /* [R,S+1] */
  int64_t* inds_out;        /* [R,S+1] or NULL */
  int32_t R, S_prev, S;
  int32_t u_mode, rand_cols, kind;
  float anneal;             /* weights ** anneal; 1.0 = off */
  float histogram_padding;  /* 0.01 */
  float eps;                /* 1e-5 */
} snerf_resample_args;
int snerf_pdf_resample(const snerf_resample_args* args, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Tiny bias-free MLP = the reference's tcnn.Network(FullyFusedMLP) call sites
 * (NS/fields/kplanes_field.py:249-273,397-407; NS/fields/nerfplayer_nerfacto_field.py:94-104,238-248,301-311).
 * d_in -> hidden x n_hidden (ReLU/None) -> d_out (None/Sigmoid); d_out <= 16; hidden in {16,64,128}.
 * Parameters: ONE flat fp32 buffer, layer l stored row-major [d_l][d_{l+1}] (input-major), layers back to back.
 * Computed in exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t d_in, hidden, n_hidden, d_out;
  int32_t hidden_act; /* 0 none, 1 ReLU */
  int32_t out_act;    /* 0 none, 1 Sigmoid */
} snerf_mlp_desc;

int64_t snerf_mlp_param_count(const snerf_mlp_desc* desc);

/* Y[N,d_out] (row stride ldy) = MLP(X[N,d_in] (row stride ldx)).  If aux_out != NULL:
 * aux_out[n] = exp(raw output column aux_col) -- trunc_exp's forward (NS/field_components/activations.py:25-41),
 * i.e. the density head of sigma_net (kplanes_field.py:308-311). */
int snerf_mlp_fwd(const snerf_mlp_desc* desc, const float* W, const float* X, int32_t ldx, int64_t N, float* Y, int32_t ldy,
                  int32_t aux_col, float* aux_out, snerf_stream_t stream);

/* Backward.  Incoming gradients: gY [N,d_out] (row stride ldgy, may be NULL = zeros) w.r.t. the activated output,
 * and/or gaux [N] w.r.t. aux_out (applies trunc_exp's clamped backward g*exp(clamp(y,-15,15))).
 * Outputs: gX [N,d_in] (row stride ldgx; NULL = not needed, overwritten otherwise) and gW (flat, ACCUMULATED atomically;
 * NULL = not needed).  The forward is recomputed tile by tile; nothing is saved between fwd and bwd. */
int snerf_mlp_bwd(const snerf_mlp_desc* desc, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                  int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SNERF_H */

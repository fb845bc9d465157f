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
#define SNERF_ABI_VERSION 14

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

/* Deterministic accumulation.  Float atomics make a sum depend on the order in which wavefronts arrive (torch's own
 * grid_sampler_2d_backward has the same property on GPUs).  The _fx entry points accumulate into 64-bit fixed-point cells instead
 * (value * 2^50; integer adds are associative, so two runs give the same bits) and snerf_fx_to_float converts a cell buffer into float
 * gradients (out = or += cells * 2^-50) and clears it.  Same layout as the float gradient buffer. */
int snerf_kplanes_gather_bwd_fx(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords,
                                int64_t N, const float* grad_out, int64_t* grad_planes_fx, snerf_stream_t stream);
int snerf_fx_to_float(int64_t* fx, float* out, int64_t n, int32_t accumulate, snerf_stream_t stream);

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

/* Backward of get_weights w.r.t. density.  accumulate != 0: grad_density += ...; else overwritten.
 * Where autograd would produce a non-finite gradient (0 * inf: an overflowed density on a zero-width bin) the kernel emits 0 and, if
 * nonfinite_flag != NULL, stores 1 there -- the `nonfinite` field of the parameter group's snerf_adam_dyn, so that the optimiser step
 * can be skipped as the reference's GradScaler does (NS/engine/trainer.py:394-408). */
int snerf_weights_bwd(const float* density, const float* ebins, const float* grad_weights, int32_t R, int32_t S,
                      float* grad_density, int32_t accumulate, int32_t* nonfinite_flag, snerf_stream_t stream);

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
 * Computed in exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32), or with 16-bit operands (v_mfma_f32_16x16x32_bf16 / _f16) when
 * desc.operands = 1 / 2: parameters, inputs, outputs and gradients stay fp32 either way.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t d_in, hidden, n_hidden, d_out;
  int32_t hidden_act; /* 0 none, 1 ReLU */
  int32_t out_act;    /* 0 none, 1 Sigmoid */
  int32_t operands;   /* 0: fp32 MFMA operands (exact fp32, the parity path); 1: bf16, 2: fp16 MFMA operands with fp32 accumulation (2 is
                         exactly tcnn FullyFusedMLP's arithmetic class); 16-bit forms: one hidden layer of 64 / 128, d_in <= 160
                         (snerf_mlp_supported tells) */
} snerf_mlp_desc;

int64_t snerf_mlp_param_count(const snerf_mlp_desc* desc);
/* 1 if the fused kernels are instantiated for this shape (else the caller composes the net from library GEMMs), 0 otherwise. */
int snerf_mlp_supported(const snerf_mlp_desc* desc);
/* Single bias-free dense layer Y[N,M] = act(X[N,K] W[K,M]) and its backward (act: 0 none, 1 ReLU, 2 Sigmoid; K, M <= 4096 -- the kernels hold one
 * 128 x 128 block of W in LDS and the entry points tile wider layers over it): the building
 * block for tcnn.Network shapes outside the fused table (full NeRFPlayer: NS/fields/nerfplayer_field.py:231-316; the linear decoder's
 * 3 -> 128 x L -> 3F basis net and F -> 1 density layer, NS/fields/kplanes_field.py:219-246).  Backward works from the
 * stored OUTPUT Y: gX[N,K] is written (may be NULL), gW[K,M] is ACCUMULATED (atomic fp32; may be NULL). */
int snerf_dense_fwd(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, float* Y, int32_t ldy,
                    snerf_stream_t stream);
int snerf_dense_bwd(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, const float* Y, int32_t ldy,
                    const float* gY, int32_t ldgy, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream);
/* ABI 13: the same single layers with 16-bit MFMA operands (operands: 1 = bf16; fp32 accumulation; X, W, dY rounded while staged into LDS, outputs
 * and gradients fp32) -- csrc/dense_lp.hip.  K, M <= 128 (the full NeRFPlayer's nets: NS/fields/nerfplayer_field.py:231-316, which the reference runs
 * in tcnn's fp16).  gW (float atomics) or gW_fx (fixed-point cells), not both; either may be NULL.  snerf_dense_lp_supported: 1 iff built for the shape. */
int snerf_dense_lp_supported(int32_t K, int32_t M, int32_t operands);
int snerf_dense_fwd_lp(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, float* Y, int32_t ldy,
                       int32_t operands, snerf_stream_t stream);
int snerf_dense_bwd_lp(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, const float* Y, int32_t ldy,
                       const float* gY, int32_t ldgy, float* gX, int32_t ldgx, float* gW, int64_t* gW_fx, int32_t operands, snerf_stream_t stream);
/* ABI 13, deterministic mode: the same with gW accumulated into 2^50-scaled 64-bit cells [K,M] (see snerf_kplanes_gather_bwd_fx; gX is per-sample
 * work and identical between runs either way). */
int snerf_dense_bwd_fx(const float* W, int32_t K, int32_t M, int32_t act, const float* X, int32_t ldx, int64_t N, const float* Y, int32_t ldy,
                       const float* gY, int32_t ldgy, float* gX, int32_t ldgx, int64_t* gW_fx, snerf_stream_t stream);

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
/* ABI 12: snerf_mlp_bwd through the workgroup-tile kernels (a 64-sample tile shared by 4 / 8 waves that split the COLUMNS of every layer)
 * whatever the default for the shape is.  Since round 5 the 64-wide nets with 16-bit operands default to the wave-owns-rows kernel
 * (csrc/mlp_rows.hip: no barrier inside the loop); this entry keeps the older kernel callable for A-B runs and as a cross-check. */
int snerf_mlp_bwd_tile(const snerf_mlp_desc* desc, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                       int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream);
/* ABI 12: weight gradients through a WORKSPACE instead of straight into gW.  Every workgroup of a backward launch ends by adding its share
 * of the weight gradient to the same few thousand addresses; 256 same-address float atomics queue up at the memory side (~25 us at the
 * end of every launch, measured, whatever the element count).  With a workspace of 16 replicas of the flat gradient, workgroup b adds into
 * replica b % 16 -- 16 atomics per address -- and snerf_mlp_gw_reduce folds the replicas into gW (ACCUMULATED) and clears them, on whatever
 * stream and at whatever later point the caller chooses (the optimiser is the only reader of gW).
 *   workspace: snerf_mlp_gw_workspace_floats(desc) floats, caller-owned, ZERO before the first use; _reduce leaves it zero again.
 *   16-bit operands only route by replica; the exact-fp32 kernels add everything into replica 0 (same results through _reduce). */
int64_t snerf_mlp_gw_workspace_floats(const snerf_mlp_desc* desc);
int snerf_mlp_bwd_ws(const snerf_mlp_desc* desc, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                     int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* workspace, snerf_stream_t stream);
int snerf_mlp_gw_reduce(const snerf_mlp_desc* desc, float* workspace, float* gW, snerf_stream_t stream);
/* Same with X given in the net's 16-bit operand type (desc.operands = 1: bf16, 2: fp16; row stride ldx in elements) -- the feature tile
 * snerf_kplanes_field_fwd wrote.  The kernels round X to that type anyway, so results equal snerf_mlp_bwd on the fp32 image of X. */
int snerf_mlp_bwd_x16(const snerf_mlp_desc* desc, const float* W, const void* X16, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                      int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, float* gW, snerf_stream_t stream);
/* snerf_mlp_bwd_x16 with the QUOTIENT EPILOGUE (ABI 11; sigma_net shapes 32 k -> 128 -> d_out, 16-bit operands): instead of gX the kernel
 * writes G[N, ldg] = gX .* X, X being the 16-bit feature tile it holds in LDS for the layer-0 weight gradient anyway -- the tensor pass B of
 * the quotient scatter divides by v_q (snerf_kplanes_scatter_quotient_scales) -- and appends to the fix list the elements whose X vanished
 * (|X| below the smallest normal float) while gX did not: {element index n * ldg + col, gX} (two int32 per entry), G = 0 there.  Replaces
 * snerf_kplanes_quotient_prepare and the gX / fp32-feature round trips between the two kernels.  fix_count / fix_count_next as in
 * snerf_kplanes_quotient_prepare.  G carries the operand rounding of X (2^-9 relative for bf16: the size of the MFMA operand roundings gX
 * went through already); ldg must be the scatter's row stride (32 n_scales). */
int snerf_mlp_bwd_x16_quotient_ws(const snerf_mlp_desc* desc, const float* W, const void* X16, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                                  int32_t aux_col, const float* gaux, float* G, int32_t ldg, int32_t* fix_list, int32_t fix_capacity,
                                  int32_t* fix_count, int32_t* fix_count_next, float* workspace, snerf_stream_t stream);
int snerf_mlp_bwd_x16_quotient(const snerf_mlp_desc* desc, const float* W, const void* X16, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                               int32_t aux_col, const float* gaux, float* G, int32_t ldg, int32_t* fix_list, int32_t fix_capacity,
                               int32_t* fix_count, int32_t* fix_count_next, float* gW, snerf_stream_t stream);
/* Same with the weight gradients accumulated into fixed-point cells (see snerf_kplanes_gather_bwd_fx). */
int snerf_mlp_bwd_fx(const snerf_mlp_desc* desc, const float* W, const float* X, int32_t ldx, int64_t N, const float* gY, int32_t ldgy,
                     int32_t aux_col, const float* gaux, float* gX, int32_t ldgx, int64_t* gW_fx, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused K-Planes field forward = KPlanesField.get_density + get_outputs (NS/fields/kplanes_field.py:275-358) in one kernel: plane gather
 * (interpolate_kplanes :77-126) -> sigma_net (32 n_scales -> 128 -> 16, :249-261) -> density = trunc_exp(column 15) (:308-311) and
 * color_net on the 15 geometry features (15 -> 64 -> 64 -> 3 Sigmoid, :263-273, disable_viewing_dependent) -> rgb.  The features, the
 * 16 sigma_net outputs and every MLP activation stay on chip.  Built for 4-D plane sets with C = 32, concatenated scales (<= 6), the two
 * net shapes above and 16-bit MFMA operands (snerf_mlp_desc.operands = 1 bf16 / 2 fp16, both nets alike): snerf_kplanes_field_fwd_supported
 * tells; the exact-fp32 path composes snerf_kplanes_gather_fwd + snerf_mlp_fwd.  Results are bit-identical to that composition run with the
 * same 16-bit operands.  The backward runs unfused on what this kernel leaves behind (snerf_mlp_bwd, snerf_mlp_bwd_x16, the sorted scatter);
 * a fused backward kernel existed in ABI v3-v7 and was removed in v8 (slower than the unfused kernels).
 *   fwd: density [N], rgb [N,3].  Optional (NULL = not written), for a training step that runs the UNFUSED backward kernels on what the
 *        forward already computed: feat16 [N, 32 n_scales] = the feature tile in the operand type (bf16 / fp16: exactly the values the
 *        MFMA consumed; 2 B per feature instead of the 4 B the unfused gather writes) for snerf_mlp_bwd_x16(sigma_net), and h [N,16] =
 *        the raw fp32 sigma_net outputs (color_net's input; column 15 = log density) for snerf_mlp_bwd(color_net); feat32
 *        [N, 32 n_scales] = the fp32 features before rounding, for the quotient form of the plane scatter (snerf_kplanes_quotient_*).
 *        The three come as a set: feat16 and h together (or neither), feat32 only with them.
 * ------------------------------------------------------------------------------------------------ */
int snerf_kplanes_field_fwd_supported(const snerf_kplanes_desc* desc, const snerf_mlp_desc* sigma, const snerf_mlp_desc* color);
int snerf_kplanes_field_fwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                            const snerf_mlp_desc* sigma, const float* W_sigma, const snerf_mlp_desc* color, const float* W_color,
                            float* density, float* rgb, void* feat16, float* h, float* feat32, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused proposal density = KPlanesDensityField.get_density (NS/fields/kplanes_field.py:410-460) as ProposalNetworkSampler calls it per level
 * (NS/model_components/ray_samplers.py:559-600) in one kernel (ABI 11): plane gather (one scale of six C = 8 planes) -> sigma_net 8 -> 64 -> 1
 * (16-bit MFMA operands) -> density = trunc_exp(.).  Bit-identical to snerf_kplanes_gather_fwd + snerf_mlp_fwd (aux = trunc_exp) with the same
 * operands.  feat (optional, NULL = not written): the [N,8] fp32 features, which the unfused backward kernels of a step that updates the
 * proposal networks read (snerf_mlp_bwd, snerf_kplanes_gather_bwd).
 * ------------------------------------------------------------------------------------------------ */
int snerf_kplanes_density_fwd_supported(const snerf_kplanes_desc* desc, const snerf_mlp_desc* net);
int snerf_kplanes_density_fwd(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N, const snerf_mlp_desc* net,
                              const float* W, float* density, float* feat, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Compositing and ray-level losses (one wavefront per ray, S <= 320).
 * ------------------------------------------------------------------------------------------------ */

/* RGBRenderer + AccumulationRenderer + DepthRenderer(median|expected) + MedianRGBRenderer in one pass
 * (NS/model_components/renderers.py:58-140,197-223,226-287,290-362).
 * bg_mode 0: per-ray colour bg[R,3] (the "random" background with the draw made an explicit input);
 *         1: "last_sample"; 2: constant colour bg[3] ("black"/"white").
 * training == 0 applies the eval-mode nan_to_num + clamp.  Optional outputs may be NULL.
 * depth_expected is sum(w*steps)/(sum(w)+1e-10) WITHOUT the reference's global clip to [steps.min(), steps.max()]. */
typedef struct {
  const float* weights;  /* [R,S] */
  const float* rgb;      /* [R,S,3] */
  const float* ebins;    /* [R,S+1] */
  const float* bg;
  int32_t R, S, bg_mode, training;
  float* rgb_out;        /* [R,3] */
  float* acc_out;        /* [R] */
  float* depth_median;   /* [R] */
  float* depth_expected; /* [R] */
  float* median_rgb;     /* [R,3] (the reference returns it shaped [R,1,3]) */
  int64_t* median_index; /* [R]: searchsorted(cumsum(w), 0.5, left) clamped -- bit-exact vs the oracle */
} snerf_render_args;
int snerf_render_fwd(const snerf_render_args* args, snerf_stream_t stream);

/* The nerf level's per-ray work of a TRAINING step in one launch: get_weights (NS/cameras/rays.py:127-149) -> RGB / accumulation / median
 * depth (renderers.py:58-140,197-223,260-270) -> MSELoss backward (kplanes.py:418) -> distortion loss + gradient (losses.py:125-144) ->
 * get_weights backward.  Bit-identical to snerf_weights_fwd + snerf_render_fwd + snerf_render_mse_bwd + snerf_distortion(accumulate = 1) +
 * snerf_weights_bwd(accumulate = 0) run one after the other (same arithmetic, same order).  bg_mode 0 ([R,3]) or 2 ([3]).
 * go_scale = 2 c_rgb / (3 R), dist_scale = c_distortion / R.  Optional outputs may be NULL: depth_median, sqerr_rays, dist_rays, g_weights
 * (the gradient w.r.t. the weights, for inspection). */
typedef struct {
  const float* density;   /* [R,S] */
  const float* ebins;     /* [R,S+1] euclidean bin edges */
  const float* sbins;     /* [R,S+1] s-space bin edges */
  const float* rgb;       /* [R,S,3] */
  const float* bg;
  const float* target;    /* [R,3] */
  int32_t R, S, bg_mode;
  float go_scale, dist_scale;
  float* weights;         /* out [R,S] */
  float* rgb_out;         /* out [R,3] */
  float* acc_out;         /* out [R] */
  float* depth_median;    /* out [R] */
  float* sqerr_rays;      /* out [R]: sum_c (rgb_out - target)^2 */
  float* dist_rays;       /* out [R]: unscaled distortion loss per ray */
  float* g_rgb;           /* out [R,S,3] */
  float* g_density;       /* out [R,S] */
  float* g_weights;       /* out [R,S] */
  int32_t* nonfinite_flag; /* as snerf_weights_bwd */
} snerf_ray_train_args;
int snerf_ray_train_fwd_bwd(const snerf_ray_train_args* args, snerf_stream_t stream);

/* Backward of rgb_out (+ optionally accumulation) w.r.t. weights [R,S] and per-sample rgb [R,S,3] (g_rgb may be NULL).
 * bg_mode 0 or 2 only (the training backgrounds).  accumulate_w != 0: g_weights += ... */
int snerf_render_bwd(const float* weights, const float* rgb, const float* bg, int32_t bg_mode, const float* g_rgb_out,
                     const float* g_acc, int32_t R, int32_t S, float* g_weights, float* g_rgb, int32_t accumulate_w,
                     snerf_stream_t stream);

/* snerf_render_bwd with the image loss folded in (MSELoss, NS/models/kplanes.py:291,418): g_rgb_out = go_scale * (rgb_out - target)
 * is formed inside the kernel (go_scale = 2 * coefficient / (3 R) for the mean over R x 3 elements); sqerr_rays[R] (may be NULL)
 * receives sum_c (rgb_out - target)^2 so that the loss VALUE is one small reduction when somebody asks for it. */
int snerf_render_mse_bwd(const float* weights, const float* rgb, const float* bg, int32_t bg_mode, const float* rgb_out, const float* target,
                         float go_scale, int32_t R, int32_t S, float* g_weights, float* g_rgb, float* sqerr_rays, snerf_stream_t stream);

/* lossfun_distortion per ray (NS/model_components/losses.py:125-136): loss_rays[R] (may be NULL) and, if g_weights != NULL,
 * g_weights (+)= grad_scale * d loss_r / d w.  The caller supplies grad_scale = coefficient / R (mean over rays, :143). */
int snerf_distortion(const float* weights, const float* sbins, int32_t R, int32_t S, float grad_scale, float* loss_rays,
                     float* g_weights, int32_t accumulate, snerf_stream_t stream);

/* One proposal level of interlevel_loss (losses.py:46-121): per-ray sum over nerf bins of lossfun_outer, and the gradient
 * w.r.t. the proposal weights g_wprop[R,Sp] (= grad_scale * d sum / d wp; may be NULL).  The nerf level is detached (:111-112). */
int snerf_interlevel(const float* c_bins, const float* w_nerf, int32_t S, const float* p_bins, const float* w_prop, int32_t Sp,
                     int32_t R, float grad_scale, float* loss_rays, float* g_wprop, snerf_stream_t stream);

/* ds_nerf_depth_loss behind depth_loss, one sampling level (NS/model_components/losses.py:213-235,261-311; called per level with
 * weight 1/3 at NS/models/kplanes.py:395-409): loss_rays[R] (may be NULL) = [D > 0] * sum_s -log(w_s + 1e-7) *
 * exp(-(t_s - D)^2 / (2 sigma)) * (e_{s+1} - e_s) with t_s the bin centre and D = termination_depth (x directions_norm if that is not
 * NULL: depth maps holding z-distances, is_euclidean_depth = False); g_weights (may be NULL) (+)= grad_scale * d loss_r / d w.  The caller
 * supplies grad_scale = coefficient / (levels * R) (torch.mean over rays). */
int snerf_depth_loss(const float* weights, const float* ebins, const float* termination_depth, const float* directions_norm, float sigma,
                     int32_t R, int32_t S, float grad_scale, float* loss_rays, float* g_weights, int32_t accumulate, snerf_stream_t stream);

/* urban_radiance_field_depth_loss behind depth_loss (DepthLossType.URF, NS/model_components/losses.py:238-274,308-309), one sampling level:
 * loss_rays[R] = [D > 0] * ((D - predicted_depth)^2 + sum over bins within +-sigma of D of (w - N(t - D; 0, sigma / 3))^2 + sum over bins in front
 * of D - sigma of w^2); D, t, grad_scale, g_weights as snerf_depth_loss.  g_predicted_depth [R] (may be NULL) = grad_scale * d loss_r / d depth. */
int snerf_urf_depth_loss(const float* weights, const float* ebins, const float* termination_depth, const float* directions_norm,
                         const float* predicted_depth, float sigma, int32_t R, int32_t S, float grad_scale, float* loss_rays, float* g_weights,
                         float* g_predicted_depth, int32_t accumulate, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense per-step sweeps.
 * ------------------------------------------------------------------------------------------------ */

/* K-Planes plane regularisers for one plane set (losses.py:356-452): accumulates UNSCALED partial sums of
 * {space_tv, time_smoothness, sparse_transients} into losses[n_slots][16] (columns 0..2; one 64-B line per slot so the
 * per-workgroup adds do not serialise on one address; the caller zeroes it and sums over slots) and, if grad != NULL, adds
 * c_space_tv * d(space_tv) + c_time_smooth * d(time_smoothness) + c_sparse * d(sparse_transients) into grad
 * (same layout as planes); overwrite != 0 STORES instead (the caller knows grad is still zero: saves the read sweep). */
int snerf_plane_reg(const snerf_kplanes_desc* desc, const float* planes, float* grad, float c_space_tv, float c_time_smooth,
                    float c_sparse, float* losses, int32_t n_slots, int32_t overwrite, snerf_stream_t stream);

/* Device-resident optimiser state of ONE parameter group (= one torch optimiser of the reference: "fields", "proposal_networks",
 * NS/models/kplanes.py:311-316).  It lets a step be skipped without a host round trip, which is how the reference treats non-finite
 * gradients: GradScaler.step(optimizer) skips optimizer.step() when found_inf is set for that optimiser (NS/engine/trainer.py:394-408,
 * NS/engine/optimizers.py:130-139), Adam's own step counter then does not advance while the LR scheduler does.
 *   producers (snerf_weights_bwd) store 1 into `nonfinite`;
 *   snerf_adam_prepare (one tiny launch per group and step, before the group's Adam kernels): skip = policy == 1 && (nonfinite ||
 *     force_nonfinite); clears nonfinite; if skipped ++skipped, else ++t and step_size = lr / (1 - beta1^t), inv_sqrt_bc2 =
 *     1 / sqrt(1 - beta2^t);
 *   the Adam entry points, given `dyn` != NULL, ignore their `step` argument, read {step_size, inv_sqrt_bc2, skip} from it and on a
 *     skipped step leave p, m, v untouched (p_out = p) and only clear the gradient; non-finite gradient ELEMENTS that still reach them
 *     are dropped and counted in `dropped`.
 * Scope of the flag (a documented difference from GradScaler, which inspects every gradient tensor of the optimiser): it is raised where
 * non-finite gradients originate on this path -- the get_weights backward (an overflowed density makes autograd's 0 * inf there) -- i.e. by
 * snerf_weights_bwd / snerf_ray_train_fwd_bwd.  A non-finite value that first appears DOWNSTREAM of those kernels (an fp16-operand MLP backward
 * overflowing its 2^13 loss scale for |g| > 8, the depth term's -k / (w + 1e-7)) does not raise it: such elements take the drop-and-count path of
 * the Adam kernels instead of skipping the step, and in deterministic mode (fixed-point accumulation) a non-finite contribution adds nothing and
 * is not counted.  With the default bf16 operands no such case has been observed in 30 k-step runs (`dropped` stays 0). */
typedef struct {
  int32_t nonfinite;
  int32_t t;           /* optimiser steps taken (Adam's state["step"]) */
  int32_t skipped;     /* steps skipped */
  int32_t dropped;     /* gradient elements dropped */
  float step_size;
  float inv_sqrt_bc2;
  int32_t skip;
  int32_t _pad;
} snerf_adam_dyn;
int snerf_adam_prepare(snerf_adam_dyn* dyn, float lr, float beta1, float beta2, int32_t policy, int32_t force_nonfinite, snerf_stream_t stream);

/* torch.optim.Adam single-tensor step (no weight decay, no amsgrad) on a flat buffer; `step` is 1-based.
 * g is first multiplied by grad_scale (e.g. 1/world_size after an all-reduce SUM) and, if zero_grad != 0, cleared.
 * The new parameters go to p_out (may equal p: in place).  dyn: NULL, or the group's device-side state (above). */
int snerf_adam_step(const float* p, float* p_out, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int32_t step, float grad_scale, int32_t zero_grad, snerf_adam_dyn* dyn, snerf_stream_t stream);

/* Adam over one K-Planes plane set with the plane regularisers (snerf_plane_reg) fused in: the regulariser gradient is
 * formed from the +-1/+-2 neighbours of the OLD parameters inside the optimiser sweep and never touches HBM; g holds the
 * data-term gradient only.  Parameters must ping-pong: p_in (old, read with neighbours) != p_out (new).  All pointers address
 * the plane set's segment (same layout as `planes`).  losses / n_slots as in snerf_plane_reg (may be NULL). */
int snerf_adam_planes_step(const snerf_kplanes_desc* desc, const float* p_in, float* p_out, float* g, float* m, float* v,
                           float c_space_tv, float c_time_smooth, float c_sparse, float* losses, int32_t n_slots, float lr, float beta1,
                           float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, snerf_adam_dyn* dyn,
                           snerf_stream_t stream);

/* Same, restricted to the floats [range_lo, range_hi) of the segment (both multiples of 4): the optimiser shard of one rank when
 * the gradient is reduce-scattered instead of all-reduced (DDP + ZeroRedundancyOptimizer semantics; the reference's DDP wrapper
 * is NS/pipelines/base_pipeline.py:244-246).  p_in must hold the WHOLE old segment (neighbours across the shard edge are read);
 * only the range of p_out, g, m, v is touched; the loss partials cover the range only. */
int snerf_adam_planes_step_range(const snerf_kplanes_desc* desc, const float* p_in, float* p_out, float* g, float* m, float* v,
                                 float c_space_tv, float c_time_smooth, float c_sparse, float* losses, int32_t n_slots, float lr, float beta1,
                                 float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, int64_t range_lo, int64_t range_hi,
                                 snerf_adam_dyn* dyn, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient exchange (multi-GPU): the reference wraps the model in DistributedDataParallel (NS/pipelines/base_pipeline.py:244-246).
 * One communicator per process (one process per GPU); snerf_allreduce_grads is ONE in-place RCCL all-reduce (SUM, fp32) of the flat
 * gradient buffer on `stream` -- the mean over ranks is folded into snerf_adam_step's grad_scale = 1 / world.
 *   rank 0: snerf_comm_unique_id(id) -> the 128 bytes reach every rank by the launcher's own means (e.g. a torch.distributed broadcast)
 *   every rank: snerf_comm_create(world, rank, id, &comm) ... snerf_allreduce_grads(comm, grads, n, stream) ... snerf_comm_destroy(comm)
 * RCCL is resolved with dlopen("librccl.so.1") at the first call (the process's already-loaded copy); errors: code 1000 + ncclResult_t.
 * ------------------------------------------------------------------------------------------------ */
#define SNERF_COMM_ID_BYTES 128
int snerf_comm_unique_id(void* id128);
int snerf_comm_create(int32_t world, int32_t rank, const void* id128, void** comm_out);
int snerf_comm_destroy(void* comm);
int snerf_allreduce_grads(void* comm, float* grads, int64_t n, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Ray generation + collider.
 * ------------------------------------------------------------------------------------------------ */

/* RayGenerator.forward -> Cameras._generate_rays_from_coords, perspective / no distortion / camera-optimiser off
 * (NS/model_components/ray_generators.py:41-59, NS/cameras/cameras.py:505-741), optionally fused with
 * AABBBoxCollider (NS/model_components/scene_colliders.py:59-95).  indices: int64 [R,3] = (camera, row, col). */
typedef struct {
  const int64_t* indices;
  const float* fx; const float* fy; const float* cx; const float* cy; /* [M] */
  const float* c2w;        /* [M,3,4] */
  const float* cam_times;  /* [M] or NULL */
  int32_t R;
  int32_t collide;         /* 1: also fill nears/fars */
  int32_t training;        /* collider: near_plane applies in training only */
  float near_plane;
  float aabb_min[3];
  float aabb_max[3];
  float* origins;          /* [R,3] */
  float* dirs;             /* [R,3] */
  float* pixel_area;       /* [R] */
  float* dir_norm;         /* [R] */
  float* times;            /* [R] or NULL */
  float* nears;            /* [R] */
  float* fars;             /* [R] */
} snerf_raygen_args;
int snerf_raygen(const snerf_raygen_args* args, snerf_stream_t stream);

/* PixelSampler.sample_method (NS/data/pixel_samplers.py:74-77): indices[R,3] = floor(u[R,3] * (M,H,W)) as int64 (image, row, col), fused
 * with collate_image_dataset_batch's gather (:111-123): target[R,3] = images[c,y,x,:] / 255 for a resident uint8 image cache
 * [M,H,W,3] (images may be NULL: indices only). */
int snerf_sample_pixels_uniform(const float* u, int32_t R, int32_t M, int32_t H, int32_t W, const uint8_t* images, int64_t* indices,
                                float* target, snerf_stream_t stream);

/* The ray batch in order of a per-image key (image_key[M] in [0, n_keys): the rank of the image's frame time).  A batch is a set (losses are
 * means over it, the per-ray draws are i.i.d.), so its order is free; with equal-time rays adjacent, the gathers of every plane that holds the
 * time axis (and the temporal hash grid's rows) become coherent.  Out of place: indices_out [R,3] (and aux_out [R,aux_cols] = the same
 * permutation of aux_in, e.g. the target colours; aux_cols = 0: none).  The order is a pure function of the batch (ties keep their order).
 * R <= 16384 (one workgroup sorts in LDS). */
int snerf_sort_rays_by_key(const int64_t* indices_in, const int32_t* image_key, int32_t n_keys, int32_t R, const float* aux_in, int32_t aux_cols,
                           int64_t* indices_out, float* aux_out, snerf_stream_t stream);

/* AABBBoxCollider alone: aabb6 = HOST pointer to {min x,y,z, max x,y,z}. */
int snerf_aabb_collide(const float* origins, const float* dirs, int32_t R, const float* aabb6, float near_plane, int32_t training,
                       float* nears, float* fars, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Temporal multi-level hash grid (NeRFPlayer).  Replaces the reference's own native extension
 * `nerfstudio_field_components_cuda` (NS/field_components/cuda/csrc/include/temporal_gridencoder.h:24-60, pybind.cu:11-13):
 *   temporal_grid_encode_forward (inputs, temporal_row_index, embeddings, offsets, outputs, B, D, grid_C, C, L, S, H, dy_dx, gridtype, align_corners)
 *   temporal_grid_encode_backward(grad, inputs, temporal_row_index, embeddings, offsets, grad_embeddings, B, D, grid_C, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners)
 * dy_dx / grad_inputs (the gradient w.r.t. the coordinates, calc_grad_inputs): snerf_tgrid_encode_fwd_dydx + snerf_tgrid_input_bwd below.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t D;             /* input dims (1..3) */
  int32_t C;             /* level_dim: output features per level (1,2,4,8) */
  int32_t L;             /* levels (<= 32) */
  int32_t grid_C;        /* columns per table row = level_dim + temporal_dim */
  int32_t H;             /* base resolution */
  int32_t gridtype;      /* 0 = hash, 1 = tiled */
  int32_t align_corners; /* 0 in the reference's models */
  float S;               /* log2(per_level_scale) */
  int32_t offsets[33];   /* row offset of each level in the table; offsets[L] = total rows */
} snerf_tgrid_desc;

/* out[B, L*C] (the layout TemporalGridEncodeFunc returns after its permute, temporal_grid.py:108).
 * coords: mode 0 -> pts[B,D] in [0,1]; mode 1 -> derived from rays, normalised by the aabb to [0,1] (rescale ignored).
 * Time: EITHER temporal_row_index [B/samples_per_row, 4*C] (the reference's rows: w_a, col_a, w_b, col_b per channel, as
 * get_temporal_index builds them, temporal_grid.py:320-330) OR times [B/samples_per_row] in [0,1] (rows derived in-kernel from
 * the closed form of the channel table).  samples_per_row: 1 = one row per sample, S = one row per ray. */
int snerf_tgrid_encode_fwd(const snerf_tgrid_desc* desc, const float* embeddings, const snerf_coords* coords,
                           const float* temporal_row_index, const float* times, int32_t samples_per_row, int64_t B, float* out,
                           snerf_stream_t stream);
/* Forward that also writes dy_dx [B, L, D, C] = d out[b, l*C + ch] / d x[b, d] (temporal_gridencoder.cu:204-273; zero for out-of-range inputs),
 * and the coordinate gradient built from it: grad_inputs[B, D] = sum_{l, ch} grad_out[b, l*C + ch] * dy_dx[b, l, d, ch] (kernel_input_backward,
 * .cu:373-398).  The reference allocates dy_dx only when the inputs require a gradient (temporal_grid.py:82-87). */
int snerf_tgrid_encode_fwd_dydx(const snerf_tgrid_desc* desc, const float* embeddings, const snerf_coords* coords,
                                const float* temporal_row_index, const float* times, int32_t samples_per_row, int64_t B, float* out, float* dy_dx,
                                snerf_stream_t stream);
int snerf_tgrid_input_bwd(const float* grad_out, const float* dy_dx, int64_t B, int32_t D, int32_t C, int32_t L, float* grad_inputs,
                          snerf_stream_t stream);
/* ACCUMULATES (atomic fp32) into grad_embeddings [rows, grid_C]; the caller zeroes it when needed (the reference allocates a
 * zeros_like(embeddings) every backward, temporal_grid.py:126). */
int snerf_tgrid_encode_bwd(const snerf_tgrid_desc* desc, const snerf_coords* coords, const float* temporal_row_index,
                           const float* times, int32_t samples_per_row, int64_t B, const float* grad_out, float* grad_embeddings,
                           snerf_stream_t stream);
/* ABI 12, deterministic mode: the same scatter ACCUMULATED into 2^50-scaled 64-bit integer cells [rows, grid_C] (integer addition is
 * associative: the sum does not depend on the order the wavefronts arrive in); snerf_fx_to_float turns the cells into floats once per step. */
int snerf_tgrid_encode_bwd_fx(const snerf_tgrid_desc* desc, const snerf_coords* coords, const float* temporal_row_index,
                              const float* times, int32_t samples_per_row, int64_t B, const float* grad_out, int64_t* grad_embeddings_fx,
                              snerf_stream_t stream);

/* ABI 14: the backward w.r.t. the table in OWNER-COMPUTES form, for D = 3 grids with per-sample / per-ray `times` (csrc/tgrid_tiles.hip).  Replaces the
 * same reference kernel (kernel_grid_backward, temporal_gridencoder.cu:283-370) and, in its fused form, the torch.optim.Adam step of the table that
 * follows it (NS/configs/method_configs.py:648-657) together with the temporal-TV gradient (NS/field_components/temporal_grid.py:352-376).  The table is
 * cut into tiles of 2^tile_rows_log2 consecutive rows; snerf_tgrid_bwd_bin files every (sample, level, corner) under the tile its row belongs to (a
 * counting sort without global atomics), and one workgroup per tile then sums the tile's gradient rows in LDS and either adds them into the dense
 * gradient buffer with plain stores (snerf_tgrid_bwd_tiles: same result as snerf_tgrid_encode_bwd up to the association order of the float sums) or
 * runs Adam for its rows straight from LDS (snerf_tgrid_bwd_tiles_adam: the dense gradient buffer is not touched).  Levels [0, first_tiled_level) --
 * few rows, thousands of samples per row -- are NOT binned: the caller runs snerf_tgrid_encode_bwd restricted to them (snerf_tgrid_encode_bwd_levels)
 * into grad_embeddings, which the fused form reads and clears for those levels' rows. */
typedef struct {
  int32_t tile_rows_log2;     /* rows per tile = 1 << tile_rows_log2 */
  int32_t n_tiles;            /* over all levels = tile_start[L] */
  int32_t n_chunks;           /* sample chunks of the binning passes = ceil(B / chunk) */
  int32_t chunk;              /* samples per chunk */
  int32_t first_tiled_level;  /* levels below it go through the atomic kernel */
  int32_t lds_bytes;          /* dynamic LDS of a tile workgroup */
  int32_t tile_start[33];     /* first tile of each level */
  int32_t _pad;
  int64_t count_ints;         /* int32 elements of `counts` */
  int64_t record_capacity;    /* uint32 elements of `records` = B * (L - first_tiled_level) * 8 */
} snerf_tgrid_tile_plan;
/* Host arithmetic only.  tile_rows_log2 <= 0: the largest tile that lets two workgroups share a CU's LDS; first_tiled_level < 0: levels with fewer than
 * 2^16 rows stay atomic. */
int snerf_tgrid_tile_plan_make(const snerf_tgrid_desc* desc, int64_t B, int32_t tile_rows_log2, int32_t first_tiled_level, snerf_tgrid_tile_plan* plan);
/* counts [count_ints] and tile_base [n_tiles + 3] are workspaces (no initialisation needed: n_tiles + 1 prefix sums, then two words the fused tile pass
 * uses to hand out tiles); records [record_capacity]; pos4 [B,4] (16-byte aligned)
 * receives (x, y, z, time) per sample -- the later passes read ONLY pos4, grad_out, tile_base and records, never `coords`, so they may run on another
 * stream while the caller's ray buffers are rewritten.  grad_out may be NULL: the pass then files every in-range sample (it depends on the sample positions
 * only, so it can run as soon as those exist -- beside the forward); with grad_out, (sample, level) pairs whose gradient is all zero are left out.  B < 2^28. */
int snerf_tgrid_bwd_bin(const snerf_tgrid_desc* desc, const snerf_tgrid_tile_plan* plan, const snerf_coords* coords, const float* times,
                        int32_t samples_per_row, int64_t B, const float* grad_out, float* pos4, int32_t* counts, int32_t* tile_base, uint32_t* records,
                        snerf_stream_t stream);
/* grad_embeddings [rows, grid_C] += the gradient of the tiled levels (16-byte aligned). */
int snerf_tgrid_bwd_tiles(const snerf_tgrid_desc* desc, const snerf_tgrid_tile_plan* plan, int64_t B, const float* grad_out, const float* pos4,
                          const int32_t* tile_base, const uint32_t* records, float* grad_embeddings, snerf_stream_t stream);
/* Adam (torch.optim.Adam: bias-corrected, eps outside the square root; step is 1-based) over the WHOLE table with gradient = tiled scatter + what
 * grad_embeddings holds for the rows of levels [0, first_tiled_level) (read and cleared; may be NULL when first_tiled_level = 0) + the temporal-TV
 * term srow[row] on column col_a and -srow[row] on col_b (col_a < 0: none; srow from snerf_tgrid_tv_sign).  A non-finite gradient element is dropped. */
int snerf_tgrid_bwd_tiles_adam(const snerf_tgrid_desc* desc, const snerf_tgrid_tile_plan* plan, int64_t B, const float* grad_out, const float* pos4,
                               int32_t* tile_base, const uint32_t* records, float* grad_embeddings, float* p, float* m, float* v, float lr,
                               float beta1, float beta2, float eps, int32_t step, int32_t col_a, int32_t col_b, const float* srow, snerf_stream_t stream);
/* snerf_tgrid_encode_bwd for levels [level_begin, level_end) only (the coarse levels beside the tiled form). */
int snerf_tgrid_encode_bwd_levels(const snerf_tgrid_desc* desc, const snerf_coords* coords, const float* temporal_row_index, const float* times,
                                  int32_t samples_per_row, int64_t B, const float* grad_out, float* grad_embeddings, int32_t level_begin,
                                  int32_t level_end, snerf_stream_t stream);

/* TemporalGridEncoder.get_temporal_tv_loss (NS/field_components/temporal_grid.py:352-376): mean over table rows of
 * |E[r, col_a] - E[r, col_b]|.  fwd ADDS per-workgroup partial sums of |.| into partial[n_slots][16] (col 0; caller zeroes, then
 * value = sum / rows); bwd ADDS g_tv[0] * sign(.) / rows into grad_embeddings[:, col_a] and subtracts it from [:, col_b]
 * (g_tv: device scalar, the upstream gradient of the loss value). */
int snerf_tgrid_tv_fwd(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, float* partial, int32_t n_slots,
                       snerf_stream_t stream);
int snerf_tgrid_tv_bwd(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, const float* g_tv,
                       float* grad_embeddings, snerf_stream_t stream);
/* Both in one pass, for callers that know the upstream gradient g_tv (the loss weight) before the value (the fused trainer). */
int snerf_tgrid_tv_fwd_bwd(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, float g_tv, float* partial,
                           int32_t n_slots, float* grad_embeddings, snerf_stream_t stream);

/* TV term folded into the optimiser (fused NeRFPlayer trainer): snerf_tgrid_tv_sign adds the partial sums of |E[r,a] - E[r,b]| into
 * partial (as snerf_tgrid_tv_fwd) and writes srow[rows] = g_tv * sign(E[r,a] - E[r,b]) / rows from the CURRENT table; then
 * snerf_adam_step_tv is snerf_adam_step (in place) over the table [rows][grid_C] with +srow[r] added to the gradient of column
 * col_a and -srow[r] to column col_b -- the dense gradient buffer is never read-modified-written for the TV term. */
int snerf_tgrid_tv_sign(const float* embeddings, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, float g_tv, float* partial,
                        int32_t n_slots, float* srow, snerf_stream_t stream);
int snerf_adam_step_tv(float* p, float* g, float* m, float* v, int64_t rows, int32_t grid_C, int32_t col_a, int32_t col_b, const float* srow,
                       float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, snerf_adam_dyn* dyn,
                       snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Static multiresolution hash grid: tcnn.Encoding(3, {"otype": "HashGrid", "n_levels", "n_features_per_level", "log2_hashmap_size",
 * "base_resolution", "per_level_scale"}) as the full NeRFPlayer constructs it (NS/fields/nerfplayer_field.py:242-252) and calls it
 * on the normalised and on the DEFORMED positions (:341-342), so the coordinate gradient is part of the contract.
 * tiny-cuda-nn (v1.6, Dockerfile:121) is third-party and absent from the reference tree: published algorithm, parity unpinned
 * (oracle/hashgrid_oracle.py).  No bounds check on coordinates, as tcnn.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t D;              /* input dims (1..3) */
  int32_t F;              /* n_features_per_level (1,2,4,8) */
  int32_t L;              /* n_levels (<= 32) */
  float scale[32];        /* filled by snerf_hashgrid_layout: exp2(l * log2(per_level_scale)) * base_resolution - 1 */
  int32_t resolution[32]; /* ceil(scale) + 1 */
  int32_t offsets[33];    /* row offset of each level; offsets[L] = total rows (each level: min(roundup8(res^D), 2^log2_hashmap_size)) */
} snerf_hashgrid_desc;
/* HOST ONLY: fills scale/resolution/offsets from (D, L already set) and returns the total row count (< 0 on bad arguments). */
int64_t snerf_hashgrid_layout(snerf_hashgrid_desc* desc, int32_t base_resolution, float per_level_scale, int32_t log2_hashmap_size);
/* out[B, L*F] = D-linear interpolation of table[rows, F] per level, level-major. */
int snerf_hashgrid_encode_fwd(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, float* out,
                              snerf_stream_t stream);
/* grad_table [rows, F] (may be NULL) and grad_x [B, D] (may be NULL; needs table) are ACCUMULATED into (atomic fp32): caller zeroes. */
int snerf_hashgrid_encode_bwd(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, const float* grad_out,
                              float* grad_table, float* grad_x, snerf_stream_t stream);
/* ABI 13, deterministic mode: grad_table_fx [rows, F] and grad_x_fx [B, D] (either may be NULL) are 2^50-scaled 64-bit cells (integer adds are
 * associative: the table scatter's arrival order and the order in which the levels' workgroups reach a sample's coordinate gradient no longer
 * matter); snerf_fx_to_float turns them into floats. */
int snerf_hashgrid_encode_bwd_fx(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, const float* grad_out,
                                 int64_t* grad_table_fx, int64_t* grad_x_fx, snerf_stream_t stream);

/* ABI 14: the backward w.r.t. the hash table in OWNER-COMPUTES form (csrc/hashgrid_tiles.hip; D = 3), as snerf_tgrid_bwd_bin / _tiles / _tiles_adam for the
 * temporal grid: the batch's (point, level, corner pair) touches are filed under tiles of 2^tile_rows_log2 consecutive table rows by a counting sort without
 * global atomics; one workgroup per tile sums its rows in LDS and either adds them into the dense gradient (snerf_hashgrid_bwd_tiles: what
 * snerf_hashgrid_encode_bwd leaves in grad_table, up to the association order of the sums) or runs torch.optim.Adam over them from there
 * (snerf_hashgrid_bwd_tiles_adam: no dense gradient for the table).  Replaces tcnn's HashGrid table backward (grid.h, kernel_grid_backward) + the optimiser
 * step of NS/fields/nerfplayer_field.py:242-252's encoding.  The coordinate gradient stays with snerf_hashgrid_encode_bwd (grad_table = NULL). */
typedef struct {
  int32_t tile_rows_log2, n_tiles, n_chunks, chunk, lds_bytes;
  int32_t first_tiled_level;  /* levels below it (few rows, every point of the batch in a handful of tiles) go through the atomic kernel */
  int32_t tile_start[33];
  int32_t _pad2;
  int64_t count_ints;       /* int32 elements of `counts` */
  int64_t record_capacity;  /* uint32 elements of `records` = B * (L - first_tiled_level) * 8 */
} snerf_hashgrid_tile_plan;
/* tile_rows_log2 <= 0: 2^11 rows (or fewer when a tile image would not fit 64 KB); first_tiled_level < 0: levels with fewer than 2^18 rows stay atomic. */
int snerf_hashgrid_tile_plan_make(const snerf_hashgrid_desc* desc, int64_t B, int32_t tile_rows_log2, int32_t first_tiled_level,
                                  snerf_hashgrid_tile_plan* plan);
/* snerf_hashgrid_encode_bwd for levels [level_begin, level_end) only. */
int snerf_hashgrid_encode_bwd_levels(const snerf_hashgrid_desc* desc, const float* table, const float* x, int64_t B, const float* grad_out, float* grad_table,
                                     float* grad_x, int32_t level_begin, int32_t level_end, snerf_stream_t stream);
/* counts [count_ints], tile_base [n_tiles + 1]: workspaces; x [B,3] and grad_out [B, L*F] must stay valid until the tile pass has run; grad_out may be NULL
 * in the binning pass (every point is filed: the pass then depends on the positions only).  B < 2^28. */
int snerf_hashgrid_bwd_bin(const snerf_hashgrid_desc* desc, const snerf_hashgrid_tile_plan* plan, const float* x, int64_t B, const float* grad_out,
                           int32_t* counts, int32_t* tile_base, uint32_t* records, snerf_stream_t stream);
int snerf_hashgrid_bwd_tiles(const snerf_hashgrid_desc* desc, const snerf_hashgrid_tile_plan* plan, const float* x, int64_t B, const float* grad_out,
                             const int32_t* tile_base, const uint32_t* records, float* grad_table, snerf_stream_t stream);
/* grad_table: what snerf_hashgrid_encode_bwd_levels left for levels [0, first_tiled_level) (read and cleared; NULL when every level is tiled). */
int snerf_hashgrid_bwd_tiles_adam(const snerf_hashgrid_desc* desc, const snerf_hashgrid_tile_plan* plan, const float* x, int64_t B, const float* grad_out,
                                  const int32_t* tile_base, const uint32_t* records, float* grad_table, float* p, float* m, float* v, float lr, float beta1,
                                  float beta2, float eps, int32_t step, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * NeRFPlayer decomposition mixing (NerfplayerField.get_density, NS/fields/nerfplayer_field.py:365-372): probs[N,3] = softmax(logits[N,3])
 * (0 = static, 1 = deforming, 2 = new) and v[N,F] = probs_0 v_static + probs_1 v_deform + probs_2 v_new; F = 4, 8, 16, 32 or 64.
 * bwd: from g_v [N,F] and (may be NULL) g_probs [N,3], the gradient that reaches probs from outside v (the rendered-probability
 * regulariser, NS/models/nerfplayer.py:336-341): g_static / g_deform / g_new [N,F] and g_logits [N,3], all overwritten.
 * ------------------------------------------------------------------------------------------------ */
int snerf_nerfplayer_mix_fwd(const float* logits, const float* v_static, const float* v_deform, const float* v_new, int64_t N, int32_t F,
                             float* probs, float* v, snerf_stream_t stream);
int snerf_nerfplayer_mix_bwd(const float* probs, const float* v_static, const float* v_deform, const float* v_new, const float* g_v,
                             const float* g_probs, int64_t N, int32_t F, float* g_static, float* g_deform, float* g_new, float* g_logits,
                             snerf_stream_t stream);

/* ABI 13: the colour head's input of the NeRFPlayer-nerfacto field (NS/fields/nerfplayer_nerfacto_field.py:350-384), one launch each way instead of ~35
 * ATen kernels.  fwd: hx[R*S, 64] = [SH degree 4 of dirs[ray] (16, the expressions of tcnn's SphericalHarmonics on unit vectors) | h[n, 1:16] (15) |
 * appearance[cams[ray]] (32; cams NULL: row 0 of `appearance` for every ray -- an average embedding; appearance NULL: zeros) | 0].
 * bwd: g_h[n, 1:16] = g_hx[n, 16:31] (the other columns of g_h are left alone) and, if asked for, the appearance gradient: for every ray the sum over its
 * samples of g_hx[n, 31:63] is ADDED to row cams[ray] of g_appearance [num_images, 32] (float atomics) or g_appearance_fx (fixed-point cells); not both. */
int snerf_nerfacto_head_input_fwd(const float* dirs, const float* h, const float* appearance, const int64_t* cams, int32_t S, int64_t R, float* hx,
                                  snerf_stream_t stream);
int snerf_nerfacto_head_input_bwd(const float* g_hx, const int64_t* cams, int32_t S, int64_t R, float* g_h, float* g_appearance,
                                  int64_t* g_appearance_fx, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * KPlanesField's linear decoder, the pointwise pieces (NS/fields/kplanes_field.py:305-311, :349-354; the dense layers are snerf_dense_*):
 * trunc_exp (NS/field_components/activations.py:25-41): y = exp(x);  gx = g * exp(clamp(x, -15, 15)).
 * basis_rgb: rgb[N,3] = sigmoid(sum_f feat[n, f] * basis[n, c * F + f]) with basis [N, 3F] = color_basis(direction) and feat [N, F] (row
 * stride ldf) the interpolated plane features; bwd from the stored rgb: g_feat [N,F] (may be NULL) and g_basis [N,3F], both overwritten.
 * F and ldf multiples of 4.
 * ------------------------------------------------------------------------------------------------ */
int snerf_trunc_exp_fwd(const float* x, int64_t n, float* y, snerf_stream_t stream);
int snerf_trunc_exp_bwd(const float* x, const float* g, int64_t n, float* gx, snerf_stream_t stream);
int snerf_basis_rgb_fwd(const float* feat, int32_t ldf, const float* basis, int64_t N, int32_t F, float* rgb, snerf_stream_t stream);
int snerf_basis_rgb_bwd(const float* feat, int32_t ldf, const float* basis, const float* rgb, const float* g_rgb, int64_t N, int32_t F,
                        float* g_feat, float* g_basis, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Ray importance sampling (IST = temporal difference).
 * ------------------------------------------------------------------------------------------------ */

/* DynamicDataset.compute_ist (NS/data/datasets/dynamic_dataset.py:328-470): out[M,H,W] (fp16) = per-pixel max over the
 * image's temporal neighbours of |img_i - img_j|, mean over RGB, zeroed at or below alpha (0.15 in the reference); all-ones for
 * an image without neighbours.  images [M,H,W,3]: image_dtype 0 = uint8 (scaled by 1/255 as base_dataset.py:82 does),
 * 1 = float32 in [0,1].  The neighbour relation (same camera id, 0.01 < |dt| <= ist_range, :426-429) is passed as CSR:
 * nbr_off [M+1], nbr_idx [nnz] (int32, device). */
int snerf_ist_maps(const void* images, int32_t image_dtype, int32_t M, int32_t H, int32_t W, const int32_t* nbr_off,
                   const int32_t* nbr_idx, float alpha, void* out_f16, snerf_stream_t stream);

/* DynamicDataset.compute_isg (NS/data/datasets/dynamic_dataset.py:215-326): out[M,H,W] (fp16) = mean over RGB of r^2 / (r^2 + gamma^2),
 * r = image - per-camera median image (torch.median over the camera's images: the LOWER median, an input element).  The cameras'
 * image lists are CSR: cam_off [n_cams+1], cam_img [M] (int32, device); img_cam [M] = camera slot of each image; max_frames = the
 * longest list (<= 128).  medians: workspace [n_cams,H,W,3] of the images' dtype (also an output). */
int snerf_isg_maps(const void* images, int32_t image_dtype, int32_t M, int32_t H, int32_t W, int32_t n_cams, const int32_t* cam_off,
                   const int32_t* cam_img, const int32_t* img_cam, int32_t max_frames, float gamma, void* medians, void* out_f16,
                   snerf_stream_t stream);

/* Weighted pixel draws of DynamicBasedPixelSampler.sample_method (NS/data/pixel_samplers.py:369-411): slot j = draws [j * per_image,
 * min((j + 1) * per_image, n)) takes its image from chosen_images[j] and pixels with probability proportional to the image's weight map,
 * by inverse-CDF on cdf[M, H*W] (fp32 inclusive prefix sums of the maps) with the uniform draws u[n].  As torch.multinomial with the
 * reference's replacement flag (:400-402): WITHOUT replacement inside a slot when nonzero_counts[image] >= the slot's draws (a drawn
 * pixel's weight leaves the distribution of the slot's later draws), with replacement otherwise (or always, if nonzero_counts is NULL).
 * min(per_image, n) <= 13000 (the slot's removed-pixel list lives in LDS; the search costs O(draws^2) per slot -- the preset draws 10).
 * indices [n,3] int64 = (image, row, col).  oracle/ist_oracle.py::sample restates the loop: identical indices for
 * identical (cdf, chosen, u). */
int snerf_ist_sample(const float* cdf, int32_t H, int32_t W, const int64_t* chosen_images, const int32_t* nonzero_counts, int32_t per_image,
                     const float* u, int32_t n, int64_t* indices, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sorted plane-gradient scatter (same result as snerf_kplanes_gather_bwd, ~6x fewer atomic requests on training batches).
 *   1. snerf_kplanes_sort_samples : counting sort of the N samples, once per PLANE, by the Z-order (Morton) code of their texel
 *      at the finest scale -- one order serves every scale, because a coarser texel is a (nearly) aligned block of fine ones.
 *      Depends only on the sample coordinates -> can run on a side stream as soon as they are known.
 *      Workspace sizes (elements) from snerf_kplanes_sort_workspace: hist[hist_cells], rank[index_elems] (int32) and
 *      sorted_rec[index_elems][4] (float: sample id bits, the two normalised plane coordinates, 0)  (index_elems = n_planes * N).
 *   2. snerf_kplanes_gradvec      : gvec[scale*n_planes+plane][N][C] = dL/d(interpolated value of that plane) per sample; elements are
 *      fp32 (gvec_bf16 = 0: exact, the parity path) or bf16 (gvec_bf16 = 1: half the bytes of the largest intermediate of the step; the
 *      scatter still accumulates in fp32).
 *   3. snerf_kplanes_scatter_sorted: walks each segment in sorted order, applies the bilinear weights, run-length-combines
 *      equal texel rows and ACCUMULATES into grad_planes with one 2*C-float atomic instruction per run.
 * ------------------------------------------------------------------------------------------------ */
int snerf_kplanes_sort_workspace(const snerf_kplanes_desc* desc, int64_t N, int64_t* hist_cells, int64_t* index_elems);
int snerf_kplanes_sort_samples(const snerf_kplanes_desc* desc, const snerf_coords* coords, int64_t N, int32_t* hist, int32_t* rank,
                               float* sorted_rec, snerf_stream_t stream);
int snerf_kplanes_gradvec(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N, const float* grad_out,
                          void* gvec, int32_t gvec_bf16, snerf_stream_t stream);
int snerf_kplanes_scatter_sorted(const snerf_kplanes_desc* desc, int64_t N, const void* gvec, int32_t gvec_bf16, const float* sorted_rec,
                                 float* grad_planes, snerf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Quotient form of the sorted scatter (C = 32, concatenated scales).  interpolate_kplanes multiplies the six planes' values of a scale
 * (NS/fields/kplanes_field.py:113-120), so the gradient w.r.t. plane q's value at a sample is
 *     g_q = gfeat .* prod_{p != q} v_p = (gfeat .* feat) ./ v_q          (feat = prod_p v_p, what the forward computed)
 * Instead of materialising 6 gradient vectors per (sample, scale) (snerf_kplanes_gradvec: a second gather of all 30 planes + 1 GB of
 * vectors at the preset), ONE tensor G = gfeat .* feat [N, 32 n_scales] is formed and pass B divides by v_q, which it re-interpolates
 * from the 4 texels of the cell it is adding into (sorted order: the reads stay in cache) -- with the forward's own arithmetic, bit for
 * bit, so that the division cancels the forward's v_q exactly even where v_q is a small difference of large texels.  Differs from the
 * product form by a few ulp.
 * Where a feature vanished (exactly 0, or below the smallest normal float -- also as the underflowing product of six normal plane values) the
 * quotient has lost the other planes' product: the producer of G writes G = 0 for such an element and, if its gradient is not zero, appends
 * {element index n * 32 n_scales + s * 32 + ch, feature gradient} to the fix list (ABI 11: two int32 per entry; rows before.  Device-side,
 * capacity in ENTRIES; entries beyond it are dropped); pass B then adds exactly 0 there and _fixup adds the exact product-form terms (one
 * vanished plane: that plane's; none, i.e. an underflowed product: every plane's).  A plane value that is subnormal but not zero beside a
 * usable G is divided by in IEEE arithmetic inside pass B (v_rcp_f32 may flush it).
 *   _prepare : G = grad_feat .* feat + the fix list (fix_list: int32[2 * fix_capacity]).  fix_count must be 0 on entry: with
 *              fix_count_next == NULL it is reset here (one memset); a caller that alternates between two counters passes the other one as
 *              fix_count_next and the kernel resets THAT one for the next step (no extra launch).  snerf_mlp_bwd_x16_quotient produces
 *              the same pair from inside the sigma_net backward.
 *   _scatter_quotient_scales : pass B over scales [scale_begin, scale_end), sorted_rec from snerf_kplanes_sort_samples; ACCUMULATES.
 *   _fixup   : exact terms of the listed elements for scales [scale_begin, scale_end); ACCUMULATES.  Launch cost only when the list is empty.
 *              ABI 12: overflow_peak (device int32, may be NULL) -- when more entries were appended than fix_capacity holds, the demanded
 *              count is max-ed into it (sticky; the caller clears it), so that lost entries are an error the host can raise, not a silent
 *              drop.  The producers keep counting beyond the capacity (fix_count then exceeds it).
 * ------------------------------------------------------------------------------------------------ */
int snerf_kplanes_quotient_supported(const snerf_kplanes_desc* desc, int64_t N);
int snerf_kplanes_quotient_prepare(const snerf_kplanes_desc* desc, int64_t N, const float* grad_feat, const float* feat, float* G,
                                   int32_t* fix_list, int32_t fix_capacity, int32_t* fix_count, int32_t* fix_count_next, snerf_stream_t stream);
int snerf_kplanes_scatter_quotient_scales(const snerf_kplanes_desc* desc, const float* planes, int64_t N, const float* G, const float* sorted_rec,
                                          float* grad_planes, int32_t scale_begin, int32_t scale_end, snerf_stream_t stream);
int snerf_kplanes_quotient_fixup(const snerf_kplanes_desc* desc, const float* planes, const snerf_coords* coords, int64_t N,
                                 const int32_t* fix_list, const int32_t* fix_count, int32_t fix_capacity, float* grad_planes,
                                 int32_t scale_begin, int32_t scale_end, int32_t* overflow_peak, snerf_stream_t stream);
/* Step 3 for the scales [scale_begin, scale_end) only: lets the caller start the optimiser sweep of the planes whose gradient is
 * complete (snerf_adam_planes_step_range) while the remaining scales are still being scattered. */
int snerf_kplanes_scatter_sorted_scales(const snerf_kplanes_desc* desc, int64_t N, const void* gvec, int32_t gvec_bf16, const float* sorted_rec,
                                        float* grad_planes, int32_t scale_begin, int32_t scale_end, snerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SNERF_H */

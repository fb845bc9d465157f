"""CPU oracle for the K-Planes hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (fp32, CPU) restatement of the reference algorithm for the path
ray batch -> samples -> field -> composite -> losses.  Every function cites the reference
file:line it follows (NS/ = /root/reference/nerfstudio/nerfstudio/).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
the product package (`soccernerfs_amd/`) never does and fails loudly without its HIP library.

Parity pinning: checked against golden vectors captured from the reference itself
(`oracle/gen_golden.py` imports /root/reference with throw-away shims; fixtures under
`tests/golden/`), and live against the imported reference in `tests/test_oracle_vs_reference.py`
whenever /root/reference is present.  The MLPs are tiny-cuda-nn networks in the reference
(third-party, v1.6, source absent): they are restated as bias-free fp32 Linear stacks
(ReLU hidden, None/Sigmoid output) -- "parity unpinned" for tcnn's fp16 numerics.

Layouts here are the reference's: planes [1, C, H, W]; Linear weights [out, in].

One deliberate, documented deviation: `pdf_sample` sums the padded weights in sequential
order (cumsum[-1]) instead of `torch.sum`, whose association order depends on the CPU ISA
(AVX2/AVX512) and is a parallel tree on GPUs.  With the sequential order the sample indices
are a pure function of the inputs, which is what the HIP kernel reproduces bit-exactly.
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

EPS = 1.0e-7  # NS/model_components/losses.py:32


# ----------------------------------------------------------------------------------------------
# geometry: ray generation, collider, positions
# ----------------------------------------------------------------------------------------------
def generate_rays_pinhole(indices, fx, fy, cx, cy, c2w, times):
    """Perspective, no distortion, camera optimiser off.

    Follows NS/cameras/cameras.py:318-319 (pixel centre +0.5), :622-633,:663-670 (camera-frame
    directions incl. +1 pixel x/y neighbours), :712-715 (rotate, normalise), :725-729 (pixel_area),
    :732 (times) and NS/model_components/ray_generators.py:41-59.

    indices [R,3] int64 (camera, row, col); fx,fy,cx,cy [M]; c2w [M,3,4]; times [M].
    """
    c, y, x = indices[:, 0], indices[:, 1], indices[:, 2]
    yy = y.to(torch.float32) + 0.5
    xx = x.to(torch.float32) + 0.5
    fxc, fyc, cxc, cyc = fx[c], fy[c], cx[c], cy[c]

    def cam_dir(px, py):
        return torch.stack([(px - cxc) / fxc, -(py - cyc) / fyc, -torch.ones_like(px)], dim=-1)

    d0 = cam_dir(xx, yy)
    dx = cam_dir(xx + 1.0, yy)
    dy = cam_dir(xx, yy + 1.0)
    rot = c2w[c][:, :3, :3]  # [R,3,3]

    def to_world(d):
        dw = torch.sum(d[:, None, :] * rot, dim=-1)
        n = torch.linalg.norm(dw, dim=-1, keepdim=True)
        return dw / n, n

    d0w, norm = to_world(d0)
    dxw, _ = to_world(dx)
    dyw, _ = to_world(dy)
    ddx = torch.sqrt(torch.sum((d0w - dxw) ** 2, dim=-1))
    ddy = torch.sqrt(torch.sum((d0w - dyw) ** 2, dim=-1))
    return {
        "origins": c2w[c][:, :3, 3],
        "directions": d0w,
        "pixel_area": (ddx * ddy)[:, None],
        "directions_norm": norm,
        "camera_indices": c[:, None],
        "times": times[c][:, None],
    }


def intersect_aabb(origins, directions, aabb, near_plane: float, training: bool):
    """NS/model_components/scene_colliders.py:59-95 (slab test, 1/(d+1e-6), clamp, far>=near+1e-6)."""
    inv = 1.0 / (directions + 1e-6)
    t_lo = (aabb[0][None, :] - origins) * inv
    t_hi = (aabb[1][None, :] - origins) * inv
    nears = torch.minimum(t_lo, t_hi).max(dim=1).values
    fars = torch.maximum(t_lo, t_hi).min(dim=1).values
    nears = torch.clamp(nears, min=near_plane if training else 0.0)
    fars = torch.maximum(fars, nears + 1e-6)
    return nears[:, None], fars[:, None]


def sample_positions(origins, directions, starts, ends):
    """Frustums.get_positions, NS/cameras/rays.py:54: o + d * (start + end) / 2."""
    return origins[:, None, :] + directions[:, None, :] * (starts + ends)[..., None] / 2


# ----------------------------------------------------------------------------------------------
# samplers
# ----------------------------------------------------------------------------------------------
def spacing_fns(kind: str):
    """'uniform' (ray_samplers.py:143-149) or 'piecewise' uniform/lin-disp (:238-246)."""
    if kind == "uniform":
        return (lambda v: v), (lambda v: v)
    if kind == "piecewise":
        return (
            lambda v: torch.where(v < 1, v / 2, 1 - 1 / (2 * v)),
            lambda v: torch.where(v < 0.5, 2 * v, 1 / (2 - 2 * v)),
        )
    raise ValueError(kind)


def spaced_bins(num_rays: int, num_samples: int, t_rand: Optional[torch.Tensor]):
    """Normalised bin edges [R, S+1]; SpacedSampler.generate_ray_samples, ray_samplers.py:101-112.

    t_rand None = eval (no jitter); else [R,S+1] or [R,1] (single_jitter) uniform draws.
    """
    bins = torch.linspace(0.0, 1.0, num_samples + 1, device=t_rand.device if t_rand is not None else None)[None, :]
    if t_rand is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    else:
        bins = bins.expand(num_rays, -1)
    return bins


def spacing_to_euclidean(bins, nears, fars, kind: str = "uniform"):
    """ray_samplers.py:114-116: fn_inv(x * fn(far) + (1 - x) * fn(near))."""
    fn, fn_inv = spacing_fns(kind)
    s_near, s_far = fn(nears), fn(fars)
    return fn_inv(bins * s_far + (1 - bins) * s_near)


def pdf_sample(weights, existing_bins, u, histogram_padding: float = 0.01, eps: float = 1e-5):
    """PDFSampler.generate_ray_samples, ray_samplers.py:302-351 with include_original=False.

    weights [R,S_prev] (already annealed), existing_bins [R,S_prev+1] (spacing domain),
    u [R,S+1] stratified draws (ray_samplers.py:316-327 builds them; see `pdf_u`).
    Returns (new bins [R,S+1], inds int64 [R,S+1], cdf [R,S_prev+1]).
    """
    w = weights + histogram_padding
    # sequential-order sum (module docstring); reference: torch.sum(weights, -1, keepdim=True) (:305)
    w_sum = torch.cumsum(w, dim=-1)[..., -1:]
    padding = torch.relu(eps - w_sum)
    w = w + padding / w.shape[-1]
    w_sum = w_sum + padding
    pdf = w / w_sum
    cdf = torch.minimum(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, side="right")
    hi = existing_bins.shape[-1] - 1
    below = torch.clamp(inds - 1, 0, hi)
    above = torch.clamp(inds, 0, hi)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(existing_bins, -1, below), torch.gather(existing_bins, -1, above)
    t = torch.clip(torch.nan_to_num((u - c0) / (c1 - c0), 0), 0, 1)
    bins = b0 + t * (b1 - b0)
    return bins.detach(), inds, cdf


def pdf_u(num_rays: int, num_samples: int, rand: Optional[torch.Tensor]):
    """u for the PDF sampler: ray_samplers.py:316-327. rand None = eval; else [R,S+1] or [R,1]."""
    nb = num_samples + 1
    u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb, device=rand.device if rand is not None else None)
    if rand is not None:
        u = u.expand(num_rays, nb) + rand / nb
    else:
        u = (u + 1.0 / (2 * nb)).expand(num_rays, nb)
    return u.contiguous()


def get_weights(deltas, densities):
    """RaySamples.get_weights, NS/cameras/rays.py:127-149. deltas, densities [R,S]."""
    dd = deltas * densities
    alphas = 1 - torch.exp(-dd)
    trans = torch.cumsum(dd[..., :-1], dim=-1)
    trans = torch.cat([torch.zeros_like(trans[..., :1]), trans], dim=-1)
    trans = torch.exp(-trans)
    return torch.nan_to_num(alphas * trans)


# ----------------------------------------------------------------------------------------------
# K-Planes interpolation + fields
# ----------------------------------------------------------------------------------------------
COO_COMBS = list(itertools.combinations(range(4), 2))  # kplanes_field.py:61-65: XY XZ XT YZ YT ZT


# True: `bilinear_plane` calls torch.nn.functional.grid_sample exactly as the reference's grid_sample_wrapper does
# (NS/utils/interpolation.py:5-33) instead of the explicit restatement below.  Used by oracle/torch_standin.py (the stock-PyTorch
# stand-in timed by bench.py) and by tests/test_oracle_golden.py, which checks the two routes against each other.
USE_GRID_SAMPLE = False


def bilinear_plane(plane, coords):
    """Bilinear, align_corners=True, padding 'border' sample of one plane.

    plane [1,C,H,W]; coords [N,2] = (x -> W axis, y -> H axis) in [-1,1].  Restates
    grid_sample_wrapper (NS/utils/interpolation.py:5-33) + ATen grid_sampler_2d semantics:
    ix = ((x+1)/2)*(W-1) clipped to [0,W-1]; corners floor/floor+1; out-of-range corners add 0.
    Returns [N,C].
    """
    if USE_GRID_SAMPLE:
        out = torch.nn.functional.grid_sample(plane, coords.view(1, -1, 1, 2), align_corners=True, mode="bilinear", padding_mode="border")
        return out[0, :, :, 0].t()  # [1,C,N,1] -> [N,C]
    _, C, H, W = plane.shape
    if plane.stride(1) == 1 and C > 1:
        return _bilinear_plane_rows(plane, coords)
    ix = ((coords[:, 0] + 1) / 2) * (W - 1)
    iy = ((coords[:, 1] + 1) / 2) * (H - 1)
    ix = torch.clamp(ix, 0, W - 1)
    iy = torch.clamp(iy, 0, H - 1)
    ix0, iy0 = torch.floor(ix), torch.floor(iy)
    ix1, iy1 = ix0 + 1, iy0 + 1
    w_nw = (ix1 - ix) * (iy1 - iy)
    w_ne = (ix - ix0) * (iy1 - iy)
    w_sw = (ix1 - ix) * (iy - iy0)
    w_se = (ix - ix0) * (iy - iy0)
    p = plane[0].permute(1, 2, 0)  # [H,W,C]

    def corner(iyc, ixc, w):
        ok = (ixc >= 0) & (ixc <= W - 1) & (iyc >= 0) & (iyc <= H - 1)
        v = p[iyc.clamp(0, H - 1).long(), ixc.clamp(0, W - 1).long()]
        return v * (w * ok)[:, None]

    return corner(iy0, ix0, w_nw) + corner(iy0, ix1, w_ne) + corner(iy1, ix0, w_sw) + corner(iy1, ix1, w_se)


def _bilinear_plane_rows(plane, coords):
    """The same restatement for a plane whose STORAGE is channel-last (a [1,C,H,W] view of an [H,W,C] tensor: oracle/torch_standin.py's
    `plane_layout="hwc"`): the four corner texels are ROWS of an [H*W, C] matrix, fetched with index_select -- whose backward is a row-wise index_add,
    128 contiguous bytes per texel -- instead of F.grid_sample's channel-strided backward (one 4-byte atomic per channel and corner, 4 MB apart at the
    finest scale).  Same arithmetic per output element (weights, products, order of the four additions) as the strided route above; only the order in
    which autograd sums a texel's gradient contributions differs.  Exists so that the stock-PyTorch stand-in of the reference's algorithm can finish
    30 000 steps inside one GPU call; checked against F.grid_sample by tests/test_standin_cpu.py."""
    _, C, H, W = plane.shape
    rows = plane[0].permute(1, 2, 0).reshape(H * W, C)  # a view: the storage is [H,W,C]
    ix = torch.clamp(((coords[:, 0] + 1) / 2) * (W - 1), 0, W - 1)
    iy = torch.clamp(((coords[:, 1] + 1) / 2) * (H - 1), 0, H - 1)
    ix0, iy0 = torch.floor(ix), torch.floor(iy)
    ix1, iy1 = ix0 + 1, iy0 + 1
    w_nw = (ix1 - ix) * (iy1 - iy)
    w_ne = (ix - ix0) * (iy1 - iy)
    w_sw = (ix1 - ix) * (iy - iy0)
    w_se = (ix - ix0) * (iy - iy0)

    def corner(iyc, ixc, w):
        ok = (ixc <= W - 1) & (iyc <= H - 1)  # the lower corners are >= 0 after the clamp; an upper corner beyond the border adds 0
        idx = iyc.clamp(0, H - 1).long() * W + ixc.clamp(0, W - 1).long()
        return rows.index_select(0, idx) * (w * ok)[:, None]

    return corner(iy0, ix0, w_nw) + corner(iy0, ix1, w_ne) + corner(iy1, ix0, w_sw) + corner(iy1, ix1, w_se)


def interpolate_kplanes(pts, ms_grids, concat_features: bool, freeze_time_planes: bool = False, freeze_space_planes: bool = False):
    """interpolate_kplanes, NS/fields/kplanes_field.py:77-126.

    pts [N,4]; ms_grids: list over scales of 6 planes [1,C,reso[b],reso[a]] for pair (a,b).
    Product over the 6 planes, then concat (or sum) over scales.  freeze_time_planes (:95-99): planes holding the time axis are skipped.
    freeze_space_planes (:101-116): a space plane is interpolated AND multiplied into the running product inside set_grad_enabled(False), which
    detaches everything accumulated so far -- only the planes after the last space plane (YT, ZT) keep a gradient.
    """
    outs = []
    for grids in ms_grids:
        prod = 1.0
        for ci, comb in enumerate(COO_COMBS):
            if freeze_time_planes and 3 in comb:
                continue
            v = bilinear_plane(grids[ci], pts[:, list(comb)])
            prod = (prod * v).detach() if (freeze_space_planes and 3 not in comb) else prod * v
        outs.append(prod)
    if concat_features:
        return torch.cat(outs, dim=-1)
    return sum(outs)


class _TruncExp(torch.autograd.Function):
    """NS/field_components/activations.py:25-41: exp fwd, g*exp(clamp(x,-15,15)) bwd."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def mlp(x, weights: Sequence[torch.Tensor], out_act: str = "None", hidden_act: str = "ReLU"):
    """Bias-free MLP standing in for tcnn.Network (FullyFusedMLP), e.g. kplanes_field.py:249-273.

    weights: list of [out,in] matrices.
    """
    for i, w in enumerate(weights):
        x = x @ w.t()
        last = i == len(weights) - 1
        if not last and hidden_act == "ReLU":
            x = torch.relu(x)
        if last and out_act == "Sigmoid":
            x = torch.sigmoid(x)
    return x


def scene_contraction_inf(x):
    """SceneContraction(order=inf), NS/field_components/spatial_distortions.py:62-69: identity inside the unit cube, (2 - 1/m) x/m with
    m = max |x_k| outside; the K-Planes fields halve the result into [-1, 1] (kplanes_field.py:278-280)."""
    mag = torch.linalg.norm(x, ord=float("inf"), dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def normalize_positions(positions, aabb):
    """SceneBox.get_normalized_positions, NS/data/scene_box.py:55-65."""
    return (positions - aabb[0]) / (aabb[1] - aabb[0])


def field_forward(positions, times, aabb, grids, sigma_w, color_w, geo_feat_dim: int = 15, **frozen):
    """KPlanesField.get_density + get_outputs, kplanes_field.py:275-358 with
    linear_decoder=False, disable_viewing_dependent=True, no appearance embedding, concat scales.

    positions [R,S,3], times [R,1].  Returns density [R,S], rgb [R,S,3].
    """
    R, S = positions.shape[:2]
    p = normalize_positions(positions, aabb) * 2.0 - 1.0  # :283-284
    t = (times * 2) - 1  # :290
    pts = torch.cat([p, t[:, None, :].expand(R, S, 1)], dim=-1).reshape(-1, 4)
    feats = interpolate_kplanes(pts, grids, concat_features=True, **frozen)
    h = mlp(feats, sigma_w)
    geo, dpre = h[:, :geo_feat_dim], h[:, geo_feat_dim:]
    density = trunc_exp(dpre).view(R, S)
    rgb = mlp(geo, color_w, out_act="Sigmoid").view(R, S, 3)
    return density, rgb


def field_forward_linear_decoder(positions, directions, times, aabb, grids, sigma_w, basis_w):
    """KPlanesField with linear_decoder=True (kplanes_field.py:219-246, :305-311, :349-354): the density is ONE linear layer on the
    interpolated features; the colour is sigmoid(features . basis_c) with a 3 x F basis produced by color_basis from the RAW direction.

    positions / directions [R,S,3], times [R,1]; sigma_w [[1,F]], basis_w the Linear-layout matrices of color_basis.  -> density [R,S], rgb [R,S,3].
    """
    R, S = positions.shape[:2]
    p = normalize_positions(positions, aabb) * 2.0 - 1.0
    t = (times * 2) - 1
    pts = torch.cat([p, t[:, None, :].expand(R, S, 1)], dim=-1).reshape(-1, 4)
    feats = interpolate_kplanes(pts, grids, concat_features=True)
    density = trunc_exp(mlp(feats, sigma_w, hidden_act="None")).view(R, S)
    basis = mlp(directions.reshape(-1, 3), basis_w).view(feats.shape[0], 3, -1)
    rgb = torch.sigmoid(torch.sum(feats[:, None, :] * basis, dim=-1)).view(R, S, 3)
    return density, rgb


def density_field_forward(positions, times, aabb, grids, sigma_w, hidden_act: str = "ReLU", **frozen):
    """KPlanesDensityField.density_fn/get_density, kplanes_field.py:410-460 (hidden_act "None": the proposal field of the linear decoder, :391-393).

    NOTE (behaviour, reproduced): positions are normalised to [0,1] and NOT rescaled to [-1,1]
    (:440), so only the upper quadrant of each proposal plane is sampled.
    """
    R, S = positions.shape[:2]
    p = normalize_positions(positions, aabb)
    t = (times * 2) - 1
    pts = torch.cat([p, t[:, None, :].expand(R, S, 1)], dim=-1).reshape(-1, 4)
    feats = interpolate_kplanes(pts, [grids], concat_features=False, **frozen)
    return trunc_exp(mlp(feats, sigma_w, hidden_act=hidden_act)).view(R, S)


# ----------------------------------------------------------------------------------------------
# renderers
# ----------------------------------------------------------------------------------------------
def render_rgb(rgb, weights, background, training: bool):
    """RGBRenderer.forward/combine_rgb, NS/model_components/renderers.py:69-140.

    rgb [R,S,3], weights [R,S]; background: [R,3] tensor (the 'random' draw made explicit),
    'last_sample', or a [3] colour.
    """
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = torch.sum(weights[..., None] * rgb, dim=-2)
    acc = torch.sum(weights, dim=-1, keepdim=True)
    bg = rgb[..., -1, :] if isinstance(background, str) and background == "last_sample" else background
    comp = comp + bg * (1.0 - acc)
    if not training:
        comp = torch.clamp(comp, 0.0, 1.0)
    return comp


def render_accumulation(weights):
    """AccumulationRenderer, renderers.py:200-223."""
    return torch.sum(weights, dim=-1, keepdim=True)


def median_index(weights):
    """searchsorted(cumsum(w), 0.5, left) clamped; renderers.py:264-267 and :312-315."""
    cw = torch.cumsum(weights, dim=-1)
    split = torch.full((weights.shape[0], 1), 0.5, device=weights.device)
    idx = torch.searchsorted(cw, split, side="left")
    return torch.clamp(idx, 0, weights.shape[-1] - 1)


def render_depth_median(weights, starts, ends):
    """DepthRenderer('median'), renderers.py:260-270."""
    steps = (starts + ends) / 2
    return torch.gather(steps, -1, median_index(weights))


def render_depth_expected(weights, starts, ends):
    """DepthRenderer('expected'), renderers.py:271-285."""
    steps = (starts + ends) / 2
    d = torch.sum(weights * steps, dim=-1, keepdim=True) / (torch.sum(weights, -1, keepdim=True) + 1e-10)
    return torch.clip(d, steps.min(), steps.max())


def render_median_rgb(rgb, weights, training: bool):
    """MedianRGBRenderer, renderers.py:301-362.  Returns [R,1,3] (reference quirk)."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    idx = median_index(weights)[..., None].expand(-1, -1, 3)
    out = torch.gather(rgb, dim=-2, index=idx)
    if not training:
        out = torch.clamp(out, 0.0, 1.0)
    return out


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    """losses.py:46-75."""
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    n = y1.shape[-1] - 1
    lo = torch.clamp(torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1, 0, n)
    hi = torch.clamp(torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right"), 0, n)
    return torch.take_along_dim(cy1[..., 1:], hi, dim=-1) - torch.take_along_dim(cy1[..., :-1], lo, dim=-1)


def lossfun_outer(t, w, t_env, w_env):
    """losses.py:78-95."""
    w_outer = outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + EPS)


def interlevel_loss(weights_list, sdist_list):
    """losses.py:106-121.  weights_list[i] [R,S_i]; sdist_list[i] [R,S_i+1] (spacing bins)."""
    c = sdist_list[-1].detach()
    w = weights_list[-1].detach()
    total = 0.0
    for sd, wp in zip(sdist_list[:-1], weights_list[:-1]):
        total = total + torch.mean(lossfun_outer(c, w, sd, wp))
    return total


def lossfun_distortion(t, w):
    """losses.py:125-136."""
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    intra = torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return inter + intra


def distortion_loss(weights, sdist):
    """losses.py:139-144 (on the last, i.e. nerf, level)."""
    return torch.mean(lossfun_distortion(sdist, weights))


def ds_nerf_depth_loss(weights, termination_depth, steps, lengths, sigma):
    """NS/model_components/losses.py:213-235.  weights, steps, lengths [R,S]; termination_depth [R]; sigma scalar."""
    mask = (termination_depth > 0).to(weights.dtype)
    loss = -torch.log(weights + EPS) * torch.exp(-((steps - termination_depth[:, None]) ** 2) / (2 * sigma)) * lengths
    return torch.mean(loss.sum(-1) * mask)


def urf_depth_loss(weights, ebins, termination_depth, predicted_depth, sigma, directions_norm=None, is_euclidean: bool = True):
    """depth_loss, URF branch = urban_radiance_field_depth_loss (NS/model_components/losses.py:238-274,308-309).  weights [R,S], ebins
    [R,S+1], termination_depth / predicted_depth [R]."""
    D = termination_depth if is_euclidean else termination_depth * directions_norm
    steps = (ebins[:, :-1] + ebins[:, 1:]) / 2
    mask = (D > 0).to(weights.dtype)
    expected = (D - predicted_depth) ** 2
    sd = sigma / 3.0  # URF_SIGMA_SCALE_FACTOR, losses.py:36
    x = steps - D[:, None]
    pdf = torch.exp(-(x**2) / (2 * sd**2) - math.log(sd) - 0.5 * math.log(2 * math.pi))
    near = ((steps <= D[:, None] + sigma) & (steps >= D[:, None] - sigma)).to(weights.dtype)
    empty = (steps < D[:, None] - sigma).to(weights.dtype)
    los = (near * (weights - pdf) ** 2).sum(-1) + (empty * weights**2).sum(-1)
    return torch.mean((expected + los) * mask)


def depth_loss(weights, ebins, termination_depth, sigma, directions_norm=None, is_euclidean: bool = True):
    """depth_loss, DS_NERF branch (losses.py:261-311): z-distance maps are scaled by the ray direction norms (:302-303); steps are the bin
    centres (:304), lengths the bin widths (:307).  ebins [R,S+1]."""
    if not is_euclidean:
        termination_depth = termination_depth * directions_norm
    starts, ends = ebins[:, :-1], ebins[:, 1:]
    return ds_nerf_depth_loss(weights, termination_depth, (starts + ends) / 2, ends - starts, sigma)


def plane_tv(t, only_w: bool = False):
    """compute_plane_tv, losses.py:356-366."""
    h_tv = torch.square(t[..., 1:, :] - t[..., :-1, :]).mean()
    w_tv = torch.square(t[..., :, 1:] - t[..., :, :-1]).mean()
    return w_tv if only_w else h_tv + w_tv


def plane_smoothness(t):
    """compute_plane_smoothness, losses.py:369-380 (second difference along H = time)."""
    d1 = t[..., 1:, :] - t[..., :-1, :]
    d2 = d1[..., 1:, :] - d1[..., :-1, :]
    return torch.square(d2).mean()


SPACE_PLANES = (0, 1, 3)
TIME_PLANES = (2, 4, 5)


def space_tv_loss(ms_grids):
    """losses.py:383-406."""
    total = 0.0
    for grids in ms_grids:
        for gi, g in enumerate(grids):
            total = total + plane_tv(g, only_w=gi not in SPACE_PLANES)
    return total


def time_smoothness_loss(ms_grids):
    """losses.py:409-428."""
    total = 0.0
    for grids in ms_grids:
        for gi in TIME_PLANES:
            total = total + plane_smoothness(grids[gi])
    return total


def sparse_transients_loss(ms_grids):
    """losses.py:431-452."""
    total = 0.0
    for grids in ms_grids:
        for gi in TIME_PLANES:
            total = total + torch.abs(1 - grids[gi]).mean()
    return total


# ----------------------------------------------------------------------------------------------
# model: one K-Planes training forward (all random draws are explicit inputs)
# ----------------------------------------------------------------------------------------------
DEFAULT_LOSS_COEF = {  # NS/configs/method_configs.py:530-541 (k-planes preset)
    "rgb_loss": 1.0,
    "interlevel_loss": 1.0,
    "distortion_loss": 0.001,
    "space_tv_loss": 0.0002,
    "time_smoothness_loss": 0.001,
    "sparse_transients_loss": 0.0001,
    "space_tv_proposal_loss": 0.0002,
    "time_smoothness_proposal_loss": 0.00001,
    "sparse_transients_proposal_loss": 0.0001,
}


def anneal_value(step: int, max_iters: int = 1000, slope: float = 10.0) -> float:
    """set_anneal callback, NS/models/kplanes.py:326-331."""
    frac = min(max(step / max_iters, 0.0), 1.0)
    return (slope * frac) / ((slope - 1) * frac + 1)


def update_schedule(step: int, warmup: int = 5000, every: int = 5) -> float:
    """NS/models/kplanes.py:254-259: clip(interp(step,[0,warmup],[0,every]),1,every)."""
    v = every * min(max(step / warmup, 0.0), 1.0)
    return min(max(v, 1.0), float(every))


def kplanes_forward(
    params: Dict,
    rays: Dict,
    rng: Dict,
    num_proposal_samples: Sequence[int] = (256, 128),
    num_nerf_samples: int = 64,
    anneal: float = 1.0,
    training: bool = True,
    proposal_requires_grad: bool = True,
    near_plane: float = 0.0,  # kplanes.py:276-277 builds AABBBoxCollider with its default near_plane=0.0
    single_jitter: bool = False,
):
    """KPlanesModel.forward = collider + get_outputs, NS/models/kplanes.py:349-388 via
    ProposalNetworkSampler.generate_ray_samples (ray_samplers.py:559-600).

    params: {"aabb","field_grids":[scale][6],"field_sigma":[..],"field_color":[..],
             "prop_grids":[level][6],"prop_sigma":[level][..]}
    rays:   {"origins","directions","times"} ([R,3],[R,3],[R,1])
    rng (training): {"t_rand":[R,S0+1], "u":[ [R,S1+1], [R,S2+1] ], "bg":[R,3]}  (uniform draws)
    """
    aabb = params["aabb"]
    o, d, times = rays["origins"], rays["directions"], rays["times"]
    R = o.shape[0]
    nears, fars = intersect_aabb(o, d, aabb, near_plane, training)
    levels = list(num_proposal_samples) + [num_nerf_samples]
    weights_list, sdist_list, eucl_list = [], [], []
    weights = None
    bins = None
    for li, S in enumerate(levels):
        if li == 0:
            bins = spaced_bins(R, S, rng["t_rand"] if training else None).to(o.device)
        else:
            annealed = torch.pow(weights, anneal)  # ray_samplers.py:584
            u = pdf_u(R, S, rng["u"][li - 1] if training else None).to(o.device)
            bins, _, _ = pdf_sample(annealed, bins, u)
        eucl = spacing_to_euclidean(bins, nears, fars)
        starts, ends = eucl[:, :-1], eucl[:, 1:]
        pos = sample_positions(o, d, starts, ends)
        if li < len(levels) - 1:
            ctx = torch.enable_grad() if proposal_requires_grad else torch.no_grad()
            with ctx:
                dens = density_field_forward(pos, times, aabb, params["prop_grids"][li], params["prop_sigma"][li])
            weights = get_weights(ends - starts, dens)
            weights_list.append(weights)
            sdist_list.append(bins)
            eucl_list.append(eucl)
    density, rgb = field_forward(pos, times, aabb, params["field_grids"], params["field_sigma"], params["field_color"])
    weights = get_weights(ends - starts, density)
    weights_list.append(weights)
    sdist_list.append(bins)
    eucl_list.append(eucl)
    bg = rng["bg"] if training else "last_sample"
    out = {
        "rgb": render_rgb(rgb, weights, bg, training),
        "accumulation": render_accumulation(weights),
        "depth": render_depth_median(weights, starts, ends),
        "median_rgb": render_median_rgb(rgb, weights, training),
        "weights_list": weights_list,
        "sdist_list": sdist_list,
        "eucl_list": eucl_list,
        "nears": nears,
        "fars": fars,
        "field_rgb": rgb,
        "field_density": density,
    }
    for i in range(len(levels) - 1):
        e = eucl_list[i]
        out[f"prop_depth_{i}"] = render_depth_median(weights_list[i], e[:, :-1], e[:, 1:])
    return out


def kplanes_loss_dict(params, out, target_rgb, coef: Dict = DEFAULT_LOSS_COEF, training: bool = True):
    """KPlanesModel.get_loss_dict, NS/models/kplanes.py:414-452 + misc.scale_dict (utils/misc.py:116-130)."""
    ld = {"rgb_loss": torch.mean((target_rgb - out["rgb"]) ** 2)}
    if training:
        ld["distortion_loss"] = distortion_loss(out["weights_list"][-1], out["sdist_list"][-1])
        ld["interlevel_loss"] = interlevel_loss(out["weights_list"], out["sdist_list"])
        fg, pg = params["field_grids"], params["prop_grids"]
        ld["space_tv_loss"] = space_tv_loss(fg)
        ld["space_tv_proposal_loss"] = space_tv_loss(pg)
        ld["sparse_transients_loss"] = sparse_transients_loss(fg)
        ld["sparse_transients_proposal_loss"] = sparse_transients_loss(pg)
        ld["time_smoothness_loss"] = time_smoothness_loss(fg)
        ld["time_smoothness_proposal_loss"] = time_smoothness_loss(pg)
    return {k: v * coef[k] if k in coef else v for k, v in ld.items()}


# ----------------------------------------------------------------------------------------------
# optimiser / schedule (NS/engine/optimizers.py, schedulers.py:126-141; values method_configs.py:546-557)
# ----------------------------------------------------------------------------------------------
def cosine_lr_factor(step: int, warm_up_end: int = 512, max_steps: int = 30000, alpha: float = 0.0) -> float:
    """CosineDecayScheduler, NS/engine/schedulers.py:126-141."""
    if step < warm_up_end:
        return step / warm_up_end
    progress = (step - warm_up_end) / (max_steps - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


def adam_step(p, g, m, v, step: int, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-12):
    """torch.optim.Adam (no weight decay, no amsgrad) single-tensor update, 1-based `step`."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------------------------
# parameter construction (init follows kplanes_field.py:47-74, :396; tcnn init is upstream-defined)
# ----------------------------------------------------------------------------------------------
def make_planes(C: int, reso: Sequence[int], a: float, b: float, gen: torch.Generator):
    planes = []
    for comb in COO_COMBS:
        shape = [1, C, reso[comb[1]], reso[comb[0]]]
        if 3 in comb:
            planes.append(torch.ones(shape))
        else:
            planes.append(torch.rand(shape, generator=gen) * (b - a) + a)
    return planes


def make_mlp(dims: Sequence[int], gen: torch.Generator):
    ws = []
    for i in range(len(dims) - 1):
        bound = math.sqrt(6.0 / (dims[i] + dims[i + 1]))
        ws.append((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound)
    return ws


def make_kplanes_params(
    base_res=(64, 64, 64, 8),
    multiscale=(1,),
    feat_dim=32,
    prop_res=((128, 128, 128, 8), (256, 256, 256, 8)),
    prop_feat=8,
    sigma_hidden=128,
    color_hidden=64,
    aabb_scale=1.5,
    seed=0,
):
    gen = torch.Generator().manual_seed(seed)
    fg = []
    for m in multiscale:
        reso = [r * m for r in base_res[:3]] + [base_res[3]]
        fg.append(make_planes(feat_dim, reso, 0.1, 0.5, gen))
    return {
        "aabb": torch.tensor([[-aabb_scale] * 3, [aabb_scale] * 3], dtype=torch.float32),
        "field_grids": fg,
        "field_sigma": make_mlp([feat_dim * len(multiscale), sigma_hidden, 16], gen),
        "field_color": make_mlp([15, color_hidden, color_hidden, 3], gen),
        "prop_grids": [make_planes(prop_feat, r, 0.1, 0.15, gen) for r in prop_res],
        "prop_sigma": [make_mlp([prop_feat, 64, 1], gen) for _ in prop_res],
    }


def all_param_tensors(params) -> List[torch.Tensor]:
    out = []
    for g in params["field_grids"]:
        out += list(g)
    out += list(params["field_sigma"]) + list(params["field_color"])
    for g in params["prop_grids"]:
        out += list(g)
    for w in params["prop_sigma"]:
        out += list(w)
    return out

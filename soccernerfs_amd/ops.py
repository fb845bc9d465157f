"""torch.autograd wrappers over the libsnerf C ABI.  Every op runs on the current torch HIP stream,
on caller-allocated torch tensors (device memory plumbing only -- the kernels are in csrc/)."""
import ctypes as C

import torch

from . import _lib
from .plane_set import PlaneSet


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a HIP device tensor (the HIP library is the only product path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def coords_from_points(pts: torch.Tensor) -> _lib.Coords:
    c = _lib.Coords()
    c.mode = 0
    c.pts = pts.data_ptr()
    return c


def coords_from_rays(origins, dirs, times, ebins, aabb, rescale: bool) -> _lib.Coords:
    c = _lib.Coords()
    c.mode = 1
    c.S = ebins.shape[-1] - 1
    c.rescale = int(rescale)
    c.origins, c.dirs, c.times, c.ebins = origins.data_ptr(), dirs.data_ptr(), times.data_ptr(), ebins.data_ptr()
    a = aabb.detach().cpu().tolist() if isinstance(aabb, torch.Tensor) else aabb
    for k in range(3):
        c.aabb_min[k] = a[0][k]
        c.aabb_max[k] = a[1][k]
    return c


class _KPlanesGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, planes, ps: PlaneSet, coords_keepalive, coords: _lib.Coords, N: int):
        out = torch.empty(N, ps.out_dim, dtype=torch.float32, device=planes.device)
        desc = ps.desc()
        _lib.check(_lib.lib().snerf_kplanes_gather_fwd(C.byref(desc), _ptr(planes), C.byref(coords), C.c_int64(N), _ptr(out), _stream()),
                   "kplanes_gather_fwd")
        ctx.ps, ctx.coords, ctx.keep, ctx.N = ps, coords, coords_keepalive, N
        ctx.save_for_backward(planes)
        return out

    @staticmethod
    def backward(ctx, gout):
        (planes,) = ctx.saved_tensors
        gout = gout.contiguous()
        gplanes = torch.zeros_like(planes)
        desc = ctx.ps.desc()
        _lib.check(_lib.lib().snerf_kplanes_gather_bwd(C.byref(desc), _ptr(planes), C.byref(ctx.coords), C.c_int64(ctx.N), _ptr(gout),
                                                       _ptr(gplanes), _stream()), "kplanes_gather_bwd")
        return gplanes, None, None, None, None


def interpolate_kplanes(pts: torch.Tensor, plane_set: PlaneSet) -> torch.Tensor:
    """Drop-in for interpolate_kplanes(pts, ms_grids, concat_features, ...) (NS/fields/kplanes_field.py:77-126).
    pts [N,4] in [-1,1]; returns [N, C*n_scales] (concat) or [N, C]."""
    pts = _f32c(pts, "pts")
    return _KPlanesGather.apply(plane_set.planes, plane_set, (pts,), coords_from_points(pts), pts.shape[0])


def interpolate_kplanes_rays(plane_set: PlaneSet, origins, dirs, times, ebins, aabb, rescale: bool) -> torch.Tensor:
    """Same gather with sample coordinates derived in-kernel from rays + euclidean bin edges [R,S+1]."""
    origins, dirs, ebins = _f32c(origins, "origins"), _f32c(dirs, "dirs"), _f32c(ebins, "ebins")
    times = _f32c(times, "times").reshape(-1)
    R, S = ebins.shape[0], ebins.shape[1] - 1
    c = coords_from_rays(origins, dirs, times, ebins, aabb, rescale)
    return _KPlanesGather.apply(plane_set.planes, plane_set, (origins, dirs, times, ebins), c, R * S)


# ----------------------------------------------------------------------------------------------
# per-ray sampling ops
# ----------------------------------------------------------------------------------------------
SPACING_KIND = {"uniform": 0, "piecewise": 1}


def spaced_bins(nears, fars, num_samples: int, t_rand=None, kind: str = "uniform"):
    """SpacedSampler bins (ray_samplers.py:79-126). nears/fars [R] or [R,1]; t_rand None | [R,S+1] | [R,1].
    Returns (sbins, ebins) [R,S+1]."""
    nears, fars = _f32c(nears, "nears").reshape(-1), _f32c(fars, "fars").reshape(-1)
    R, S = nears.shape[0], num_samples
    sb = torch.empty(R, S + 1, dtype=torch.float32, device=nears.device)
    eb = torch.empty_like(sb)
    cols = 0
    if t_rand is not None:
        t_rand = _f32c(t_rand, "t_rand")
        cols = t_rand.shape[-1]
    _lib.check(_lib.lib().snerf_spaced_bins(_ptr(nears), _ptr(fars), _ptr(t_rand) if t_rand is not None else None, cols, R, S,
                                            SPACING_KIND[kind], _ptr(sb), _ptr(eb), _stream()), "spaced_bins")
    return sb, eb


class _Weights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, density, ebins):
        R, S = density.shape
        w = torch.empty_like(density)
        _lib.check(_lib.lib().snerf_weights_fwd(_ptr(density), _ptr(ebins), R, S, _ptr(w), _stream()), "weights_fwd")
        ctx.save_for_backward(density, ebins)
        return w

    @staticmethod
    def backward(ctx, gw):
        density, ebins = ctx.saved_tensors
        R, S = density.shape
        gw = gw.contiguous()
        gd = torch.empty_like(density)
        _lib.check(_lib.lib().snerf_weights_bwd(_ptr(density), _ptr(ebins), _ptr(gw), R, S, _ptr(gd), 0, _stream()), "weights_bwd")
        return gd, None


def get_weights(density, ebins):
    """RaySamples.get_weights (NS/cameras/rays.py:127-149). density [R,S], ebins [R,S+1] -> [R,S]."""
    return _Weights.apply(_f32c(density, "density"), _f32c(ebins, "ebins"))


def pdf_resample(sbins_prev, nears, fars, num_samples: int, weights=None, density=None, ebins_prev=None, u=None, rand=None,
                 anneal: float = 1.0, kind: str = "uniform", histogram_padding: float = 0.01, eps: float = 1e-5,
                 return_inds: bool = False, return_weights: bool = False):
    """PDFSampler (ray_samplers.py:274-369, include_original=False) with annealing (:584).
    Exactly one of `weights` [R,Sp] or (`density`, `ebins_prev`); exactly one of u [R,S+1] (explicit),
    rand [R,S+1]|[R,1] (training draws) or neither (eval).  Returns (sbins, ebins[, inds][, weights])."""
    sbins_prev = _f32c(sbins_prev, "sbins_prev")
    nears, fars = _f32c(nears, "nears").reshape(-1), _f32c(fars, "fars").reshape(-1)
    R, Sp, S = sbins_prev.shape[0], sbins_prev.shape[1] - 1, num_samples
    dev = sbins_prev.device
    a = _lib.ResampleArgs()
    keep = []
    if density is not None:
        density, ebins_prev = _f32c(density, "density"), _f32c(ebins_prev, "ebins_prev")
        a.density, a.ebins_prev = density.data_ptr(), ebins_prev.data_ptr()
        keep += [density, ebins_prev]
    else:
        weights = _f32c(weights.detach(), "weights")
        a.weights_in = weights.data_ptr()
        keep.append(weights)
    wout = None
    if return_weights:
        wout = torch.empty(R, Sp, dtype=torch.float32, device=dev)
        a.weights_out = wout.data_ptr()
    if u is not None:
        u = _f32c(u, "u")
        a.u_mode, a.u_or_rand = 0, u.data_ptr()
    elif rand is not None:
        rand = _f32c(rand, "rand")
        a.u_mode, a.u_or_rand, a.rand_cols = 1, rand.data_ptr(), rand.shape[-1]
    else:
        a.u_mode = 2
    sb = torch.empty(R, S + 1, dtype=torch.float32, device=dev)
    eb = torch.empty_like(sb)
    inds = torch.empty(R, S + 1, dtype=torch.int64, device=dev) if return_inds else None
    a.sbins_prev, a.nears, a.fars = sbins_prev.data_ptr(), nears.data_ptr(), fars.data_ptr()
    a.sbins_out, a.ebins_out = sb.data_ptr(), eb.data_ptr()
    a.inds_out = inds.data_ptr() if inds is not None else None
    a.R, a.S_prev, a.S, a.kind = R, Sp, S, SPACING_KIND[kind]
    a.anneal, a.histogram_padding, a.eps = anneal, histogram_padding, eps
    _lib.check(_lib.lib().snerf_pdf_resample(C.byref(a), _stream()), "pdf_resample")
    out = [sb, eb]
    if return_inds:
        out.append(inds)
    if return_weights:
        out.append(wout)
    return tuple(out)


# ----------------------------------------------------------------------------------------------
# tiny MLP
# ----------------------------------------------------------------------------------------------
class _MLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, desc, aux_col, need_gx):
        N = x.shape[0]
        y = torch.empty(N, desc.d_out, dtype=torch.float32, device=x.device)
        aux = torch.empty(N, dtype=torch.float32, device=x.device) if aux_col >= 0 else None
        _lib.check(_lib.lib().snerf_mlp_fwd(C.byref(desc), _ptr(params), _ptr(x), x.stride(0), C.c_int64(N), _ptr(y), desc.d_out,
                                            aux_col, _ptr(aux) if aux is not None else None, _stream()), "mlp_fwd")
        ctx.desc, ctx.aux_col, ctx.need_gx = desc, aux_col, need_gx
        ctx.save_for_backward(x, params)
        if aux is None:
            return y
        return y, aux

    @staticmethod
    def backward(ctx, gy, gaux=None):
        x, params = ctx.saved_tensors
        desc = ctx.desc
        N = x.shape[0]
        gy = gy.contiguous() if gy is not None else None
        gaux = gaux.contiguous() if gaux is not None else None
        gx = torch.empty(N, desc.d_in, dtype=torch.float32, device=x.device) if ctx.need_gx else None
        gw = torch.zeros_like(params)
        _lib.check(_lib.lib().snerf_mlp_bwd(C.byref(desc), _ptr(params), _ptr(x), x.stride(0), C.c_int64(N),
                                            _ptr(gy) if gy is not None else None, desc.d_out, ctx.aux_col,
                                            _ptr(gaux) if gaux is not None else None,
                                            _ptr(gx) if gx is not None else None, desc.d_in, _ptr(gw), _stream()), "mlp_bwd")
        return gx, gw, None, None, None


def mlp_forward(x, params, desc: _lib.MlpDesc, aux_col: int = -1):
    """x [N,d_in] (unit inner stride; row stride may exceed d_in), params flat -> y [N,d_out] (and exp(raw[:,aux_col]))."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("mlp: expected a float32 HIP device tensor (the HIP library is the only product path)")
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.reshape(-1, x.shape[-1]).contiguous()
    return _MLP.apply(x, params, desc, aux_col, x.requires_grad)

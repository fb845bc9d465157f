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

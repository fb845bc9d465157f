"""torch.autograd wrappers over the libsnerf C ABI.  Every op runs on the current torch HIP stream,
on caller-allocated torch tensors (device memory plumbing only -- the kernels are in csrc/)."""
import ctypes as C

from typing import Optional

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


def aabb_host(aabb):
    """[[min x,y,z],[max x,y,z]] as host floats.  A device tensor (the nerfstudio-shaped modules hold scene_box.aabb as an nn.Parameter) is
    copied to the host ONCE and the copy is kept on the tensor object -- every later call would otherwise be a device-to-host
    synchronisation in the middle of the stream pipeline (four per model forward).  The cache follows in-place edits through `_version`."""
    if not isinstance(aabb, torch.Tensor):
        return aabb
    cached = getattr(aabb, "_snerf_host", None)
    if cached is None or cached[0] != aabb._version:
        cached = (aabb._version, aabb.detach().cpu().tolist())
        aabb._snerf_host = cached
    return cached[1]


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
    a = aabb_host(aabb)
    for k in range(3):
        c.aabb_min[k] = a[0][k]
        c.aabb_max[k] = a[1][k]
    return c


class _KPlanesGather(torch.autograd.Function):
    """freeze: 0 none; bit 0 = freeze_time_planes (the time planes are skipped: the gather runs on the static-scene view of the buffer); bit 1 =
    freeze_space_planes (kplanes_field.py:101-116: the space planes are interpolated with autograd off AND multiplied into the running product with
    autograd off, so everything up to the last space plane -- XY, XZ, XT, YZ -- is cut off; only YT and ZT receive a gradient.  Reproduced)."""

    @staticmethod
    def forward(ctx, planes, ps: PlaneSet, coords_keepalive, coords: _lib.Coords, N: int, freeze: int = 0):
        out = torch.empty(N, ps.out_dim, dtype=torch.float32, device=planes.device)
        desc = ps.space_desc() if freeze & 1 else ps.desc()
        _lib.check(_lib.lib().snerf_kplanes_gather_fwd(C.byref(desc), _ptr(planes), C.byref(coords), C.c_int64(N), _ptr(out), _stream()),
                   "kplanes_gather_fwd")
        ctx.ps, ctx.coords, ctx.keep, ctx.N, ctx.freeze, ctx.desc = ps, coords, coords_keepalive, N, freeze, desc
        ctx.save_for_backward(planes)
        return out

    @staticmethod
    def backward(ctx, gout):
        (planes,) = ctx.saved_tensors
        ps = ctx.ps
        if ctx.freeze & 2 and (ctx.freeze & 1 or ps.n_coords == 3):  # only space planes take part and they are frozen
            return None, None, None, None, None, None
        gout = gout.contiguous()
        gplanes = torch.zeros_like(planes)
        _lib.check(_lib.lib().snerf_kplanes_gather_bwd(C.byref(ctx.desc), _ptr(planes), C.byref(ctx.coords), C.c_int64(ctx.N), _ptr(gout),
                                                       _ptr(gplanes), _stream()), "kplanes_gather_bwd")
        if ctx.freeze & 2:
            for s in range(len(ps.resolutions)):
                gplanes[ps.offsets[s][0]: ps.offsets[s][4]].zero_()  # XY XZ XT YZ are contiguous in the buffer
        return gplanes, None, None, None, None, None


class SortedScatter:
    """Workspace + driver of the sorted plane-gradient scatter (csrc/kplanes_sorted.hip) for a fixed sample count N."""

    def __init__(self, ps: PlaneSet, N: int, device, gvec_dtype: torch.dtype = torch.float32, quotient: bool = False,
                 fix_capacity: Optional[int] = None):
        """gvec_dtype: element type of the per-plane gradient vectors between pass A and pass B -- float32 (exact) or bfloat16 (half the
        bytes of the step's largest intermediate; the scatter accumulates in fp32 either way).  quotient: the quotient form
        (scatter_quotient: one [N, C n_scales] tensor instead of the vectors; C = 32, concatenated scales) -- the vector buffer is then
        not allocated."""
        if gvec_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("gvec_dtype must be torch.float32 or torch.bfloat16")
        self.ps, self.N, self.desc = ps, N, ps.desc()
        self.gvec_bf16 = int(gvec_dtype == torch.bfloat16)
        hc, ie = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().snerf_kplanes_sort_workspace(C.byref(self.desc), C.c_int64(N), C.byref(hc), C.byref(ie)), "sort_workspace")
        self.hist = torch.empty(hc.value, dtype=torch.int32, device=device)
        self.rank = torch.empty(ie.value, dtype=torch.int32, device=device)
        self.sorted_rec = torch.empty(ie.value, 4, dtype=torch.float32, device=device)  # {sample id bits, pixel x, pixel y, 0}
        self.quotient = bool(quotient)
        if self.quotient:
            if not _lib.lib().snerf_kplanes_quotient_supported(C.byref(self.desc), C.c_int64(N)):
                raise ValueError("the quotient scatter is built for C = 32, concatenated scales and N * 32 * n_scales < 2^31")
            rows = N * len(ps.resolutions)
            self.G = torch.empty(N, ps.out_dim, dtype=torch.float32, device=device)       # gfeat .* feat
            # fix list: {element index, feature gradient} per vanished feature with a non-zero gradient (2 int32 per entry).  In training it
            # stays empty (planes initialised in [0.1, 0.5] / [1, 1]: no feature is exactly 0), but imported planes (a checkpoint, restart(params),
            # all-zero planes) can make EVERY feature vanish, and an entry that does not fit is a lost gradient term.  (r06, ADVICE) The default
            # therefore holds the worst case again -- N x C n_scales entries: 335 MB at the preset, 0.1 % of the HBM -- so that no input can
            # overflow it; a smaller fix_capacity is an explicit choice of a memory-constrained caller, and then the fix-up kernel records the
            # demanded count in fix_peak and check_fix_overflow() / the trainer (every optimizer_step) raise on it.
            worst = max(N * ps.out_dim, 1)
            self.fix_capacity = worst if fix_capacity is None else min(max(int(fix_capacity), 1), worst)
            self.fix_list = torch.empty(2 * self.fix_capacity, dtype=torch.int32, device=device)
            self.fix_peak = torch.zeros(1, dtype=torch.int32, device=device)  # sticky: the largest entry count that did not fit
            self.fix_counts = torch.zeros(2, dtype=torch.int32, device=device)  # used alternately: a prepare resets the other one for the next step
            self._fix_parity = 0
            self.fix_count = self.fix_counts[0:1]
            self.gvec = None
        else:
            self.gvec = torch.empty(len(ps.resolutions) * ie.value * ps.C, dtype=gvec_dtype, device=device)  # [scale*planes+plane][N][C]

    def sort(self, coords: _lib.Coords, stream=None):
        _lib.check(_lib.lib().snerf_kplanes_sort_samples(C.byref(self.desc), C.byref(coords), C.c_int64(self.N), _ptr(self.hist), _ptr(self.rank),
                                                         _ptr(self.sorted_rec), stream if stream is not None else _stream()), "sort_samples")

    def scatter(self, planes, coords: _lib.Coords, gout, gplanes, stream=None):
        st = stream if stream is not None else _stream()
        _lib.check(_lib.lib().snerf_kplanes_gradvec(C.byref(self.desc), _ptr(planes), C.byref(coords), C.c_int64(self.N), _ptr(gout), _ptr(self.gvec), self.gvec_bf16, st),
                   "gradvec")
        _lib.check(_lib.lib().snerf_kplanes_scatter_sorted(C.byref(self.desc), C.c_int64(self.N), _ptr(self.gvec), self.gvec_bf16, _ptr(self.sorted_rec), _ptr(gplanes), st),
                   "scatter_sorted")


    # ---- quotient form: g_q = (gfeat .* feat) ./ v_q (include/snerf.h) ----
    def next_fix_counter(self) -> int:
        """The two list counters are used alternately (the producer of G resets the OTHER one for the next step): makes counter k the current
        one (self.fix_count) and returns k."""
        k = self._fix_parity
        self._fix_parity = 1 - k
        self.fix_count = self.fix_counts[k:k + 1]
        return k

    def quotient_prepare(self, gfeat, feat, stream=None):
        st = stream if stream is not None else _stream()
        k = self.next_fix_counter()
        _lib.check(_lib.lib().snerf_kplanes_quotient_prepare(C.byref(self.desc), C.c_int64(self.N), _ptr(gfeat), _ptr(feat), _ptr(self.G), _ptr(self.fix_list),
                                                             self.fix_capacity, _ptr(self.fix_count), _ptr(self.fix_counts[1 - k:2 - k]), st),
                   "quotient_prepare")

    def quotient_fixup_scales(self, planes, coords: _lib.Coords, gplanes, scale_begin: int, scale_end: int, stream=None):
        """The exact terms of the listed (vanished-feature) elements for scales [scale_begin, scale_end): reads the sample coordinates, so it belongs on
        the stream that owns the ray buffers."""
        st = stream if stream is not None else _stream()
        _lib.check(_lib.lib().snerf_kplanes_quotient_fixup(C.byref(self.desc), _ptr(planes), C.byref(coords), C.c_int64(self.N), _ptr(self.fix_list),
                                                           _ptr(self.fix_count), self.fix_capacity, _ptr(gplanes), scale_begin, scale_end,
                                                           _ptr(self.fix_peak), st),
                   "quotient_fixup")

    def check_fix_overflow(self, peak: Optional[int] = None):
        """Raises if a fix-up ever found more listed entries than the list holds (their gradient terms were lost).  Reads the device
        counter (synchronises) unless the caller hands over a value it copied itself."""
        if not self.quotient:
            return
        if self.fix_capacity >= self.N * self.ps.out_dim:
            return  # worst-case list (the default): cannot overflow
        peak = int(self.fix_peak.item()) if peak is None else int(peak)
        if peak > self.fix_capacity:
            raise RuntimeError(f"quotient scatter: {peak} vanished-feature entries in one step but the fix list holds {self.fix_capacity}: the gradient terms "
                               f"of the rest were dropped.  Construct SortedScatter / the trainer with fix_capacity >= {peak} (worst case N * C * n_scales = "
                               f"{self.N * self.ps.out_dim}), or use the product-form scatter (quotient_scatter=False) for planes with this many exact zeros")

    def clear_fix_overflow(self):
        """Forget a recorded overflow (the counter is sticky): restart() / a parameter import start a new run."""
        if self.quotient:
            self.fix_peak.zero_()

    def quotient_pass_b_scales(self, planes, gplanes, scale_begin: int, scale_end: int, stream=None):
        """Pass B for scales [scale_begin, scale_end): reads only the sorted records, G and the planes -- never the ray buffers."""
        st = stream if stream is not None else _stream()
        _lib.check(_lib.lib().snerf_kplanes_scatter_quotient_scales(C.byref(self.desc), _ptr(planes), C.c_int64(self.N), _ptr(self.G), _ptr(self.sorted_rec),
                                                                    _ptr(gplanes), scale_begin, scale_end, st), "scatter_quotient")

    def quotient_scatter_scales(self, planes, coords: _lib.Coords, gfeat, gplanes, scale_begin: int, scale_end: int, stream=None):
        """Fix-up + pass B for scales [scale_begin, scale_end); G and the fix list must have been produced (quotient_prepare, or the sigma_net backward's
        epilogue: snerf_mlp_bwd_x16_quotient).  gfeat is unused since ABI 11 (the list carries the gradients it needs).  The fix-up goes first: both
        kernels only ADD to the gradient planes, and this way pass B -- not a 5 us launch behind it -- is what the optimiser sweep's stream waits for."""
        self.quotient_fixup_scales(planes, coords, gplanes, scale_begin, scale_end, stream)
        self.quotient_pass_b_scales(planes, gplanes, scale_begin, scale_end, stream)

    def scatter_quotient(self, planes, coords: _lib.Coords, gfeat, feat, gplanes, stream=None):
        """gplanes += d(sum gfeat . features)/d planes, with feat = the forward's features [N, C n_scales] (fp32)."""
        self.quotient_prepare(gfeat, feat, stream)
        self.quotient_scatter_scales(planes, coords, gfeat, gplanes, 0, len(self.ps.resolutions), stream)


def interpolate_kplanes(pts: torch.Tensor, plane_set: PlaneSet, freeze_time_planes: bool = False, freeze_space_planes: bool = False) -> torch.Tensor:
    """Drop-in for interpolate_kplanes(pts, ms_grids, concat_features, freeze_time_planes, freeze_space_planes) (NS/fields/kplanes_field.py:77-126).
    pts [N,4] in [-1,1]; returns [N, C*n_scales] (concat) or [N, C]."""
    pts = _f32c(pts, "pts")
    freeze = int(bool(freeze_time_planes)) | (int(bool(freeze_space_planes)) << 1)
    if freeze & 1 and plane_set.n_coords == 4:
        pts = pts[:, :3].contiguous()  # the static-scene view reads [N,3] points
    return _KPlanesGather.apply(plane_set.planes, plane_set, (pts,), coords_from_points(pts), pts.shape[0], freeze)


def interpolate_kplanes_rays(plane_set: PlaneSet, origins, dirs, times, ebins, aabb, rescale: bool, freeze_time_planes: bool = False,
                             freeze_space_planes: bool = False) -> torch.Tensor:
    """Same gather with sample coordinates derived in-kernel from rays + euclidean bin edges [R,S+1]."""
    origins, dirs, ebins = _f32c(origins, "origins"), _f32c(dirs, "dirs"), _f32c(ebins, "ebins")
    times = _f32c(times, "times").reshape(-1)
    R, S = ebins.shape[0], ebins.shape[1] - 1
    c = coords_from_rays(origins, dirs, times, ebins, aabb, rescale)
    freeze = int(bool(freeze_time_planes)) | (int(bool(freeze_space_planes)) << 1)
    return _KPlanesGather.apply(plane_set.planes, plane_set, (origins, dirs, times, ebins), c, R * S, freeze)


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
        _lib.check(_lib.lib().snerf_weights_bwd(_ptr(density), _ptr(ebins), _ptr(gw), R, S, _ptr(gd), 0, None, _stream()), "weights_bwd")
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


_ACT_ID = {"None": 0, "ReLU": 1, "Sigmoid": 2}


class _DenseNet(torch.autograd.Function):
    """A bias-free MLP of arbitrary depth chained from single dense-layer kernels (snerf_dense_fwd/bwd): for tcnn.Network shapes the fused
    kernels are not instantiated for.  params: one flat vector, layer l row-major [d_l][d_{l+1}] (tcnn_compat.Network's layout)."""

    @staticmethod
    def forward(ctx, x, params, dims, acts, operands=0):
        N = x.shape[0]
        l = _lib.lib()
        saved, h, off = [x], x, 0
        for i in range(len(dims) - 1):
            K, M = dims[i], dims[i + 1]
            y = torch.empty(N, M, dtype=torch.float32, device=x.device)
            if operands:  # 16-bit MFMA operands (csrc/dense_lp.hip)
                _lib.check(l.snerf_dense_fwd_lp(C.c_void_p(params.data_ptr() + 4 * off), K, M, acts[i], _ptr(h), h.stride(0), C.c_int64(N), _ptr(y), M,
                                                operands, _stream()), "dense_fwd_lp")
            else:
                _lib.check(l.snerf_dense_fwd(C.c_void_p(params.data_ptr() + 4 * off), K, M, acts[i], _ptr(h), h.stride(0), C.c_int64(N), _ptr(y), M,
                                             _stream()), "dense_fwd")
            saved.append(y)
            h = y
            off += K * M
        ctx.dims, ctx.acts, ctx.operands = dims, acts, operands
        ctx.save_for_backward(params, *saved)
        return h

    @staticmethod
    def backward(ctx, g):
        params, *saved = ctx.saved_tensors
        dims, acts = ctx.dims, ctx.acts
        N = saved[0].shape[0]
        l = _lib.lib()
        gw = torch.zeros_like(params)
        offs = [0]
        for i in range(len(dims) - 1):
            offs.append(offs[-1] + dims[i] * dims[i + 1])
        g = g.contiguous()
        for i in reversed(range(len(dims) - 1)):
            K, M = dims[i], dims[i + 1]
            x, y = saved[i], saved[i + 1]
            need_gx = i > 0 or ctx.needs_input_grad[0]
            gx = torch.empty(N, K, dtype=torch.float32, device=g.device) if need_gx else None
            if ctx.operands:
                _lib.check(l.snerf_dense_bwd_lp(C.c_void_p(params.data_ptr() + 4 * offs[i]), K, M, acts[i], _ptr(x), x.stride(0), C.c_int64(N), _ptr(y), M,
                                                _ptr(g), M, _ptr(gx) if gx is not None else None, K, C.c_void_p(gw.data_ptr() + 4 * offs[i]), None,
                                                ctx.operands, _stream()), "dense_bwd_lp")
            else:
                _lib.check(l.snerf_dense_bwd(C.c_void_p(params.data_ptr() + 4 * offs[i]), K, M, acts[i], _ptr(x), x.stride(0), C.c_int64(N), _ptr(y), M,
                                             _ptr(g), M, _ptr(gx) if gx is not None else None, K, C.c_void_p(gw.data_ptr() + 4 * offs[i]), _stream()),
                           "dense_bwd")
            g = gx
        return g, gw, None, None, None


def dense_net_forward(x, params, dims, hidden_act: str, out_act: str, operands: int = 0):
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("mlp: expected a float32 HIP device tensor (the HIP library is the only product path)")
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.reshape(-1, x.shape[-1]).contiguous()
    n = len(dims) - 1
    acts = tuple(_ACT_ID[out_act] if i == n - 1 else _ACT_ID[hidden_act] for i in range(n))
    return _DenseNet.apply(x, _f32c(params, "mlp params"), tuple(dims), acts, int(operands))


# ----------------------------------------------------------------------------------------------
# KPlanesField's linear decoder: the pointwise pieces (csrc/linear_decoder.hip)
# ----------------------------------------------------------------------------------------------
class _TruncExp(torch.autograd.Function):
    """trunc_exp (NS/field_components/activations.py:25-41): exp forward, the backward's exponent clamped to [-15, 15]."""

    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _lib.check(_lib.lib().snerf_trunc_exp_fwd(_ptr(x), x.numel(), _ptr(y), _stream()), "trunc_exp_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().snerf_trunc_exp_bwd(_ptr(x), _ptr(g), x.numel(), _ptr(gx), _stream()), "trunc_exp_bwd")
        return gx


def trunc_exp(x: torch.Tensor) -> torch.Tensor:
    return _TruncExp.apply(_f32c(x, "trunc_exp input"))


class _BasisRgb(torch.autograd.Function):
    """rgb = sigmoid(sum_f feat[:, None, f] * basis.view(N, 3, F)) (kplanes_field.py:349-354) without the [N, 3, F] temporary."""

    @staticmethod
    def forward(ctx, feat, basis):
        N, F = feat.shape
        rgb = torch.empty(N, 3, dtype=torch.float32, device=feat.device)
        _lib.check(_lib.lib().snerf_basis_rgb_fwd(_ptr(feat), feat.stride(0), _ptr(basis), N, F, _ptr(rgb), _stream()), "basis_rgb_fwd")
        ctx.save_for_backward(feat, basis, rgb)
        return rgb

    @staticmethod
    def backward(ctx, g):
        feat, basis, rgb = ctx.saved_tensors
        N, F = feat.shape
        g = g.contiguous()
        gf = torch.empty(N, F, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(basis)
        _lib.check(_lib.lib().snerf_basis_rgb_bwd(_ptr(feat), feat.stride(0), _ptr(basis), _ptr(rgb), _ptr(g), N, F, _ptr(gf) if gf is not None else None,
                                                  _ptr(gb), _stream()), "basis_rgb_bwd")
        return gf, gb


def basis_rgb(feat: torch.Tensor, basis: torch.Tensor) -> torch.Tensor:
    """feat [N,F], basis [N,3F] -> rgb [N,3]."""
    feat, basis = _f32c(feat, "basis_rgb features"), _f32c(basis, "basis_rgb basis")
    if feat.dim() != 2 or basis.shape != (feat.shape[0], 3 * feat.shape[1]) or feat.shape[1] % 4:
        raise ValueError(f"basis_rgb: features {tuple(feat.shape)} and basis {tuple(basis.shape)} (expected [N,F] and [N,3F], F a multiple of 4)")
    return _BasisRgb.apply(feat, basis)


# ----------------------------------------------------------------------------------------------
# static multiresolution hash grid (tcnn HashGrid)
# ----------------------------------------------------------------------------------------------
def hashgrid_desc(n_input_dims: int, n_levels: int, n_features_per_level: int, base_resolution: int, per_level_scale: float,
                  log2_hashmap_size: int):
    """-> (descriptor, total rows): the level geometry is computed once, on the host, by the library."""
    d = _lib.HashgridDesc()
    d.D, d.F, d.L = n_input_dims, n_features_per_level, n_levels
    rows = _lib.lib().snerf_hashgrid_layout(C.byref(d), base_resolution, per_level_scale, log2_hashmap_size)
    if rows < 0:
        _lib.check(-1, "hashgrid_layout")
    return d, int(rows)


class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, desc):
        B = x.shape[0]
        out = torch.empty(B, desc.L * desc.F, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().snerf_hashgrid_encode_fwd(C.byref(desc), _ptr(table), _ptr(x), C.c_int64(B), _ptr(out), _stream()), "hashgrid_encode_fwd")
        ctx.desc = desc
        ctx.save_for_backward(x, table)
        return out

    @staticmethod
    def backward(ctx, g):
        x, table = ctx.saved_tensors
        g = g.contiguous()
        gt = torch.zeros_like(table) if ctx.needs_input_grad[1] else None
        gx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        if gt is None and gx is None:
            return None, None, None
        _lib.check(_lib.lib().snerf_hashgrid_encode_bwd(C.byref(ctx.desc), _ptr(table), _ptr(x), C.c_int64(x.shape[0]), _ptr(g),
                                                        _ptr(gt) if gt is not None else None, _ptr(gx) if gx is not None else None, _stream()),
                   "hashgrid_encode_bwd")
        return gx, gt, None


def hashgrid_encode(x: torch.Tensor, table: torch.Tensor, desc: _lib.HashgridDesc) -> torch.Tensor:
    """x [B, D] -> [B, L*F]; differentiable w.r.t. the flat table and the coordinates."""
    x = _f32c(x, "hashgrid x")
    table = _f32c(table, "hashgrid table")
    if x.dim() != 2 or x.shape[1] != desc.D:
        raise RuntimeError(f"hashgrid: x must be [B, {desc.D}], got {tuple(x.shape)}")
    if table.numel() != desc.offsets[desc.L] * desc.F:
        raise RuntimeError(f"hashgrid: table has {table.numel()} floats, the layout needs {desc.offsets[desc.L] * desc.F}")
    return _HashGrid.apply(x, table, desc)


# ----------------------------------------------------------------------------------------------
# compositing + ray-level losses
# ----------------------------------------------------------------------------------------------
BG_MODES = {"random": 0, "last_sample": 1, "constant": 2}


def _bg_args(background, R, dev):
    """background: [R,3] tensor (explicit 'random' draw), 'last_sample', or a [3] tensor / 'black' / 'white'."""
    if isinstance(background, str):
        if background == "last_sample":
            return 1, None
        if background in ("black", "white"):
            return 2, torch.full((3,), 0.0 if background == "black" else 1.0, dtype=torch.float32, device=dev)
        raise ValueError(f"background {background!r}: pass the random colours as a tensor [R,3]")
    bg = _f32c(background, "background")
    if bg.dim() == 1:
        return 2, bg
    assert bg.shape == (R, 3)
    return 0, bg


class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, rgb, ebins, bg, bg_mode, training):
        R, S = weights.shape
        dev = weights.device
        out = torch.empty(R, 3, dtype=torch.float32, device=dev)
        acc = torch.empty(R, dtype=torch.float32, device=dev)
        dmed = torch.empty(R, dtype=torch.float32, device=dev)
        dexp = torch.empty(R, dtype=torch.float32, device=dev)
        mrgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
        midx = torch.empty(R, dtype=torch.int64, device=dev)
        a = _lib.RenderArgs()
        a.weights, a.rgb, a.ebins = weights.data_ptr(), rgb.data_ptr(), ebins.data_ptr()
        a.bg = bg.data_ptr() if bg is not None else None
        a.R, a.S, a.bg_mode, a.training = R, S, bg_mode, int(training)
        a.rgb_out, a.acc_out, a.depth_median, a.depth_expected = out.data_ptr(), acc.data_ptr(), dmed.data_ptr(), dexp.data_ptr()
        a.median_rgb, a.median_index = mrgb.data_ptr(), midx.data_ptr()
        _lib.check(_lib.lib().snerf_render_fwd(C.byref(a), _stream()), "render_fwd")
        ctx.bg_mode = bg_mode
        ctx.save_for_backward(weights, rgb, bg if bg is not None else weights.new_zeros(3))
        ctx.mark_non_differentiable(dmed, dexp, mrgb, midx)
        return out, acc, dmed, dexp, mrgb, midx

    @staticmethod
    def backward(ctx, g_out, g_acc, *_):
        weights, rgb, bg = ctx.saved_tensors
        if ctx.bg_mode == 1:
            raise RuntimeError("render backward with 'last_sample' background is not a training configuration")
        R, S = weights.shape
        g_out = g_out.contiguous()
        gw = torch.empty_like(weights)
        grgb = torch.empty_like(rgb)
        g_acc_p = g_acc.contiguous() if g_acc is not None else None
        _lib.check(_lib.lib().snerf_render_bwd(_ptr(weights), _ptr(rgb), _ptr(bg), ctx.bg_mode, _ptr(g_out),
                                               _ptr(g_acc_p) if g_acc_p is not None else None, R, S, _ptr(gw), _ptr(grgb), 0, _stream()),
                   "render_bwd")
        return gw, grgb, None, None, None, None


def render(weights, rgb, ebins, background, training: bool = True):
    """One-pass renderers (renderers.py): returns dict(rgb [R,3], accumulation [R], depth_median [R], depth_expected [R]
    (unclipped), median_rgb [R,3], median_index [R] int64).  weights [R,S], rgb [R,S,3], ebins [R,S+1]."""
    weights, rgb, ebins = _f32c(weights, "weights"), _f32c(rgb, "rgb"), _f32c(ebins, "ebins")
    mode, bg = _bg_args(background, weights.shape[0], weights.device)
    out, acc, dmed, dexp, mrgb, midx = _Render.apply(weights, rgb, ebins, bg, mode, training)
    return {"rgb": out, "accumulation": acc, "depth_median": dmed, "depth_expected": dexp, "median_rgb": mrgb, "median_index": midx}


class _Distortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, sbins):
        R, S = weights.shape
        loss_rays = torch.empty(R, dtype=torch.float32, device=weights.device)
        gw = torch.empty_like(weights)
        _lib.check(_lib.lib().snerf_distortion(_ptr(weights), _ptr(sbins), R, S, 1.0 / R, _ptr(loss_rays), _ptr(gw), 0, _stream()), "distortion")
        ctx.save_for_backward(gw)
        return loss_rays.mean()

    @staticmethod
    def backward(ctx, g):
        (gw,) = ctx.saved_tensors
        return gw * g, None


def distortion_loss(weights, sbins):
    """distortion_loss (losses.py:139-144) on the nerf level: weights [R,S], sbins [R,S+1] -> scalar."""
    return _Distortion.apply(_f32c(weights, "weights"), _f32c(sbins, "sbins"))


class _Interlevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_prop, p_bins, w_nerf, c_bins):
        R, Sp = w_prop.shape
        S = w_nerf.shape[1]
        loss_rays = torch.empty(R, dtype=torch.float32, device=w_prop.device)
        g = torch.empty_like(w_prop)
        _lib.check(_lib.lib().snerf_interlevel(_ptr(c_bins), _ptr(w_nerf), S, _ptr(p_bins), _ptr(w_prop), Sp, R, 1.0 / (R * S),
                                               _ptr(loss_rays), _ptr(g), _stream()), "interlevel")
        ctx.save_for_backward(g)
        return loss_rays.sum() / (R * S)

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * gout, None, None, None


def interlevel_loss(weights_list, sbins_list):
    """interlevel_loss (losses.py:106-121): lists over levels, the last entry is the (detached) nerf level."""
    c = _f32c(sbins_list[-1].detach(), "sbins")
    w = _f32c(weights_list[-1].detach(), "weights")
    total = 0.0
    for wp, sp in zip(weights_list[:-1], sbins_list[:-1]):
        total = total + _Interlevel.apply(_f32c(wp, "weights"), _f32c(sp.detach(), "sbins"), w, c)
    return total


class _DepthLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, ebins, term, dnorm, sigma: float):
        R, S = weights.shape
        loss_rays = torch.empty(R, dtype=torch.float32, device=weights.device)
        gw = torch.empty_like(weights)
        _lib.check(_lib.lib().snerf_depth_loss(_ptr(weights), _ptr(ebins), _ptr(term), _ptr(dnorm) if dnorm is not None else None, sigma, R, S, 1.0 / R,
                                               _ptr(loss_rays), _ptr(gw), 0, _stream()), "depth_loss")
        ctx.save_for_backward(gw)
        return loss_rays.mean()

    @staticmethod
    def backward(ctx, g):
        (gw,) = ctx.saved_tensors
        return gw * g, None, None, None, None


def ds_nerf_depth_loss(weights, ebins, termination_depth, sigma: float, directions_norm=None):
    """depth_loss with DepthLossType.DS_NERF for one sampling level (losses.py:213-235,261-311): weights [R,S], euclidean bin edges
    [R,S+1], termination_depth [R]; directions_norm [R] when the depth maps hold z-distances (is_euclidean = False), else None."""
    dn = _f32c(directions_norm, "directions_norm").reshape(-1) if directions_norm is not None else None
    return _DepthLoss.apply(_f32c(weights, "weights"), _f32c(ebins.detach(), "ebins"), _f32c(termination_depth, "termination_depth").reshape(-1), dn,
                            float(sigma))


class _UrfDepthLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, ebins, term, dnorm, pred, sigma: float):
        R, S = weights.shape
        loss_rays = torch.empty(R, dtype=torch.float32, device=weights.device)
        gw, gp = torch.empty_like(weights), torch.empty(R, dtype=torch.float32, device=weights.device)
        _lib.check(_lib.lib().snerf_urf_depth_loss(_ptr(weights), _ptr(ebins), _ptr(term), _ptr(dnorm) if dnorm is not None else None, _ptr(pred), sigma, R, S,
                                                   1.0 / R, _ptr(loss_rays), _ptr(gw), _ptr(gp), 0, _stream()), "urf_depth_loss")
        ctx.save_for_backward(gw, gp)
        ctx.pred_shape = pred.shape
        return loss_rays.mean()

    @staticmethod
    def backward(ctx, g):
        gw, gp = ctx.saved_tensors
        return gw * g, None, None, None, (gp * g).view(ctx.pred_shape), None


def urf_depth_loss(weights, ebins, termination_depth, predicted_depth, sigma: float, directions_norm=None):
    """depth_loss with DepthLossType.URF for one sampling level (losses.py:238-274): arguments as ds_nerf_depth_loss + the level's predicted
    depth [R] (differentiable: the expected-depth term)."""
    dn = _f32c(directions_norm, "directions_norm").reshape(-1) if directions_norm is not None else None
    return _UrfDepthLoss.apply(_f32c(weights, "weights"), _f32c(ebins.detach(), "ebins"), _f32c(termination_depth, "termination_depth").reshape(-1), dn,
                               _f32c(predicted_depth, "predicted_depth").reshape(-1), float(sigma))


REG_SLOTS = 1024  # partial-sum slots of the regulariser values (one 64-B line each)


class _PlaneReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, planes, ps: PlaneSet):
        parts = torch.zeros(REG_SLOTS, 16, dtype=torch.float32, device=planes.device)
        desc = ps.desc()
        _lib.check(_lib.lib().snerf_plane_reg(C.byref(desc), _ptr(planes), None, 0.0, 0.0, 0.0, _ptr(parts), REG_SLOTS, 0, _stream()), "plane_reg")
        ctx.ps = ps
        ctx.save_for_backward(planes)
        return parts[:, :3].sum(0)

    @staticmethod
    def backward(ctx, g):
        (planes,) = ctx.saved_tensors
        c = g.detach().cpu().tolist()  # three coefficients (host sync; the fused trainer passes them directly)
        grad = torch.empty_like(planes)  # overwrite mode: every element of every plane is written
        desc = ctx.ps.desc()
        _lib.check(_lib.lib().snerf_plane_reg(C.byref(desc), _ptr(planes), _ptr(grad), c[0], c[1], c[2], None, 0, 1, _stream()), "plane_reg")
        return grad, None


def plane_regularizers(ps: PlaneSet) -> torch.Tensor:
    """[space_tv, time_smoothness, sparse_transients] of one plane set (losses.py:383-452), differentiable."""
    return _PlaneReg.apply(ps.planes, ps)


def new_adam_dyn(device) -> torch.Tensor:
    """Device-resident optimiser state of one parameter group (snerf_adam_dyn, include/snerf.h): int32[8], zero-initialised.
    [0] nonfinite flag, [1] Adam steps taken, [2] steps skipped, [3] gradient elements dropped."""
    return torch.zeros(8, dtype=torch.int32, device=device)


def adam_prepare(dyn: torch.Tensor, lr: float, betas=(0.9, 0.999), policy: str = "skip_step", force_nonfinite: bool = False):
    """Once per parameter group and step, before its Adam kernels: decides whether the step is skipped (GradScaler semantics,
    NS/engine/trainer.py:394-408), advances the device-side step counter and the bias corrections."""
    _lib.check(_lib.lib().snerf_adam_prepare(_ptr(dyn), lr, betas[0], betas[1], {"drop_elements": 0, "skip_step": 1}[policy], int(force_nonfinite),
                                             _stream()), "adam_prepare")


def adam_step(p, g, m, v, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-12, grad_scale: float = 1.0, zero_grad: bool = False,
              p_out=None, dyn: Optional[torch.Tensor] = None):
    """Fused Adam on flat fp32 buffers (1-based step); in place unless p_out is given.  dyn: the group's device-side state
    (adam_prepare) -- `step` is then ignored."""
    for t in (p, g, m, v):
        _f32c(t, "adam buffer")
    _lib.check(_lib.lib().snerf_adam_step(_ptr(p), _ptr(p if p_out is None else p_out), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, betas[0], betas[1], eps,
                                          step, grad_scale, int(zero_grad), _ptr(dyn) if dyn is not None else None, _stream()), "adam_step")


def adam_planes_step(ps: PlaneSet, p_in, p_out, g, m, v, coefs, losses, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-12,
                     grad_scale: float = 1.0, zero_grad: bool = True, shard_range=None, dyn: Optional[torch.Tensor] = None):
    """Adam over one plane set with the plane regularisers fused in (ping-pong p_in -> p_out). coefs = (space_tv, time_smooth, sparse).
    shard_range = (lo, hi): update only floats [lo, hi) of the segment (this rank's optimiser shard); all tensors are whole segments."""
    desc = ps.desc()
    lo, hi = (0, ps.numel + 3 & ~3) if shard_range is None else shard_range
    _lib.check(_lib.lib().snerf_adam_planes_step_range(C.byref(desc), _ptr(p_in), _ptr(p_out), _ptr(g), _ptr(m), _ptr(v), coefs[0], coefs[1], coefs[2],
                                                       _ptr(losses) if losses is not None else None, REG_SLOTS if losses is not None else 0, lr,
                                                       betas[0], betas[1], eps, step, grad_scale, int(zero_grad), int(lo), int(hi),
                                                       _ptr(dyn) if dyn is not None else None, _stream()),
               "adam_planes_step")


def fx_to_float(fx: torch.Tensor, out: torch.Tensor, accumulate: bool = False):
    """Deterministic mode: fixed-point gradient cells (int64, value * 2^50) -> float gradients; clears the cells."""
    if fx.dtype != torch.int64 or out.dtype != torch.float32 or fx.numel() != out.numel():
        raise RuntimeError("fx_to_float: need an int64 cell buffer and a float32 buffer of the same length")
    _lib.check(_lib.lib().snerf_fx_to_float(_ptr(fx), _ptr(out), fx.numel(), int(accumulate), _stream()), "fx_to_float")


def generate_rays(indices, fx, fy, cx, cy, c2w, cam_times=None, aabb=None, near_plane: float = 0.0, training: bool = True):
    """RayGenerator.forward (+ AABBBoxCollider when aabb is given).  indices int64 [R,3]; per-camera fx,fy,cx,cy [M],
    c2w [M,3,4], cam_times [M].  Returns dict of origins, directions, pixel_area, directions_norm, times, (nears, fars)."""
    if not indices.is_cuda or indices.dtype != torch.int64:
        raise RuntimeError("generate_rays: indices must be an int64 HIP device tensor")
    indices = indices.contiguous()
    R, dev = indices.shape[0], indices.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = {"origins": f(R, 3), "directions": f(R, 3), "pixel_area": f(R, 1), "directions_norm": f(R, 1), "times": f(R, 1)}
    a = _lib.RaygenArgs()
    a.indices = indices.data_ptr()
    keep = [_f32c(t, "camera table") for t in (fx, fy, cx, cy, c2w)]
    a.fx, a.fy, a.cx, a.cy, a.c2w = [t.data_ptr() for t in keep]
    if cam_times is not None:
        cam_times = _f32c(cam_times, "cam_times")
        a.cam_times = cam_times.data_ptr()
    a.R = R
    a.origins, a.dirs, a.pixel_area, a.dir_norm, a.times = [out[k].data_ptr() for k in ("origins", "directions", "pixel_area", "directions_norm", "times")]
    if aabb is not None:
        ab = aabb_host(aabb)
        a.collide, a.training, a.near_plane = 1, int(training), near_plane
        for k in range(3):
            a.aabb_min[k], a.aabb_max[k] = ab[0][k], ab[1][k]
        out["nears"], out["fars"] = f(R, 1), f(R, 1)
        a.nears, a.fars = out["nears"].data_ptr(), out["fars"].data_ptr()
    _lib.check(_lib.lib().snerf_raygen(C.byref(a), _stream()), "raygen")
    out["camera_indices"] = indices[:, 0:1]
    return out


def image_time_keys(times: torch.Tensor):
    """-> (key [M] int32: the rank of each image's time among the distinct times, number of distinct times): the key table of sort_rays_by_time."""
    uniq, inv = torch.unique(times.reshape(-1), sorted=True, return_inverse=True)
    return inv.to(torch.int32).contiguous(), int(uniq.numel())


def sort_rays_by_time(indices: torch.Tensor, image_key: torch.Tensor, n_keys: int, aux: Optional[torch.Tensor] = None):
    """The batch's pixel indices [R,3] (image, row, col) -- and, with them, aux [R,k] (e.g. the target colours) -- in order of the images' frame
    time (snerf_sort_rays_by_key: why, and why that is free).  Batches beyond 16384 rays are returned as they are."""
    R = indices.shape[0]
    if R > 16384 or R == 0:
        return (indices, aux) if aux is not None else indices
    if not indices.is_cuda or indices.dtype != torch.int64 or image_key.dtype != torch.int32 or not image_key.is_cuda:
        raise RuntimeError("sort_rays_by_time: indices must be an int64 HIP tensor [R,3] and image_key an int32 HIP tensor")
    indices = indices.contiguous()
    out = torch.empty_like(indices)
    aux_c = _f32c(aux, "sort_rays_by_time aux").reshape(R, -1) if aux is not None else None
    aux_out = torch.empty_like(aux_c) if aux_c is not None else None
    _lib.check(_lib.lib().snerf_sort_rays_by_key(_ptr(indices), _ptr(image_key), n_keys, R, _ptr(aux_c) if aux_c is not None else None,
                                                 aux_c.shape[1] if aux_c is not None else 0, _ptr(out), _ptr(aux_out) if aux_out is not None else None,
                                                 _stream()), "sort_rays_by_key")
    return (out, aux_out.view(aux.shape)) if aux is not None else out


def sample_pixels_uniform(u: torch.Tensor, num_images: int, height: int, width: int, images: Optional[torch.Tensor] = None):
    """PixelSampler.sample_method's uniform draw (pixel_samplers.py:74-77) from u = rand(R,3), fused with the image gather of
    collate_image_dataset_batch (:111-123) when the uint8 image cache [M,H,W,3] is given.  Returns (indices int64 [R,3], target fp32 [R,3] | None)."""
    u = _f32c(u, "u")
    R = u.shape[0]
    idx = torch.empty(R, 3, dtype=torch.int64, device=u.device)
    target = None
    if images is not None:
        if not images.is_cuda or images.dtype != torch.uint8 or not images.is_contiguous() or tuple(images.shape) != (num_images, height, width, 3):
            raise RuntimeError("sample_pixels_uniform: images must be a contiguous uint8 HIP tensor [M,H,W,3]")
        target = torch.empty(R, 3, dtype=torch.float32, device=u.device)
    _lib.check(_lib.lib().snerf_sample_pixels_uniform(_ptr(u), R, num_images, height, width, _ptr(images) if images is not None else None, _ptr(idx),
                                                      _ptr(target) if target is not None else None, _stream()), "sample_pixels_uniform")
    return idx, target


def aabb_collide(origins, directions, aabb, near_plane: float = 0.0, training: bool = True):
    """AABBBoxCollider._intersect_with_aabb (scene_colliders.py:59-95) -> (nears [R,1], fars [R,1])."""
    origins, directions = _f32c(origins, "origins"), _f32c(directions, "directions")
    R = origins.shape[0]
    nears = torch.empty(R, 1, dtype=torch.float32, device=origins.device)
    fars = torch.empty_like(nears)
    arr = (C.c_float * 6)(*[v for row in aabb_host(aabb) for v in row])
    _lib.check(_lib.lib().snerf_aabb_collide(_ptr(origins), _ptr(directions), R, arr, near_plane, int(training), _ptr(nears), _ptr(fars),
                                             _stream()), "aabb_collide")
    return nears, fars

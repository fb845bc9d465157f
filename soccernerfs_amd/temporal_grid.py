"""`TemporalGridEncoder` with the constructor, buffers and methods of NS/field_components/temporal_grid.py:159-376,
running on libsnerf's gfx950 kernels (csrc/tgrid.hip) instead of the reference's CUDA extension."""
import ctypes as C
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib, ops


def channel_table(temporal_dim: int, level_dim: int):
    """Closed form of the table init_parameters builds (temporal_grid.py:231-308): row r blends, in channel p = r mod C,
    the column that currently occupies position p with the entering column C + r; the other channels read their occupant."""
    Cn, T = level_dim, temporal_dim
    rows = max(T - 1, 1)
    samp = torch.zeros(rows, 4 * Cn, dtype=torch.long)
    mask_a = torch.zeros(rows, 4 * Cn, dtype=torch.bool)
    mask_b = torch.zeros(rows, 4 * Cn, dtype=torch.bool)
    index_list = torch.zeros(rows, 2, dtype=torch.long)
    occ = lambda q, r: (Cn + q + Cn * ((r - 1 - q) // Cn)) if r > q else q
    for r in range(rows):
        p = r % Cn
        for q in range(Cn):
            samp[r, 4 * q], samp[r, 4 * q + 1] = 1, occ(q, r)
        samp[r, 4 * p + 3] = Cn + r
        mask_a[r, 4 * p], mask_b[r, 4 * p + 2] = True, True
        index_list[r, 0], index_list[r, 1] = occ(p, r), Cn + r
    return samp, mask_a, mask_b, index_list


class _Encode(torch.autograd.Function):
    """Encoder forward/backward.  Optionally carries the temporal-TV term of the same table (columns tv_a, tv_b) as a second output,
    so that the table receives ONE gradient tensor per step: as separate autograd nodes the two column slices cost two dense
    zero-filled [rows, 66] tensors plus the adds that merge them with the scatter gradient (2.9 ms of a 12 ms step on the 1.54 GB
    table, profiles/r01_kernels.md)."""

    @staticmethod
    def forward(ctx, embeddings, enc, coords_keep, coords, trow, times, spr, B, tv_a, tv_b, xyz=None):
        out = torch.empty(B, enc.output_dim, dtype=torch.float32, device=embeddings.device)
        ctx.dy_dx = None
        if xyz is not None and xyz.requires_grad:
            # calc_grad_inputs (temporal_grid.py:82-87): d out / d xyz is kept for the backward, [B, L, D, C]
            ctx.dy_dx = torch.empty(B, enc.num_levels, enc.input_dim, enc.level_dim, dtype=torch.float32, device=embeddings.device)
            _lib.check(_lib.lib().snerf_tgrid_encode_fwd_dydx(C.byref(enc.desc), ops._ptr(embeddings), C.byref(coords),
                                                              ops._ptr(trow) if trow is not None else None, ops._ptr(times) if times is not None else None,
                                                              spr, C.c_int64(B), ops._ptr(out), ops._ptr(ctx.dy_dx), ops._stream()), "tgrid_encode_fwd_dydx")
        else:
            _lib.check(_lib.lib().snerf_tgrid_encode_fwd(C.byref(enc.desc), ops._ptr(embeddings), C.byref(coords),
                                                         ops._ptr(trow) if trow is not None else None, ops._ptr(times) if times is not None else None,
                                                         spr, C.c_int64(B), ops._ptr(out), ops._stream()), "tgrid_encode_fwd")
        ctx.enc, ctx.coords, ctx.keep, ctx.trow, ctx.times, ctx.spr, ctx.B = enc, coords, coords_keep, trow, times, spr, B
        ctx.shape, ctx.tv = embeddings.shape, None
        if tv_a is None:
            return out, None
        part = torch.zeros(64, 16, dtype=torch.float32, device=embeddings.device)
        _lib.check(_lib.lib().snerf_tgrid_tv_fwd(ops._ptr(embeddings), C.c_int64(embeddings.shape[0]), embeddings.shape[1], tv_a, tv_b, ops._ptr(part), 64,
                                                 ops._stream()), "tgrid_tv_fwd")
        ctx.tv = (tv_a, tv_b)
        ctx.save_for_backward(embeddings)
        return out, part[:, 0].sum() / embeddings.shape[0]

    @staticmethod
    def backward(ctx, g, g_tv):
        enc = ctx.enc
        acc = enc.embeddings.grad if getattr(enc, "accumulate_into_grad", False) else None
        gemb = acc if acc is not None else torch.zeros(ctx.shape, dtype=torch.float32, device=enc.embeddings.device)
        gxyz = None
        if g is not None:
            g = g.contiguous()
            if ctx.dy_dx is not None:  # kernel_input_backward (temporal_gridencoder.cu:373-398)
                gxyz = torch.empty(ctx.B, enc.input_dim, dtype=torch.float32, device=g.device)
                _lib.check(_lib.lib().snerf_tgrid_input_bwd(ops._ptr(g), ops._ptr(ctx.dy_dx), C.c_int64(ctx.B), enc.input_dim, enc.level_dim, enc.num_levels,
                                                            ops._ptr(gxyz), ops._stream()), "tgrid_input_bwd")
            _lib.check(_lib.lib().snerf_tgrid_encode_bwd(C.byref(enc.desc), C.byref(ctx.coords), ops._ptr(ctx.trow) if ctx.trow is not None else None,
                                                         ops._ptr(ctx.times) if ctx.times is not None else None, ctx.spr, C.c_int64(ctx.B), ops._ptr(g),
                                                         ops._ptr(gemb), ops._stream()), "tgrid_encode_bwd")
        if g_tv is not None and ctx.tv is not None:
            (emb,) = ctx.saved_tensors
            gt = g_tv.detach().reshape(1).float().contiguous()
            _lib.check(_lib.lib().snerf_tgrid_tv_bwd(ops._ptr(emb), C.c_int64(ctx.shape[0]), ctx.shape[1], ctx.tv[0], ctx.tv[1], ops._ptr(gt), ops._ptr(gemb),
                                                     ops._stream()), "tgrid_tv_bwd")
        # accumulate_into_grad (set by optimizers.FusedAdam): the scatter went straight into the persistent, optimiser-cleared
        # .grad buffer; returning None keeps autograd from allocating / adding a second dense tensor
        return (None if acc is not None else gemb), None, None, None, None, None, None, None, None, None, gxyz


class TemporalGridEncoder(nn.Module):
    def __init__(self, temporal_dim: int = 64, input_dim: int = 3, num_levels: int = 16, level_dim: int = 2, per_level_scale: float = 2.0,
                 base_resolution: int = 16, log2_hashmap_size: int = 19, desired_resolution: Optional[float] = None, gridtype: str = "hash",
                 align_corners: bool = False) -> None:
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = float(np.exp2(np.log2(float(desired_resolution) / base_resolution) / (num_levels - 1)))
        self.temporal_dim, self.input_dim, self.num_levels, self.level_dim = temporal_dim, input_dim, num_levels, level_dim
        self.per_level_scale, self.log2_hashmap_size, self.base_resolution = per_level_scale, log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id, self.align_corners = gridtype, {"hash": 0, "tiled": 1}[gridtype], align_corners
        offsets, off = [], 0
        self.max_params = 2**log2_hashmap_size
        for i in range(num_levels):
            res = int(np.ceil(base_resolution * per_level_scale**i))
            n = min(self.max_params, (res if align_corners else res + 1) ** input_dim)
            n = int(np.ceil(n / 8) * 8)
            offsets.append(off)
            off += n
        offsets.append(off)
        self.register_buffer("offsets", torch.tensor(offsets, dtype=torch.int32))
        self.n_params = off * level_dim
        self.embeddings = nn.Parameter(torch.empty(off, level_dim + temporal_dim).uniform_(-1e-4, 1e-4))  # init_parameters :249-250
        samp, ma, mb, il = channel_table(temporal_dim, level_dim)
        self.register_buffer("sampling_index", samp)
        self.register_buffer("index_a_mask", ma)
        self.register_buffer("index_b_mask", mb)
        self.register_buffer("index_list", il)  # columns (A, B) of each row: what get_temporal_tv_loss reads
        self._index_list_host = il.tolist()
        self.fuse_tv, self._tv_cached = True, None
        self.tv_row_override = None  # parity hook: fixed table row instead of the random draw
        self.accumulate_into_grad = False
        d = _lib.TgridDesc()
        d.D, d.C, d.L, d.grid_C, d.H = input_dim, level_dim, num_levels, level_dim + temporal_dim, base_resolution
        d.gridtype, d.align_corners, d.S = self.gridtype_id, int(align_corners), float(np.log2(per_level_scale))
        for i, o in enumerate(offsets):
            d.offsets[i] = o
        self.desc = d

    def get_temporal_index(self, time: torch.Tensor) -> torch.Tensor:
        """temporal_grid.py:320-330: time [B] -> rows [B, 4*level_dim] (w_a, col_a, w_b, col_b per channel)."""
        n = len(self.sampling_index) - 1
        v = time * n
        r = v.long()
        r[time == 1] = n
        out = self.sampling_index[r].float()
        out[self.index_a_mask[r]] = r + 1 - v
        out[self.index_b_mask[r]] = v - r
        return out

    def forward(self, xyz: torch.Tensor, time: torch.Tensor, explicit_rows: bool = False) -> torch.Tensor:
        """xyz [B, input_dim] in [0,1]; time [B,1] in [0,1] -> [B, num_levels*level_dim].  explicit_rows=True passes the
        reference's temporal_row_index tensor to the kernel; default derives it in-kernel from `time` (same values).  When xyz requires a
        gradient (deformed positions, camera optimisation) the forward also keeps d out / d xyz and the backward returns the coordinate gradient
        (calc_grad_inputs of TemporalGridEncodeFunc, temporal_grid.py:82-87,139-150)."""
        xyz = ops._f32c(xyz, "xyz")
        t = ops._f32c(time, "time").reshape(-1)
        B = xyz.shape[0]
        co = ops.coords_from_points(xyz)
        tv_a, tv_b = self._draw_tv_columns()
        if explicit_rows:
            rows = self.get_temporal_index(t).contiguous()
            out, tv = _Encode.apply(self.embeddings, self, (xyz, rows), co, rows, None, 1, B, tv_a, tv_b, xyz)
        else:
            out, tv = _Encode.apply(self.embeddings, self, (xyz, t), co, None, t, 1, B, tv_a, tv_b, xyz)
        self._tv_cached = tv
        return out

    def forward_rays(self, origins, directions, ray_times, ebins, aabb) -> torch.Tensor:
        """Same encoding with sample coordinates derived in-kernel from rays (one time per ray): [R*S, L*C]."""
        origins, directions, ebins = ops._f32c(origins, "origins"), ops._f32c(directions, "directions"), ops._f32c(ebins, "ebins")
        t = ops._f32c(ray_times, "times").reshape(-1)
        R, S = ebins.shape[0], ebins.shape[1] - 1
        co = ops.coords_from_rays(origins, directions, t, ebins, aabb, rescale=False)
        tv_a, tv_b = self._draw_tv_columns()
        out, tv = _Encode.apply(self.embeddings, self, (origins, directions, t, ebins), co, None, t, S, R * S, tv_a, tv_b)
        self._tv_cached = tv
        return out

    def _draw_tv_columns(self):
        """Training forward: draw the table row get_temporal_tv_loss would draw (temporal_grid.py:371-373) so that the TV term rides on
        the encoder's autograd node; its value is handed out by the next get_temporal_tv_loss() call."""
        if not (self.fuse_tv and self.training and torch.is_grad_enabled() and self.embeddings.requires_grad):
            return None, None
        row = self.tv_row_override if self.tv_row_override is not None else int(torch.randint(0, len(self.index_list), [1]).item())
        return tuple(self._index_list_host[row])

    def get_temporal_tv_loss(self, row_idx: Optional[int] = None) -> torch.Tensor:
        """temporal_grid.py:352-376: mean |emb[:, A] - emb[:, B]| for a random (or given) table row."""
        if row_idx is None and getattr(self, "_tv_cached", None) is not None:
            tv, self._tv_cached = self._tv_cached, None
            return tv
        if row_idx is None:
            row_idx = int(torch.randint(0, len(self.index_list), [1]).item())
        a, b = self.index_list[row_idx].tolist()
        return (self.embeddings[:, a] - self.embeddings[:, b]).abs().mean()


class TiledTableBackward:
    """Owner-computes backward of one TemporalGridEncoder table for a fixed batch size (csrc/tgrid_tiles.hip, ABI 14): `bin` files the batch's (sample,
    level, corner) touches under tiles of consecutive table rows, then either `scatter` adds the tiles into a dense gradient buffer (= what
    snerf_tgrid_encode_bwd leaves there, up to the association order of the float sums) or `scatter_adam` runs torch.optim.Adam for the whole table
    straight from the tiles' LDS images (+ the temporal-TV step), without a dense gradient.  Levels below plan.first_tiled_level go through the run-length
    atomic kernel into the gradient buffer (`coarse_levels`, in both forms).  After `bin` the tile passes read only this object's buffers and `gout`.
    Replaces NS/field_components/cuda/csrc/temporal_gridencoder.cu:283-370 + the optimiser step of the table."""

    def __init__(self, enc: "TemporalGridEncoder", B: int, tile_rows_log2: int = 0, first_tiled_level: int = -1):
        self.enc, self.B = enc, int(B)
        self.plan = _lib.TgridTilePlan()
        _lib.check(_lib.lib().snerf_tgrid_tile_plan_make(C.byref(enc.desc), C.c_int64(B), tile_rows_log2, first_tiled_level, C.byref(self.plan)), "tile_plan_make")
        dev = enc.embeddings.device
        self.counts = torch.empty(max(int(self.plan.count_ints), 1), dtype=torch.int32, device=dev)
        self.tile_base = torch.zeros(self.plan.n_tiles + 3, dtype=torch.int32, device=dev)  # prefix sums [n_tiles + 1] + the fused pass's ticket words
        self.records = torch.empty(max(int(self.plan.record_capacity), 1), dtype=torch.int32, device=dev)
        self.pos4 = torch.empty(max(self.B, 1), 4, dtype=torch.float32, device=dev)

    def share_bins_of(self, other: "TiledTableBackward"):
        """Two tables of the same geometry evaluated at the SAME positions and times (the full NeRFPlayer's newness and decomposition grids) need one binning
        pass between them when it runs without the gradient filter (bin(gout=None)): alias `other`'s buffers; only `other.bin` is then called."""
        a, b = self.enc.desc, other.enc.desc
        same = (a.D, a.C, a.L, a.grid_C, a.H, a.gridtype, a.align_corners, a.S, list(a.offsets)) == (b.D, b.C, b.L, b.grid_C, b.H, b.gridtype, b.align_corners, b.S, list(b.offsets))
        if not same or self.B != other.B or self.plan.tile_rows_log2 != other.plan.tile_rows_log2 or self.plan.first_tiled_level != other.plan.first_tiled_level:
            raise ValueError("share_bins_of: the two tables' geometry / batch / tiling differ")
        self.counts, self.tile_base, self.records, self.pos4 = other.counts, other.tile_base, other.records, other.pos4

    def bin(self, coords: _lib.Coords, times: torch.Tensor, spr: int, gout: Optional[torch.Tensor], stream=None):
        """gout = None: file every in-range sample (the pass then needs the sample positions only and can run beside the forward)."""
        st = stream if stream is not None else ops._stream()
        _lib.check(_lib.lib().snerf_tgrid_bwd_bin(C.byref(self.enc.desc), C.byref(self.plan), C.byref(coords), ops._ptr(times), spr, C.c_int64(self.B),
                                                  ops._ptr(gout) if gout is not None else None, ops._ptr(self.pos4), ops._ptr(self.counts), ops._ptr(self.tile_base), ops._ptr(self.records), st), "tgrid_bwd_bin")

    def coarse_levels(self, coords: _lib.Coords, times: torch.Tensor, spr: int, gout: torch.Tensor, gtable: torch.Tensor, stream=None):
        """Levels [0, first_tiled_level) through the run-length atomic kernel into gtable; reads the ray buffers, so it belongs on their stream."""
        lc = self.plan.first_tiled_level
        if lc > 0:
            st = stream if stream is not None else ops._stream()
            _lib.check(_lib.lib().snerf_tgrid_encode_bwd_levels(C.byref(self.enc.desc), C.byref(coords), None, ops._ptr(times), spr, C.c_int64(self.B), ops._ptr(gout),
                                                                ops._ptr(gtable), 0, lc, st), "tgrid_encode_bwd_levels")

    def scatter(self, gout: torch.Tensor, gtable: torch.Tensor, stream=None):
        """gtable += the tiled levels' share of d loss / d table (after `bin` of the same batch; `coarse_levels` adds the rest)."""
        st = stream if stream is not None else ops._stream()
        _lib.check(_lib.lib().snerf_tgrid_bwd_tiles(C.byref(self.enc.desc), C.byref(self.plan), C.c_int64(self.B), ops._ptr(gout), ops._ptr(self.pos4),
                                                    ops._ptr(self.tile_base), ops._ptr(self.records), ops._ptr(gtable), st), "tgrid_bwd_tiles")

    def scatter_adam(self, gout: torch.Tensor, gtable: Optional[torch.Tensor], p: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, step: int, eps: float,
                     tv_cols=None, srow: Optional[torch.Tensor] = None, betas=(0.9, 0.999), stream=None):
        """Adam step `step` (1-based) of the whole table with gradient = tiles + gtable's coarse-level rows (cleared) + the TV term; after `bin` and
        `coarse_levels` of the same batch."""
        st = stream if stream is not None else ops._stream()
        ca, cb = tv_cols if tv_cols is not None else (-1, -1)
        _lib.check(_lib.lib().snerf_tgrid_bwd_tiles_adam(C.byref(self.enc.desc), C.byref(self.plan), C.c_int64(self.B), ops._ptr(gout), ops._ptr(self.pos4),
                                                         ops._ptr(self.tile_base), ops._ptr(self.records), ops._ptr(gtable) if gtable is not None else None,
                                                         ops._ptr(p), ops._ptr(m), ops._ptr(v), lr, betas[0], betas[1], eps, step, ca, cb,
                                                         ops._ptr(srow) if srow is not None else None, st), "tgrid_bwd_tiles_adam")

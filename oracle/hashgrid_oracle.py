"""CPU restatement of tiny-cuda-nn's multiresolution HashGrid encoding (`tcnn.Encoding(3, {"otype": "HashGrid", ...})`), the static
"stationary field" of the full NeRFPlayer (call site: NS/fields/nerfplayer_field.py:242-252, evaluated at :341-342).

TEST INFRASTRUCTURE ONLY: imported by tests/, oracle/shims/tinycudann (golden generation) -- never by the product path.

PARITY UNPINNED: tiny-cuda-nn v1.6 (pinned by the reference's Dockerfile:121) is a third-party dependency whose source is not under
/root/reference and the reference's tests hold no values for it.  This file restates the PUBLISHED algorithm (Mueller et al. 2022,
"Instant Neural Graphics Primitives", section 3, and tiny-cuda-nn include/tiny-cuda-nn/encodings/grid.h):

  level l:  scale_l = exp2(l * log2(per_level_scale)) * base_resolution - 1        (grid_scale)
            res_l   = ceil(scale_l) + 1                                             (grid_resolution)
            rows_l  = min(next_multiple(res_l ** D, 8), 2 ** log2_hashmap_size)     (offset table, Hash grid type)
  position: pos = fma(x, scale_l, 0.5); cell = floor(pos) (as uint32); frac = pos - floor(pos)   (pos_fract; no bounds check: coordinates
            outside [0, 1] wrap through the uint32 cast / the hash)
  corner index (grid_index): dense  index = sum_d cell_d * res_l ** d  while the running stride <= rows_l;
            if the stride overflowed rows_l: index = XOR_d (cell_d * prime_d)   (coherent_prime_hash, primes 1, 2654435761, 805459861)
            row = index % rows_l
  value:    D-linear interpolation of the 2**D corner rows; outputs laid out [B, L * F] level-major.
  init:     table ~ U(-1e-4, 1e-4).

Cross-check available in the reference itself: the prime constants and the XOR hash agree with NS/field_components/encodings.py:301
(`HashEncoding.hash_fn`, an independent pure-torch hash grid) and with temporal_gridencoder.cu:46-59 (pinned by G9/G9b).
"""
import math

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)
M32 = 0xFFFFFFFF


def level_geometry(n_levels: int, base_resolution: int, per_level_scale: float, log2_hashmap_size: int, n_dims: int = 3):
    """-> (scales[L] float32, resolutions[L], offsets[L+1]) exactly as the encoding's constructor derives them."""
    log2_pls = np.float32(math.log2(float(np.float32(per_level_scale))))  # log2f of the float member
    scales, ress, offsets = [], [], [0]
    for l in range(n_levels):
        # float32 steps of grid_scale; exp2f evaluated in double and rounded (numpy's float32 exp2 is 1 ulp off libm's for some l)
        e = np.float32(2.0 ** float(np.float32(l) * log2_pls))
        scale = np.float32(np.float32(e * np.float32(base_resolution)) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        params = min(res ** n_dims, M32 // 2)
        params = (params + 7) // 8 * 8
        params = min(params, 1 << log2_hashmap_size)
        scales.append(scale)
        ress.append(res)
        offsets.append(offsets[-1] + params)
    return np.asarray(scales, np.float32), ress, offsets


def _rows(cell, res: int, rows: int):
    """cell: int64 [B, D] already reduced mod 2**32.  grid_index."""
    D = cell.shape[1]
    stride, index = 1, torch.zeros_like(cell[:, 0])
    d = 0
    while d < D and stride <= rows:
        index = (index + cell[:, d] * stride) & M32
        stride *= res
        d += 1
    if rows < stride:
        index = torch.zeros_like(cell[:, 0])
        for k in range(D):
            index = index ^ ((cell[:, k] * PRIMES[k]) & M32)
    return index % rows


def encode(x: torch.Tensor, table: torch.Tensor, n_levels: int, n_features: int, base_resolution: int, per_level_scale: float,
           log2_hashmap_size: int) -> torch.Tensor:
    """x [B, D] float32, table [rows_total, F] -> [B, L*F]; differentiable w.r.t. table and x (the x-derivative is tcnn's dy_dx:
    scale * finite difference of the corner values along the axis, frac treated as linear inside a cell)."""
    B, D = x.shape
    scales, ress, offsets = level_geometry(n_levels, base_resolution, per_level_scale, log2_hashmap_size, D)
    outs = []
    for l in range(n_levels):
        scale = float(scales[l])
        pos = (x.double() * scale + 0.5).to(x.dtype)  # fmaf(scale, x, 0.5): one rounding (the product of two floats is exact in double)
        fl = torch.floor(pos)
        frac = pos - fl
        cell0 = fl.detach().to(torch.int64) & M32  # (uint32)(int)floorf
        rows = offsets[l + 1] - offsets[l]
        acc = torch.zeros(B, n_features, dtype=table.dtype)
        for corner in range(1 << D):
            w = torch.ones(B, dtype=x.dtype)
            cell = cell0.clone()
            for d in range(D):
                if corner >> d & 1:
                    w = w * frac[:, d]
                    cell[:, d] = (cell[:, d] + 1) & M32
                else:
                    w = w * (1.0 - frac[:, d])
            r = _rows(cell, ress[l], rows) + offsets[l]
            acc = acc + w[:, None] * table[r]
        outs.append(acc)
    return torch.cat(outs, dim=1)


def nerfplayer_field_forward(positions, times, aabb, grid_cfg, tgrid_enc, params):
    """NerfplayerField.get_density + get_outputs (NS/fields/nerfplayer_field.py:330-414) with view dependence disabled (the model's
    default, NS/models/nerfplayer.py:90): positions [R,S,3] world, times [R,1].
    grid_cfg: (n_levels, n_features, base_resolution, per_level_scale, log2_hashmap_size) of the static grid; tgrid_enc: encoder dict of
    oracle.tgrid_oracle for the newness / decomposition grids; params: dict name -> tensor(s) as the reference's state_dict names them.
    -> density [R,S], rgb [R,S,3], probs [R,S,3]."""
    from oracle import kplanes_oracle as KO
    from oracle import tgrid_oracle as TO

    R, S = positions.shape[:2]
    p = KO.normalize_positions(positions, aabb).reshape(-1, 3)
    t = times.expand(R, S).reshape(-1, 1)
    lin = lambda name, n: [params[f"field.{name}.layers.{i}.weight"] for i in range(n)]
    table = params["field.stationary_field.params"].view(-1, grid_cfg[1])
    deformed = p + KO.mlp(p, lin("deformation_field", 4))                                             # :336-339
    v_stat = KO.mlp(torch.cat([encode(p, table, *grid_cfg), t], -1), lin("stationary_field_mlp", 2))         # :341-344
    v_deform = KO.mlp(torch.cat([encode(deformed, table, *grid_cfg), t], -1), lin("stationary_field_mlp", 2))
    trow = TO.temporal_index(t[:, 0], tgrid_enc["table"])
    tg = lambda name: TO.encode(p, trow, params[f"field.{name}.embeddings"], tgrid_enc["offsets"], tgrid_enc["log2_scale"], tgrid_enc["base_res"],
                                tgrid_enc["gridtype"], tgrid_enc["level_dim"])
    v_new = tg("newness_field")                                                                        # :347
    probs = torch.softmax(KO.mlp(tg("decomposition_field"), lin("decomposition_mlp", 2)), dim=-1)     # :350-353
    v = probs[:, 0:1] * v_stat + probs[:, 1:2] * v_deform + probs[:, 2:3] * v_new                     # :359-363
    h = KO.mlp(v, lin("mlp_base_decode", 3))
    density = KO.trunc_exp(h[:, :1]).view(R, S)
    rgb = KO.mlp(h[:, 1:], lin("mlp_head", 4), out_act="Sigmoid").view(R, S, 3)
    return density, rgb, probs.view(R, S, 3)

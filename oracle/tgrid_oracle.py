"""CPU oracle for the temporal hash-grid encoder (config 4) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates NS/field_components/temporal_grid.py (TemporalGridEncoder: level offsets :211-228, the channel
combination table built by init_parameters :231-308, get_temporal_index :320-330, get_temporal_tv_loss :352-376) and the
reference's CUDA kernels NS/field_components/cuda/csrc/temporal_gridencoder.cu (fast_hash :46-59, get_grid_index
:62-88, kernel_grid forward :107-203; the backward :299-369 is obtained here by autograd on the forward).

Pinning: the Python half of the reference IS importable, so offsets / sampling_index / index_list / masks /
get_temporal_index are checked against golden vectors captured from it (tests/golden/g9_tgrid.npz).  The CUDA half
cannot be compiled here (CUDA-only, no nvcc): the forward restatement is pinned by the reference's own known-answer test
(NSR/tests/field_components/test_temporal_grid.py:15-40: out == 0.5, gradient sparsity) and, for the hash function, by the
reference's independent pure-torch HashEncoding.hash_fn (NS/field_components/encodings.py:289-306) -- otherwise
"parity unpinned" for the kernel arithmetic beyond what that KAT fixes.
"""
import math
from typing import Dict

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)  # temporal_gridencoder.cu:49


def level_offsets(num_levels, base_resolution, per_level_scale, log2_hashmap_size, input_dim=3, align_corners=False):
    """temporal_grid.py:211-228: rows per level = min(2^log2T, (res(+1))^D) rounded up to a multiple of 8."""
    max_params = 2**log2_hashmap_size
    offs, off = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale**i))
        n = min(max_params, (res if align_corners else res + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return offs


def resolve_scale(num_levels, base_resolution, per_level_scale, desired_resolution):
    """temporal_grid.py:194-196."""
    if desired_resolution is not None:
        return float(np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1)))
    return float(per_level_scale)


def channel_table(temporal_dim: int, level_dim: int) -> Dict[str, torch.Tensor]:
    """The sliding-window channel combination table (what init_parameters :231-308 constructs procedurally).

    Closed form: the embedding has level_dim + temporal_dim columns.  Row r (r = 0 .. temporal_dim-2) covers one time
    interval.  Output channel q normally reads one column, occ(q, r): the newest column that entered position q before
    row r (column level_dim + k for the largest k < r with k = q mod level_dim), or q itself if none has.  The one
    channel p = r mod level_dim is in transition: it blends column A = occ(p, r) (leaving) with B = level_dim + r (entering).
    sampling_index[r] = per channel (w_a, col_a, w_b, col_b); index_list[r][:2] = (A, B).
    """
    C, T = level_dim, temporal_dim
    rows = max(T - 1, 1)
    samp = torch.zeros(rows, 4 * C, dtype=torch.long)
    mask_a = torch.zeros(rows, 4 * C, dtype=torch.bool)
    mask_b = torch.zeros(rows, 4 * C, dtype=torch.bool)
    ab = torch.zeros(rows, 2, dtype=torch.long)

    def occ(q, r):
        ks = [k for k in range(r) if k % C == q]
        return C + ks[-1] if ks else q

    for r in range(rows):
        p = r % C
        for q in range(C):
            samp[r, 4 * q + 0] = 1
            samp[r, 4 * q + 1] = occ(q, r)
        samp[r, 4 * p + 3] = C + r
        mask_a[r, 4 * p] = True
        mask_b[r, 4 * p + 2] = True
        ab[r, 0], ab[r, 1] = occ(p, r), C + r
    return {"sampling_index": samp, "index_a_mask": mask_a, "index_b_mask": mask_b, "index_ab": ab}


def temporal_index(time: torch.Tensor, table: Dict[str, torch.Tensor]) -> torch.Tensor:
    """get_temporal_index, temporal_grid.py:320-330: time [B] in [0,1] -> [B, 4*C] float rows."""
    samp = table["sampling_index"]
    n = samp.shape[0] - 1
    v = time * n
    r = v.long()
    r[time == 1] = n
    out = samp[r].float()
    out[table["index_a_mask"][r]] = (r + 1 - v)
    out[table["index_b_mask"][r]] = (v - r)
    return out


def fast_hash(pos_grid: torch.Tensor) -> torch.Tensor:
    """temporal_gridencoder.cu:46-59: XOR of pos*prime in uint32 arithmetic. pos_grid int64 [..., D] (values < 2^32)."""
    res = torch.zeros(pos_grid.shape[:-1], dtype=torch.int64)
    for i in range(pos_grid.shape[-1]):
        res = res ^ ((pos_grid[..., i] * PRIMES[i]) & 0xFFFFFFFF)
    return res


def grid_row_index(pos_grid, hashmap_size, resolution, gridtype, align_corners=False):
    """get_grid_index, temporal_gridencoder.cu:62-88 without the channel term: row index in [0, hashmap_size)."""
    D = pos_grid.shape[-1]
    stride, index = 1, torch.zeros(pos_grid.shape[:-1], dtype=torch.int64)
    for d in range(D):
        if stride > hashmap_size:
            break
        index = (index + pos_grid[..., d] * stride) & 0xFFFFFFFF
        stride = (stride * (resolution if align_corners else resolution + 1)) & 0xFFFFFFFF
    if gridtype == 0 and stride > hashmap_size:
        index = fast_hash(pos_grid)
    return index % hashmap_size


def encode(x, trow, embeddings, offsets, log2_scale, base_res, gridtype, level_dim, align_corners=False):
    """kernel_grid forward, temporal_gridencoder.cu:107-203.  x [B,D] in [0,1]; trow [B,4C]; embeddings [rows, grid_C].
    Returns [B, L*C] (the layout TemporalGridEncodeFunc returns after its permute, temporal_grid.py:108)."""
    B, D = x.shape
    C, L = level_dim, len(offsets) - 1
    oob = ((x < 0) | (x > 1)).any(dim=-1)
    outs = []
    for lvl in range(L):
        hsize = offsets[lvl + 1] - offsets[lvl]
        scale = float(np.float32(np.exp2(np.float32(lvl * log2_scale))) * np.float32(base_res) - np.float32(1.0))
        resolution = int(math.ceil(scale)) + 1
        pos = x * scale + (0.0 if align_corners else 0.5)
        pg = torch.floor(pos)
        frac = pos - pg
        pg = pg.long()
        res = torch.zeros(B, C, dtype=embeddings.dtype)
        for corner in range(1 << D):
            w = torch.ones(B, dtype=x.dtype)
            pgl = pg.clone()
            for d in range(D):
                if corner & (1 << d):
                    w = w * frac[:, d]
                    pgl[:, d] += 1
                else:
                    w = w * (1 - frac[:, d])
            row = offsets[lvl] + grid_row_index(pgl, hsize, resolution, gridtype, align_corners)
            for ch in range(C):
                wa, ca, wb, cb = trow[:, 4 * ch], trow[:, 4 * ch + 1].round().long(), trow[:, 4 * ch + 2], trow[:, 4 * ch + 3].round().long()
                ga, gb = embeddings[row, ca], embeddings[row, cb]
                single = wa == 1
                val = torch.where(single, ga, ga * wa + gb * wb)
                res[:, ch] = res[:, ch] + w * val
        res = torch.where(oob[:, None], torch.zeros_like(res), res)
        outs.append(res)
    return torch.stack(outs, dim=1).reshape(B, L * C)


def temporal_tv_loss(embeddings, index_ab_row):
    """get_temporal_tv_loss, temporal_grid.py:352-376 for a given (not random) table row (A, B)."""
    return (embeddings[:, index_ab_row[0]] - embeddings[:, index_ab_row[1]]).abs().mean()


# ----------------------------------------------------------------------------------------------
# NeRFPlayer-nerfacto fields on top of the encoder (NS/fields/nerfplayer_nerfacto_field.py)
# ----------------------------------------------------------------------------------------------
def sh4(d):
    """tcnn SphericalHarmonics degree 4 on unit directions (the field shifts to [0,1], tcnn maps back: base_field.py:131-137)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        0.28209479177387814 * torch.ones_like(x), -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], dim=-1)


def density_field_forward(positions, times, aabb, enc, emb, linear_w):
    """TemporalHashMLPDensityField.get_density (:133-146). enc: dict(offsets, log2_scale, base_res, gridtype, level_dim, table)."""
    from oracle import kplanes_oracle as KO

    R, S = positions.shape[:2]
    p = KO.normalize_positions(positions, aabb).reshape(-1, 3)
    t = times.expand(R, S).reshape(-1) if times.dim() == 2 else times
    x = encode(p, temporal_index(t, enc["table"]), emb, enc["offsets"], enc["log2_scale"], enc["base_res"], enc["gridtype"], enc["level_dim"])
    return KO.trunc_exp(KO.mlp(x, linear_w)).view(R, S)


def main_field_forward(positions, directions, times, aabb, enc, emb, decode_w, head_w, appearance):
    """NerfplayerNerfactoField.get_density/get_outputs (:313-409). directions [R,3] unit; appearance [R,32]."""
    from oracle import kplanes_oracle as KO

    R, S = positions.shape[:2]
    p = KO.normalize_positions(positions, aabb).reshape(-1, 3)
    t = times.expand(R, S).reshape(-1)
    x = encode(p, temporal_index(t, enc["table"]), emb, enc["offsets"], enc["log2_scale"], enc["base_res"], enc["gridtype"], enc["level_dim"])
    h = KO.mlp(x, decode_w)
    density = KO.trunc_exp(h[:, :1]).view(R, S)
    ex = lambda v: v[:, None, :].expand(R, S, v.shape[-1]).reshape(R * S, -1)
    rgb = KO.mlp(torch.cat([ex(sh4(directions)), h[:, 1:], ex(appearance)], dim=-1), head_w, out_act="Sigmoid").view(R, S, 3)
    return density, rgb

"""G9c: temporal-grid INTERPOLATION pinned through reference-executed code (TEST INFRASTRUCTURE; build container only).

The reference's temporal hash-grid kernel is CUDA-only (NS/field_components/cuda/csrc/temporal_gridencoder.cu) and cannot run here, but
the same tree holds an independent pure-torch multiresolution hash encoder, HashEncoding.pytorch_fwd
(NS/field_components/encodings.py:289-347): floor / ceil corners, trilinear weights, the Instant-NGP hash.  For a HASHED level of the
temporal grid the two coincide once two conventions are mapped onto each other:
  * position: the temporal grid looks up pos = x * scale_l + 0.5 (align_corners = False, .cu:137-139), HashEncoding pos' = x' * scaling;
    feeding x' = pos / 16 to a HashEncoding whose level-0 scaling is 16 gives pos' == pos exactly (a power-of-two division);
  * channels: output channel c of the temporal grid is w_a * E[row, col_a] + w_b * E[row, col_b] with (w_a, col_a, w_b, col_b) from
    TemporalGridEncoder.get_temporal_index (importable, temporal_grid.py:320-330) -- linear in the table, so per sample group with one
    time value the HashEncoding table is that column combination of the level's rows.
Expected outputs are therefore produced by the REFERENCE's pytorch_fwd + get_temporal_index; only x -> pos (scale_l = 2^(l S) H - 1,
.cu:128-131) is restated.  tests/test_oracle_tgrid.py checks the oracle's encoder against it, tests/test_gpu_tgrid.py the HIP kernel.

    python -m oracle.gen_golden_tgrid_interp      ->  tests/golden/g9c_tgrid_interp.npz
"""
import os

import numpy as np
import torch

from oracle._refimport import import_reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g9c_tgrid_interp.npz")


def main():
    import_reference()
    from nerfstudio.field_components.encodings import HashEncoding
    from nerfstudio.field_components.temporal_grid import TemporalGridEncoder

    gen = torch.Generator().manual_seed(4242)
    cases = {"a": dict(temporal_dim=4, level_dim=2, num_levels=3, log2_hashmap_size=7, base_resolution=8, per_level_scale=1.5),
             "b": dict(temporal_dim=8, level_dim=4, num_levels=2, log2_hashmap_size=9, base_resolution=16, per_level_scale=2.0)}
    g = {}
    for name, kw in cases.items():
        enc = TemporalGridEncoder(**kw)
        C, L, T = kw["level_dim"], kw["num_levels"], 2 ** kw["log2_hashmap_size"]
        offsets = enc.offsets.tolist()
        assert all(offsets[l + 1] - offsets[l] == T for l in range(L)), "every level must be hashed (full-size table)"
        rows, grid_C = enc.embeddings.shape
        emb = torch.rand(rows, grid_C, generator=gen) * 2 - 1
        times = torch.tensor([0.0, 0.37, 0.5, 0.81, 1.0])
        per_t = 40
        x = torch.rand(len(times) * per_t, 3, generator=gen)
        x[0] = torch.tensor([0.0, 0.5, 1.0])  # box faces
        t_all = times.repeat_interleave(per_t)
        trow = enc.get_temporal_index(t_all)  # [B, 4C], the reference's own rows
        S = float(np.log2(kw["per_level_scale"]))
        out = torch.zeros(x.shape[0], L * C)
        for l in range(L):
            scale = np.float32(np.exp2(np.float32(l * S))) * np.float32(kw["base_resolution"]) - np.float32(1.0)
            pos = x * float(scale) + 0.5  # fp32, as the kernel forms it
            for ti in range(len(times)):
                sl = slice(ti * per_t, (ti + 1) * per_t)
                r = trow[ti * per_t]
                table = torch.stack([r[4 * c] * emb[offsets[l]:offsets[l + 1], int(r[4 * c + 1])] +
                                     r[4 * c + 2] * emb[offsets[l]:offsets[l + 1], int(r[4 * c + 3])] for c in range(C)], dim=1)
                he = HashEncoding(num_levels=2, min_res=16, max_res=32, log2_hashmap_size=kw["log2_hashmap_size"], features_per_level=C,
                                  implementation="torch")
                with torch.no_grad():
                    he.hash_table[:T] = table
                    out[sl, l * C:(l + 1) * C] = he.pytorch_fwd(pos[sl] / 16.0)[:, :C]
        g.update({f"{name}_x": x, f"{name}_times": t_all, f"{name}_trow": trow, f"{name}_emb": emb, f"{name}_offsets": enc.offsets,
                  f"{name}_out": out, f"{name}_cfg": torch.tensor([kw["temporal_dim"], C, L, kw["log2_hashmap_size"], kw["base_resolution"]]),
                  f"{name}_per_level_scale": torch.tensor(float(kw["per_level_scale"]))})
    np.savez_compressed(OUT, **{k: v.numpy() for k, v in g.items()})
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

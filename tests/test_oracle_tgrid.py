"""CPU: temporal-grid oracle vs golden vectors from the reference's Python half, and vs the reference's own known-answer
test of its CUDA kernel (NSR/tests/field_components/test_temporal_grid.py:15-40)."""
import numpy as np
import pytest
import torch

from oracle import tgrid_oracle as TO
from tests.conftest import load_golden

CASES = {
    "main": dict(temporal_dim=64, level_dim=2, num_levels=16, log2_hashmap_size=19, base_resolution=16, desired_resolution=2048, input_dim=3),
    "prop0": dict(temporal_dim=32, level_dim=2, num_levels=5, log2_hashmap_size=17, base_resolution=16,
                  per_level_scale=float(np.exp((np.log(64) - np.log(16)) / 4)), input_dim=3),
    "prop1": dict(temporal_dim=32, level_dim=2, num_levels=5, log2_hashmap_size=17, base_resolution=16,
                  per_level_scale=float(np.exp((np.log(256) - np.log(16)) / 4)), input_dim=3),
    "kat": dict(temporal_dim=2, level_dim=1, num_levels=1, log2_hashmap_size=2, base_resolution=1, per_level_scale=1, input_dim=1),
    "c4": dict(temporal_dim=8, level_dim=4, num_levels=3, log2_hashmap_size=10, base_resolution=4, per_level_scale=2.0, input_dim=3),
    "c3": dict(temporal_dim=5, level_dim=3, num_levels=2, log2_hashmap_size=8, base_resolution=4, per_level_scale=2.0, input_dim=2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_tables_match_reference(name):
    g = load_golden("g9_tgrid")
    kw = CASES[name]
    scale = TO.resolve_scale(kw["num_levels"], kw["base_resolution"], kw.get("per_level_scale", 2.0), kw.get("desired_resolution"))
    assert abs(scale - float(g[f"{name}_per_level_scale"])) < 1e-12
    offs = TO.level_offsets(kw["num_levels"], kw["base_resolution"], scale, kw["log2_hashmap_size"], kw["input_dim"])
    assert offs == g[f"{name}_offsets"].tolist()
    assert [offs[-1], kw["level_dim"] + kw["temporal_dim"]] == g[f"{name}_embed_shape"].tolist()
    tab = TO.channel_table(kw["temporal_dim"], kw["level_dim"])
    assert torch.equal(tab["sampling_index"], g[f"{name}_sampling_index"])
    assert torch.equal(tab["index_a_mask"], g[f"{name}_mask_a"]) and torch.equal(tab["index_b_mask"], g[f"{name}_mask_b"])
    assert torch.equal(tab["index_ab"], g[f"{name}_index_list"][:, :2])
    torch.testing.assert_close(TO.temporal_index(g["times"], tab), g[f"{name}_trow"], rtol=0, atol=0)


def test_main_table_sizes_of_config4():
    """main table [6 119 864, 66], offsets [0, 4920, 18744, 51512, 136696, ...]."""
    g = load_golden("g9_tgrid")
    assert g["main_embed_shape"].tolist() == [6119864, 66]
    assert g["main_offsets"].tolist()[:5] == [0, 4920, 18744, 51512, 136696]
    assert g["prop0_embed_shape"].tolist() == [289584, 34] and g["prop1_embed_shape"].tolist() == [434080, 34]


def test_known_answer_of_the_reference_cuda_test():
    """temporal_dim=2, input_dim=1, 1 level, level_dim=1, base_resolution=1, log2_hashmap_size=2, 'tiled';
    embeddings[:,0] = arange(8), x = 0, t = 0  =>  out == 0.5 everywhere; grad only in rows 0,1 of column 0."""
    kw = CASES["kat"]
    offs = TO.level_offsets(1, 1, 1.0, 2, input_dim=1)
    tab = TO.channel_table(2, 1)
    emb = torch.rand(offs[-1], 3)
    emb[:, 0] = torch.arange(8.0)
    emb.requires_grad_(True)
    x, t = torch.zeros(1024, 1), torch.zeros(1024)
    out = TO.encode(x, TO.temporal_index(t, tab), emb, offs, 0.0, 1, gridtype=1, level_dim=1)
    assert torch.all(out == 0.5)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    assert abs(float(emb.grad.sum() - w.sum())) < 0.01
    assert torch.all(emb.grad[2:, :] == 0) and torch.all(emb.grad[:, 1:] == 0)


def test_hash_primes_match_reference_pure_torch_hash():
    """NS/field_components/encodings.py:301 uses the same first three primes."""
    assert TO.PRIMES[:3] == (1, 2654435761, 805459861)
    pg = torch.tensor([[1, 1, 1], [0, 1, 0], [1, 0, 1]])
    assert TO.fast_hash(pg).tolist() == [(1 ^ 2654435761 ^ 805459861), 2654435761, 1 ^ 805459861]


def test_oob_and_interpolation_properties():
    gen = torch.Generator().manual_seed(0)
    offs = TO.level_offsets(3, 4, 2.0, 10)
    tab = TO.channel_table(8, 2)
    emb = torch.rand(offs[-1], 10, generator=gen)
    x = torch.rand(50, 3, generator=gen)
    x[0, 1] = 1.2
    x[1, 0] = -0.01
    t = torch.rand(50, generator=gen)
    out = TO.encode(x, TO.temporal_index(t, tab), emb, offs, 1.0, 4, 0, 2)
    assert out.shape == (50, 6) and torch.all(out[:2] == 0)
    # constant table => constant output (interpolation weights sum to 1 in space and time)
    out2 = TO.encode(x[2:], TO.temporal_index(t[2:], tab), torch.full_like(emb, 0.7), offs, 1.0, 4, 0, 2)
    torch.testing.assert_close(out2, torch.full_like(out2, 0.7), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("log2", [8, 17, 19])
def test_fast_hash_matches_reference_hash_encoding_on_random_corners(log2):
    """G9b (oracle/gen_golden_hash.py): the reference's importable pure-torch hash on 512 random corners == fast_hash mod 2^k."""
    g = load_golden("g9b_hash")
    pos, want = g[f"pos_{log2}"].long(), g[f"hash_{log2}"].long()
    got = TO.fast_hash(pos) % (1 << log2)
    assert torch.equal(got, want)


def _g12_enc(g, prefix, temporal_dim, num_levels, log2, max_res=None):
    """Encoder description dict of oracle.tgrid_oracle.encode for one of G12's tables."""
    if max_res is None:  # main field: desired_resolution 2048 * scene scale 1 (nerfplayer_nerfacto_field.py:250-262)
        scale = TO.resolve_scale(num_levels, 16, 2.0, 2048)
    else:
        scale = float(np.exp((np.log(max_res) - np.log(16)) / (num_levels - 1)))
    offsets = TO.level_offsets(num_levels, 16, scale, log2)
    assert offsets[-1] == g["param_" + prefix + ".embeddings"].shape[0]
    return {"offsets": offsets, "log2_scale": float(np.log2(scale)), "base_res": 16, "gridtype": 0, "level_dim": 2,
            "table": TO.channel_table(temporal_dim, 2)}


def test_oracle_fields_reproduce_reference_model_golden():
    """G12 (the reference's NerfplayerNerfactoModel run on the CPU): from the stored sample bins, the oracle's proposal density field
    and main field give the reference's weights at every level and its composited rgb."""
    from oracle import kplanes_oracle as KO

    g = load_golden("g12_nerfplayer")
    o, d, times = g["origins"], g["directions"], g["times"]
    aabb = torch.tensor([[-1.0] * 3, [1.0] * 3])
    lin = lambda prefix, n: [g[f"param_{prefix}.layers.{i}.weight"] for i in range(n)]
    for lvl in range(2):
        eb = g[f"ebins_{lvl}"]
        pos = o[:, None, :] + d[:, None, :] * ((eb[:, :-1] + eb[:, 1:]) / 2)[..., None]
        enc = _g12_enc(g, f"proposal_networks.{lvl}.encoding", 4, 3, 9, max_res=(32, 64)[lvl])
        dens = TO.density_field_forward(pos, times, aabb, enc, g[f"param_proposal_networks.{lvl}.encoding.embeddings"], lin(f"proposal_networks.{lvl}.linear", 2))
        w = KO.get_weights(eb[:, 1:] - eb[:, :-1], dens)
        torch.testing.assert_close(w, g[f"weights_{lvl}"], rtol=1e-4, atol=1e-6)
    eb = g["ebins_2"]
    pos = o[:, None, :] + d[:, None, :] * ((eb[:, :-1] + eb[:, 1:]) / 2)[..., None]
    enc = _g12_enc(g, "field.mlp_base", 8, 4, 10)
    app = g["param_field.embedding_appearance.embedding.weight"][g["cams"][:, 0].long()]
    dens, rgb = TO.main_field_forward(pos, d, times, aabb, enc, g["param_field.mlp_base.embeddings"], lin("field.mlp_base_decode", 2),
                                      lin("field.mlp_head", 3), app)
    w = KO.get_weights(eb[:, 1:] - eb[:, :-1], dens)
    torch.testing.assert_close(w, g["weights_2"], rtol=1e-4, atol=1e-6)
    comp = (w[..., None] * rgb).sum(1) + g["bg"] * (1 - w.sum(1, keepdim=True))
    torch.testing.assert_close(comp, g["rgb"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(w.sum(1, keepdim=True), g["accumulation"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["a", "b"])
def test_interpolation_matches_reference_hash_encoding(name):
    """G9c: the oracle's encoder on hashed levels == the reference's own HashEncoding.pytorch_fwd + get_temporal_index (corner order,
    trilinear weights, hash, channel blending all executed by reference code; oracle/gen_golden_tgrid_interp.py)."""
    g = load_golden("g9c_tgrid_interp")
    tdim, C, L, log2T, H = [int(v) for v in g[f"{name}_cfg"]]
    pls = float(g[f"{name}_per_level_scale"])
    offs = TO.level_offsets(L, H, pls, log2T, 3)
    assert offs == g[f"{name}_offsets"].tolist()
    tab = TO.channel_table(tdim, C)
    trow = TO.temporal_index(g[f"{name}_times"], tab)
    torch.testing.assert_close(trow, g[f"{name}_trow"], rtol=0, atol=0)
    out = TO.encode(g[f"{name}_x"], trow, g[f"{name}_emb"], offs, float(np.log2(pls)), H, gridtype=0, level_dim=C)
    torch.testing.assert_close(out, g[f"{name}_out"], rtol=1e-5, atol=2e-6)

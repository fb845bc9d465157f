"""GPU parity: temporal hash-grid encoder (csrc/tgrid.hip) vs the CPU oracle and the reference's own known-answer test;
NeRFPlayer-nerfacto fields/model built on it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _enc_dict(enc):
    from oracle import tgrid_oracle as TO

    return {"offsets": enc.offsets.tolist(), "log2_scale": float(np.log2(enc.per_level_scale)), "base_res": enc.base_resolution,
            "gridtype": enc.gridtype_id, "level_dim": enc.level_dim, "table": TO.channel_table(enc.temporal_dim, enc.level_dim)}


def test_reference_known_answer():
    """NSR/tests/field_components/test_temporal_grid.py:15-40, verbatim parameters and assertions."""
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder

    model = TemporalGridEncoder(temporal_dim=2, input_dim=1, num_levels=1, level_dim=1, per_level_scale=1, base_resolution=1,
                                log2_hashmap_size=2, desired_resolution=None, gridtype="tiled", align_corners=False).to(DEV)
    emb = torch.rand_like(model.embeddings)
    emb[:, 0] = torch.arange(8).to(emb)
    model.embeddings = torch.nn.Parameter(emb, requires_grad=True)
    x, t = torch.zeros([1024, 1], device=DEV), torch.zeros([1024, 1], device=DEV)
    for explicit in (False, True):
        model.embeddings.grad = None
        out = model(x, t, explicit_rows=explicit)
        weight = torch.randn_like(out)
        (out * weight).sum().backward()
        assert torch.all(out == 0.5)
        assert model.embeddings.grad.sum() - weight.sum() < 0.01
        assert torch.all(model.embeddings.grad[2:, :] == 0)
        assert torch.all(model.embeddings.grad[:, 1:] == 0)
    model.get_temporal_tv_loss()


@pytest.mark.parametrize("gridtype", ["hash", "tiled"])
def test_dense_3d_levels_hand_computed_kat(gridtype):
    """tgrid_kernel on DENSE 3-D levels against numbers computed by hand from the reference kernel's text (tests/tgrid_dense_kat.py): the
    levels config 4 takes with (res + 1)^3 <= 2^19 were pinned only by oracle == kernel before."""
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder
    from tests import tgrid_dense_kat as K

    model = TemporalGridEncoder(gridtype=gridtype, **K.KW).to(DEV)
    assert model.offsets.tolist() == K.OFFSETS
    x, t, want = K.inputs(torch)
    for explicit in (False, True):
        model.embeddings = torch.nn.Parameter(K.embedding(torch).to(DEV), requires_grad=True)
        out = model(x.to(DEV), t.to(DEV), explicit_rows=explicit)
        assert torch.equal(out.cpu(), want), out
        model(x[1:2].to(DEV), t[1:2].to(DEV), explicit_rows=explicit).sum().backward()
        assert torch.equal(model.embeddings.grad.cpu(), K.expected_grad(torch))


@pytest.mark.parametrize("kw", [
    dict(temporal_dim=8, level_dim=2, num_levels=4, log2_hashmap_size=10, base_resolution=4, per_level_scale=1.7),            # hashed levels
    dict(temporal_dim=6, level_dim=4, num_levels=3, log2_hashmap_size=12, base_resolution=3, per_level_scale=2.0),
    dict(temporal_dim=5, level_dim=1, num_levels=2, log2_hashmap_size=8, base_resolution=4, per_level_scale=2.0, gridtype="tiled"),
    dict(temporal_dim=16, level_dim=8, num_levels=2, log2_hashmap_size=9, base_resolution=5, per_level_scale=1.5),
    dict(temporal_dim=4, level_dim=2, num_levels=2, log2_hashmap_size=8, base_resolution=4, per_level_scale=2.0, input_dim=2),
])
def test_encode_fwd_bwd_matches_oracle(kw):
    from oracle import tgrid_oracle as TO
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder

    gen = torch.Generator().manual_seed(1)
    enc = TemporalGridEncoder(**kw)
    D = enc.input_dim
    with torch.no_grad():
        enc.embeddings.copy_(torch.rand(enc.embeddings.shape, generator=gen) - 0.5)
    B = 777
    x = torch.rand(B, D, generator=gen)
    x[0, 0], x[1, -1], x[2] = 1.0, 0.0, -0.1  # borders + one out-of-bounds point
    t = torch.rand(B, 1, generator=gen)
    t[3], t[4] = 1.0, 0.0
    emb = enc.embeddings.detach().clone().requires_grad_(True)
    ed = _enc_dict(enc)
    ref = TO.encode(x, TO.temporal_index(t[:, 0], ed["table"]), emb, ed["offsets"], ed["log2_scale"], ed["base_res"], ed["gridtype"], ed["level_dim"])
    go = torch.rand(ref.shape, generator=gen) - 0.5
    ref.backward(go)
    enc = enc.to(DEV)
    for explicit in (False, True):
        enc.embeddings.grad = None
        out = enc(x.to(DEV), t.to(DEV), explicit_rows=explicit)
        torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        assert torch.all(out[2] == 0)
        out.backward(go.to(DEV))
        torch.testing.assert_close(enc.embeddings.grad.cpu(), emb.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(enc.get_temporal_index(t[:, 0].to(DEV)).cpu(), TO.temporal_index(t[:, 0], ed["table"]), rtol=0, atol=0)
    # coordinate gradient (calc_grad_inputs: dy_dx in the forward, kernel_input_backward, temporal_gridencoder.cu:204-273,373-398) against autograd
    # through the oracle's interpolation weights (floor has no gradient: the same piecewise-linear derivative); zero for the out-of-range point
    xo = x.clone().requires_grad_(True)
    ref2 = TO.encode(xo, TO.temporal_index(t[:, 0], ed["table"]), emb.detach(), ed["offsets"], ed["log2_scale"], ed["base_res"], ed["gridtype"], ed["level_dim"])
    ref2.backward(go)
    for explicit in (False, True):
        xd = x.to(DEV).requires_grad_(True)
        enc.embeddings.grad = None
        out = enc(xd, t.to(DEV), explicit_rows=explicit)
        torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        out.backward(go.to(DEV))
        torch.testing.assert_close(xd.grad.cpu(), xo.grad, rtol=2e-4, atol=1e-5 * float(xo.grad.abs().max()))
        assert torch.all(xd.grad[2] == 0)
        torch.testing.assert_close(enc.embeddings.grad.cpu(), emb.grad, rtol=1e-4, atol=1e-6)  # the table's gradient is unchanged by it


def _make_model(num_images=5):
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModel, NerfplayerNerfactoModelConfig
    from soccernerfs_amd.scene_colliders import SceneBox

    cfg = NerfplayerNerfactoModelConfig(
        num_levels=6, log2_hashmap_size=12, temporal_dim=16,
        proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 32},
                                {"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 64}],
        num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=16)
    torch.manual_seed(0)
    model = NerfplayerNerfactoModel(cfg, SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=num_images)
    with torch.no_grad():  # the 1e-4 init gives a featureless field: use O(1) tables for a meaningful comparison
        for e in [model.field.mlp_base] + [p.encoding for p in model.proposal_networks]:
            e.embeddings.uniform_(-1.0, 1.0)
    return model


def test_nerfplayer_fields_match_oracle():
    from oracle import tgrid_oracle as TO
    from soccernerfs_amd.kplanes_field import FieldHeadNames
    from soccernerfs_amd.ray_samplers import UniformSampler
    from soccernerfs_amd.rays import RayBundle

    model = _make_model()
    gen = torch.Generator().manual_seed(4)
    R, S = 30, 16
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.5
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    times = torch.rand(R, 1, generator=gen)
    cams = torch.randint(0, 5, (R, 1), generator=gen)
    aabb = model.scene_box.aabb
    nears = torch.rand(R, 1, generator=gen) * 0.1
    fars = nears + 0.5
    sb = torch.linspace(0.0, 1.0, S + 1)[None, :].expand(R, -1)
    eb = sb * fars + (1 - sb) * nears
    pos = o[:, None, :] + d[:, None, :] * ((eb[:, :-1] + eb[:, 1:]) / 2)[..., None]
    f = model.field
    ref_d, ref_rgb = TO.main_field_forward(pos, d, times, aabb, _enc_dict(f.mlp_base), f.mlp_base.embeddings.detach(),
                                            f.mlp_base_decode.linear_weights(), f.mlp_head.linear_weights(),
                                            f.embedding_appearance.weight.detach()[cams[:, 0]])
    pn = model.proposal_networks[0]
    ref_pd = TO.density_field_forward(pos, times, aabb, _enc_dict(pn.encoding), pn.encoding.embeddings.detach(), pn.linear.linear_weights())
    model = model.to(DEV).train()
    model.scene_box.aabb = aabb.to(DEV)
    rb = RayBundle(origins=o.to(DEV), directions=d.to(DEV), pixel_area=torch.ones(R, 1, device=DEV), camera_indices=cams.to(DEV),
                   nears=nears.to(DEV), fars=fars.to(DEV), times=times.to(DEV))
    smp = UniformSampler()
    smp.eval()  # no jitter: bins = linspace, the same sample positions as above
    rs = smp(rb, num_samples=S)
    torch.testing.assert_close(rs._compact["ebins"].cpu(), eb, rtol=1e-6, atol=1e-7)
    out = model.field(rs)
    torch.testing.assert_close(out[FieldHeadNames.DENSITY][..., 0].cpu(), ref_d, rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(out[FieldHeadNames.RGB].cpu(), ref_rgb, rtol=1e-4, atol=1e-5)
    # proposal field: positions path (as the reference calls density_fn) and in-kernel ray path
    pd1 = model.proposal_networks[0].density_fn(pos.to(DEV), times=times.to(DEV))
    pd2 = model.proposal_networks[0].density_from_ray_samples(rs)
    torch.testing.assert_close(pd1[..., 0].cpu(), ref_pd, rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(pd2[..., 0].cpu(), ref_pd, rtol=2e-4, atol=1e-6)


def test_nerfplayer_model_trains():
    """End-to-end plugin surface: forward, metrics, loss dict keys, backward reaches every trainable tensor, loss decreases."""
    from soccernerfs_amd.rays import RayBundle

    model = _make_model().to(DEV).train()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    gen = torch.Generator().manual_seed(9)
    R = 64
    o = ((torch.rand(R, 3, generator=gen) * 2 - 1) * 0.3).to(DEV)
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1).to(DEV)
    cams = torch.randint(0, 5, (R, 1), generator=gen).to(DEV)
    times = torch.rand(R, 1, generator=gen).to(DEV)
    target = torch.rand(R, 3, generator=gen).to(DEV)
    opt = torch.optim.Adam([p for g in model.get_param_groups().values() for p in g if p.requires_grad], lr=1e-2, eps=1e-12)
    losses = []
    for step in range(8):
        for where, fn in model.get_training_callbacks():
            if where == "before":
                fn(step)
        out = model(RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1, device=DEV), camera_indices=cams, times=times))
        md = model.get_metrics_dict(out, {"image": target})
        ld = model.get_loss_dict(out, {"image": target}, md)
        assert set(ld) == {"rgb_loss", "interlevel_loss", "distortion_loss", "temporal_tv_loss"}
        loss = sum(ld.values())
        opt.zero_grad()
        loss.backward()
        if step == 0:
            missing = [n for n, p in model.named_parameters() if p.requires_grad and p.numel() and p.grad is None]
            assert not missing, missing
        opt.step()
        for where, fn in model.get_training_callbacks():
            if where == "after":
                fn(step)
        losses.append(float(ld["rgb_loss"]))
    assert out["depth"].shape == (R, 1) and out["rgb"].shape == (R, 3)
    assert losses[-1] < losses[0], losses


def test_fused_tv_and_fused_adam_match_torch_adam():
    """The temporal-TV term riding on the encoder's autograd node + optimizers.FusedAdam (gradient scattered straight into the
    persistent .grad, cleared by the optimiser sweep) reproduce separate TV autograd + torch.optim.Adam over three steps."""
    import copy
    from soccernerfs_amd.optimizers import FusedAdam
    from soccernerfs_amd.rays import RayBundle

    base = _make_model().to(DEV).train()
    base.scene_box.aabb = base.scene_box.aabb.to(DEV)
    models = [base, copy.deepcopy(base)]
    encs = [[m.field.mlp_base] + [p.encoding for p in m.proposal_networks] for m in models]
    for e in encs[0]:
        e.fuse_tv = False
    models[0].tv_row_fn = lambda enc: 3
    for e in encs[1]:
        e.tv_row_override = 3
    plist = lambda m: [p for g in m.get_param_groups().values() for p in g if p.requires_grad]
    opts = [torch.optim.Adam(plist(models[0]), lr=1e-2, eps=1e-15), FusedAdam(plist(models[1]), lr=1e-2, eps=1e-15, encoders=encs[1])]
    gen = torch.Generator().manual_seed(5)
    R = 64
    for step in range(3):
        o = ((torch.rand(R, 3, generator=gen) * 2 - 1) * 0.3).to(DEV)
        d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1).to(DEV)
        cams = torch.randint(0, 5, (R, 1), generator=gen).to(DEV)
        times = torch.rand(R, 1, generator=gen).to(DEV)
        target = torch.rand(R, 3, generator=gen).to(DEV)
        losses = []
        for m, opt in zip(models, opts):
            torch.manual_seed(100 + step)  # same sampler jitter / background colours in both models
            for where, fn in m.get_training_callbacks():
                if where == "before":
                    fn(step)
            out = m(RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1, device=DEV), camera_indices=cams, times=times))
            ld = m.get_loss_dict(out, {"image": target}, m.get_metrics_dict(out, {"image": target}))
            loss = sum(ld.values())
            opt.zero_grad()
            loss.backward()
            opt.step()
            for where, fn in m.get_training_callbacks():
                if where == "after":
                    fn(step)
            losses.append({k: float(v.detach()) for k, v in ld.items()})
        for k in losses[0]:
            assert abs(losses[0][k] - losses[1][k]) <= 1e-4 * abs(losses[0][k]) + 1e-9, (step, k, losses)
    for (n0, p0), (n1, p1) in zip(models[0].named_parameters(), models[1].named_parameters()):
        if p0.numel():
            bad = ((p0 - p1).abs() > 2e-4).float().mean()  # Adam's first steps move by ~lr*sign(g): rounding-noise gradients may flip
            assert float(bad) < 2e-3, (n0, float(bad))
    for e in encs[1]:
        assert float(e.embeddings.grad.abs().max()) == 0.0  # cleared by the optimiser sweep


def test_nerfplayer_model_matches_reference_golden():
    """G12 (oracle/gen_golden_nerfplayer.py): the reference's own NerfplayerNerfactoModel, run on the CPU with explicit random draws
    -- outputs, sample bins, every loss term and per-tensor gradient checksums."""
    from tests.conftest import load_golden
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModel, NerfplayerNerfactoModelConfig
    from soccernerfs_amd.rays import RayBundle
    from soccernerfs_amd.scene_colliders import SceneBox

    g = load_golden("g12_nerfplayer")
    cfg = NerfplayerNerfactoModelConfig(
        num_levels=4, features_per_level=2, log2_hashmap_size=10, temporal_dim=8,
        proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 32},
                                {"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 64}],
        num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8)
    model = NerfplayerNerfactoModel(cfg, SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=int(g["num_images"]))
    P = lambda name: g["param_" + name]
    with torch.no_grad():
        model.field.embedding_appearance.weight.copy_(P("field.embedding_appearance.embedding.weight"))
        model.field.mlp_base.embeddings.copy_(P("field.mlp_base.embeddings"))
        model.field.mlp_base_decode.load_linear_weights([P(f"field.mlp_base_decode.layers.{i}.weight") for i in range(2)])
        model.field.mlp_head.load_linear_weights([P(f"field.mlp_head.layers.{i}.weight") for i in range(3)])
        for k, pn in enumerate(model.proposal_networks):
            pn.encoding.embeddings.copy_(P(f"proposal_networks.{k}.encoding.embeddings"))
            pn.linear.load_linear_weights([P(f"proposal_networks.{k}.linear.layers.{i}.weight") for i in range(2)])
    model = model.to(DEV).train()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    t = lambda k: g[k].to(DEV).contiguous()
    draws = [t("t_rand"), t("u0"), t("u1"), t("bg")]
    model.set_rand_fn(lambda shape, device=None: draws.pop(0))
    model.tv_row_fn = lambda enc: int(g["tv_row"])
    for e in [model.field.mlp_base] + [p.encoding for p in model.proposal_networks]:
        e.fuse_tv = False
    model.proposal_sampler.set_anneal(float(g["anneal"]))
    R = int(g["R"])
    rb = RayBundle(origins=t("origins"), directions=t("directions"), pixel_area=torch.ones(R, 1, device=DEV), camera_indices=t("cams"), times=t("times"))
    out = model(rb)
    assert not draws
    for i in range(3):
        torch.testing.assert_close(out["ray_samples_list"][i]._compact["ebins"].cpu(), g[f"ebins_{i}"], rtol=0, atol=3e-5)
        torch.testing.assert_close(out["weights_list"][i][..., 0].cpu() if out["weights_list"][i].dim() == 3 else out["weights_list"][i].cpu(),
                                   g[f"weights_{i}"], rtol=2e-3, atol=2e-5)
    torch.testing.assert_close(out["rgb"].cpu(), g["rgb"], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(out["accumulation"].cpu(), g["accumulation"], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(out["depth"].cpu(), g["depth"], rtol=1e-3, atol=1e-4)
    for i in range(2):
        torch.testing.assert_close(out[f"prop_depth_{i}"].cpu(), g[f"prop_depth_{i}"], rtol=1e-3, atol=1e-4)
    target = t("target")
    md = model.get_metrics_dict(out, {"image": target})
    ld = model.get_loss_dict(out, {"image": target}, md)
    torch.testing.assert_close(md["distortion"].detach().cpu(), torch.as_tensor(g["distortion"]), rtol=2e-3, atol=1e-9)
    assert set(ld) == {"rgb_loss", "interlevel_loss", "distortion_loss", "temporal_tv_loss"}
    for k, v in ld.items():
        torch.testing.assert_close(v.detach().cpu(), torch.as_tensor(g["loss_" + k]), rtol=2e-3, atol=1e-9)
    sum(ld.values()).backward()
    mine = {"field.embedding_appearance.embedding.weight": model.field.embedding_appearance.weight.grad,
            "field.mlp_base.embeddings": model.field.mlp_base.embeddings.grad}
    for i, w in enumerate(model.field.mlp_base_decode.linear_weights(model.field.mlp_base_decode.params.grad)):
        mine[f"field.mlp_base_decode.layers.{i}.weight"] = w
    for i, w in enumerate(model.field.mlp_head.linear_weights(model.field.mlp_head.params.grad)):
        mine[f"field.mlp_head.layers.{i}.weight"] = w
    for k, pn in enumerate(model.proposal_networks):
        mine[f"proposal_networks.{k}.encoding.embeddings"] = pn.encoding.embeddings.grad
        for i, w in enumerate(pn.linear.linear_weights(pn.linear.params.grad)):
            mine[f"proposal_networks.{k}.linear.layers.{i}.weight"] = w
    for name in [str(n) for n in g["param_names"]]:
        got = mine[name].cpu()
        gabs = float(g["gabs_" + name])
        assert abs(float(got.double().sum()) - float(g["gsum_" + name])) <= 3e-3 * gabs + 1e-9, name
        assert abs(float(got.double().abs().sum()) - gabs) <= 3e-3 * gabs + 1e-9, name
        probe = got.flatten()[:: max(1, got.numel() // 64)][:64]
        torch.testing.assert_close(probe, g["gprobe_" + name], rtol=5e-3, atol=1e-7 + 2e-3 * float(g["gprobe_" + name].abs().max()))


@pytest.mark.parametrize("name", ["a", "b"])
def test_interpolation_matches_reference_hash_encoding(name):
    """G9c: csrc/tgrid.hip on hashed levels == the reference's own HashEncoding.pytorch_fwd + get_temporal_index (corner order, trilinear
    weights, hash and channel blending executed by reference code; oracle/gen_golden_tgrid_interp.py), with the time rows derived in-kernel
    and passed explicitly."""
    from tests.conftest import load_golden
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder

    g = load_golden("g9c_tgrid_interp")
    tdim, C, L, log2T, H = [int(v) for v in g[f"{name}_cfg"]]
    enc = TemporalGridEncoder(temporal_dim=tdim, level_dim=C, num_levels=L, log2_hashmap_size=log2T, base_resolution=H,
                              per_level_scale=float(g[f"{name}_per_level_scale"]))
    assert enc.offsets.tolist() == g[f"{name}_offsets"].tolist()
    with torch.no_grad():
        enc.embeddings.copy_(g[f"{name}_emb"])
    enc = enc.to(DEV)
    x, t = g[f"{name}_x"].to(DEV), g[f"{name}_times"][:, None].to(DEV)
    for explicit in (False, True):
        out = enc(x, t, explicit_rows=explicit)
        torch.testing.assert_close(out.cpu(), g[f"{name}_out"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("mode", ["rays", "points"])
def test_run_length_forward_is_bit_identical_to_the_per_sample_kernel(mode):
    """tgrid_fwd_runs_kernel (round 5: the eight corner values of the current cell stay in registers while consecutive samples of a ray remain in it)
    against tgrid_kernel<false> (one gather of all corners per sample; SNERF_TGRID_RUNS=0 selects it): the same products summed in the same order, so the
    outputs must agree BIT FOR BIT -- on the preset's main grid (16 levels, hashed above level 5) with ragged segments (S = 48 -> 2 x 24, S = 37 ->
    2 x 19 / 18), rays that leave the box, and the explicit-point form the full NeRFPlayer feeds with deformed positions."""
    import ctypes as C
    import os

    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder

    gen = torch.Generator().manual_seed(21)
    enc = TemporalGridEncoder(input_dim=3, temporal_dim=64, num_levels=16, level_dim=2, log2_hashmap_size=15, desired_resolution=2048).to(DEV)
    with torch.no_grad():
        enc.embeddings.copy_((torch.rand(enc.embeddings.shape, generator=gen) - 0.5).to(DEV))
    L = _lib.lib()
    for R, S in ((300, 48), (129, 37)):
        o = ((torch.rand(R, 3, generator=gen) * 2 - 1) * 0.9).to(DEV)
        d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1).to(DEV)
        # clustered samples (as after proposal resampling): runs of samples inside one cell on the coarse levels, new cells on the fine ones
        edges = torch.sort(torch.rand(R, S + 1, generator=gen) ** 3 * 1.6, dim=-1).values.to(DEV).contiguous()
        times = torch.rand(R, generator=gen).to(DEV)
        times[0], times[1] = 0.0, 1.0
        B = R * S
        if mode == "rays":
            co, spr, keep = ops.coords_from_rays(o, d, times, edges, [[-1.0] * 3, [1.0] * 3], False), S, None
        else:
            mid = (edges[:, :-1] + edges[:, 1:]) / 2
            pts = (((o[:, None, :] + d[:, None, :] * mid[..., None]) + 1.0) / 2.0).reshape(B, 3).contiguous()
            co, spr, keep = ops.coords_from_points(pts), S, pts
        outs = []
        for runs in ("1", "0"):
            os.environ["SNERF_TGRID_RUNS"] = runs
            try:
                out = torch.full((B, enc.output_dim), 7.0, device=DEV)
                _lib.check(L.snerf_tgrid_encode_fwd(C.byref(enc.desc), ops._ptr(enc.embeddings), C.byref(co), None, ops._ptr(times), spr, C.c_int64(B), ops._ptr(out),
                                                    ops._stream()), "tgrid_encode_fwd")
                torch.cuda.synchronize()
            finally:
                os.environ.pop("SNERF_TGRID_RUNS", None)
            outs.append(out)
        assert float(outs[1].abs().max()) > 0 and bool((outs[1] == 0).all(dim=1).any())  # some samples lie outside the box (all-zero rows)
        assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())

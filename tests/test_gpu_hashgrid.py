"""GPU parity: static multiresolution hash grid (csrc/hashgrid.hip, tcnn HashGrid) vs the CPU oracle (value, table gradient and
coordinate gradient), generic-shape tcnn-style networks, and the full NeRFPlayer model against the reference's own model (G13)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("cfg", [
    dict(n_levels=16, n_features_per_level=2, base_resolution=16, per_level_scale=1.4472692012786865, log2_hashmap_size=15),  # NeRFPlayer's
    dict(n_levels=4, n_features_per_level=2, base_resolution=16, per_level_scale=1.4472692012786865, log2_hashmap_size=13),   # dense + hashed
    dict(n_levels=3, n_features_per_level=4, base_resolution=4, per_level_scale=2.0, log2_hashmap_size=12),                  # all dense
    dict(n_levels=5, n_features_per_level=8, base_resolution=5, per_level_scale=1.5, log2_hashmap_size=9),
    dict(n_levels=6, n_features_per_level=1, base_resolution=3, per_level_scale=1.7, log2_hashmap_size=10, D=2),
])
def test_hashgrid_fwd_bwd_matches_oracle(cfg):
    from oracle import hashgrid_oracle as HG
    from soccernerfs_amd.tcnn_compat import Encoding

    cfg = dict(cfg)
    D = cfg.pop("D", 3)
    gen = torch.Generator().manual_seed(3)
    enc = Encoding(D, {"otype": "HashGrid", **cfg})
    assert float(enc.params.detach().abs().max()) <= 1e-4  # tcnn's U(-1e-4, 1e-4) init
    with torch.no_grad():
        enc.params.copy_(torch.rand(enc.params.shape, generator=gen) - 0.5)
    B = 777
    x = torch.rand(B, D, generator=gen)
    x[0, 0], x[1, -1], x[2], x[3] = 1.0, 0.0, -0.13, 1.21  # borders and points outside [0,1] (no bounds check: wraps / hashes)
    geo = (cfg["n_levels"], cfg["n_features_per_level"], cfg["base_resolution"], cfg["per_level_scale"], cfg["log2_hashmap_size"])
    xr = x.clone().requires_grad_(True)
    tr = enc.params.detach().clone().view(-1, geo[1]).requires_grad_(True)
    ref = HG.encode(xr, tr, *geo)
    go = torch.rand(ref.shape, generator=gen) - 0.5
    ref.backward(go)
    enc = enc.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = enc(xg)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    out.backward(go.to(DEV))
    torch.testing.assert_close(enc.params.grad.cpu().view(-1, geo[1]), tr.grad, rtol=1e-4, atol=1e-6)
    # coordinate gradient: products of O(scale) terms, summed over levels
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max()))
    # table-only and coordinate-only backward paths
    enc.params.grad = None
    enc(x.to(DEV)).backward(go.to(DEV))
    torch.testing.assert_close(enc.params.grad.cpu().view(-1, geo[1]), tr.grad, rtol=1e-4, atol=1e-6)
    xg2 = x.to(DEV).requires_grad_(True)
    from soccernerfs_amd import ops
    ops.hashgrid_encode(xg2, enc.params.detach(), enc.desc).backward(go.to(DEV))
    torch.testing.assert_close(xg2.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max()))


def test_hashgrid_argument_errors():
    from soccernerfs_amd import ops
    from soccernerfs_amd.tcnn_compat import Encoding

    enc = Encoding(3, {"otype": "HashGrid", "n_levels": 2, "n_features_per_level": 2, "base_resolution": 4, "per_level_scale": 2.0,
                       "log2_hashmap_size": 8}).to(DEV)
    with pytest.raises(RuntimeError):
        enc(torch.rand(4, 3))  # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.hashgrid_encode(torch.rand(4, 2, device=DEV), enc.params, enc.desc)
    with pytest.raises(RuntimeError):
        ops.hashgrid_encode(torch.rand(4, 3, device=DEV), enc.params[:-2], enc.desc)
    assert enc(torch.rand(0, 3, device=DEV)).shape == (0, 4)
    with pytest.raises(ValueError):
        Encoding(3, {"otype": "DenseGrid"})


@pytest.mark.parametrize("shape", [(3, 3, 128, 3, "None"), (33, 32, 64, 1, "None"), (15, 3, 64, 3, "Sigmoid"), (40, 7, 32, 4, "None"), (100, 20, 128, 1, "Sigmoid")])
def test_generic_shape_network_matches_torch(shape):
    """Nets outside the fused kernels' table (three or more hidden layers / more than 16 outputs), chained from the dense-layer kernels:
    same flat parameter layout, values and gradients as a bias-free torch Linear stack."""
    from soccernerfs_amd.tcnn_compat import Network

    d_in, d_out, hidden, nh, out_act = shape
    net = Network(d_in, d_out, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": out_act, "n_neurons": hidden, "n_hidden_layers": nh})
    assert not net.fused
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(333, d_in, generator=gen) - 0.5  # ragged: not a multiple of the 64-row tile
    ws = [w.clone().requires_grad_(True) for w in net.linear_weights()]
    h = x.clone().requires_grad_(True)
    y = h
    for i, w in enumerate(ws):
        y = y @ w.t()
        y = torch.relu(y) if i < len(ws) - 1 else (torch.sigmoid(y) if out_act == "Sigmoid" else y)
    go = torch.rand(y.shape, generator=gen) - 0.5
    y.backward(go)
    net = net.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = net(xg)
    torch.testing.assert_close(out.cpu(), y.detach(), rtol=1e-5, atol=1e-6)
    out.backward(go.to(DEV))
    torch.testing.assert_close(xg.grad.cpu(), h.grad, rtol=1e-4, atol=1e-6)
    for got, w in zip(net.linear_weights(net.params.grad), ws):
        torch.testing.assert_close(got.cpu(), w.grad, rtol=1e-4, atol=1e-5)
    # inputs that do not need a gradient, strided rows
    wide = torch.zeros(333, d_in + 5, device=DEV)
    wide[:, :d_in] = x.to(DEV)
    torch.testing.assert_close(net(wide[:, :d_in]).cpu(), y.detach(), rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        net(x)  # CPU tensor


def _full_model(g):
    from soccernerfs_amd.nerfplayer import NerfplayerModel, NerfplayerModelConfig
    from soccernerfs_amd.scene_colliders import SceneBox

    cfg = NerfplayerModelConfig(
        num_levels=4, features_per_level=2, log2_hashmap_size=12, temporal_dim=8,
        proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 32},
                                {"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 64}],
        num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8, prob_reg_loss_mult=0.1)
    model = NerfplayerModel(cfg, SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=int(g["num_images"]))
    P = lambda name: g["param_" + name]
    f = model.field
    nets = {"deformation_field": 4, "stationary_field_mlp": 2, "decomposition_mlp": 2, "mlp_base_decode": 3, "mlp_head": 4}
    with torch.no_grad():
        f.embedding_appearance.weight.copy_(P("field.embedding_appearance.embedding.weight"))
        f.stationary_field.params.copy_(P("field.stationary_field.params"))
        f.newness_field.embeddings.copy_(P("field.newness_field.embeddings"))
        f.decomposition_field.embeddings.copy_(P("field.decomposition_field.embeddings"))
        for name, n in nets.items():
            getattr(f, name).load_linear_weights([P(f"field.{name}.layers.{i}.weight") for i in range(n)])
        for k, pn in enumerate(model.proposal_networks):
            pn.encoding.embeddings.copy_(P(f"proposal_networks.{k}.encoding.embeddings"))
            pn.linear.load_linear_weights([P(f"proposal_networks.{k}.linear.layers.{i}.weight") for i in range(2)])
    model = model.to(DEV).train()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    return model, nets


def test_full_nerfplayer_model_matches_reference_golden():
    """G13 (oracle/gen_golden_nerfplayer_full.py): the reference's own NerfplayerModel run on the CPU with explicit random draws --
    outputs (incl. the rendered decomposition probabilities), sample bins, every loss term, per-tensor gradient checksums."""
    from tests.conftest import load_golden
    from soccernerfs_amd.rays import RayBundle

    g = load_golden("g13_nerfplayer_full")
    model, nets = _full_model(g)
    f = model.field
    t = lambda k: g[k].to(DEV).contiguous()
    draws = [t("t_rand"), t("u0"), t("u1"), t("bg")]
    model.set_rand_fn(lambda shape, device=None: draws.pop(0))
    model.tv_row_fn = lambda enc: int(g["tv_row"])
    encs = [f.newness_field, f.decomposition_field] + [p.encoding for p in model.proposal_networks]
    for e in encs:
        e.fuse_tv = False
    model.proposal_sampler.set_anneal(float(g["anneal"]))
    R = int(g["R"])
    rb = RayBundle(origins=t("origins"), directions=t("directions"), pixel_area=torch.ones(R, 1, device=DEV), camera_indices=t("cams"), times=t("times"))
    out = model(rb)
    assert not draws
    for i in range(3):
        torch.testing.assert_close(out["ray_samples_list"][i]._compact["ebins"].cpu(), g[f"ebins_{i}"], rtol=0, atol=3e-5)
        w = out["weights_list"][i]
        torch.testing.assert_close((w[..., 0] if w.dim() == 3 else w).cpu(), g[f"weights_{i}"], rtol=2e-3, atol=2e-5)
    torch.testing.assert_close(out["rgb"].cpu(), g["rgb"], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(out["probs"].cpu(), g["probs"], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(out["accumulation"].cpu(), g["accumulation"], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(out["depth"].cpu(), g["depth"], rtol=1e-3, atol=1e-4)
    for i in range(2):
        torch.testing.assert_close(out[f"prop_depth_{i}"].cpu(), g[f"prop_depth_{i}"], rtol=1e-3, atol=1e-4)
    target = t("target")
    md = model.get_metrics_dict(out, {"image": target})
    ld = model.get_loss_dict(out, {"image": target}, md)
    assert set(ld) == {"rgb_loss", "interlevel_loss", "distortion_loss", "temporal_tv_loss", "prob_loss"}
    for k, v in ld.items():
        torch.testing.assert_close(v.detach().cpu(), torch.as_tensor(g["loss_" + k]), rtol=2e-3, atol=1e-9)
    sum(ld.values()).backward()
    mine = {"field.stationary_field.params": f.stationary_field.params.grad, "field.newness_field.embeddings": f.newness_field.embeddings.grad,
            "field.decomposition_field.embeddings": f.decomposition_field.embeddings.grad}
    for name in nets:
        net = getattr(f, name)
        for i, w in enumerate(net.linear_weights(net.params.grad)):
            mine[f"field.{name}.layers.{i}.weight"] = w
    for k, pn in enumerate(model.proposal_networks):
        mine[f"proposal_networks.{k}.encoding.embeddings"] = pn.encoding.embeddings.grad
        for i, w in enumerate(pn.linear.linear_weights(pn.linear.params.grad)):
            mine[f"proposal_networks.{k}.linear.layers.{i}.weight"] = w
    assert f.embedding_appearance.weight.grad is None and float(g["gabs_field.embedding_appearance.embedding.weight"]) == 0.0  # unused by the field
    for name in [str(n) for n in g["param_names"]]:
        if name == "field.embedding_appearance.embedding.weight":
            continue
        got = mine[name].cpu()
        gabs = float(g["gabs_" + name])
        assert abs(float(got.double().sum()) - float(g["gsum_" + name])) <= 3e-3 * gabs + 1e-9, name
        assert abs(float(got.double().abs().sum()) - gabs) <= 3e-3 * gabs + 1e-9, name
        probe = got.flatten()[:: max(1, got.numel() // 64)][:64]
        torch.testing.assert_close(probe, g["gprobe_" + name], rtol=5e-3, atol=1e-7 + 2e-3 * float(g["gprobe_" + name].abs().max()), msg=name)


def test_full_nerfplayer_preset_trains():
    """`nerfplayer` preset sizes (log2_hashmap_size 18, 4096 rays): a few optimiser steps on a constant-colour target lower the loss and
    keep every parameter finite; eval mode renders on the white background without the training-only outputs."""
    from soccernerfs_amd.nerfplayer import NerfplayerModel, NerfplayerModelConfig
    from soccernerfs_amd.optimizers import FusedAdam
    from soccernerfs_amd.rays import RayBundle
    from soccernerfs_amd.scene_colliders import SceneBox

    torch.manual_seed(0)
    model = NerfplayerModel(NerfplayerModelConfig(), SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=8).to(DEV).train()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    groups = model.get_param_groups()
    assert set(groups) == {"proposal_networks", "fields"}
    opt = FusedAdam([p for ps in groups.values() for p in ps if p.requires_grad], lr=1e-2, eps=1e-6)
    R = 4096
    o = (torch.rand(R, 3, device=DEV) * 2 - 1) * 0.3
    d = torch.nn.functional.normalize(torch.rand(R, 3, device=DEV) * 2 - 1, dim=-1)
    rb = lambda: RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1, device=DEV), camera_indices=torch.zeros(R, 1, dtype=torch.long, device=DEV),
                           times=torch.rand(R, 1, device=DEV))
    target = torch.full((R, 3), 0.25, device=DEV)
    cbs = model.get_training_callbacks()
    losses = []
    for step in range(12):
        for where, fn in cbs:
            if where == "before":
                fn(step)
        out = model(rb())
        md = model.get_metrics_dict(out, {"image": target})
        ld = model.get_loss_dict(out, {"image": target}, md)
        sum(ld.values()).backward()  # FusedAdam's sweep clears the gradients
        opt.step()
        for where, fn in cbs:
            if where == "after":
                fn(step)
        losses.append(float(ld["rgb_loss"].detach()))
    assert losses[-1] < 0.7 * losses[0], losses
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    model.eval()
    with torch.no_grad():
        out = model(rb())
    assert "weights_list" not in out and out["probs"].shape == (R, 3) and out["rgb"].shape == (R, 3)
    torch.testing.assert_close(out["probs"].sum(-1), out["accumulation"][:, 0], rtol=1e-4, atol=1e-5)  # softmax rows sum to 1


@pytest.mark.parametrize("cfg,sh,lc", [
    (dict(n_levels=16, n_features_per_level=2, base_resolution=16, per_level_scale=1.4472692012786865, log2_hashmap_size=15), 0, 0),  # NeRFPlayer's, default tile size
    (dict(n_levels=16, n_features_per_level=2, base_resolution=16, per_level_scale=1.4472692012786865, log2_hashmap_size=15), 6, 3),  # small tiles; levels 0-2 atomic
    (dict(n_levels=3, n_features_per_level=4, base_resolution=4, per_level_scale=2.0, log2_hashmap_size=12), 4, 0),                   # all dense
    (dict(n_levels=5, n_features_per_level=8, base_resolution=5, per_level_scale=1.5, log2_hashmap_size=9), 3, 1),
    (dict(n_levels=4, n_features_per_level=1, base_resolution=7, per_level_scale=1.6, log2_hashmap_size=11), 5, 2),
])
def test_tiled_table_backward_equals_the_atomic_kernel_and_its_fused_adam_equals_scatter_then_adam(cfg, sh, lc):
    """csrc/hashgrid_tiles.hip (round 6, ABI 14): the owner-computes table backward against hashgrid_kernel's atomic scatter (itself pinned against the oracle's
    autograd above) -- same entries, summation-order accuracy, ACCUMULATING -- and its fused Adam form against scatter -> snerf_adam_step over three steps."""
    import ctypes as C

    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.tcnn_compat import Encoding, TiledHashTableBackward

    gen = torch.Generator().manual_seed(8)
    enc = Encoding(3, {"otype": "HashGrid", **cfg}).to(DEV)
    F = cfg["n_features_per_level"]
    L = _lib.lib()
    for B in (5000, 1237):
        x = torch.rand(B, 3, generator=gen)
        x[0, 0], x[1, 2], x[2], x[3] = 1.0, 0.0, -0.13, 1.21  # no bounds check in tcnn's grid: wraps / hashes
        x = x.to(DEV)
        gout = (torch.rand(B, cfg["n_levels"] * F, generator=gen) - 0.5).to(DEV)
        gout[7:11] = 0.0
        ref = torch.zeros_like(enc.params)
        _lib.check(L.snerf_hashgrid_encode_bwd(C.byref(enc.desc), None, ops._ptr(x), C.c_int64(B), ops._ptr(gout), ops._ptr(ref), None, ops._stream()))
        tb = TiledHashTableBackward(enc, B, tile_rows_log2=sh, first_tiled_level=lc)
        assert tb.plan.n_tiles == tb.plan.tile_start[cfg["n_levels"]] and (sh == 0 or tb.plan.tile_rows_log2 == sh) and tb.plan.first_tiled_level == lc
        got = torch.zeros_like(ref)
        tb.bin(x, gout)
        tb.coarse_levels(x, gout, got)
        tb.scatter(x, gout, got)
        torch.cuda.synchronize()
        assert 0 < int(tb.tile_base[-1]) <= tb.plan.record_capacity
        scale = float(ref.abs().max())
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6 * scale)
        assert bool(((got != 0) == (ref != 0)).all())
        tb.coarse_levels(x, gout, got)
        tb.scatter(x, gout, got)
        torch.testing.assert_close(got, 2 * ref, rtol=1e-5, atol=4e-6 * scale)
    # fused Adam
    p_ref, p_new = enc.params.detach().clone(), enc.params.detach().clone()
    z = lambda: torch.zeros_like(p_ref)
    m_ref, v_ref, m_new, v_new, g_ref, g_new = z(), z(), z(), z(), z(), z()
    B = 3000
    tb = TiledHashTableBackward(enc, B, tile_rows_log2=sh, first_tiled_level=lc)
    for step in range(1, 4):
        x = torch.rand(B, 3, generator=gen).to(DEV)
        gout = ((torch.rand(B, cfg["n_levels"] * F, generator=gen) - 0.5) * 1e-3).to(DEV)
        _lib.check(L.snerf_hashgrid_encode_bwd(C.byref(enc.desc), None, ops._ptr(x), C.c_int64(B), ops._ptr(gout), ops._ptr(g_ref), None, ops._stream()))
        ops.adam_step(p_ref, g_ref, m_ref, v_ref, step, 1e-2, eps=1e-6, zero_grad=True)
        tb.bin(x, gout)
        tb.coarse_levels(x, gout, g_new)
        tb.scatter_adam(x, gout, g_new if lc > 0 else None, p_new, m_new, v_new, 1e-2, step, 1e-6)
        torch.cuda.synchronize()
        assert float(g_new.abs().max()) == 0.0  # the coarse levels' gradient was read AND cleared
        torch.testing.assert_close(m_new, m_ref, rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(v_new, v_ref, rtol=2e-4, atol=1e-15)
        assert float(((p_new - p_ref).abs() > 1e-5).float().mean()) < 1e-4
    assert float((p_new - enc.params.detach()).abs().max()) > 1e-3

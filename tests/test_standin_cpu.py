"""The stock-PyTorch stand-in trainer (oracle/torch_standin.StandinTrainer; baseline code for PSNR runs) on the CPU: its step equals
the oracle's forward / loss dict / adam_step applied by hand, proposal updates follow the sampler's schedule, eval forward runs."""
import torch

from oracle import kplanes_oracle as KO
from oracle import torch_standin as TS

TINY = dict(base_res=(8, 8, 8, 4), multiscale=(1, 2), prop_res=((8, 8, 8, 4), (16, 16, 16, 4)))


def _rays(R, gen):
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.9
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    return {"origins": o, "directions": d, "times": torch.rand(R, 1, generator=gen)}


def test_standin_step_is_the_oracle_step_with_two_adams():
    R, S = 32, (16, 8, 4)
    tr = TS.StandinTrainer("cpu", R, seed=3, model=TINY, samples=S)
    P = KO.make_kplanes_params(seed=3, **TINY)
    leaves = KO.all_param_tensors(P)
    for x in leaves:
        x.requires_grad_(True)
    m = [torch.zeros_like(x) for x in leaves]
    v = [torch.zeros_like(x) for x in leaves]
    gen = torch.Generator().manual_seed(3)  # the trainer's own draw stream, replayed
    g2 = torch.Generator().manual_seed(11)
    for step in range(3):
        rays, target = _rays(R, g2), torch.rand(R, 3, generator=g2)
        rgb = tr.train_step(rays, target)
        rnd = lambda *s: torch.rand(*s, generator=gen)
        rng = {"t_rand": rnd(R, S[0] + 1), "u": [rnd(R, S[1] + 1), rnd(R, S[2] + 1)], "bg": rnd(R, 3)}
        out = KO.kplanes_forward(P, rays, rng, S[:2], S[2], anneal=KO.anneal_value(step))
        torch.testing.assert_close(rgb, out["rgb"].detach())
        loss = sum(KO.kplanes_loss_dict(P, out, target).values())
        grads = torch.autograd.grad(loss, leaves)
        with torch.no_grad():
            for x, g, mm, vv in zip(leaves, grads, m, v):
                KO.adam_step(x, g, mm, vv, step + 1, 1e-2 * KO.cosine_lr_factor(step))
    mine = torch.cat([x.detach().reshape(-1) for x in [t for sc in P["prop_grids"] for t in sc] + [w for lv in P["prop_sigma"] for w in lv]
                      + [t for sc in P["field_grids"] for t in sc] + list(P["field_sigma"]) + list(P["field_color"])])
    torch.testing.assert_close(tr.params, mine, rtol=1e-4, atol=1e-6)
    assert tr.step == 3 and tr.skipped_steps() == {"proposal_networks": 0, "fields": 0}
    img = tr.forward(_rays(R, g2), None, 1.0, training=False)
    assert img.shape == (R, 3) and bool(torch.isfinite(img).all()) and float(img.min()) >= 0 and float(img.max()) <= 1


def test_standin_skips_an_optimiser_whose_gradient_is_not_finite():
    tr = TS.StandinTrainer("cpu", 8, seed=0, model=TINY, samples=(8, 4, 4))
    g = torch.Generator().manual_seed(0)
    before = tr.params.clone()
    tr.train_step(_rays(8, g), torch.full((8, 3), float("nan")))
    assert tr.skipped_steps()["fields"] == 1
    n_prop = sum(x.numel() for x in tr.groups["proposal_networks"])
    torch.testing.assert_close(tr.params[n_prop:], before[n_prop:], rtol=0, atol=0)


def test_channel_last_row_gather_equals_grid_sample():
    """KO._bilinear_plane_rows (planes stored [H,W,C], four corner ROWS via index_select: the layout that lets the stand-in finish 30 000 steps inside one
    GPU call) against F.grid_sample as the reference calls it (NS/utils/interpolation.py:5-33): values and plane gradients, points on and beyond the
    borders included (padding 'border', align_corners=True)."""
    gen = torch.Generator().manual_seed(0)
    for C, H, W in ((32, 9, 13), (8, 4, 33)):
        plane = torch.rand(1, C, H, W, generator=gen) - 0.3
        pts = torch.rand(500, 2, generator=gen) * 2.4 - 1.2
        pts[:4] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [1.0, -1.0], [0.0, 1.0]])
        go = torch.rand(500, C, generator=gen) - 0.5
        a = plane.clone().requires_grad_(True)
        ref = torch.nn.functional.grid_sample(a, pts.view(1, -1, 1, 2), align_corners=True, mode="bilinear", padding_mode="border")[0, :, :, 0].t()
        ref.backward(go)
        leaf = plane[0].permute(1, 2, 0).contiguous().requires_grad_(True)  # [H,W,C] storage
        view = leaf.permute(2, 0, 1).unsqueeze(0)
        assert view.shape == plane.shape and view.stride(1) == 1
        prev, KO.USE_GRID_SAMPLE = KO.USE_GRID_SAMPLE, False
        try:
            got = KO.bilinear_plane(view, pts)
        finally:
            KO.USE_GRID_SAMPLE = prev
        got.backward(go)
        torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(leaf.grad.permute(2, 0, 1).unsqueeze(0), a.grad, rtol=1e-5, atol=1e-6)


def test_channel_last_standin_trains_like_the_grid_sample_standin():
    """StandinTrainer(plane_layout="hwc") against the default layout with F.grid_sample: same seed, same batches, three steps -- same rendered colours,
    same loss terms, same parameters (plane values compared in the reference's [1,C,H,W] shape)."""
    R, S = 32, (16, 8, 4)
    prev, KO.USE_GRID_SAMPLE = KO.USE_GRID_SAMPLE, True
    try:
        a = TS.StandinTrainer("cpu", R, seed=5, model=TINY, samples=S)
        g2 = torch.Generator().manual_seed(21)
        batches = [(_rays(R, g2), torch.rand(R, 3, generator=g2)) for _ in range(3)]
        rgb_a = [a.train_step(r, t) for r, t in batches]
        ld_a = {k: float(v) for k, v in a.loss_dict().items()}
    finally:
        KO.USE_GRID_SAMPLE = prev
    KO.USE_GRID_SAMPLE = False
    try:
        b = TS.StandinTrainer("cpu", R, seed=5, model=TINY, samples=S, plane_layout="hwc")
        rgb_b = [b.train_step(r, t) for r, t in batches]
        ld_b = {k: float(v) for k, v in b.loss_dict().items()}
        b._rebuild()
    finally:
        KO.USE_GRID_SAMPLE = prev
    for x, y in zip(rgb_a, rgb_b):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-6)
    for k in ld_a:
        assert abs(ld_a[k] - ld_b[k]) <= 1e-5 * abs(ld_a[k]) + 1e-9, (k, ld_a[k], ld_b[k])
    for sa, sb in zip(a.P["field_grids"], b.P["field_grids"]):
        for pa, pb in zip(sa, sb):
            torch.testing.assert_close(pb.detach(), pa.detach(), rtol=0, atol=2e-5)
    for la, lb in zip(a.P["prop_grids"], b.P["prop_grids"]):
        for pa, pb in zip(la, lb):
            torch.testing.assert_close(pb.detach(), pa.detach(), rtol=0, atol=2e-5)
    # eval render through the channel-last planes
    gen = torch.Generator().manual_seed(9)
    KO.USE_GRID_SAMPLE = False
    try:
        out = b.forward(_rays(R, gen), None, 1.0, training=False)
    finally:
        KO.USE_GRID_SAMPLE = prev
    assert out.shape == (R, 3) and bool(torch.isfinite(out).all())


def test_nerfplayer_standin_grid_equals_the_tgrid_oracle():
    """oracle/nerfplayer_standin.TorchTemporalGrid (the eight corners vectorised, per-level leaf tables: what lets the config-4 stand-in train on the device)
    against oracle/tgrid_oracle.encode + temporal_index -- the restatement pinned by the reference's known-answer test and HashEncoding -- value for value,
    on dense AND hashed levels, with border / out-of-range points and t = 0 / 1; its TV term against temporal_tv_loss; gradients flow to every level."""
    from oracle import nerfplayer_standin as NS_
    from oracle import tgrid_oracle as TO

    gen = torch.Generator().manual_seed(4)
    for C, T, L, log2T, H, scale in ((2, 16, 6, 10, 4, 1.6), (4, 9, 3, 8, 3, 2.0), (1, 5, 2, 12, 5, 1.5)):
        offs = TO.level_offsets(L, H, scale, log2T)
        emb = torch.rand(offs[-1], C + T, generator=gen) - 0.5
        g = NS_.TorchTemporalGrid(emb, offs, float(torch.log2(torch.tensor(scale))), H, C)
        assert any(m[2] for m in g.meta) or C == 1  # hashed levels present in the first two cases
        B = 300
        x = torch.rand(B, 3, generator=gen)
        x[0, 0], x[1, 2], x[2] = 1.0, 0.0, torch.tensor([-0.1, 0.5, 0.5])
        t = torch.rand(B, generator=gen)
        t[3], t[4] = 1.0, 0.0
        table = TO.channel_table(T, C)
        want = TO.encode(x, TO.temporal_index(t, table), emb, offs, float(torch.log2(torch.tensor(scale))), H, 0, C)
        got = g.encode(x, t)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
        assert float(got[2].abs().max()) == 0.0  # the out-of-range point
        row = 3 % len(g.index_ab)
        torch.testing.assert_close(g.tv_loss(row), TO.temporal_tv_loss(emb, table["index_ab"][row].tolist()))
        (got.square().sum() + g.tv_loss(row)).backward()
        assert all(lv.grad is not None and float(lv.grad.abs().max()) > 0 for lv in g.levels)


def test_nerfplayer_standin_reproduces_the_reference_models_own_step():
    """oracle/nerfplayer_standin.NerfplayerStandinTrainer -- the reference-ALGORITHM arm of config 4's PSNR -- against golden G12 = the
    REFERENCE'S OWN NerfplayerNerfactoModel run on the CPU (oracle/gen_golden_nerfplayer.py imports it): built from G12's parameters and fed G12's rays, uniform
    draws, anneal value and TV row, the stand-in must give the reference's sample bins and weights at all three levels, its rendered colours, every term of its
    loss dict and the gradient of every parameter tensor (sum, sum of magnitudes, 64 probes)."""
    from types import SimpleNamespace as NS

    import numpy as np

    from oracle import tgrid_oracle as TO
    from oracle.nerfplayer_standin import NerfplayerStandinTrainer
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModelConfig
    from tests.conftest import load_golden

    g = load_golden("g12_nerfplayer")
    # the golden's model configuration (oracle/gen_golden_nerfplayer.py: CFG; restated here so that the test never imports the generator, which imports the reference)
    CFG = dict(disable_scene_contraction=True, num_levels=4, features_per_level=2, log2_hashmap_size=10, temporal_dim=8,
               proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 32},
                                       {"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 64}],
               num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8)
    cfg = NerfplayerNerfactoModelConfig(**CFG)

    def enc(prefix, num_levels, log2, max_res=None):
        scale = float(np.exp2(np.log2(2048 / 16) / (num_levels - 1))) if max_res is None else float(np.exp((np.log(max_res) - np.log(16)) / (num_levels - 1)))
        offs = TO.level_offsets(num_levels, 16, scale, log2)
        emb = g[f"param_{prefix}.embeddings"]
        assert offs[-1] == emb.shape[0]
        return NS(embeddings=emb, offsets=torch.tensor(offs), per_level_scale=scale, base_resolution=16, level_dim=2, gridtype_id=0)

    lin = lambda prefix, n: NS(linear_weights=lambda: [g[f"param_{prefix}.layers.{i}.weight"] for i in range(n)])
    R = int(g["R"])
    src = NS(cfg=cfg, R=R, S=(32, 16, 8), aabb=[[-1.0] * 3, [1.0] * 3],
             prop_enc=[enc(f"proposal_networks.{i}.encoding", 3, 9, max_res=(32, 64)[i]) for i in range(2)], enc=enc("field.mlp_base", 4, 10),
             prop_mlp=[lin(f"proposal_networks.{i}.linear", 2) for i in range(2)], decode=lin("field.mlp_base_decode", 2), head=lin("field.mlp_head", 3),
             appearance=NS(weight=g["param_field.embedding_appearance.embedding.weight"]))
    tr = NerfplayerStandinTrainer(src, "cpu")
    tr.tv_rows = [int(g["tv_row"])] * 3
    rays = {"origins": g["origins"], "directions": g["directions"], "times": g["times"]}
    rng = {"t_rand": g["t_rand"], "u": [g["u0"], g["u1"]], "bg": g["bg"]}
    rgb, ld = tr.loss_and_backward(rays, g["cams"], g["target"], rng, float(g["anneal"]), True)
    for i in range(3):
        torch.testing.assert_close(tr._last_ebins[i], g[f"ebins_{i}"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(tr._last["weights"][i], g[f"weights_{i}"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rgb.detach(), g["rgb"], rtol=1e-4, atol=1e-6)
    assert set(ld) == {"rgb_loss", "interlevel_loss", "distortion_loss", "temporal_tv_loss"}
    for k, v in ld.items():
        torch.testing.assert_close(v.detach(), torch.as_tensor(g["loss_" + k]), rtol=1e-4, atol=1e-9, msg=lambda m: f"{k}: {m}")
    torch.testing.assert_close(sum(ld.values()).detach(), torch.as_tensor(g["loss_total"]), rtol=1e-4, atol=1e-9)
    table_grad = lambda grid: torch.cat([lv.grad for lv in grid.levels], 0)
    mine = {"field.embedding_appearance.embedding.weight": tr.appearance.grad, "field.mlp_base.embeddings": table_grad(tr.grid)}
    for i, w in enumerate(tr.decode_w):
        mine[f"field.mlp_base_decode.layers.{i}.weight"] = w.grad
    for i, w in enumerate(tr.head_w):
        mine[f"field.mlp_head.layers.{i}.weight"] = w.grad
    for k in range(2):
        mine[f"proposal_networks.{k}.encoding.embeddings"] = table_grad(tr.prop_grid[k])
        for i, w in enumerate(tr.prop_w[k]):
            mine[f"proposal_networks.{k}.linear.layers.{i}.weight"] = w.grad
    names = [str(n) for n in g["param_names"]]
    assert set(names) == set(mine)
    for name in names:
        got = mine[name]
        gabs = float(g["gabs_" + name])
        assert abs(float(got.double().sum()) - float(g["gsum_" + name])) <= 1e-4 * gabs + 1e-10, name
        assert abs(float(got.double().abs().sum()) - gabs) <= 1e-4 * gabs + 1e-10, name
        probe = got.flatten()[:: max(1, got.numel() // 64)][:64]
        torch.testing.assert_close(probe, g["gprobe_" + name], rtol=1e-3, atol=1e-7 + 1e-4 * float(g["gprobe_" + name].abs().max()), msg=lambda m: f"{name}: {m}")

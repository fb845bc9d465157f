"""GPU parity: renderers, ray-level losses, plane regularisers, Adam, ray generation vs golden vectors / oracle."""
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().cpu(), torch.as_tensor(b).detach().cpu(), rtol=rtol, atol=atol)


def test_render_golden():
    from soccernerfs_amd import ops

    g = load_golden("g7_render")
    w, rgb, eb = g["weights"].to(DEV), g["rgb"].to(DEV), g["ebins"].to(DEV)
    o = ops.render(w, rgb, eb, g["bg"].to(DEV), training=True)
    close(o["rgb"], g["rgb_random_train"])
    close(o["accumulation"], g["accumulation"][:, 0])
    assert torch.equal(o["median_index"].cpu(), g["median_index"][:, 0])  # bit-exact
    close(o["depth_median"], g["depth_median"][:, 0], rtol=0, atol=1e-7)
    close(o["median_rgb"], g["median_rgb"][:, 0], rtol=0, atol=0)
    close(ops.render(w, rgb, eb, "black", True)["rgb"], g["rgb_black_train"])
    close(ops.render(w, rgb, eb, "white", False)["rgb"], g["rgb_white_eval"])
    close(ops.render(w, rgb, eb, "last_sample", True)["rgb"], g["rgb_last_sample_train"])
    close(ops.render(w, rgb, eb, "last_sample", False)["rgb"], g["rgb_last_sample_eval"])
    steps = (g["ebins"][:, :-1] + g["ebins"][:, 1:]) / 2
    close(torch.clip(o["depth_expected"].cpu(), steps.min(), steps.max()), g["depth_expected"][:, 0])


def test_render_backward_vs_oracle():
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(2)
    R, S = 33, 64
    w = (torch.rand(R, S, generator=gen) * 0.03).requires_grad_(True)
    rgb = torch.rand(R, S, 3, generator=gen).requires_grad_(True)
    eb = torch.cumsum(torch.rand(R, S + 1, generator=gen) * 0.05, -1)
    bg = torch.rand(R, 3, generator=gen)
    go, ga = torch.rand(R, 3, generator=gen) - 0.5, torch.rand(R, generator=gen)
    ref = KO.render_rgb(rgb, w, bg, True)
    ((ref * go).sum() + (KO.render_accumulation(w)[:, 0] * ga).sum()).backward()
    w2, rgb2 = w.detach().to(DEV).requires_grad_(True), rgb.detach().to(DEV).requires_grad_(True)
    o = ops.render(w2, rgb2, eb.to(DEV), bg.to(DEV), True)
    ((o["rgb"] * go.to(DEV)).sum() + (o["accumulation"] * ga.to(DEV)).sum()).backward()
    close(w2.grad, w.grad, rtol=1e-4, atol=1e-6)
    close(rgb2.grad, rgb.grad, rtol=1e-4, atol=1e-7)


def test_losses_golden():
    from soccernerfs_amd import ops

    g = load_golden("g8_losses")
    ws = [g[f"w_{i}"].to(DEV).requires_grad_(True) for i in range(3)]
    sb = [g[f"sbins_{i}"].to(DEV) for i in range(3)]
    li = ops.interlevel_loss(ws, sb)
    close(li, g["interlevel"], rtol=1e-5, atol=1e-8)
    li.backward()
    close(ws[0].grad, g["grad_w0"], rtol=1e-4, atol=1e-8)
    close(ws[1].grad, g["grad_w1"], rtol=1e-4, atol=1e-8)
    assert ws[2].grad is None
    ld = ops.distortion_loss(ws[2], sb[2])
    close(ld, g["distortion"], rtol=1e-5, atol=1e-8)
    ld.backward()
    close(ws[2].grad, g["grad_w2_distortion"], rtol=1e-4, atol=1e-8)


def test_plane_regularizers_golden():
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    g = load_golden("g8_losses")
    planes = [g[f"reg_plane_{p}"] for p in range(6)]
    ps = PlaneSet(8, [[12, 10, 9, 7]], concat=False)
    ps.load_reference([planes])
    ps = ps.to(DEV)
    vals = ops.plane_regularizers(ps)
    for k, nm in enumerate(("space_tv", "time_smooth", "sparse_transients")):
        close(vals[k], g[nm], rtol=1e-5, atol=1e-8)
    for k, nm in enumerate(("space_tv", "time_smooth", "sparse_transients")):
        ps.planes.grad = None
        ops.plane_regularizers(ps)[k].backward()
        got = ps.to_reference(ps.planes.grad.cpu())[0]
        for p in range(6):
            close(got[p], g[f"{nm}_grad_{p}"], rtol=1e-4, atol=1e-9)


def test_plane_regularizers_multiscale_vs_oracle():
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    gen = torch.Generator().manual_seed(4)
    ps = PlaneSet(32, [[9, 7, 6, 5], [18, 14, 12, 5], [36, 28, 24, 5]], concat=True, generator=gen)
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, generator=gen) * 2)
    grids = [[t.clone().requires_grad_(True) for t in sc] for sc in ps.to_reference()]
    ref = 0.3 * KO.space_tv_loss(grids) + 0.7 * KO.time_smoothness_loss(grids) + 1.1 * KO.sparse_transients_loss(grids)
    ref.backward()
    ps = ps.to(DEV)
    v = ops.plane_regularizers(ps)
    (0.3 * v[0] + 0.7 * v[1] + 1.1 * v[2]).backward()
    close(0.3 * v[0] + 0.7 * v[1] + 1.1 * v[2], ref, rtol=1e-5, atol=1e-8)
    got = ps.to_reference(ps.planes.grad.cpu())
    for s in range(3):
        for p in range(6):
            close(got[s][p], grids[s][p].grad, rtol=1e-4, atol=1e-9)


def test_adam_matches_torch():
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(8)
    n = 100003  # not a multiple of 4
    p0 = torch.rand(n, generator=gen)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-12)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        grad = torch.rand(n, generator=gen) - 0.5
        ref.grad = grad.clone()
        opt.step()
        g = grad.to(DEV)
        ops.adam_step(p, g, m, v, step, 1e-2, zero_grad=True)
        assert float(g.abs().max()) == 0.0
        close(p, ref, rtol=1e-5, atol=1e-6)


def test_raygen_and_collider_golden():
    from soccernerfs_amd import ops

    g = load_golden("g1_raygen")
    t = lambda k: g[k].to(DEV)
    out = ops.generate_rays(t("indices"), t("fx"), t("fy"), t("cx"), t("cy"), t("c2w"), t("times"))
    close(out["origins"], g["origins"], rtol=0, atol=0)
    close(out["directions"], g["directions"], rtol=1e-6, atol=2e-7)
    close(out["pixel_area"], g["pixel_area"], rtol=2e-4, atol=1e-9)
    close(out["directions_norm"], g["directions_norm"], rtol=1e-6, atol=1e-7)
    close(out["times"], g["ray_times"], rtol=0, atol=0)
    c = load_golden("g2_collider")
    for mode in ("train", "eval"):
        n, f = ops.aabb_collide(c["origins"].to(DEV), c["directions"].to(DEV), c["aabb"], float(c["near_plane"]), mode == "train")
        close(n[:, 0], c[f"nears_{mode}"], rtol=1e-5, atol=1e-6)
        close(f[:, 0], c[f"fars_{mode}"], rtol=1e-5, atol=1e-6)
    # fused variant
    out = ops.generate_rays(t("indices"), t("fx"), t("fy"), t("cx"), t("cy"), t("c2w"), t("times"), aabb=c["aabb"], near_plane=0.05)
    n, f = ops.aabb_collide(out["origins"], out["directions"], c["aabb"], 0.05, True)
    close(out["nears"], n, rtol=0, atol=0)
    close(out["fars"], f, rtol=0, atol=0)


def test_render_mse_bwd_equals_render_bwd_of_mse_gradient():
    """snerf_render_mse_bwd == snerf_render_bwd fed with 2c/(3R) (rgb_out - target), plus the per-ray squared error."""
    import ctypes as C
    from soccernerfs_amd import _lib, ops

    dev = "cuda:0"
    gen = torch.Generator().manual_seed(12)
    R, S, coef = 77, 48, 0.7
    w = torch.rand(R, S, generator=gen).to(dev)
    rgb = torch.rand(R, S, 3, generator=gen).to(dev)
    bg = torch.rand(R, 3, generator=gen).to(dev)
    out = torch.rand(R, 3, generator=gen).to(dev)
    target = torch.rand(R, 3, generator=gen).to(dev)
    go = (out - target) * (2 * coef / (3 * R))
    gw_ref, grgb_ref = torch.empty_like(w), torch.empty_like(rgb)
    L, p = _lib.lib(), ops._ptr
    _lib.check(L.snerf_render_bwd(p(w), p(rgb), p(bg), 0, p(go), None, R, S, p(gw_ref), p(grgb_ref), 0, ops._stream()))
    gw, grgb, sq = torch.empty_like(w), torch.empty_like(rgb), torch.empty(R, device=dev)
    _lib.check(L.snerf_render_mse_bwd(p(w), p(rgb), p(bg), 0, p(out), p(target), 2 * coef / (3 * R), R, S, p(gw), p(grgb), p(sq), ops._stream()))
    torch.testing.assert_close(gw, gw_ref, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(grgb, grgb_ref, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(sq, ((out - target) ** 2).sum(-1), rtol=1e-6, atol=0)
    torch.testing.assert_close(sq.sum() / (3 * R) * coef, torch.nn.functional.mse_loss(out, target) * coef, rtol=1e-5, atol=0)


def test_depth_loss_golden():
    """snerf_depth_loss (DS-NeRF depth supervision, losses.py:213-235,261-311) vs the reference's own values and gradients (G8b)."""
    from soccernerfs_amd import ops

    g = load_golden("g8b_depth")
    d = lambda k: g[k].to(DEV).contiguous()
    for tag, eucl in (("eucl_s001", True), ("eucl_s02", True), ("z_s02", False)):
        w = d("weights").requires_grad_(True)
        val = ops.ds_nerf_depth_loss(w, d("bins"), d("termination_depth"), float(g["sigma_" + tag]), None if eucl else d("directions_norm"))
        (val * 2.0).backward()
        torch.testing.assert_close(val.detach().cpu(), torch.as_tensor(g["loss_" + tag]), rtol=2e-5, atol=1e-8)
        # 1 / (w + 1e-7) at w = 0 or 1e-9 amplifies the last ulp of expf: relative tolerance only
        torch.testing.assert_close(w.grad.cpu() / 2.0, g["grad_" + tag], rtol=2e-5, atol=1e-9)


def test_urf_depth_loss_golden():
    """snerf_urf_depth_loss (Urban Radiance Fields lidar losses, losses.py:238-274) through the reference-shaped losses.depth_loss vs the
    reference's own values and gradients w.r.t. weights and predicted depth (G8b)."""
    from soccernerfs_amd import losses
    from soccernerfs_amd.rays import Frustums, RaySamples

    g = load_golden("g8b_depth")
    d = lambda k: g[k].to(DEV).contiguous()
    bins = d("bins")
    R, S = g["weights"].shape
    o = torch.zeros(R, S, 3, device=DEV)
    rs = RaySamples(frustums=Frustums(origins=o, directions=torch.ones_like(o), starts=bins[:, :-1, None], ends=bins[:, 1:, None],
                                      pixel_area=torch.ones(R, S, 1, device=DEV)))
    for tag, eucl in (("urf_eucl_s02", True), ("urf_z_s05", False), ("urf_eucl_s001", True)):
        w = d("weights")[..., None].clone().requires_grad_(True)
        pd = d("predicted_depth")[:, None].clone().requires_grad_(True)
        val = losses.depth_loss(w, rs, d("termination_depth")[:, None], pd, float(g["sigma_" + tag]), d("directions_norm")[:, None], eucl,
                                losses.DepthLossType.URF)
        (val * 3.0).backward()
        torch.testing.assert_close(val.detach().cpu(), torch.as_tensor(g["loss_" + tag]), rtol=2e-5, atol=1e-8)
        torch.testing.assert_close(w.grad[..., 0].cpu() / 3.0, g["grad_" + tag], rtol=2e-5, atol=1e-8)
        torch.testing.assert_close(pd.grad[:, 0].cpu() / 3.0, g["gpred_" + tag], rtol=2e-5, atol=1e-8)


def test_trainer_depth_supervision_matches_oracle():
    """Fused trainer with termination depths: loss value and every gradient segment equal the oracle with the depth term on all three levels."""
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    E = dict(base_res=(16, 16, 16, 4), multiscale=(1, 2), feat_dim=32, prop_res=((24, 24, 24, 4), (32, 32, 32, 4)), prop_feat=8,
             sigma_hidden=128, color_hidden=64, aabb_scale=1.5, seed=9)
    P = KO.make_kplanes_params(**E)
    leaves = KO.all_param_tensors(P)
    for x in leaves:
        x.requires_grad_(True)
    R = 48
    cfg = KPlanesTrainConfig(aabb_scale=1.5, spacetime_resolution=E["base_res"], multiscale_res=E["multiscale"], feature_dim=32,
                             proposal_resolutions=E["prop_res"], proposal_feature_dim=8, num_proposal_samples_per_ray=(64, 32),
                             num_nerf_samples_per_ray=16, depth_sigma=0.05, mlp_operands="fp32")
    tr = KPlanesTrainer(cfg, R, DEV)
    tr.load_oracle_params(P)
    gen = torch.Generator().manual_seed(12)
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 1.2
    dd = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    times, target = torch.rand(R, 1, generator=gen), torch.rand(R, 3, generator=gen)
    depth = torch.rand(R, generator=gen) * 2 + 0.3
    depth[::7] = 0.0  # rays without a depth value
    rng = {"t_rand": torch.rand(R, 65, generator=gen), "u": [torch.rand(R, 33, generator=gen), torch.rand(R, 17, generator=gen)],
           "bg": torch.rand(R, 3, generator=gen)}
    out = KO.kplanes_forward(P, {"origins": o, "directions": dd, "times": times}, rng, (64, 32), 16, anneal=0.3)
    ld = KO.kplanes_loss_dict(P, out, target)
    ld["depth_loss"] = 0.05 * sum(KO.depth_loss(w, e, depth, 0.05) for w, e in zip(out["weights_list"], out["eucl_list"])) / 3
    sum(ld.values()).backward()
    dv = lambda z: z.to(DEV).contiguous()
    rngd = {"t_rand": dv(rng["t_rand"]), "u": [dv(rng["u"][0]), dv(rng["u"][1])], "bg": dv(rng["bg"])}
    tr.forward({"origins": dv(o), "directions": dv(dd), "times": dv(times)}, rngd, 0.3, training=True)
    tr.backward(dv(target), rngd, proposal_grads=True, depth=dv(depth))
    got = tr.loss_dict()
    torch.testing.assert_close(got["depth_loss"].cpu(), ld["depth_loss"].detach(), rtol=1e-3, atol=1e-8)
    torch.testing.assert_close(sum(got.values()).cpu(), sum(ld.values()).detach(), rtol=2e-3, atol=1e-7)
    torch.cuda.synchronize()
    gref = tr.field_planes.to_reference(tr.gviews["field.planes"])
    for s in range(2):
        for p in range(6):
            want = P["field_grids"][s][p].grad
            torch.testing.assert_close(gref[s][p].cpu(), want, rtol=5e-3, atol=2e-3 * float(want.abs().max()) + 1e-9)
    for lvl in range(2):
        gp = tr.prop_planes[lvl].to_reference(tr.gviews[f"prop{lvl}.planes"])[0]
        for p in range(6):
            want = P["prop_grids"][lvl][p].grad
            torch.testing.assert_close(gp[p].cpu(), want, rtol=5e-3, atol=2e-3 * float(want.abs().max()) + 1e-9)
    # without depths the term is absent and the gradient differs
    assert "depth_loss" in got


@pytest.mark.parametrize("R,S", [(37, 64), (5, 48), (130, 320), (4, 1)])
def test_ray_train_kernel_equals_the_five_kernels(R, S):
    """snerf_ray_train_fwd_bwd (the fused trainer's one launch for the nerf level's per-ray work) against snerf_weights_fwd + snerf_render_fwd +
    snerf_render_mse_bwd + snerf_distortion(accumulate) + snerf_weights_bwd run one after the other: every output bit for bit, including rays
    whose densities overflow (non-finite weights -> the skip-step flag) and zero-width bins."""
    import ctypes as C

    from soccernerfs_amd import _lib

    L = _lib.lib()
    gen = torch.Generator().manual_seed(R * 1000 + S)
    g = lambda *sh: torch.rand(*sh, generator=gen)
    dens = (g(R, S) ** 4 * 60).to(DEV)
    if R > 3:
        dens[1, S // 2] = float("inf")   # exp overflow in the field
        dens[2] = 0.0
        dens[3, : max(1, S // 3)] = 3e38
    eb = torch.cumsum(g(R, S + 1) * 0.05 + 1e-3, -1)
    if R > 3 and S > 4:
        eb[1, S // 2 + 1] = eb[1, S // 2]  # a zero-width bin under the infinite density: 0 * inf
    eb = eb.to(DEV)
    sb = torch.sort(g(R, S + 1), -1).values.to(DEV)
    rgb, bg, target = g(R, S, 3).to(DEV), g(R, 3).to(DEV), g(R, 3).to(DEV)
    go_scale, dist_scale = 2.0 / (3 * R), 1e-3 / R
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    new = lambda *sh: torch.full(sh, -7.0, device=DEV)
    # ---- the five kernels ----
    w, rgb_out, acc, depth = new(R, S), new(R, 3), new(R), new(R)
    gw, grgb, sqerr, dist, gdens = new(R, S), new(R, S, 3), new(R), new(R), new(R, S)
    flag = torch.zeros(4, dtype=torch.int32, device=DEV)
    _lib.check(L.snerf_weights_fwd(p(dens), p(eb), R, S, p(w), st))
    a = _lib.RenderArgs()
    a.weights, a.rgb, a.ebins, a.bg, a.R, a.S, a.bg_mode, a.training = w.data_ptr(), rgb.data_ptr(), eb.data_ptr(), bg.data_ptr(), R, S, 0, 1
    a.rgb_out, a.acc_out, a.depth_median = rgb_out.data_ptr(), acc.data_ptr(), depth.data_ptr()
    _lib.check(L.snerf_render_fwd(C.byref(a), st))
    _lib.check(L.snerf_render_mse_bwd(p(w), p(rgb), p(bg), 0, p(rgb_out), p(target), go_scale, R, S, p(gw), p(grgb), p(sqerr), st))
    _lib.check(L.snerf_distortion(p(w), p(sb), R, S, dist_scale, p(dist), p(gw), 1, st))
    _lib.check(L.snerf_weights_bwd(p(dens), p(eb), p(gw), R, S, p(gdens), 0, p(flag), st))
    # ---- one launch ----
    w2, rgb_out2, acc2, depth2 = new(R, S), new(R, 3), new(R), new(R)
    gw2, grgb2, sqerr2, dist2, gdens2 = new(R, S), new(R, S, 3), new(R), new(R), new(R, S)
    flag2 = torch.zeros(4, dtype=torch.int32, device=DEV)
    ra = _lib.RayTrainArgs()
    ra.density, ra.ebins, ra.sbins, ra.rgb, ra.bg, ra.target = dens.data_ptr(), eb.data_ptr(), sb.data_ptr(), rgb.data_ptr(), bg.data_ptr(), target.data_ptr()
    ra.R, ra.S, ra.bg_mode, ra.go_scale, ra.dist_scale = R, S, 0, go_scale, dist_scale
    ra.weights, ra.rgb_out, ra.acc_out, ra.depth_median = w2.data_ptr(), rgb_out2.data_ptr(), acc2.data_ptr(), depth2.data_ptr()
    ra.sqerr_rays, ra.dist_rays, ra.g_rgb, ra.g_density = sqerr2.data_ptr(), dist2.data_ptr(), grgb2.data_ptr(), gdens2.data_ptr()
    ra.g_weights, ra.nonfinite_flag = gw2.data_ptr(), flag2.data_ptr()
    _lib.check(L.snerf_ray_train_fwd_bwd(C.byref(ra), st))
    eq = lambda x, y: torch.equal(torch.nan_to_num(x, nan=12345.0), torch.nan_to_num(y, nan=12345.0))
    for name, x, y in (("weights", w, w2), ("rgb_out", rgb_out, rgb_out2), ("acc", acc, acc2), ("depth", depth, depth2), ("g_weights", gw, gw2),
                       ("g_rgb", grgb, grgb2), ("sqerr", sqerr, sqerr2), ("dist", dist, dist2), ("g_density", gdens, gdens2)):
        assert eq(x, y), name
    assert int(flag[0]) == int(flag2[0]) and (int(flag[0]) == 1) == (R > 3 and S > 4)

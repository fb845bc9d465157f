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

"""GPU parity of the owner-computes scatter + optimiser kernel (csrc/kplanes_tile_adam.hip: pass B of the quotient scatter, plane regularisers
and Adam for the finest scale in one launch, gradient tile in LDS) against the two-kernel path it replaces
(snerf_kplanes_scatter_quotient_scales -> gradient plane in HBM -> snerf_adam_planes_step_range), the sample-major scatter + torch Adam, and --
through whole training steps -- the trainer with the kernel switched off.  Reference: autograd of interpolate_kplanes
(NS/fields/kplanes_field.py:77-126) + losses.py:356-452 + torch.optim.Adam (NS/configs/method_configs.py:546-557)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COEFS = (2e-4, 1e-3, 1e-4)


def _setup(base, ms, N, seed=0, spread=1.05, zero_texels=False):
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    gen = torch.Generator(device=DEV).manual_seed(seed)
    ps = PlaneSet(32, [[r * m for r in base[:3]] + [base[3]] for m in ms], concat=True, device=DEV)
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, device=DEV, generator=gen) * 1.3 - 0.3)  # some negative values, products of six stay in range
    pts = (torch.rand(N, 4, device=DEV, generator=gen) * 2 - 1) * spread  # a few samples outside the box: border clamp
    pts[: N // 8, :3] = pts[: N // 8, :3] * 0.05 + 0.3                    # a hot spot: hundreds of entries in a handful of cells
    pts[N // 8: N // 4, 3] = 0.2                                            # one time value: the time planes' entries share two rows
    if zero_texels:
        s = len(ms) - 1
        res = ps.resolutions[s]
        with torch.no_grad():  # exact zeros at the finest scale: all four texels of a cell, so every sample inside it sees v = 0 exactly
            ps.plane_view(s, 0)[1:3, 2:4, :] = 0.0   # XY plane: y texels 1-2, x texels 2-3 (one zero plane per sample: the fix-up's case)
            ps.plane_view(s, 2)[1:3, 4:6, :] = 0.0   # XT plane: t texels 1-2, x texels 4-5
        tex = lambda i, n: 2.0 * i / (n - 1) - 1.0
        pts[0, 0], pts[0, 1] = tex(2.5, res[0]), tex(1.5, res[1])
        pts[1, 0], pts[1, 3] = tex(4.5, res[0]), tex(1.5, res[3])
    gfeat = (torch.rand(N, ps.out_dim, device=DEV, generator=gen) - 0.5) * 1e-2
    feat = ops.interpolate_kplanes(pts.contiguous(), ps).detach()
    return ps, pts.contiguous(), gfeat, feat


def _two_kernel_path(ps, ss, co, gfeat, m0, v0, step, lr, dyn=None):
    """pass B of every scale (+ fix-up) into a gradient plane, then the plain sweep."""
    from soccernerfs_amd import ops

    g = torch.zeros_like(ps.planes)
    ns = len(ps.resolutions)
    ss.quotient_scatter_scales(ps.planes, co, gfeat, g, 0, ns)
    p_out, m, v = torch.zeros_like(ps.planes), m0.clone(), v0.clone()
    losses = torch.zeros(ops.REG_SLOTS, 16, device=DEV)
    ops.adam_planes_step(ps, ps.planes.detach(), p_out, g, m, v, COEFS, losses, step, lr, dyn=dyn)
    return p_out, m, v, losses, g


def _tile_path(ps, ss, co, gfeat, m0, v0, step, lr, tile_shape, dyn=None):
    from soccernerfs_amd import ops

    ns = len(ps.resolutions)
    fin = ns - 1
    lo = int(ps.desc().off[fin][0])
    g = torch.zeros_like(ps.planes)
    ss.quotient_scatter_scales(ps.planes, co, gfeat, g, 0, fin)  # pass B + fix-up of the coarser scales
    ss.quotient_fixup_scales(ps.planes, co, gfeat, g, fin, ns)   # exact-zero rows of the finest scale: the only thing its gradient plane ever holds
    p_out, m, v = torch.zeros_like(ps.planes), m0.clone(), v0.clone()
    losses = torch.zeros(ops.REG_SLOTS, 16, device=DEV)
    ss.scatter_adam_scale(fin, ps.planes.detach(), p_out, g, m, v, COEFS, losses, step, lr, dyn=dyn, tile_shape=tile_shape)
    assert float(g[lo:].abs().max()) == 0.0  # the finest scale's gradient plane is clean again
    if lo > 0:
        ops.adam_planes_step(ps, ps.planes.detach(), p_out, g, m, v, COEFS, losses, step, lr, shard_range=(0, lo), dyn=dyn)
    return p_out, m, v, losses, g


@pytest.mark.parametrize("tile_shape", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("base,ms,N", [((12, 10, 9, 6), (1, 2, 4), 20000), ((33, 17, 40, 7), (1,), 9000), ((16, 16, 16, 5), (1, 2), 257)])
def test_tile_kernel_equals_pass_b_plus_sweep(base, ms, N, tile_shape):
    from soccernerfs_amd import ops

    ps, pts, gfeat, feat = _setup(base, ms, N, seed=len(ms))
    co = ops.coords_from_points(pts)
    ss = ops.SortedScatter(ps, N, DEV, quotient=True)
    assert ss.scatter_adam_supported(len(ms) - 1) and (len(ms) == 1 or not ss.scatter_adam_supported(0))
    ss.sort(co)
    ss.quotient_prepare(gfeat, feat)
    gen = torch.Generator(device=DEV).manual_seed(5)
    m0 = (torch.rand(ps.numel, device=DEV, generator=gen) - 0.5) * 1e-3
    v0 = torch.rand(ps.numel, device=DEV, generator=gen) * 1e-6
    step, lr = 7, 3e-3
    pa, ma, va, la, ga = _two_kernel_path(ps, ss, co, gfeat, m0, v0, step, lr)
    pb, mb, vb, lb, _ = _tile_path(ps, ss, co, gfeat, m0, v0, step, lr, tile_shape)
    torch.cuda.synchronize()
    # moments: linear / quadratic in the gradient, whose sums differ only in their order
    torch.testing.assert_close(mb, ma, rtol=2e-4, atol=1e-9)
    torch.testing.assert_close(vb, va, rtol=4e-4, atol=1e-14)
    # parameters: the update is lr * m / (sqrt(v) + eps), a ratio -> compare the UPDATE relative to lr
    torch.testing.assert_close(pb - ps.planes.detach(), pa - ps.planes.detach(), rtol=1e-3, atol=lr * 1e-4)
    torch.testing.assert_close(lb[:, :3].sum(0), la[:, :3].sum(0), rtol=1e-4, atol=1e-9)  # regulariser values
    # and against an independent route: the sample-major scatter + torch's own Adam arithmetic (no regularisers: coefficients 0)
    desc = ps.desc()
    direct = torch.zeros_like(ps.planes)
    from soccernerfs_amd import _lib

    _lib.check(_lib.lib().snerf_kplanes_gather_bwd(C.byref(desc), ops._ptr(ps.planes), C.byref(co), C.c_int64(N), ops._ptr(gfeat), ops._ptr(direct), ops._stream()))
    fin = len(ms) - 1
    lo = int(desc.off[fin][0])
    p_out, m, v = torch.zeros_like(ps.planes), m0.clone(), v0.clone()
    g = torch.zeros_like(ps.planes)
    ss.scatter_adam_scale(fin, ps.planes.detach(), p_out, g, m, v, (0.0, 0.0, 0.0), None, step, lr, tile_shape=tile_shape)
    m_ref = 0.9 * m0[lo:] + 0.1 * direct[lo:]
    v_ref = 0.999 * v0[lo:] + 0.001 * direct[lo:] ** 2
    torch.testing.assert_close(m[lo:], m_ref, rtol=2e-4, atol=1e-9)
    p_ref = ps.planes.detach()[lo:] - (lr / (1 - 0.9 ** step)) * m_ref / (v_ref.sqrt() / (1 - 0.999 ** step) ** 0.5 + 1e-12)
    torch.testing.assert_close(p_out[lo:] - ps.planes.detach()[lo:], p_ref - ps.planes.detach()[lo:], rtol=1e-3, atol=lr * 1e-4)
    assert torch.equal(p_out[:lo], torch.zeros_like(p_out[:lo])) and torch.equal(m[:lo], m0[:lo])  # the other scales are not touched


def test_tile_kernel_with_exact_zero_features_and_skip_flag():
    """Exact-zero texels under samples: the fix-up's terms travel through the gradient plane and the tile kernel adds + clears them.  A raised
    skip flag (non-finite gradient in the parameter group, GradScaler semantics) leaves p, m, v untouched."""
    from soccernerfs_amd import ops

    ps, pts, gfeat, feat = _setup((12, 10, 9, 6), (1, 2, 4), 6000, seed=9, zero_texels=True)
    co = ops.coords_from_points(pts)
    ss = ops.SortedScatter(ps, pts.shape[0], DEV, quotient=True)
    ss.sort(co)
    ss.quotient_prepare(gfeat, feat)
    assert int(ss.fix_count.item()) > 0
    m0, v0 = torch.zeros(ps.numel, device=DEV), torch.zeros(ps.numel, device=DEV)
    pa, ma, va, _, _ = _two_kernel_path(ps, ss, co, gfeat, m0, v0, 1, 1e-2)
    pb, mb, vb, _, _ = _tile_path(ps, ss, co, gfeat, m0, v0, 1, 1e-2, 0)
    torch.testing.assert_close(mb, ma, rtol=2e-4, atol=1e-9)
    torch.testing.assert_close(vb, va, rtol=4e-4, atol=1e-14)
    # skip flag
    dyn = ops.new_adam_dyn(DEV)
    dyn[0] = 1  # nonfinite
    ops.adam_prepare(dyn, 1e-2, policy="skip_step")
    pc, mc, vc, _, _ = _tile_path(ps, ss, co, gfeat, m0, v0, 1, 1e-2, 0, dyn=dyn)
    assert torch.equal(pc, ps.planes.detach()) and torch.equal(mc, m0) and torch.equal(vc, v0)


def test_training_steps_with_and_without_the_tile_kernel_agree():
    """Five train steps of the default trainer (tile kernel on) and with cfg.tile_adam = False: same rendered colours, losses and parameters
    up to accumulation order."""
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    small = dict(aabb_scale=1.5, spacetime_resolution=(16, 16, 16, 4), multiscale_res=(1, 2, 4), feature_dim=32,
                 proposal_resolutions=((24, 24, 24, 4), (32, 32, 32, 4)), proposal_feature_dim=8, num_proposal_samples_per_ray=(64, 32),
                 num_nerf_samples_per_ray=16, warm_up_end=2, mlp_operands="fp32")
    R = 512
    trs = [KPlanesTrainer(KPlanesTrainConfig(**small, tile_adam=t), R, DEV) for t in (True, False)]
    assert trs[0].tile_adam and not trs[1].tile_adam
    gen = torch.Generator().manual_seed(2)
    g = lambda z: z.to(DEV).contiguous()
    for step in range(5):
        o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 1.2
        d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
        rays = {"origins": g(o), "directions": g(d), "times": g(torch.rand(R, 1, generator=gen))}
        target = g(torch.rand(R, 3, generator=gen))
        rng = {"t_rand": g(torch.rand(R, 65, generator=gen)), "u": [g(torch.rand(R, 33, generator=gen)), g(torch.rand(R, 17, generator=gen))],
               "bg": g(torch.rand(R, 3, generator=gen))}
        outs = [tr.train_step(rays, target, rng).clone() for tr in trs]
        if step == 0:
            assert torch.equal(outs[0], outs[1])
        else:
            torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=2e-3)
    for tr in trs:
        tr.synchronize()
    ld0, ld1 = trs[0].loss_dict(), trs[1].loss_dict()
    for k in ld0:
        torch.testing.assert_close(ld0[k], ld1[k], rtol=2e-2, atol=1e-8)
    assert float((trs[0].params - trs[1].params).abs().mean()) < 2e-5
    assert float(trs[0].grads.abs().max()) == 0.0 and trs[0].step == 5

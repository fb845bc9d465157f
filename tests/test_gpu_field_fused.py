"""GPU parity of the fused K-Planes field forward (csrc/field_fused.hip: gather -> sigma_net -> colour net in one kernel) against

* the UNFUSED composition with the same 16-bit MFMA operands (snerf_kplanes_gather_fwd + snerf_mlp_fwd x 2): density / rgb and the
  tensors left behind for the unfused backward bit for bit;
* the fp32 CPU oracle restating KPlanesField (NS/fields/kplanes_field.py:275-358) within SURVEY 8d's bf16 tolerance
  (density rtol 2e-2, rgb atol 4e-3);
* whole training steps: the fused trainer path against the unfused one."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(ms, N, operands, seed=0, ragged=True):
    from soccernerfs_amd.plane_set import PlaneSet
    from soccernerfs_amd.tcnn_compat import Network

    gen = torch.Generator().manual_seed(seed)
    base = (12, 10, 9, 6)
    ps = PlaneSet(32, [[r * m for r in base[:3]] + [base[3]] for m in ms], concat=True, generator=gen)
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, generator=gen) * 0.9 + 0.3)
    mk = lambda i, o, h, nh, act: Network(i, o, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": h,
                                                   "n_hidden_layers": nh}, seed=seed + 7 * i, operands=operands)
    sigma, color = mk(32 * len(ms), 16, 128, 1, "None"), mk(15, 3, 64, 2, "Sigmoid")
    pts = torch.rand(N, 4, generator=gen) * 2.2 - 1.1  # some points outside the box (border clamp)
    return ps.to(DEV), sigma.to(DEV), color.to(DEV), pts.to(DEV)


def _unfused_forward(ps, sigma, color, pts):
    from soccernerfs_amd import _lib, ops

    N = pts.shape[0]
    L = _lib.lib()
    co = ops.coords_from_points(pts)
    desc = ps.desc()
    feat = torch.empty(N, ps.out_dim, device=DEV)
    _lib.check(L.snerf_kplanes_gather_fwd(C.byref(desc), ops._ptr(ps.planes), C.byref(co), C.c_int64(N), ops._ptr(feat), ops._stream()))
    h, dens, rgb = torch.empty(N, 16, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
    _lib.check(L.snerf_mlp_fwd(C.byref(sigma.desc), ops._ptr(sigma.params), ops._ptr(feat), ps.out_dim, C.c_int64(N), ops._ptr(h), 16, 15, ops._ptr(dens),
                               ops._stream()))
    _lib.check(L.snerf_mlp_fwd(C.byref(color.desc), ops._ptr(color.params), ops._ptr(h), 16, C.c_int64(N), ops._ptr(rgb), 3, -1, None, ops._stream()))
    return feat, h, dens, rgb


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
@pytest.mark.parametrize("ms,N", [((1, 2, 4, 8, 16), 5000), ((1, 2), 1031), ((1,), 64), ((1, 2, 4), 33), ((1, 2, 3, 4, 6, 8), 2100)])
def test_fused_field_forward(ms, N, operands):
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import _lib, ops

    ps, sigma, color, pts = _setup(ms, N, operands)
    L = _lib.lib()
    desc = ps.desc()
    assert L.snerf_kplanes_field_fwd_supported(C.byref(desc), C.byref(sigma.desc), C.byref(color.desc)) == 1
    co = ops.coords_from_points(pts)
    dens, rgb = torch.full((N,), -1.0, device=DEV), torch.full((N, 3), -1.0, device=DEV)
    _lib.check(L.snerf_kplanes_field_fwd(C.byref(desc), ops._ptr(ps.planes), C.byref(co), C.c_int64(N), C.byref(sigma.desc), ops._ptr(sigma.params),
                                         C.byref(color.desc), ops._ptr(color.params), ops._ptr(dens), ops._ptr(rgb), None, None, None, ops._stream()))
    feat_u, h_u, dens_u, rgb_u = _unfused_forward(ps, sigma, color, pts)
    assert torch.equal(dens, dens_u) and torch.equal(rgb, rgb_u)  # same arithmetic, same order: bit for bit
    # optional outputs for an unfused backward: the operand-typed feature tile and the raw sigma_net outputs
    dt = torch.bfloat16 if operands == "bf16" else torch.float16
    feat16, h = torch.full((N, 32 * len(ms)), -1.0, device=DEV, dtype=dt), torch.full((N, 16), -1.0, device=DEV)
    feat32 = torch.full((N, 32 * len(ms)), -1.0, device=DEV)
    dens2, rgb2 = torch.empty_like(dens), torch.empty_like(rgb)
    _lib.check(L.snerf_kplanes_field_fwd(C.byref(desc), ops._ptr(ps.planes), C.byref(co), C.c_int64(N), C.byref(sigma.desc), ops._ptr(sigma.params),
                                         C.byref(color.desc), ops._ptr(color.params), ops._ptr(dens2), ops._ptr(rgb2), ops._ptr(feat16), ops._ptr(h),
                                         ops._ptr(feat32), ops._stream()))
    assert torch.equal(dens2, dens) and torch.equal(rgb2, rgb)
    assert torch.equal(feat16, feat_u.to(dt)) and torch.equal(h, h_u) and torch.equal(feat32, feat_u)
    # fp32 oracle of the field (kplanes_field.py:275-358)
    grids = [[t.cpu() for t in sc] for sc in ps.to_reference()]
    feat = KO.interpolate_kplanes(pts.cpu(), grids, True)
    hh = KO.mlp(feat, [w.cpu() for w in sigma.linear_weights()])
    want_d = torch.exp(hh[:, 15])
    want_rgb = KO.mlp(hh[:, :15], [w.cpu() for w in color.linear_weights()], out_act="Sigmoid")
    torch.testing.assert_close(dens.cpu(), want_d, rtol=2e-2, atol=1e-3)
    torch.testing.assert_close(rgb.cpu(), want_rgb, rtol=0, atol=4e-3)


def test_fused_and_unfused_training_steps_agree():
    """Three train steps (bf16 operands) with the fused forward + unfused backward (the default) and with cfg.fused_field = False:
    rendered colours bit for bit at step 0, parameters to atomic-order noise afterwards."""
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    small = dict(aabb_scale=1.5, spacetime_resolution=(16, 16, 16, 4), multiscale_res=(1, 2), feature_dim=32,
                 proposal_resolutions=((24, 24, 24, 4), (32, 32, 32, 4)), proposal_feature_dim=8, num_proposal_samples_per_ray=(64, 32),
                 num_nerf_samples_per_ray=16, warm_up_end=2, mlp_operands="bf16")
    R = 256
    trs = [KPlanesTrainer(KPlanesTrainConfig(**small, fused_field=f), R, DEV) for f in (True, False)]
    assert trs[0].fused_field and not trs[1].fused_field
    gen = torch.Generator().manual_seed(2)
    g = lambda z: z.to(DEV).contiguous()
    for step in range(3):
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
    ld1 = trs[1].loss_dict()
    for tr in (trs[0],):
        assert float((tr.params - trs[1].params).abs().mean()) < 2e-5
        ld0 = tr.loss_dict()
        for k in ld0:
            torch.testing.assert_close(ld0[k], ld1[k], rtol=2e-2, atol=1e-7)
    # eval forward (no backward): fused as well
    rgb = [tr.forward(rays, None, 1.0, training=False).clone() for tr in trs]
    torch.testing.assert_close(rgb[0], rgb[1], rtol=0, atol=2e-3)

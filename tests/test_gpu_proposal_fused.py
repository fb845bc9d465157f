"""GPU parity of the fused proposal density (csrc/proposal_fused.hip: plane gather -> 8 -> 64 -> 1 net -> trunc_exp in one kernel per level;
KPlanesDensityField.get_density, NS/fields/kplanes_field.py:410-460) against

* the UNFUSED composition with the same 16-bit operands (snerf_kplanes_gather_fwd + snerf_mlp_fwd with the trunc_exp aux output): densities and
  the [N,8] features bit for bit, for explicit points and for samples derived in-kernel from rays + bin edges, ragged N, both operand types, with and
  without the hidden ReLU (the linear-decoder model's proposal fields have none);
* the fp32 CPU oracle (density_field_forward) within SURVEY 8d's 16-bit tolerance;
* a training step of the fused trainer with the switch on and off."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(res, operands, act="ReLU", seed=0):
    from soccernerfs_amd.plane_set import PlaneSet
    from soccernerfs_amd.tcnn_compat import Network

    gen = torch.Generator().manual_seed(seed)
    ps = PlaneSet(8, [list(res)], concat=False, a=0.1, b=0.9, generator=gen)
    net = Network(8, 1, {"otype": "FullyFusedMLP", "activation": act, "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}, seed=seed + 3,
                  operands=operands)
    return ps.to(DEV), net.to(DEV), gen


def _unfused(ps, net, co, N):
    from soccernerfs_amd import _lib, ops

    L = _lib.lib()
    desc = ps.desc()
    feat, out, dens = torch.empty(N, 8, device=DEV), torch.empty(N, 1, device=DEV), torch.empty(N, device=DEV)
    _lib.check(L.snerf_kplanes_gather_fwd(C.byref(desc), ops._ptr(ps.planes), C.byref(co), C.c_int64(N), ops._ptr(feat), ops._stream()))
    _lib.check(L.snerf_mlp_fwd(C.byref(net.desc), ops._ptr(net.params), ops._ptr(feat), 8, C.c_int64(N), ops._ptr(out), 1, 0, ops._ptr(dens), ops._stream()))
    return feat, dens


def _fused(ps, net, co, N, keep=True):
    from soccernerfs_amd import _lib, ops

    L = _lib.lib()
    desc = ps.desc()
    assert L.snerf_kplanes_density_fwd_supported(C.byref(desc), C.byref(net.desc)) == 1
    feat, dens = torch.full((N, 8), -7.0, device=DEV), torch.full((N,), -7.0, device=DEV)
    _lib.check(L.snerf_kplanes_density_fwd(C.byref(desc), ops._ptr(ps.planes), C.byref(co), C.c_int64(N), C.byref(net.desc), ops._ptr(net.params), ops._ptr(dens),
                                           ops._ptr(feat) if keep else None, ops._stream()))
    torch.cuda.synchronize()
    return feat, dens


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
@pytest.mark.parametrize("res,N,act", [((128, 128, 128, 100), 70000, "ReLU"), ((24, 20, 18, 5), 1031, "ReLU"), ((9, 7, 5, 3), 64, "None"), ((16, 16, 16, 4), 1, "ReLU")])
def test_fused_density_equals_gather_plus_net_on_points(res, N, act, operands):
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    ps, net, gen = _setup(res, operands, act)
    pts = (torch.rand(N, 4, generator=gen) * 2.2 - 1.1).to(DEV)  # some outside the box: border clamp
    co = ops.coords_from_points(pts)
    feat_u, dens_u = _unfused(ps, net, co, N)
    feat_f, dens_f = _fused(ps, net, co, N)
    assert torch.equal(feat_f, feat_u) and torch.equal(dens_f, dens_u)
    _, dens_n = _fused(ps, net, co, N, keep=False)  # without the feature output: same densities, buffer untouched
    assert torch.equal(dens_n, dens_u)
    # fp32 oracle (exact net): SURVEY 8d's tolerance for 16-bit operands
    grids = [t.cpu() for t in ps.to_reference()[0]]
    feats = KO.interpolate_kplanes(pts.cpu(), [grids], concat_features=False)
    y = KO.mlp(feats, [w.cpu() for w in net.linear_weights()], hidden_act=act)
    torch.testing.assert_close(dens_f.cpu(), torch.exp(y[:, 0]), rtol=2e-2, atol=1e-6)


def test_fused_density_on_ray_samples_equals_unfused():
    """coords mode 1: positions derived in-kernel from origins, directions and euclidean bin edges (what the trainer hands over), [0,1] proposal
    coordinates (rescale False)."""
    from soccernerfs_amd import ops

    ps, net, gen = _setup((128, 128, 128, 100), "bf16")
    R, S = 300, 256
    o = ((torch.rand(R, 3, generator=gen) * 2 - 1) * 1.2).to(DEV)
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1).to(DEV)
    t = torch.rand(R, generator=gen).to(DEV)
    eb = torch.sort(torch.rand(R, S + 1, generator=gen) * 3.0, dim=-1).values.to(DEV)
    for rescale in (False, True):
        co = ops.coords_from_rays(o, d, t, eb, [[-1.5] * 3, [1.5] * 3], rescale)
        feat_u, dens_u = _unfused(ps, net, co, R * S)
        feat_f, dens_f = _fused(ps, net, co, R * S)
        assert torch.equal(feat_f, feat_u) and torch.equal(dens_f, dens_u)


def test_other_shapes_are_refused():
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.plane_set import PlaneSet
    from soccernerfs_amd.tcnn_compat import Network

    L = _lib.lib()
    ps, net, _ = _setup((16, 16, 16, 4), "bf16")
    net32 = Network(8, 1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}).to(DEV)
    ps32 = PlaneSet(32, [[8, 8, 8, 4]], concat=True).to(DEV)
    d8, d32 = ps.desc(), ps32.desc()
    assert L.snerf_kplanes_density_fwd_supported(C.byref(d8), C.byref(net32.desc)) == 0
    assert L.snerf_kplanes_density_fwd_supported(C.byref(d32), C.byref(net.desc)) == 0
    co = ops.coords_from_points(torch.zeros(4, 4, device=DEV))
    rc = L.snerf_kplanes_density_fwd(C.byref(d8), ops._ptr(ps.planes), C.byref(co), C.c_int64(4), C.byref(net32.desc), ops._ptr(net32.params),
                                     ops._ptr(torch.zeros(4, device=DEV)), None, ops._stream())
    assert rc != 0 and b"16-bit operands" in L.snerf_last_error()


def test_train_step_with_and_without_the_fused_proposal_levels():
    """The fused trainer with fused_proposal on / off: the forward's bins, weights and rgb are bit-identical (the densities are), and so is everything the
    backward derives from them on a step that updates the proposal networks."""
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    small = dict(aabb_scale=1.5, spacetime_resolution=(16, 16, 16, 4), multiscale_res=(1, 2), proposal_resolutions=((24, 24, 24, 4), (32, 32, 32, 4)),
                 num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=16, deterministic=True, quotient_scatter=False)
    R = 96
    outs = []
    for fused in (True, False):
        tr = KPlanesTrainer(KPlanesTrainConfig(fused_proposal=fused, **small), R, DEV)
        assert tr.fused_proposal == fused
        gen = torch.Generator(device=DEV).manual_seed(3)
        o = (torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1) * 0.8
        d = torch.nn.functional.normalize(torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1, dim=-1)
        rays = {"origins": o.contiguous(), "directions": d.contiguous(), "times": torch.rand(R, 1, device=DEV, generator=gen)}
        target = torch.rand(R, 3, device=DEV, generator=gen)
        rng = {"t_rand": torch.rand(R, 65, device=DEV, generator=gen), "u": [torch.rand(R, 33, device=DEV, generator=gen), torch.rand(R, 17, device=DEV, generator=gen)],
               "bg": torch.rand(R, 3, device=DEV, generator=gen)}
        rgb = tr.train_step(rays, target, rng).clone()
        tr.synchronize()
        outs.append((rgb, [w.clone() for w in tr.buf["w"]], tr.params.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][2], outs[1][2])  # deterministic mode: fixed-point accumulation, so one Adam step lands on the same bits

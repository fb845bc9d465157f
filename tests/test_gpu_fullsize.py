"""GPU parity at BASELINE.json's full sizes for configs 3 and 4, where the CPU oracle is too slow to be the checker:
size-independent properties, agreement between independent HIP code paths, and a plain-torch fp32 restatement evaluated on the
GPU (test-only; the product never calls it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_interp(pts, grids, concat):
    """fp32 torch restatement of interpolate_kplanes (NS/fields/kplanes_field.py:77-126) on whatever device pts lives on."""
    import itertools
    import torch.nn.functional as F

    combs = list(itertools.combinations(range(pts.shape[-1]), 2))
    outs = []
    acc = 0.0
    for sc in grids:
        f = 1.0
        for ci, (a, b) in enumerate(combs):
            g = sc[ci]  # [1, C, H, W]
            xy = torch.stack([pts[:, a], pts[:, b]], -1)[None, :, None, :]
            v = F.grid_sample(g, xy, align_corners=True, mode="bilinear", padding_mode="border")  # [1, C, N, 1]
            f = f * v[0, :, :, 0].T
        if concat:
            outs.append(f)
        else:
            acc = acc + f
    return torch.cat(outs, -1) if concat else acc


def test_config3_gather_and_scatter_full_size():
    """Config 3: multiscale_res (1,2,4,8,16,32), C = 32, 64 x 4096 samples, 575 447 040 plane parameters: the preset's OWN time resolution 100
    (the reference's README changes multiscale-res / ist-range / fps-downsample only; NS/configs/method_configs.py:515 keeps (64,64,64,100))."""
    import ctypes as Ct
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.plane_set import PlaneSet

    gen = torch.Generator(device=DEV).manual_seed(5)
    reso = [[64 * m, 64 * m, 64 * m, 100] for m in (1, 2, 4, 8, 16, 32)]
    ps = PlaneSet(32, reso, concat=True, device=DEV)
    assert ps.numel == 575_447_040
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, device=DEV, generator=gen) * 0.8 + 0.6)
    N = 64 * 4096
    pts = torch.rand(N, 4, device=DEV, generator=gen) * 2.1 - 1.05
    out = ops.interpolate_kplanes(pts, ps)
    assert out.shape == (N, 192)
    # forward against torch's grid_sample on the reference layout, one scale at a time to bound memory
    sub = slice(0, 32768)
    for s in range(6):
        grids = [[ps.plane_view(s, p).permute(2, 0, 1)[None].contiguous() for p in range(6)]]
        ref = _torch_interp(pts[sub], grids, True)
        torch.testing.assert_close(out[sub, s * 32:(s + 1) * 32], ref, rtol=2e-5, atol=1e-6)
        del grids, ref
    # backward: the sorted scatter and the sample-major scatter are independent kernels and must agree
    gout = torch.rand(N, 192, device=DEV, generator=gen) - 0.5
    co = ops.coords_from_points(pts)
    desc = ps.desc()
    direct = torch.zeros_like(ps.planes)
    _lib.check(_lib.lib().snerf_kplanes_gather_bwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(gout), ops._ptr(direct),
                                                   ops._stream()))
    ss = ops.SortedScatter(ps, N, DEV)
    ss.sort(co)
    got = torch.zeros_like(ps.planes)
    ss.scatter(ps.planes, co, gout, got)
    torch.testing.assert_close(got, direct, rtol=1e-4, atol=2e-5)
    del ss
    # the quotient form (g_q = (gout .* features) ./ v_q, v_q recomputed in pass B) at the same size: features from the forward above
    ssq = ops.SortedScatter(ps, N, DEV, quotient=True)
    ssq.sort(co)
    got.zero_()
    ssq.scatter_quotient(ps.planes, co, gout, out, got)
    assert int(ssq.fix_count.item()) == 0  # positive planes: no feature vanishes
    torch.testing.assert_close(got, direct, rtol=1e-4, atol=2e-5)
    assert float((got - direct).norm() / direct.norm()) < 2e-6
    del ssq
    # conservation on the finest scale: sum over texels of dL/dplane = sum_n gout * prod(other planes), checked against autograd of the torch restatement
    s = 5
    grids = [[ps.plane_view(s, p).detach().permute(2, 0, 1)[None].contiguous().requires_grad_(True) for p in range(6)]]
    ref = _torch_interp(pts[sub], grids, True)
    ref.backward(gout[sub, s * 32:(s + 1) * 32])
    part = torch.zeros_like(ps.planes)
    co_sub = ops.coords_from_points(pts[sub].contiguous())
    gsub = torch.zeros(32768, 192, device=DEV)
    gsub[:, s * 32:(s + 1) * 32] = gout[sub, s * 32:(s + 1) * 32]
    _lib.check(_lib.lib().snerf_kplanes_gather_bwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co_sub), Ct.c_int64(32768), ops._ptr(gsub), ops._ptr(part),
                                                   ops._stream()))
    for p in range(6):
        torch.testing.assert_close(ps.plane_view(s, p, part), grids[0][p].grad[0].permute(1, 2, 0), rtol=1e-4, atol=1e-5)


def test_config3_fused_train_steps():
    """Config 3 through the fused trainer (sigma net 192 -> 128 -> 16): both backward variants give the same gradients; a few
    steps run and reduce the loss on a fixed batch."""
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    R = 2048
    cfg = KPlanesTrainConfig(mlp_operands="fp32", multiscale_res=(1, 2, 4, 8, 16, 32), spacetime_resolution=(64, 64, 64, 100),
                             proposal_resolutions=((128, 128, 128, 100), (256, 256, 256, 100)))
    tr = KPlanesTrainer(cfg, R, DEV)
    assert tr.n_params == 578_367_744  # config 3 = the preset with six scales
    gen = torch.Generator(device=DEV).manual_seed(1)
    o = (torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1) * 0.8
    d = torch.nn.functional.normalize(torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1, dim=-1)
    rays = {"origins": o.contiguous(), "directions": d.contiguous(), "times": torch.rand(R, 1, device=DEV, generator=gen)}
    target = torch.rand(R, 3, device=DEV, generator=gen)
    rng = tr.random_draws()
    grads = {}
    for sorted_scatter in (True, False):
        tr.sorted_scatter = sorted_scatter
        tr.grads.zero_()
        tr.forward(rays, rng, 0.5, training=True)
        tr.backward(target, rng, proposal_grads=True, include_reg=True)
        torch.cuda.synchronize()
        grads[sorted_scatter] = tr.gviews["field.planes"].clone()
        assert bool(torch.isfinite(tr.grads).all())
    scale = float(grads[False].abs().max())
    assert scale > 0
    torch.testing.assert_close(grads[True], grads[False], rtol=1e-3, atol=1e-5 * scale)
    tr.grads.zero_()
    tr.sorted_scatter = True
    losses = []
    for _ in range(6):
        tr.train_step(rays, target, rng)
        losses.append(float(tr.loss_dict()["rgb_loss"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_config3_ist_sampler_full_size():
    """Config 3's sampler inputs: 19 cameras x 25 frames (fps-downsample 4) of 960x540, ist_range 0.75."""
    from soccernerfs_amd import synthetic
    from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler, compute_ist

    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 4)
    assert len(times) == 25
    data = synthetic.render_dataset(cams, times, list(range(19)), torch.device(DEV), chunk_rows=540)
    M, H, W = data["images"].shape[:3]
    assert (M, H, W) == (19 * 25, 540, 960)
    maps = compute_ist(data["images"], data["cam_id"], data["times"], 0.75)
    assert maps.shape == (M, H, W)
    nz = (maps > 0).float().mean()
    assert 0 < float(nz) < 0.5  # only what the moving player / ball sweep within the +-0.75 time window is non-zero
    # static pixels (ground far from the players) carry zero weight in every frame
    assert float(maps[:, :8, :8].float().abs().max()) == 0.0
    R = 4096
    smp = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
    batch = {"image": data["images"], "image_idx": torch.arange(M, device=DEV), "ist_weights": maps, "iter_steps": 5000}
    idx = smp.sample_method(R, M, H, W, batch=batch, device=DEV)
    num_ist = int(0.15 * R)
    ist = idx[:num_ist]
    assert bool((maps[ist[:, 0], ist[:, 1], ist[:, 2]] > 0).all())
    assert int(idx[:, 0].max()) < M and int(idx[:, 1].max()) < H and int(idx[:, 2].max()) < W


def test_config4_temporal_grid_full_size():
    """Config 4's main table: 16 levels x 2 features, 2^19 rows, temporal_dim 64 (2^19 x 16 x ... = 403.9 M floats)."""
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder

    enc = TemporalGridEncoder(temporal_dim=64, input_dim=3, num_levels=16, level_dim=2, per_level_scale=float(np.exp(np.log(2048 / 16) / 15)),
                              base_resolution=16, log2_hashmap_size=19).to(DEV)
    assert enc.embeddings.numel() > 400_000_000
    B = 48 * 4096
    gen = torch.Generator(device=DEV).manual_seed(2)
    x = torch.rand(B, 3, device=DEV, generator=gen)
    t = torch.rand(B, 1, device=DEV, generator=gen)
    k = 0.375
    with torch.no_grad():
        enc.embeddings.fill_(k)
    out = enc(x, t)
    assert out.shape == (B, 32)
    torch.testing.assert_close(out, torch.full_like(out, k), rtol=1e-6, atol=0)  # interpolation weights sum to one at every level
    go = torch.rand_like(out) - 0.3
    enc.embeddings.grad = None
    out.backward(go)
    g = enc.embeddings.grad
    # gradient mass is conserved: each output element distributes its gradient with weights that sum to one
    torch.testing.assert_close(g.double().sum(), go.double().sum(), rtol=1e-5, atol=1e-3)
    # explicit temporal rows give the same result as in-kernel rows
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    a = enc(x[:8192], t[:8192])
    b = enc(x[:8192], t[:8192], explicit_rows=True)
    torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_config4_model_trains_at_preset_size():
    """nerfplayer-nerfacto preset (method_configs.py:616-660): R = 4096, samples (256, 96, 48), full-size tables."""
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModel, NerfplayerNerfactoModelConfig
    from soccernerfs_amd.rays import RayBundle
    from soccernerfs_amd.scene_colliders import SceneBox

    torch.manual_seed(0)
    model = NerfplayerNerfactoModel(NerfplayerNerfactoModelConfig(), SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=30 * 100).to(DEV).train()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params > 420_000_000
    R = 4096
    gen = torch.Generator(device=DEV).manual_seed(3)
    o = (torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1) * 0.3
    d = torch.nn.functional.normalize(torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1, dim=-1)
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1, device=DEV), camera_indices=torch.randint(0, 3000, (R, 1), device=DEV, generator=gen),
                   times=torch.rand(R, 1, device=DEV, generator=gen))
    target = torch.rand(R, 3, device=DEV, generator=gen) * 0.5
    opt = torch.optim.Adam([p for g in model.get_param_groups().values() for p in g if p.requires_grad], lr=1e-2, eps=1e-15)
    losses = []
    for step in range(4):
        for where, fn in model.get_training_callbacks():
            if where == "before":
                fn(step)
        out = model(rb)
        assert out["rgb"].shape == (R, 3)
        ld = model.get_loss_dict(out, {"image": target}, model.get_metrics_dict(out, {"image": target}))
        loss = sum(ld.values())
        assert bool(torch.isfinite(loss))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(ld["rgb_loss"].detach()))
    assert losses[-1] < losses[0], losses


def test_full_nerfplayer_hashgrid_full_size():
    """The full NeRFPlayer's static grid at the `nerfplayer` preset's size (16 levels x 2 features, 2^18-row levels, 48 x 4096 samples),
    through size-independent properties: a constant table is reproduced exactly (interpolation weights sum to one at every level, inside
    and outside [0,1]); its coordinate gradient vanishes; gradient mass is conserved by the table scatter; the encoding is linear in the
    table; the coordinate gradient matches a central finite difference of the encoding itself."""
    from soccernerfs_amd.tcnn_compat import Encoding

    enc = Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 18, "base_resolution": 16,
                       "per_level_scale": 1.4472692012786865}).to(DEV)
    B = 48 * 4096
    gen = torch.Generator(device=DEV).manual_seed(4)
    x = (torch.rand(B, 3, device=DEV, generator=gen) * 1.4 - 0.2).requires_grad_(True)  # a fifth of the points leave [0,1] (deformed positions do)
    k = -0.625
    with torch.no_grad():
        enc.params.fill_(k)
    out = enc(x)
    assert out.shape == (B, 32)
    torch.testing.assert_close(out, torch.full_like(out, k), rtol=1e-6, atol=0)
    go = torch.rand(B, 32, device=DEV, generator=gen) - 0.3
    out.backward(go)
    torch.testing.assert_close(enc.params.grad.double().sum(), go.double().sum(), rtol=1e-5, atol=1e-3)
    assert float(x.grad.abs().max()) <= 1e-3 * 4096 * abs(k) * 1e-3  # d(constant)/dx = 0 up to the cancellation of O(scale * k) terms
    # linearity in the table
    with torch.no_grad():
        t1 = torch.rand(enc.params.shape, device=DEV, generator=gen) - 0.5
        t2 = torch.rand(enc.params.shape, device=DEV, generator=gen) - 0.5
        xs = x.detach()[:65536]
        enc.params.copy_(t1); a = enc(xs)
        enc.params.copy_(t2); b = enc(xs)
        enc.params.copy_(2.0 * t1 - 3.0 * t2); c = enc(xs)
    torch.testing.assert_close(c, 2.0 * a - 3.0 * b, rtol=1e-4, atol=1e-5)
    # coordinate gradient against a finite difference on the coarse levels (cells wide enough for the step)
    with torch.no_grad():
        enc.params.copy_(t1)
    xq = (torch.rand(4096, 3, device=DEV, generator=gen) * 0.9 + 0.05).requires_grad_(True)
    w = torch.zeros(32, device=DEV)
    w[:6] = torch.tensor([1.0, -2.0, 0.5, 1.5, -1.0, 2.0], device=DEV)  # levels 0..2: resolution 16, 24, 34
    (enc(xq) * w).sum().backward()
    eps = 1e-4
    for d in range(3):
        dx = torch.zeros(1, 3, device=DEV)
        dx[0, d] = eps
        with torch.no_grad():
            fd = ((enc(xq.detach() + dx) - enc(xq.detach() - dx)) * w).sum(-1) / (2 * eps)
        # points whose +-eps neighbourhood crosses a cell border see a kink: compare the robust bulk
        err = (fd - xq.grad[:, d]).abs()
        assert float(err.median()) < 2e-2 * float(fd.abs().median() + 1e-6) and float((err < 0.05 * (fd.abs() + 1.0)).float().mean()) > 0.97


def test_16bit_operand_mlp_full_size():
    """Opt-in bf16 / fp16 MFMA operands at the K-Planes preset's sizes (sigma_net 160 -> 128 -> 16 over 64 x 4096 samples, proposal net
    8 -> 64 -> 1 over 256 x 4096): deterministic (two launches agree bit for bit on every atomics-free output), row-permutation
    equivariant, and within the stated tolerance of the exact fp32 kernels."""
    from soccernerfs_amd.tcnn_compat import Network

    for d_in, d_out, hidden, N in ((160, 16, 128, 64 * 4096), (8, 1, 64, 256 * 4096)):
        cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": hidden, "n_hidden_layers": 1}
        gen = torch.Generator(device=DEV).manual_seed(5)
        x = torch.rand(N, d_in, device=DEV, generator=gen) - 0.3
        go = torch.rand(N, d_out, device=DEV, generator=gen) - 0.5
        ref = Network(d_in, d_out, cfg).to(DEV)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        yr.backward(go)
        perm = torch.randperm(N, device=DEV, generator=gen)
        for operands in ("bf16", "fp16"):
            net = Network(d_in, d_out, cfg, operands=operands).to(DEV)
            with torch.no_grad():
                net.params.copy_(ref.params)
            xg = x.clone().requires_grad_(True)
            y = net(xg)
            y.backward(go)
            gw = net.params.grad.clone()
            net.params.grad = None
            xg2 = x.clone().requires_grad_(True)
            y2 = net(xg2)
            y2.backward(go)
            assert torch.equal(y, y2) and torch.equal(xg.grad, xg2.grad)
            torch.testing.assert_close(net.params.grad, gw, rtol=1e-4, atol=1e-5 * float(gw.abs().max()))  # flushed with atomics
            with torch.no_grad():
                torch.testing.assert_close(net(x[perm]), y.detach()[perm], rtol=0, atol=0)
            # raw (pre-activation) outputs of O(1): operand rounding of 2^-9 (bf16) / 2^-12 (fp16) relative, summed over 128 hidden units
            torch.testing.assert_close(y, yr, rtol=2e-2, atol=(1e-2 if operands == "bf16" else 2e-3) * max(1.0, float(yr.detach().abs().max())))
            rel = lambda u, v: float((u - v).norm() / (v.norm() + 1e-20))
            assert rel(y, yr) < (5e-3 if operands == "bf16" else 1e-3)
            lim = 8e-2 if operands == "bf16" else 2e-2
            assert rel(xg.grad, xr.grad) < lim and rel(gw, ref.params.grad) < lim, (operands, rel(xg.grad, xr.grad), rel(gw, ref.params.grad))

"""GPU: the fused NeRFPlayer-nerfacto trainer (soccernerfs_amd.nerfplayer_trainer) against the nerfstudio-shaped autograd model on the same
HIP kernels -- which is itself pinned against the reference's own model by golden G12 (tests/test_gpu_tgrid.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg():
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModelConfig

    return NerfplayerNerfactoModelConfig(
        num_levels=6, log2_hashmap_size=12, temporal_dim=16,
        proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 32},
                                {"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 64}],
        num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=16)


def _batch(R, n_img, seed):
    gen = torch.Generator().manual_seed(seed)
    o = ((torch.rand(R, 3, generator=gen) * 2 - 1) * 0.3).to(DEV)
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1).to(DEV)
    times = torch.rand(R, 1, generator=gen).to(DEV)
    cams = torch.randint(0, n_img, (R,), generator=gen).to(DEV)
    target = torch.rand(R, 3, generator=gen).to(DEV)
    rng = {"t_rand": torch.rand(R, 1, generator=gen).to(DEV), "u": [torch.rand(R, 1, generator=gen).to(DEV), torch.rand(R, 1, generator=gen).to(DEV)],
           "bg": torch.rand(R, 3, generator=gen).to(DEV)}
    return {"origins": o, "directions": d, "times": times}, cams, target, rng


def test_fused_step_equals_autograd_model():
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModel
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer
    from soccernerfs_amd.rays import RayBundle
    from soccernerfs_amd.scene_colliders import SceneBox

    R, n_img = 96, 7
    cfg = _cfg()
    tr = NerfplayerTrainer(cfg, R, n_img, aabb_scale=1.0, device=DEV, seed=3)
    with torch.no_grad():  # O(1) tables: the 1e-4 init gives a featureless field
        for name in ("field.table", "prop0.table", "prop1.table"):
            tr.views[name].uniform_(-1.0, 1.0)
    model = NerfplayerNerfactoModel(cfg, SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=n_img).to(DEV).train()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    pairs = {"field.table": model.field.mlp_base.embeddings, "field.decode": model.field.mlp_base_decode.params, "field.head": model.field.mlp_head.params,
             "field.appearance": model.field.embedding_appearance.weight}
    for i, pn in enumerate(model.proposal_networks):
        pairs[f"prop{i}.table"], pairs[f"prop{i}.mlp"] = pn.encoding.embeddings, pn.linear.params
    with torch.no_grad():
        for name, p in pairs.items():
            p.copy_(tr.views[name].view(p.shape))
    rays, cams, target, rng = _batch(R, n_img, 11)
    anneal, rows = 0.4, [2, 1, 3]
    # ---- autograd model ----
    draws = [rng["t_rand"], rng["u"][0], rng["u"][1], rng["bg"]]
    model.set_rand_fn(lambda shape, device=None: draws.pop(0))
    encs = [model.field.mlp_base] + [p.encoding for p in model.proposal_networks]
    for e in encs:
        e.fuse_tv = False
    model.tv_row_fn = lambda enc: rows[[id(e) for e in encs].index(id(enc))]
    model.proposal_sampler.set_anneal(anneal)
    out = model(RayBundle(origins=rays["origins"], directions=rays["directions"], pixel_area=torch.ones(R, 1, device=DEV), camera_indices=cams[:, None],
                          times=rays["times"]))
    ld = model.get_loss_dict(out, {"image": target}, model.get_metrics_dict(out, {"image": target}))
    sum(ld.values()).backward()
    # ---- fused trainer ----
    tr.tv_rows = rows
    rgb = tr.forward(rays, cams, rng, anneal)
    tr.backward(target, rng, proposal_grads=True)
    tr.materialize_tv_gradient()  # the Adam sweep adds it on the fly; make it explicit for the comparison
    torch.testing.assert_close(rgb, out["rgb"].detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(tr.buf["acc"], out["accumulation"].detach()[:, 0], rtol=1e-4, atol=1e-6)
    for i in range(3):
        torch.testing.assert_close(tr.buf["eb"][i], out["ray_samples_list"][i]._compact["ebins"], rtol=0, atol=1e-6)
    mine = tr.loss_dict()
    assert set(mine) == set(ld)
    for k in ld:
        torch.testing.assert_close(mine[k], ld[k].detach(), rtol=1e-4, atol=1e-9)
    for name, p in pairs.items():
        g_ref = p.grad.reshape(-1)
        g = tr.gviews[name].reshape(-1)
        scale = float(g_ref.abs().max())
        assert scale > 0, name
        torch.testing.assert_close(g, g_ref, rtol=1e-3, atol=1e-5 * scale, msg=lambda m: f"{name}: {m}")


def test_adam_with_tv_folded_in_equals_explicit_gradient():
    """snerf_tgrid_tv_sign + snerf_adam_step_tv == TV gradient added to the gradient buffer, then plain Adam."""
    import ctypes as C
    from soccernerfs_amd import _lib, ops

    gen = torch.Generator(device=DEV).manual_seed(4)
    rows, gc, a, b, w = 1000, 10, 3, 7, 0.8
    E = torch.rand(rows, gc, device=DEV, generator=gen) - 0.5
    E[5, a] = E[5, b]  # sign(0) = 0
    g = torch.rand(rows, gc, device=DEV, generator=gen) - 0.5
    m = torch.rand(rows, gc, device=DEV, generator=gen) * 0.1
    v = torch.rand(rows, gc, device=DEV, generator=gen) * 0.01
    # reference: explicit gradient, plain Adam
    d = E[:, a] - E[:, b]
    g2 = g.clone()
    g2[:, a] += torch.sign(d) * (w / rows)
    g2[:, b] -= torch.sign(d) * (w / rows)
    p_ref, m_ref, v_ref = E.clone().view(-1), m.clone().view(-1), v.clone().view(-1)
    ops.adam_step(p_ref, g2.view(-1), m_ref, v_ref, 3, 1e-2, eps=1e-12, zero_grad=True)
    # fused
    part, srow = torch.zeros(64, 16, device=DEV), torch.empty(rows, device=DEV)
    L, P = _lib.lib(), ops._ptr
    _lib.check(L.snerf_tgrid_tv_sign(P(E), C.c_int64(rows), gc, a, b, w, P(part), 64, P(srow), ops._stream()))
    torch.testing.assert_close(part[:, 0].sum() / rows, d.abs().mean(), rtol=1e-5, atol=0)
    p2, g3, m2, v2 = E.clone(), g.clone(), m.clone(), v.clone()
    _lib.check(L.snerf_adam_step_tv(P(p2), P(g3), P(m2), P(v2), C.c_int64(rows), gc, a, b, P(srow), 1e-2, 0.9, 0.999, 1e-12, 3, 1.0, 1, None, ops._stream()))
    torch.testing.assert_close(p2.view(-1), p_ref, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(m2.view(-1), m_ref, rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(v2.view(-1), v_ref, rtol=1e-6, atol=1e-9)
    assert float(g3.abs().max()) == 0.0


def test_fused_training_reduces_loss_and_keeps_parameters_finite():
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R, n_img = 256, 5
    tr = NerfplayerTrainer(_cfg(), R, n_img, device=DEV, warm_up_end=1, seed=1)
    with torch.no_grad():
        for name in ("field.table", "prop0.table", "prop1.table"):
            tr.views[name].uniform_(-0.5, 0.5)
    rays, cams, target, rng = _batch(R, n_img, 5)
    target = target * 0.5
    losses = []
    for _ in range(12):
        tr.train_step(rays, cams, target, rng)
        losses.append(float(tr.loss_dict()["rgb_loss"]))
    assert tr.step == 12 and losses[-1] < losses[1], losses
    assert bool(torch.isfinite(tr.params).all()) and float(tr.grads.abs().max()) == 0.0


@pytest.mark.parametrize("tv", [0.0, 1.0])
def test_optimizer_step_sweeps_every_parameter_exactly_once(tv):
    """with the temporal-TV term off the sweep used to step the leading segments two and three times.  One optimiser step on a
    known gradient must equal ONE plain Adam sweep over the whole flat buffer (plus, with TV on, the folded-in TV gradient)."""
    import dataclasses

    from soccernerfs_amd import ops
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    cfg = dataclasses.replace(_cfg(), temporal_tv_weight=tv)
    tr = NerfplayerTrainer(cfg, 32, 5, device=DEV, seed=1)
    gen = torch.Generator(device=DEV).manual_seed(4)
    tr.params.copy_(torch.rand(tr.n_params, device=DEV, generator=gen) - 0.5)
    tr.grads.copy_(torch.rand(tr.n_params, device=DEV, generator=gen) - 0.5)
    tr.exp_avg.copy_(torch.rand(tr.n_params, device=DEV, generator=gen) * 0.1)
    tr.exp_avg_sq.copy_(torch.rand(tr.n_params, device=DEV, generator=gen) * 0.01)
    tr.step = 700  # past the warm-up: a non-zero learning rate
    p, g, m, v = tr.params.clone(), tr.grads.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone()
    if tv > 0:  # the per-row TV steps the sweep folds in: make them explicit in the reference gradient
        for k, name in enumerate(("field.table", "prop0.table", "prop1.table")):
            tr._srow[k].copy_(torch.rand(tr._srow[k].shape, device=DEV, generator=gen) - 0.5)
            o, n = next((o, n) for nm, _, _, o, n in tr.segments if nm == name)
            ca, cb = tr._tv_cols[k]
            gv = g[o:o + n].view(tr.views[name].shape)
            gv[:, ca] += tr._srow[k]
            gv[:, cb] -= tr._srow[k]
    from soccernerfs_amd.trainer import cosine_lr_factor
    lr = tr.lr * cosine_lr_factor(700, tr.warm_up_end, tr.max_steps, 0.0)
    ops.adam_step(p, g, m, v, 701, lr, eps=tr.adam_eps, zero_grad=True)
    tr.optimizer_step()
    torch.cuda.synchronize()
    assert tr.step == 701 and float(tr.grads.abs().max()) == 0.0
    torch.testing.assert_close(tr.params, p, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(tr.exp_avg, m, rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(tr.exp_avg_sq, v, rtol=1e-6, atol=1e-9)


def test_deterministic_mode_two_runs_bit_identical_and_close_to_the_default():
    """NerfplayerTrainer(deterministic=True): temporal-grid scatters (snerf_tgrid_encode_bwd_fx), MLP weight gradients (snerf_mlp_bwd_fx) and the appearance
    embedding's gradient accumulate 2^50-scaled integers -- two runs of 12 steps from the same seed on the same batches give the same bits (ADVICE r03:
    the float-atomic scatters made this trainer's runs irreproducible); the default float-atomic run stays within float-summation noise of it."""
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R, n_img, steps = 128, 9, 12
    cfg = _cfg()

    def run(det):
        tr = NerfplayerTrainer(cfg, R, n_img, aabb_scale=1.0, device=DEV, seed=5, deterministic=det, warm_up_end=4)
        with torch.no_grad():
            g = torch.Generator().manual_seed(1)
            for name in ("field.table", "prop0.table", "prop1.table"):
                tr.views[name].copy_((torch.rand(tr.views[name].shape, generator=g) * 2 - 1).to(DEV))
        tr.tv_rows = [2, 1, 3]
        for k in range(steps):
            rays, cams, target, rng = _batch(R, n_img, 100 + k)
            tr.train_step(rays, cams, target, rng)
        torch.cuda.synchronize()
        return tr.params.clone(), {k: float(v) for k, v in tr.loss_dict().items()}

    pa, la = run(True)
    pb, lb = run(True)
    assert torch.equal(pa, pb)
    for k in la:
        if k != "temporal_tv_loss":  # its VALUE is summed with float atomics over workgroups (logging only; its gradient is per-row and exact)
            assert la[k] == lb[k], k
    pc, lc = run(False)
    assert bool(torch.isfinite(pa).all())
    d = (pa - pc).abs()
    # Adam this early moves a parameter by ~lr * sign(g): a gradient that is rounding noise around zero may flip its sign between the two summation
    # orders; everything else agrees closely
    assert float((d > 1e-4).float().mean()) < 2e-3, float((d > 1e-4).float().mean())
    assert abs(la["rgb_loss"] - lc["rgb_loss"]) <= 2e-3 * abs(lc["rgb_loss"])


@pytest.mark.parametrize("tv", [1.0, 0.0])
def test_async_field_sweep_gives_the_same_bits(tv):
    """async_field_sweep=True (the field table's optimiser sweep on a side stream right behind its gradient scatter, joined in front of the next forward's
    field level) against the in-order step, both in deterministic mode so that the comparison is bit for bit: 14 steps crossing updated and non-updated
    proposal steps; parameters, both Adam moments and the loss terms of every step are identical, and an evaluation forward right after the last step
    (which must wait for the sweep by itself) renders the same colours."""
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R, n_img, steps = 128, 9, 14
    cfg = _cfg()
    cfg.temporal_tv_weight = tv

    def run(async_sweep):
        tr = NerfplayerTrainer(cfg, R, n_img, aabb_scale=1.0, device=DEV, seed=5, deterministic=True, warm_up_end=4, async_field_sweep=async_sweep)
        with torch.no_grad():
            g = torch.Generator().manual_seed(1)
            for name in ("field.table", "prop0.table", "prop1.table"):
                tr.views[name].copy_((torch.rand(tr.views[name].shape, generator=g) * 2 - 1).to(DEV))
        tr.tv_rows = [2, 1, 3]
        losses = []
        for k in range(steps):
            rays, cams, target, rng = _batch(R, n_img, 100 + k)
            tr.train_step(rays, cams, target, rng)
            losses.append({k_: float(v) for k_, v in tr.loss_dict().items() if k_ != "temporal_tv_loss"})
        rays, cams, target, rng = _batch(R, n_img, 999)
        rgb = tr.forward(rays, None, rng, 1.0, training=False).clone()  # no explicit join: forward() waits for the side stream itself
        tr.synchronize()
        return tr.params.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), losses, rgb

    a, b = run(True), run(False)
    assert a[3] == b[3]
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert torch.equal(a[4], b[4]) and bool(torch.isfinite(a[4]).all())


def test_bf16_operands_close_to_fp32_and_train():
    """NerfplayerTrainer(mlp_operands="bf16"): the decode net and the colour head on bf16 MFMA operands (at a shape the 16-bit kernels are built for).  One
    forward / backward from the same state against the exact-fp32 trainer: colours, loss terms and gradients at operand-rounding distance; twelve training
    steps lower the loss like the fp32 run."""
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModelConfig
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    cfg = NerfplayerNerfactoModelConfig(
        num_levels=16, log2_hashmap_size=12, temporal_dim=16,  # 32 grid features: the preset's decode net 32 -> 64 -> 16 and head 63 -> 64 -> 64 -> 3
        proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 32},
                                {"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 64}],
        num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=16)
    R, n_img = 128, 9
    trs = {}
    for op in ("fp32", "bf16"):
        tr = NerfplayerTrainer(cfg, R, n_img, aabb_scale=1.0, device=DEV, seed=5, warm_up_end=4, mlp_operands=op)
        with torch.no_grad():
            g = torch.Generator().manual_seed(1)
            for name in ("field.table", "prop0.table", "prop1.table"):
                tr.views[name].copy_((torch.rand(tr.views[name].shape, generator=g) * 2 - 1).to(DEV))
        tr.tv_rows = [2, 1, 3]
        trs[op] = tr
    assert trs["bf16"].decode.desc.operands == 1 and trs["bf16"].head.desc.operands == 1 and trs["fp32"].head.desc.operands == 0
    with torch.no_grad():
        trs["bf16"].params.copy_(trs["fp32"].params)
    rays, cams, target, rng = _batch(R, n_img, 100)
    out = {}
    for op, tr in trs.items():
        rgb = tr.forward(rays, cams, rng, 1.0).clone()
        tr.backward(target, rng, proposal_grads=True)
        out[op] = (rgb, {k: float(v) for k, v in tr.loss_dict().items()}, {k: v.clone() for k, v in tr.gviews.items()})
        tr.grads.zero_()
    (ra, la, ga), (rb, lb, gb) = out["fp32"], out["bf16"]
    assert float((ra - rb).abs().max()) <= 2e-2, float((ra - rb).abs().max())
    for k in la:
        assert abs(la[k] - lb[k]) <= 3e-2 * max(abs(la[k]), 1e-6), (k, la[k], lb[k])
    rel = {k: float((ga[k] - gb[k]).norm() / ga[k].norm().clamp_min(1e-20)) for k in ga}
    print("NerfplayerTrainer bf16 vs fp32 operands: max |rgb| deviation", float((ra - rb).abs().max()), {k: round(v, 4) for k, v in rel.items()})
    assert max(rel.values()) <= 0.1, rel
    first, last = {}, {}
    for op, tr in trs.items():
        for k in range(12):
            b = _batch(R, n_img, 100)  # the same batch every step: the loss must fall
            tr.train_step(b[0], b[1], b[2], b[3])
            if k == 0:
                first[op] = float(tr.loss_dict()["rgb_loss"])
        last[op] = float(tr.loss_dict()["rgb_loss"])
    assert last["bf16"] < 0.8 * first["bf16"] and abs(last["bf16"] - last["fp32"]) <= 0.25 * last["fp32"], (first, last)


def test_head_input_kernels_match_the_torch_expressions():
    """snerf_nerfacto_head_input_fwd / _bwd against the ~35 torch ops they replace: the SH block is the bits soccernerfs_amd.sh.sh4_from_unit_dirs gives, the
    copies are exact, the appearance gradient is the per-ray sum added to the cameras' rows (float and fixed-point accumulators)."""
    import ctypes as C

    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.sh import sh4_from_unit_dirs

    L = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    gen = torch.Generator().manual_seed(4)
    R, S, M = 333, 7, 11
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1).to(DEV)
    h = torch.rand(R * S, 16, generator=gen).to(DEV)
    app = torch.rand(M, 32, generator=gen).to(DEV)
    cams = torch.randint(0, M, (R,), generator=gen).to(DEV)
    st = ops._stream()
    hx = torch.full((R * S, 64), 9.0, device=DEV)
    _lib.check(L.snerf_nerfacto_head_input_fwd(p(d), p(h), p(app), p(cams), S, C.c_int64(R), p(hx), st), "fwd")
    v = hx.view(R, S, 64)
    assert torch.equal(v[:, :, 0:16], sh4_from_unit_dirs(d)[:, None, :].expand(R, S, 16))
    assert torch.equal(v[:, :, 16:31], h.view(R, S, 16)[:, :, 1:16]) and torch.equal(v[:, :, 31:63], app[cams][:, None, :].expand(R, S, 32))
    assert bool((v[:, :, 63] == 0).all())
    # evaluation variants: one constant row for every ray / no embedding at all
    avg = app.mean(0, keepdim=True).contiguous()
    _lib.check(L.snerf_nerfacto_head_input_fwd(p(d), p(h), p(avg), None, S, C.c_int64(R), p(hx), st), "fwd avg")
    assert torch.equal(hx.view(R, S, 64)[:, :, 31:63], avg[None].expand(R, S, 32))
    _lib.check(L.snerf_nerfacto_head_input_fwd(p(d), p(h), None, None, S, C.c_int64(R), p(hx), st), "fwd zero")
    assert bool((hx.view(R, S, 64)[:, :, 31:] == 0).all())
    assert L.snerf_nerfacto_head_input_fwd(p(d), p(h), None, p(cams), S, C.c_int64(R), p(hx), st) != 0  # indices without a table
    # backward
    ghx = (torch.rand(R * S, 64, generator=gen) - 0.5).to(DEV)
    gh = torch.full((R * S, 16), 3.0, device=DEV)
    gapp = torch.zeros(M, 32, device=DEV)
    _lib.check(L.snerf_nerfacto_head_input_bwd(p(ghx), p(cams), S, C.c_int64(R), p(gh), p(gapp), None, st), "bwd")
    g3 = ghx.view(R, S, 64)
    assert torch.equal(gh.view(R, S, 16)[:, :, 1:16], g3[:, :, 16:31]) and bool((gh[:, 0] == 3.0).all())
    ref = torch.zeros(M, 32, device=DEV, dtype=torch.float64).index_add_(0, cams, g3[:, :, 31:63].double().sum(1))
    torch.testing.assert_close(gapp.double(), ref, rtol=1e-5, atol=1e-5)
    cells = []
    for _ in range(2):
        fx = torch.zeros(M * 32, dtype=torch.int64, device=DEV)
        _lib.check(L.snerf_nerfacto_head_input_bwd(p(ghx), p(cams), S, C.c_int64(R), p(gh), None, p(fx), st), "bwd fx")
        cells.append(fx)
    assert torch.equal(cells[0], cells[1])
    out = torch.empty(M * 32, device=DEV)
    ops.fx_to_float(cells[0], out)
    torch.testing.assert_close(out.view(M, 32).double(), ref, rtol=1e-5, atol=1e-5)
    assert L.snerf_nerfacto_head_input_bwd(p(ghx), p(cams), S, C.c_int64(R), p(gh), p(gapp), p(cells[0]), st) != 0  # both accumulators
    assert L.snerf_nerfacto_head_input_bwd(p(ghx), None, S, C.c_int64(R), p(gh), None, None, st) == 0             # geometry columns only


@pytest.mark.parametrize("async_sweep", [True, False])
@pytest.mark.parametrize("tv,first_level", [(1.0, 0), (0.0, 0), (1.0, 2)])
def test_tiled_field_backward_is_the_same_step(tv, first_level, async_sweep):
    """tiled_field_backward=True (round 6: the field table's gradient scatter, temporal-TV step and Adam sweep as ONE owner-computes pass over tiles of table
    rows, csrc/tgrid_tiles.hip) against the atomic scatter + dense sweep, from the same state on the same batches and draws.  After ONE step the field
    table's first moment IS the gradient (m = 0.1 g) and the second its square: both equal to summation-order accuracy, every other segment bit for bit
    (nothing else changed); over 14 steps crossing updated and non-updated proposal steps the loss terms stay together (two runs of either form differ by
    the order of their float sums), and an evaluation forward right after the last step waits for the side-stream pass by itself."""
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R, n_img, steps = 128, 9, 14
    cfg = _cfg()
    cfg.temporal_tv_weight = tv

    def make(tiled):
        tr = NerfplayerTrainer(cfg, R, n_img, aabb_scale=1.0, device=DEV, seed=5, warm_up_end=4, async_field_sweep=async_sweep, tiled_field_backward=tiled,
                               tiled_first_level=first_level)
        with torch.no_grad():
            g = torch.Generator().manual_seed(1)
            for name in ("field.table", "prop0.table", "prop1.table"):
                tr.views[name].copy_((torch.rand(tr.views[name].shape, generator=g) * 2 - 1).to(DEV))
        tr.tv_rows = [2, 1, 3]
        return tr

    a, b = make(True), make(False)
    assert a._tiled is not None and b._tiled is None
    assert a._tiled.plan.first_tiled_level == first_level
    a.early_bin = first_level == 0  # the binning pass beside the field forward (default) / behind the decode net's backward with the zero-gradient filter
    off = {name: (o, n) for name, _, _, o, n in a.segments}
    fo, fn = off["field.table"]
    rays, cams, target, rng = _batch(R, n_img, 100)
    for tr in (a, b):
        tr.train_step(rays, cams, target, rng)
        tr.synchronize()
    assert float(a.grads.abs().max()) == 0.0 and float(b.grads.abs().max()) == 0.0  # both leave cleared gradients (the tiled form never wrote the table's)
    ma, mb = a.exp_avg[fo:fo + fn], b.exp_avg[fo:fo + fn]
    assert float(mb.abs().max()) > 0
    torch.testing.assert_close(ma, mb, rtol=1e-4, atol=1e-6 * float(mb.abs().max()))
    assert bool(((ma != 0) == (mb != 0)).all())
    torch.testing.assert_close(a.exp_avg_sq[fo:fo + fn], b.exp_avg_sq[fo:fo + fn], rtol=2e-4, atol=1e-12 * float(mb.abs().max()) ** 2)
    mask = torch.ones(a.n_params, dtype=torch.bool, device=DEV)
    mask[fo:fo + fn] = False
    for x, y in ((a.params, b.params), (a.exp_avg, b.exp_avg), (a.exp_avg_sq, b.exp_avg_sq)):
        # the other segments' kernels are untouched; their float atomics (MLP weight gradients, proposal tables) are order-dependent in both forms
        torch.testing.assert_close(x[mask], y[mask], rtol=1e-3, atol=1e-6)
    # Adam's first step moves every touched parameter by ~lr * sign(g): equal wherever the gradient is not at rounding distance from zero
    moved = (a.params[fo:fo + fn] - b.params[fo:fo + fn]).abs() > 1e-6
    assert float(moved.float().mean()) < 1e-3
    la_all, lb_all = [], []
    for k in range(1, steps):
        rays, cams, target, rng = _batch(R, n_img, 100 + k)
        for tr, acc in ((a, la_all), (b, lb_all)):
            tr.train_step(rays, cams, target, rng)
            acc.append({k_: float(v) for k_, v in tr.loss_dict().items()})
    for la, lb in zip(la_all, lb_all):
        for k_ in lb:
            assert abs(la[k_] - lb[k_]) <= 3e-2 * abs(lb[k_]) + 1e-7, (k_, la[k_], lb[k_])
    rays, cams, target, rng = _batch(R, n_img, 999)
    ra = a.forward(rays, None, rng, 1.0, training=False).clone()  # no explicit join: forward() waits for the side stream itself
    rb = b.forward(rays, None, rng, 1.0, training=False).clone()
    a.synchronize(); b.synchronize()
    assert bool(torch.isfinite(ra).all()) and float((ra - rb).abs().max()) < 5e-2
    assert float((a.params[fo:fo + fn] - b.params[fo:fo + fn]).abs().mean()) < 2e-3


def test_stock_pytorch_standin_and_hip_trainer_take_the_same_steps():
    """oracle/nerfplayer_standin.NerfplayerStandinTrainer (the reference's nerfplayer-nerfacto algorithm in stock PyTorch: the checker that anchors config 4's PSNR)
    against the fused HIP trainer from the same parameters on the same batches, draws and TV rows: every loss term of six consecutive steps (updated proposal
    steps, annealed PDF weights, the cosine warm-up) and the parameters afterwards.  Exact fp32 HIP path; float sums in different orders on the two sides."""
    from oracle.nerfplayer_standin import NerfplayerStandinTrainer
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R, n_img, steps = 128, 9, 6
    cfg = _cfg()
    tr = NerfplayerTrainer(cfg, R, n_img, aabb_scale=1.0, device=DEV, seed=5, warm_up_end=4)
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for name in ("field.table", "prop0.table", "prop1.table"):
            tr.views[name].copy_(((torch.rand(tr.views[name].shape, generator=g) * 2 - 1) * 0.5).to(DEV))
    ref = NerfplayerStandinTrainer(tr, DEV, warm_up_end=4)
    tr.tv_rows = ref.tv_rows = [2, 1, 3]
    for k in range(steps):
        rays, cams, target, rng = _batch(R, n_img, 300 + k)
        rgb_h = tr.train_step(rays, cams, target, rng).clone()
        rgb_r = ref.train_step(rays, cams, target, rng)
        torch.testing.assert_close(rgb_h, rgb_r, rtol=0, atol=2e-4 * (k + 1))
        lh, lr = {k_: float(v) for k_, v in tr.loss_dict().items()}, {k_: float(v) for k_, v in ref.loss_dict().items()}
        assert set(lh) == set(lr)
        for k_ in lr:
            assert abs(lh[k_] - lr[k_]) <= 2e-3 * (k + 1) * abs(lr[k_]) + 1e-8, (k, k_, lh[k_], lr[k_])
    tr.synchronize()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(tr.views["field.table"], ref.grid.table()) < 2e-3
    for i in range(2):
        assert rel(tr.views[f"prop{i}.table"], ref.prop_grid[i].table()) < 2e-3
    for mine, theirs in zip(tr.head.linear_weights(), ref.head_w):
        assert rel(mine, theirs.detach()) < 5e-3
    assert rel(tr.appearance.weight.detach(), ref.appearance.detach()) < 1e-3
    # eval forward: average appearance embedding, no jitter, clamped colours
    rays, cams, target, rng = _batch(R, n_img, 999)
    eh = tr.forward(rays, None, rng, 1.0, training=False).clone()
    er = ref.forward(rays, None, rng, 1.0, training=False)
    torch.testing.assert_close(eh, er, rtol=0, atol=5e-3)


def test_one_launch_ray_kernel_in_train_step_gives_the_same_bits():
    """(r06) train_step runs the nerf level's per-ray work -- get_weights, compositing, MSE backward, distortion loss + gradient, get_weights backward -- as ONE
    launch (snerf_ray_train_fwd_bwd, already pinned bit-identical to the five kernels in tests/test_gpu_render_loss.py) instead of five: deterministic mode, eight
    steps, parameters / moments / every loss term identical to the five-kernel step, rendered colours of the batch identical."""
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R, n_img = 128, 9
    cfg = _cfg()

    def run(fused):
        tr = NerfplayerTrainer(cfg, R, n_img, aabb_scale=1.0, device=DEV, seed=5, deterministic=True, warm_up_end=4)
        tr.fused_ray_loss = fused
        with torch.no_grad():
            g = torch.Generator().manual_seed(1)
            for name in ("field.table", "prop0.table", "prop1.table"):
                tr.views[name].copy_((torch.rand(tr.views[name].shape, generator=g) * 2 - 1).to(DEV))
        tr.tv_rows = [2, 1, 3]
        losses, rgbs = [], []
        for k in range(8):
            rays, cams, target, rng = _batch(R, n_img, 400 + k)
            rgbs.append(tr.train_step(rays, cams, target, rng).clone())
            # (the temporal-TV VALUE is a float-atomic sum of per-workgroup partials in every mode: order-dependent in its last bit, excluded as in the
            # asynchronous-sweep test above; its GRADIENT -- the per-row signs -- is exact and enters the parameters compared below)
            losses.append({k_: float(v) for k_, v in tr.loss_dict().items() if k_ != "temporal_tv_loss"})
        tr.synchronize()
        return tr.params.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), losses, torch.stack(rgbs)

    a, b = run(True), run(False)
    assert a[3] == b[3]
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert torch.equal(a[4], b[4]) and bool(torch.isfinite(a[4]).all())

"""GPU: the fused full-NeRFPlayer trainer (soccernerfs_amd.nerfplayer_full_trainer) against the nerfstudio-shaped autograd model on the same HIP
kernels -- which is itself pinned against the reference's own NerfplayerModel by golden G13 (tests/test_gpu_hashgrid.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(tv=1.0):
    from soccernerfs_amd.nerfplayer import NerfplayerModelConfig

    return NerfplayerModelConfig(
        num_levels=16, log2_hashmap_size=12, temporal_dim=16, temporal_tv_weight=tv,
        proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 32},
                                {"hidden_dim": 16, "temporal_dim": 8, "log2_hashmap_size": 10, "num_levels": 4, "max_res": 64}],
        num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=16)


def _batch(R, seed):
    gen = torch.Generator().manual_seed(seed)
    g = lambda z: z.to(DEV)
    o = g((torch.rand(R, 3, generator=gen) * 2 - 1) * 0.3)
    d = g(torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1))
    times, target = g(torch.rand(R, 1, generator=gen)), g(torch.rand(R, 3, generator=gen))
    rng = {"t_rand": g(torch.rand(R, 1, generator=gen)), "u": [g(torch.rand(R, 1, generator=gen)), g(torch.rand(R, 1, generator=gen))],
           "bg": g(torch.rand(R, 3, generator=gen))}
    return {"origins": o, "directions": d, "times": times}, target, rng


def _pairs(model):
    f = model.field
    pairs = {"field.deform": f.deformation_field.params, "field.hash": f.stationary_field.params, "field.stat_mlp": f.stationary_field_mlp.params,
             "field.newness": f.newness_field.embeddings, "field.decomp": f.decomposition_field.embeddings, "field.decomp_mlp": f.decomposition_mlp.params,
             "field.decode": f.mlp_base_decode.params, "field.head": f.mlp_head.params}
    for i, pn in enumerate(model.proposal_networks):
        pairs[f"prop{i}.table"], pairs[f"prop{i}.mlp"] = pn.encoding.embeddings, pn.linear.params
    return pairs


def _make(R, tv=1.0):
    from soccernerfs_amd.nerfplayer import NerfplayerModel
    from soccernerfs_amd.nerfplayer_full_trainer import NerfplayerFullTrainer
    from soccernerfs_amd.scene_colliders import SceneBox

    cfg = _cfg(tv)
    tr = NerfplayerFullTrainer(cfg, R, aabb_scale=1.0, device=DEV, seed=3)
    gen = torch.Generator(device=DEV).manual_seed(5)
    with torch.no_grad():  # O(1) tables: the 1e-4 initialisation gives a featureless field
        for name in ("field.hash", "field.newness", "field.decomp", "prop0.table", "prop1.table"):
            tr.views[name].copy_(torch.rand(tr.views[name].shape, device=DEV, generator=gen) * 2 - 1)
    model = NerfplayerModel(cfg, SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=4).to(DEV).train()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    pairs = _pairs(model)
    assert set(pairs) == set(tr.views)
    with torch.no_grad():
        for name, p in pairs.items():
            p.copy_(tr.views[name].view(p.shape))
    return cfg, tr, model, pairs


def test_fused_step_equals_autograd_model():
    from soccernerfs_amd.rays import RayBundle

    R = 96
    cfg, tr, model, pairs = _make(R)
    rays, target, rng = _batch(R, 11)
    anneal, rows = 0.4, [2, 1, 3, 0]
    draws = [rng["t_rand"], rng["u"][0], rng["u"][1], rng["bg"]]
    model.set_rand_fn(lambda shape, device=None: draws.pop(0))
    f = model.field
    encs = [f.newness_field, f.decomposition_field] + [p.encoding for p in model.proposal_networks]
    for e in encs:
        e.fuse_tv = False
    model.tv_row_fn = lambda enc: rows[[id(e) for e in encs].index(id(enc))]
    model.proposal_sampler.set_anneal(anneal)
    out = model(RayBundle(origins=rays["origins"], directions=rays["directions"], pixel_area=torch.ones(R, 1, device=DEV),
                          camera_indices=torch.zeros(R, 1, dtype=torch.long, device=DEV), times=rays["times"]))
    ld = model.get_loss_dict(out, {"image": target}, model.get_metrics_dict(out, {"image": target}))
    sum(ld.values()).backward()
    tr.tv_rows = rows
    rgb = tr.forward(rays, rng, anneal)
    tr.backward(target, rng, proposal_grads=True)
    tr.materialize_tv_gradient()
    torch.testing.assert_close(rgb, out["rgb"].detach(), rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(tr.buf["acc"], out["accumulation"].detach()[:, 0], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(tr.rendered_probs(), out["probs"].detach(), rtol=1e-4, atol=2e-6)
    for i in range(3):
        torch.testing.assert_close(tr.buf["eb"][i], out["ray_samples_list"][i]._compact["ebins"], rtol=0, atol=1e-6)
    mine = tr.loss_dict()
    assert set(mine) == set(ld)
    for k in ld:
        torch.testing.assert_close(mine[k], ld[k].detach(), rtol=1e-4, atol=1e-9)
    for name, p in pairs.items():
        g_ref, g = p.grad.reshape(-1), tr.gviews[name].reshape(-1)
        scale = float(g_ref.abs().max())
        assert scale > 0, name
        torch.testing.assert_close(g, g_ref, rtol=2e-3, atol=2e-5 * scale, msg=lambda m: f"{name}: {m}")
    assert tr.launches < 90  # libsnerf launches of forward + backward (the autograd model issues ~285 kernels per step)


@pytest.mark.parametrize("tv", [1.0, 0.0])
def test_training_steps_track_autograd_model_with_torch_adam(tv):
    """Three fused steps (Adam, TV folded into the sweeps, every parameter stepped exactly once) against the autograd model + torch.optim.Adam."""
    from soccernerfs_amd.rays import RayBundle

    R = 64
    cfg, tr, model, pairs = _make(R, tv)
    tr.warm_up_end, tr.max_steps = 1, 30000
    f = model.field
    encs = [f.newness_field, f.decomposition_field] + [p.encoding for p in model.proposal_networks]
    for e in encs:
        e.fuse_tv = False
    rows = [1, 2, 0, 3]
    model.tv_row_fn = lambda enc: rows[[id(e) for e in encs].index(id(enc))]
    tr.tv_rows = rows
    opt = torch.optim.Adam(list(pairs.values()), lr=1e-2, eps=1e-6)
    from soccernerfs_amd.trainer import anneal_value, cosine_lr_factor
    for step in range(3):
        rays, target, rng = _batch(R, 100 + step)
        draws = [rng["t_rand"], rng["u"][0], rng["u"][1], rng["bg"]]
        model.set_rand_fn(lambda shape, device=None: draws.pop(0))
        model.proposal_sampler.set_anneal(anneal_value(step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope))
        for g_ in opt.param_groups:
            g_["lr"] = 1e-2 * cosine_lr_factor(step, 1, 30000, 0.0)
        opt.zero_grad(set_to_none=False)
        out = model(RayBundle(origins=rays["origins"], directions=rays["directions"], pixel_area=torch.ones(R, 1, device=DEV),
                              camera_indices=torch.zeros(R, 1, dtype=torch.long, device=DEV), times=rays["times"]))
        ld = model.get_loss_dict(out, {"image": target}, model.get_metrics_dict(out, {"image": target}))
        sum(ld.values()).backward()
        opt.step()
        rgb = tr.train_step(rays, target, rng)
        torch.testing.assert_close(rgb, out["rgb"].detach(), rtol=5e-3, atol=1e-4)
    assert tr.step == 3 and float(tr.grads.abs().max()) == 0.0
    worst = {}
    for name, p in pairs.items():
        a, b = tr.views[name].reshape(-1), p.detach().reshape(-1)
        # Adam's first steps move a parameter by ~lr * sign(g): where a gradient is accumulation-order noise around zero (the deformation MLP
        # sees only the hash grid's atomically accumulated coordinate gradient) the step can flip; count such elements instead of bounding all
        worst[name] = float(((a - b).abs() > 2e-4 + 1e-3 * b.abs()).float().mean())
        assert float((a - b).abs().mean()) < 2e-4, (name, float((a - b).abs().mean()))
    assert max(worst.values()) < 8e-2, worst  # measured 0-4 % on field.deform run to run (atomic order), 0 on every other tensor


WINDOWS = [(5, 14), (14, 23), (23, 32), (32, 41), (41, 50)]  # steps whose means are compared with the reference's run (G13b)
# allowed ratio of this trainer's window mean to the reference's (either way).  Measured over 10 runs (tools/g13b_spread.py, round 3): rgb 0.98-1.04,
# interlevel 0.63-1.92, distortion 0.61-1.13 (two families of runs: the branch taken at step 4), temporal TV 1.00, probability loss 0.93-1.36
WINDOW_FACTOR = {"rgb_loss": 1.12, "interlevel_loss": 2.5, "distortion_loss": 1.9, "temporal_tv_loss": 1.05, "prob_loss": 1.6}


N_RUNS = 4  # repeated runs of the 50 steps: the window means are asserted on their MEAN (the original bounds are the contract)


def test_fifty_training_steps_track_the_reference_models_own_run():
    """G13b (oracle/gen_golden_nerfplayer_dynamics.py): 50 optimiser steps of the REFERENCE's NerfplayerModel on the CPU, run as its Trainer
    runs them (set_anneal -> forward -> losses -> backward -> Adam per group -> cosine schedule -> step_cb), every draw stored.  The fused
    HIP trainer, started from the same (G13) parameters and fed the same batch, draws and TV rows, must follow the reference's per-step loss
    terms, PSNR and mean rendered decomposition probabilities -- including the drift towards the static branch (0.20 -> 0.85 within 50
    steps in the reference itself: the probability regulariser at work, not a defect of this build).

    Steps 0-4 are compared value by value in EVERY run.  Beyond them the run is chaotic (float-atomic order: repeated runs of this trainer differ from
    each other by 10-40 % per step, tools/g13b_spread.py), so what is compared is the course of the run -- window means -- and, since round 5, on the
    MEAN over N_RUNS repeated runs with the bounds this test had before single outlier runs made round 4 widen them (probabilities within 0.12, the
    run ending more than 0.7 static): averaging removes the run-to-run part of the deviation instead of the bound giving way to it."""
    from tests.conftest import load_golden
    from tests.test_gpu_hashgrid import _full_model
    from soccernerfs_amd.nerfplayer_full_trainer import NerfplayerFullTrainer

    g, gb = load_golden("g13_nerfplayer_full"), load_golden("g13b_nerfplayer_dynamics")
    model, _ = _full_model(g)
    R = int(g["R"])
    pairs = _pairs(model)
    t = lambda k: g[k].to(DEV).contiguous()
    rays = {"origins": t("origins"), "directions": t("directions"), "times": t("times")}
    target = t("target")
    steps = int(gb["steps"])
    keys = ["rgb_loss", "interlevel_loss", "distortion_loss", "temporal_tv_loss", "prob_loss"]

    def one_run():
        tr = NerfplayerFullTrainer(model.config, R, aabb_scale=1.0, device=DEV, lr=float(gb["lr0"]), adam_eps=float(gb["eps"]), warm_up_end=int(gb["warm_up_end"]),
                                   max_steps=int(gb["max_steps"]), seed=0)
        assert set(pairs) == set(tr.views)
        with torch.no_grad():
            for name, p in pairs.items():
                tr.views[name].copy_(p.detach().reshape(tr.views[name].shape))
        history = {k: [] for k in keys + ["probs"]}
        for step in range(steps):
            rng = {"t_rand": gb["t_rand"][step].to(DEV), "u": [gb["u0"][step].to(DEV), gb["u1"][step].to(DEV)], "bg": gb["bg"][step].to(DEV)}
            tr.tv_rows = [int(x) for x in gb["tv_rows"][step]]
            tr.train_step(rays, target, rng)
            ld = tr.loss_dict()
            assert set(ld) == set(keys)
            # early steps: fp32 agreement, value by value.  Later the two runs are 10-50 Adam steps apart from a common start: each step moves every
            # parameter by ~lr whatever the gradient's size, so rounding-level differences (atomics order) grow, terms like the interlevel loss rise by
            # four orders of magnitude within ten steps, and now and then a run takes a visibly different branch from step 4 on
            for k in keys:
                curve = gb["loss_" + k]
                ref, got = float(curve[step]), float(ld[k])
                floor = max(1e-2 * float(curve.abs().max()), 1e-7)  # a term four orders below its later size is noise
                if step < 5:
                    rtol = 2e-3 if step == 0 else 5e-2
                    assert abs(got - ref) <= rtol * max(abs(ref), floor), (step, k, got, ref)
                history[k].append(got)
            probs = tr.rendered_probs().mean(0).cpu()
            dp = float((probs - gb["probs_mean"][step]).abs().max())
            psnr = float(-10.0 * torch.log10(ld["rgb_loss"]))
            if step < 5:  # value by value while the runs are still together
                assert dp <= (1e-4 if step == 0 else 2e-2), (step, probs, gb["probs_mean"][step])
                assert abs(psnr - float(gb["psnr"][step])) <= (1e-2 if step == 0 else 0.1), (step, psnr, float(gb["psnr"][step]))
            history["probs"].append(probs)
        return history

    runs = [one_run() for _ in range(N_RUNS)]
    worst = {k: 0.0 for k in keys + ["probs"]}
    # the course of the run, window by window: the mean over the runs of each term's window mean against the reference's
    for k in keys:
        curve = gb["loss_" + k]
        floor = max(1e-2 * float(curve.abs().max()), 1e-7)
        fac = WINDOW_FACTOR[k]
        for lo, hi in WINDOWS:
            ref = float(curve[lo:hi].mean())
            got = sum(sum(h[k][lo:hi]) / (hi - lo) for h in runs) / N_RUNS
            assert ref / fac - floor <= got <= ref * fac + floor, (k, (lo, hi), got, ref)
            worst[k] = max(worst[k], abs(got - ref) / max(abs(ref), floor))
    # the rendered decomposition probabilities drift towards "static" as the reference's do (0.20 -> 0.85 in 50 steps): mean over the runs of the window
    # means within 0.12 of the reference's (single runs: up to 0.125 away, ending between 0.67 and 0.88 static)
    for lo, hi in WINDOWS:
        got = torch.stack([torch.stack(h["probs"][lo:hi]).mean(0) for h in runs]).mean(0)
        ref = gb["probs_mean"][lo:hi].mean(0)
        worst["probs"] = max(worst["probs"], float((got - ref).abs().max()))
        assert float((got - ref).abs().max()) <= 0.12, ((lo, hi), got, ref)
    # the runs end where the reference's ends: loss lower than at the start, decomposition mostly static (the reference ends at 0.85)
    end_rgb = sum(h["rgb_loss"][-1] for h in runs) / N_RUNS
    end_static = sum(float(h["probs"][-1][0]) for h in runs) / N_RUNS
    assert end_rgb < 0.8 * float(gb["loss_rgb_loss"][0]) and end_static > 0.7, (end_rgb, end_static)
    print(f"G13b: worst deviation of the mean (over {N_RUNS} runs) window means from the reference's run (steps 5..49):", {k: round(v, 4) for k, v in worst.items()},
          "end static probability per run:", [round(float(h["probs"][-1][0]), 3) for h in runs])


@pytest.mark.parametrize("async_sweeps", [True, False])
def test_tiled_table_backward_is_the_same_step(async_sweeps):
    """tiled_table_backward=True (round 6: the newness and decomposition tables' gradient scatter on the DEFORMED positions, temporal-TV step and Adam sweep as
    one owner-computes pass each, csrc/tgrid_tiles.hip) against the atomic scatter + dense sweeps from the same state, batch and draws: after one step both
    tables' first moments (= 0.1 x gradient, TV term included) and second moments agree to summation-order accuracy, the same entries are touched, the
    gradient buffer is left cleared; six more steps keep the loss terms together and an evaluation forward waits for the side-stream pass by itself."""
    from soccernerfs_amd.nerfplayer_full_trainer import NerfplayerFullTrainer

    R = 96
    cfg = _cfg(1.0)

    def make(tiled):
        tr = NerfplayerFullTrainer(cfg, R, aabb_scale=1.0, device=DEV, seed=3, async_table_sweeps=async_sweeps, tiled_table_backward=tiled, tiled_hash_backward=tiled)
        gen = torch.Generator(device=DEV).manual_seed(5)
        with torch.no_grad():
            for name in ("field.hash", "field.newness", "field.decomp", "prop0.table", "prop1.table"):
                tr.views[name].copy_(torch.rand(tr.views[name].shape, device=DEV, generator=gen) * 2 - 1)
        tr.tv_rows = [2, 1, 3, 0]
        return tr

    a, b = make(True), make(False)
    assert a._tiled is not None and b._tiled is None
    a.early_bin = async_sweeps  # binning beside the forward (default) / on the caller's stream in front of the tile passes; one shared pass either way
    assert a._tiled[1].records.data_ptr() == a._tiled[0].records.data_ptr()
    rays, target, rng = _batch(R, 11)
    for tr in (a, b):
        tr.train_step(rays, target, rng)
        tr.synchronize()
    assert float(a.grads.abs().max()) == 0.0 and float(b.grads.abs().max()) == 0.0
    off = {name: (o, n) for name, _, _, o, n in a.segments}
    assert a._tiled_hash is not None
    for name in ("field.newness", "field.decomp", "field.hash"):  # the three tables the owner-computes passes step
        o, n = off[name]
        ma, mb = a.exp_avg[o:o + n], b.exp_avg[o:o + n]
        scale = float(mb.abs().max())
        assert scale > 0
        torch.testing.assert_close(ma, mb, rtol=1e-4, atol=2e-6 * scale)
        assert float(((ma != 0) != (mb != 0)).float().mean()) < 1e-5  # (a sum that cancels to exactly 0 in one association order only)
        torch.testing.assert_close(a.exp_avg_sq[o:o + n], b.exp_avg_sq[o:o + n], rtol=4e-4, atol=1e-11 * scale * scale)
    la_all, lb_all = [], []
    for k in range(6):
        rays, target, rng = _batch(R, 20 + k)
        for tr, acc in ((a, la_all), (b, lb_all)):
            tr.train_step(rays, target, rng)
            acc.append({k_: float(v) for k_, v in tr.loss_dict().items()})
    for la, lb in zip(la_all[:3], lb_all[:3]):  # (this model's trajectories separate within ~5 steps whatever the arithmetic: DESIGN section 5)
        for k_ in lb:
            assert abs(la[k_] - lb[k_]) <= 5e-2 * abs(lb[k_]) + 1e-6, (k_, la[k_], lb[k_])
    rays, target, rng = _batch(R, 99)
    ra = a.forward(rays, None, 1.0, training=False).clone()
    a.synchronize(); b.synchronize()
    assert bool(torch.isfinite(ra).all()) and bool(torch.isfinite(a.params).all())

"""G13b: TRAINING DYNAMICS of the reference's own full NeRFPlayer model -- 50 optimiser steps of NS/models/nerfplayer.py on the CPU, run as
the reference's Trainer runs them (NS/engine/trainer.py:383-412: BEFORE_TRAIN_ITERATION set_anneal -> forward -> get_metrics_dict ->
get_loss_dict -> backward -> one torch.optim.Adam per parameter group ("proposal_networks", "fields": lr 1e-2, eps 1e-6,
method_configs.py:597-606) -> cosine schedule (schedulers.py:126-141) -> AFTER_TRAIN_ITERATION step_cb), every random draw replaced by a
stored tensor (sampler jitter, background colour, the temporal-TV rows of the four grids).

G13 pins the model's WIRING on one batch; this fixture pins what the wiring does over time: the per-step loss dict, PSNR and the mean rendered
decomposition probabilities (static / deformable / new).  Same tiny configuration, same batch and same initial parameters as G13
(oracle/gen_golden_nerfplayer_full.py; checked against the committed fixture below), warm-up shortened to 2 steps so that the learning
rate is 1e-2 for the whole run.

    python oracle/gen_golden_nerfplayer_dynamics.py        # build container only; writes tests/golden/g13b_nerfplayer_dynamics.npz

How far can ANY other arithmetic follow this run value by value?  The reference's own run answers: SNERF_G13B_THREADS=<n> (torch.set_num_threads: another
summation order inside ATen's reductions and GEMMs) and / or SNERF_G13B_PERTURB=<eps> (every initial parameter multiplied by 1 + eps * (+-1), eps ~ 1e-7 =
one fp32 rounding) with SNERF_G13B_OUT=<path> write a second run elsewhere; tools/g13b_reference_spread.py compares it with the committed one
(profiles/r05_g13b_reference_spread.json).

TEST INFRASTRUCTURE ONLY (header as oracle/_refimport.py)."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_nerfplayer_full as G13  # noqa: E402  (imports the reference and its temporal-grid backend)
from oracle.gen_golden import RandQueue, npy  # noqa: E402

NP = G13.NP
STEPS, WARM_UP_END, MAX_STEPS, LR, EPS = 50, 2, 30000, 1e-2, 1e-6


def cosine(step):  # CosineDecayScheduler (NS/engine/schedulers.py:126-141), alpha 0
    if step < WARM_UP_END:
        return step / WARM_UP_END
    return (math.cos(math.pi * (step - WARM_UP_END) / (MAX_STEPS - WARM_UP_END)) + 1.0) * 0.5


def main():
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox

    # ---- exactly G13's model, parameters and batch (same seeds, same call order) ----
    torch.manual_seed(11)
    gen = torch.Generator().manual_seed(11)
    cfg = NP.NerfplayerModelConfig(**G13.CFG)
    model = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=G13.NUM_IMAGES)
    model.train()
    f = model.field
    with torch.no_grad():
        for enc in [f.newness_field, f.decomposition_field] + [p.encoding for p in model.proposal_networks]:
            enc.embeddings.copy_(torch.rand(enc.embeddings.shape, generator=gen) * 2 - 1)
        f.stationary_field.params.copy_(torch.rand(f.stationary_field.params.shape, generator=gen) * 2 - 1)
        for l in f.deformation_field.layers:
            l.weight.mul_(1.5)
    if os.environ.get("SNERF_G13B_THREADS"):
        torch.set_num_threads(int(os.environ["SNERF_G13B_THREADS"]))
    R = 20
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.4
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    times = torch.rand(R, 1, generator=gen)
    cams = torch.randint(0, G13.NUM_IMAGES, (R, 1), generator=gen)
    target = torch.rand(R, 3, generator=gen)
    g13 = np.load(os.path.join(ROOT, "tests", "golden", "g13_nerfplayer_full.npz"))
    for name, p in model.named_parameters():
        if p.requires_grad and p.numel():
            assert np.array_equal(g13["param_" + name], p.detach().numpy()), name  # the run starts from the committed G13 parameters
    assert np.array_equal(g13["origins"], o.numpy()) and np.array_equal(g13["target"], target.numpy())
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1), camera_indices=cams, times=times)
    eps_p = float(os.environ.get("SNERF_G13B_PERTURB", "0"))
    if eps_p:  # AFTER the check against G13: a one-rounding perturbation of the start, to measure how fast the reference's own run forgets it
        genp = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for p_ in model.parameters():
                if p_.requires_grad and p_.numel():
                    p_.mul_(1.0 + eps_p * (torch.randint(0, 2, p_.shape, generator=genp).float() * 2 - 1))

    groups = model.get_param_groups()  # {"proposal_networks": [...], "fields": [...]}
    opts = {k: torch.optim.Adam(v, lr=LR, eps=EPS) for k, v in groups.items()}
    N, slope = cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope
    gen2 = torch.Generator().manual_seed(1311)
    n_rows = [len(e.index_list) for e in [f.newness_field, f.decomposition_field] + [p.encoding for p in model.proposal_networks]]
    rec = {k: [] for k in ("t_rand", "u0", "u1", "bg", "tv_rows", "psnr", "probs_mean", "lr", "updated")}
    losses = {}
    orig_randint = torch.randint
    for step in range(STEPS):
        # BEFORE_TRAIN_ITERATION: set_anneal (nerfacto.py:243-248)
        frac = float(np.clip(step / N, 0, 1))
        model.proposal_sampler.set_anneal((slope * frac) / ((slope - 1) * frac + 1))
        draws = [torch.rand(R, 1, generator=gen2), torch.rand(R, 1, generator=gen2), torch.rand(R, 1, generator=gen2), torch.rand(R, 3, generator=gen2)]
        rows = [int(torch.randint(0, n, (1,), generator=gen2)) for n in n_rows]  # order of get_temporal_tv_loss calls, nerfplayer.py:325-329
        queue = list(rows)
        torch.randint = lambda *a, **k: torch.tensor([queue.pop(0)])
        lr = LR * cosine(step)
        for opt in opts.values():
            for grp in opt.param_groups:
                grp["lr"] = lr
            opt.zero_grad()
        ps = model.proposal_sampler
        rec["updated"].append(int(ps._steps_since_update > ps.update_sched(ps._step) or ps._step < 10))
        try:
            with RandQueue([x.clone() for x in draws]):
                out = model(rb)
            metrics = model.get_metrics_dict(out, {"image": target})
            loss_dict = model.get_loss_dict(out, {"image": target}, metrics)
        finally:
            torch.randint = orig_randint
        assert not queue
        sum(loss_dict.values()).backward()
        for opt in opts.values():
            opt.step()
        ps.step_cb(step)  # AFTER_TRAIN_ITERATION
        for k, v in zip(("t_rand", "u0", "u1", "bg"), draws):
            rec[k].append(v)
        rec["tv_rows"].append(torch.tensor(rows))
        rec["psnr"].append(metrics["psnr"].detach())
        rec["probs_mean"].append(out["probs"].detach().mean(0))
        rec["lr"].append(torch.tensor(lr))
        for k, v in loss_dict.items():
            losses.setdefault(k, []).append(v.detach())
        if step % 10 == 0 or step == STEPS - 1:
            print(step, {k: round(float(v), 6) for k, v in loss_dict.items()}, "probs", [round(float(x), 4) for x in out["probs"].mean(0)])
    g = {"steps": STEPS, "warm_up_end": WARM_UP_END, "max_steps": MAX_STEPS, "lr0": LR, "eps": EPS}
    for k, v in rec.items():
        g[k] = torch.stack([torch.as_tensor(x) for x in v])
    for k, v in losses.items():
        g["loss_" + k] = torch.stack(v)
    for name, p in model.named_parameters():
        if p.requires_grad and p.numel():
            g["psum_" + name] = p.detach().double().sum()
            g["pabs_" + name] = p.detach().double().abs().sum()
    path = os.environ.get("SNERF_G13B_OUT") or os.path.join(ROOT, "tests", "golden", "g13b_nerfplayer_dynamics.npz")
    if os.environ.get("SNERF_G13B_THREADS") or eps_p:
        assert os.environ.get("SNERF_G13B_OUT"), "a run with another thread count / a perturbed start must not overwrite the committed fixture"
    np.savez_compressed(path, **{k: npy(v) for k, v in g.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

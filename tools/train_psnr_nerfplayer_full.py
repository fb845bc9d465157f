#!/usr/bin/env python3
"""Convergence run of the FULL NeRFPlayer (`nerfplayer` preset: deformation MLP + static hash grid + newness / decomposition temporal grids)
through the nerfstudio-shaped model on the HIP ops, on the synthetic multi-view clip (30 training cameras + 6 held out, as
tools/train_psnr_nerfplayer.py).  Reports PSNR / SSIM on training views and held-out cameras and the mean rendered decomposition
probabilities (static, deform, new).  Not the bench line.

    python tools/train_psnr_nerfplayer_full.py --steps 8000 --out gpurun_out/psnr_nerfplayer_full.json
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic
from soccernerfs_amd.metrics import psnr, structural_similarity_index_measure as ssim
from soccernerfs_amd.nerfplayer import NerfplayerModel, NerfplayerModelConfig
from soccernerfs_amd.optimizers import FusedAdam
from soccernerfs_amd.rays import RayBundle
from soccernerfs_amd.scene_colliders import SceneBox
from soccernerfs_amd.trainer import cosine_lr_factor


def bundle(rays, cam_idx):
    return RayBundle(origins=rays["origins"], directions=rays["directions"], pixel_area=rays["pixel_area"], camera_indices=cam_idx,
                     times=rays["times"], metadata={"directions_norm": rays["directions_norm"]})


@torch.no_grad()
def evaluate(model, data, n_images, R=8192):
    imgs = data["images"]; M, H, W = imgs.shape[:3]
    ys, xs = torch.meshgrid(torch.arange(H, device=imgs.device), torch.arange(W, device=imgs.device), indexing="ij")
    ps, ss, probs = [], [], []
    model.eval()
    for m in torch.linspace(0, M - 1, n_images).long().tolist():
        idx = torch.stack([torch.full_like(ys, m), ys, xs], -1).reshape(-1, 3)
        out, pr = [], []
        for i in range(0, idx.shape[0], R):
            rays = ops.generate_rays(idx[i:i + R].contiguous(), data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"])
            o = model(bundle(rays, idx[i:i + R, :1].contiguous()))
            out.append(o["rgb"]); pr.append(o["probs"])
        img = torch.cat(out).view(H, W, 3)
        gt = imgs[m].float() / 255.0
        chw = lambda t: t.permute(2, 0, 1)[None]
        ps.append(float(psnr(chw(img), chw(gt)))); ss.append(float(ssim(chw(gt), chw(img))))
        probs.append(torch.cat(pr).mean(0).tolist())
    model.train()
    mean = lambda v: sum(v) / len(v)
    return mean(ps), mean(ss), [mean([p[k] for p in probs]) for k in range(3)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--out", default="gpurun_out/psnr_nerfplayer_full.json")
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--frames", type=int, default=25)
    args = ap.parse_args()
    dev = torch.device("cuda:0"); torch.manual_seed(20231029)
    R = 4096
    Wd, Hd = args.width, args.width * 9 // 16
    cams = synthetic.make_cameras(36, Wd, Hd)
    times = synthetic.frame_times(100, 100 // args.frames)
    train = synthetic.render_dataset(cams, times, list(range(30)), dev, chunk_rows=Hd)
    held = synthetic.render_dataset(cams, times, list(range(30, 36)), dev, chunk_rows=Hd)
    M, H, W = train["images"].shape[:3]
    model = NerfplayerModel(NerfplayerModelConfig(), SceneBox(aabb=torch.tensor([[-1.5] * 3, [1.5] * 3])), num_train_data=M).to(dev).train()
    model.scene_box.aabb = model.scene_box.aabb.to(dev)
    params = [p for g in model.get_param_groups().values() for p in g if p.requires_grad]
    encoders = [model.field.newness_field, model.field.decomposition_field] + [p.encoding for p in model.proposal_networks]
    opt = FusedAdam(params, lr=1e-2, eps=1e-6, encoders=encoders)  # method_configs.py:601-610
    cbs = model.get_training_callbacks()
    log = {"config": f"nerfplayer preset (full NeRFPlayer), nerfstudio-shaped model on the HIP ops, synthetic clip ({M} training images {W}x{H}, 6 cameras held out)",
           "evals": []}
    t0 = time.time()
    for step in range(args.steps):
        for g in opt.param_groups:
            g["lr"] = 1e-2 * cosine_lr_factor(step, 512, args.steps, 0.0)
        for where, fn in cbs:
            if where == "before":
                fn(step)
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, train["images"])
        rays = ops.generate_rays(idx, train["fx"], train["fy"], train["cx"], train["cy"], train["c2w"], train["times"])
        out = model(bundle(rays, idx[:, :1].contiguous()))
        md = model.get_metrics_dict(out, {"image": target})
        ld = model.get_loss_dict(out, {"image": target}, md)
        sum(ld.values()).backward()
        opt.step()
        for where, fn in cbs:
            if where == "after":
                fn(step)
        if step % 1000 == 999:
            print(f"step {step + 1}: " + "  ".join(f"{k} {float(v.detach()):.3e}" for k, v in ld.items()), flush=True)
    torch.cuda.synchronize()
    log["train_seconds"] = time.time() - t0
    finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    p_tr, s_tr, pr_tr = evaluate(model, train, 4)
    p_he, s_he, pr_he = evaluate(model, held, 6)
    log["evals"].append({"step": args.steps, "psnr_train_views_mean": p_tr, "ssim_train_views_mean": s_tr, "psnr_heldout_mean": p_he, "ssim_heldout_mean": s_he,
                         "mean_probs_static_deform_new_train_views": pr_tr, "all_parameters_finite": finite})
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(log, open(args.out, "w"), indent=1)
    print(json.dumps(log["evals"][-1]), f"train {log['train_seconds']:.0f} s")


if __name__ == "__main__":
    main()

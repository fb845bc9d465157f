#!/usr/bin/env python3
"""Convergence run of the FULL NeRFPlayer (`nerfplayer` preset: deformation MLP + static hash grid + newness / decomposition temporal grids)
through the fused flat-buffer trainer (soccernerfs_amd.nerfplayer_full_trainer) on the synthetic multi-view clip: 36 training cameras
(the stadium-style arc), evaluation on 3 evaluation-only cameras between training cameras (interpolated views) and on 4 training
images.  Reports PSNR / SSIM, the step time, and the rendered decomposition probabilities (static, deforming, new) separately on DYNAMIC
pixels (temporal-difference mask of the clip) and on static ones -- the decomposition is only expected to leave "static" where something
moves, and the preset's regulariser (prob_reg_loss_mult = 0.1, NS/models/nerfplayer.py:336-341) prices the other two branches.

    python tools/train_psnr_nerfplayer_full.py --steps 8000 --out gpurun_out/psnr_nerfplayer_full.json
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.metrics import psnr, structural_similarity_index_measure as ssim  # noqa: E402
from soccernerfs_amd.nerfplayer import NerfplayerModelConfig  # noqa: E402
from soccernerfs_amd.nerfplayer_full_trainer import NerfplayerFullTrainer  # noqa: E402
from soccernerfs_amd.pixel_samplers import compute_ist  # noqa: E402


@torch.no_grad()
def evaluate(tr, data, image_ids, dyn_maps=None):
    imgs = data["images"]
    H, W = imgs.shape[1:3]
    R = tr.R
    ys, xs = torch.meshgrid(torch.arange(H, device=imgs.device), torch.arange(W, device=imgs.device), indexing="ij")
    ps, ss = [], []
    acc = {"dynamic": torch.zeros(4, device=imgs.device), "static": torch.zeros(4, device=imgs.device)}  # sum of probs (3) + pixel count
    for m in image_ids:
        idx = torch.stack([torch.full_like(ys, m), ys, xs], -1).reshape(-1, 3)
        n = idx.shape[0]
        out, prob = torch.empty(n, 3, device=imgs.device), torch.empty(n, 3, device=imgs.device)
        for i in range(0, n, R):
            chunk = idx[i:i + R]
            k = chunk.shape[0]
            if k < R:  # the trainer's buffers hold exactly R rays: pad the last chunk
                chunk = torch.cat([chunk, chunk[-1:].expand(R - k, 3)])
            rays = ops.generate_rays(chunk.contiguous(), data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"])
            rgb = tr.forward(rays, None, 1.0, training=False)
            out[i:i + k] = rgb[:k]
            prob[i:i + k] = tr.rendered_probs()[:k]
        gt = imgs[m].float() / 255.0
        chw = lambda t: t.view(H, W, 3).permute(2, 0, 1)[None]
        ps.append(float(psnr(chw(out), chw(gt.reshape(-1, 3)))))
        ss.append(float(ssim(chw(gt.reshape(-1, 3)), chw(out))))
        if dyn_maps is not None:
            dyn = dyn_maps[m].reshape(-1) > 0
            for name, mask in (("dynamic", dyn), ("static", ~dyn)):
                acc[name][:3] += prob[mask].sum(0)
                acc[name][3] += mask.sum()
    res = {"psnr_mean": sum(ps) / len(ps), "ssim_mean": sum(ss) / len(ss), "images": len(ps)}
    if dyn_maps is not None:
        for name in acc:
            res[f"mean_rendered_probs_static_deform_new_on_{name}_pixels"] = (acc[name][:3] / acc[name][3].clamp_min(1)).tolist()
            res[f"{name}_pixel_fraction"] = float(acc[name][3] / sum(a[3] for a in acc.values()))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--out", default="gpurun_out/psnr_nerfplayer_full.json")
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--prob-reg", type=float, default=0.1, help="prob_reg_loss_mult (preset: 0.1)")
    ap.add_argument("--seed", type=int, default=20231029)
    ap.add_argument("--mlp-operands", default="fp32", choices=["fp32", "bf16"], help="NerfplayerFullTrainer(mlp_operands=...)")
    ap.add_argument("--async-sweeps", action="store_true", help="NerfplayerFullTrainer(async_table_sweeps=True)")
    ap.add_argument("--tiled", action="store_true", help="NerfplayerFullTrainer(tiled_table_backward=True) (round 6)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    R = 4096
    Wd, Hd = args.width, args.width * 9 // 16
    cams = synthetic.make_cameras(36, Wd, Hd)
    times = synthetic.frame_times(100, 100 // args.frames)
    train = synthetic.render_dataset(cams, times, list(range(36)), dev, chunk_rows=Hd)
    novel = synthetic.render_dataset(synthetic.make_novel_cameras(3, Wd, Hd, n_train_cams=36), times, [0, 1, 2], dev, chunk_rows=Hd)
    M, H, W = train["images"].shape[:3]
    cfg = NerfplayerModelConfig(prob_reg_loss_mult=args.prob_reg)
    tr = NerfplayerFullTrainer(cfg, R, aabb_scale=1.5, device=dev, max_steps=args.steps, seed=args.seed, mlp_operands=args.mlp_operands,
                               async_table_sweeps=args.async_sweeps, tiled_table_backward=args.tiled)
    log = {"config": f"nerfplayer preset (full NeRFPlayer), fused flat-buffer trainer, synthetic clip ({M} training images {W}x{H} from 36 cameras; "
                     f"3 evaluation-only cameras between them), prob_reg_loss_mult {args.prob_reg}", "params": int(tr.n_params), "mlp_operands": args.mlp_operands,
           "async_table_sweeps": bool(args.async_sweeps), "evals": []}
    torch.cuda.synchronize()
    t0 = time.time()
    for step in range(args.steps):
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, train["images"])
        rays = ops.generate_rays(idx, train["fx"], train["fy"], train["cx"], train["cy"], train["c2w"], train["times"])
        tr.train_step(rays, target)
        if step % 1000 == 999:
            torch.cuda.synchronize()
            ld = {k: float(v) for k, v in tr.loss_dict().items()}
            print(f"step {step + 1}: {(time.time() - t0) / (step + 1) * 1e3:.2f} ms/step  " + "  ".join(f"{k} {v:.3e}" for k, v in ld.items()), flush=True)
    torch.cuda.synchronize()
    log["train_seconds"] = time.time() - t0
    log["ms_per_step"] = log["train_seconds"] / args.steps * 1e3
    log["rays_per_s"] = R * args.steps / log["train_seconds"]
    log["launches_per_step"] = tr.launches
    log["all_parameters_finite"] = bool(torch.isfinite(tr.params).all())
    dyn = compute_ist(novel["images"], novel["cam_id"], novel["times"], ist_range=1.0)
    ev = {"step": args.steps,
          "novel": evaluate(tr, novel, list(range(0, novel["images"].shape[0], max(1, novel["images"].shape[0] // 12))), dyn),
          "train": evaluate(tr, train, torch.linspace(0, M - 1, 4).long().tolist())}
    log["evals"].append(ev)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(log, open(args.out, "w"), indent=1)
    print(json.dumps(ev), f"train {log['train_seconds']:.0f} s, {log['ms_per_step']:.2f} ms/step, {tr.launches} launches/step")


if __name__ == "__main__":
    main()

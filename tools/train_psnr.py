#!/usr/bin/env python3
"""PSNR@N-iterations run of the k-planes preset on the synthetic Broadcast-style scene (BASELINE.json metric, 2nd half).

Trains with the fused HIP trainer exactly as the reference schedules it (4096 rays/step, uniform pixels for the first
`iters_to_start_is` = 2000 steps, then 15 % IST importance rays; proposal-weight annealing; proposal updates every n steps;
Adam + cosine schedule), then renders the held-out camera (full images, eval-mode sampler, 'last_sample' background, clamp) and
reports PSNR = 10 log10(1/MSE) per image and averaged (NS/models/kplanes.py:291,472; NS/pipelines/base_pipeline.py:323-362).

    python tools/train_psnr.py --steps 30000 --out gpurun_out/psnr_r01.json
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler, compute_ist  # noqa: E402
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer  # noqa: E402


@torch.no_grad()
def eval_psnr(trainer, data, n_images, anneal):
    imgs = data["images"]
    M, H, W = imgs.shape[:3]
    pick = torch.linspace(0, M - 1, n_images).long().tolist()
    R = trainer.R
    ys, xs = torch.meshgrid(torch.arange(H, device=imgs.device), torch.arange(W, device=imgs.device), indexing="ij")
    from soccernerfs_amd.metrics import structural_similarity_index_measure as ssim_fn

    psnrs, ssims = [], []
    for m in pick:
        idx = torch.stack([torch.full_like(ys, m), ys, xs], -1).reshape(-1, 3)
        out = torch.empty(H * W, 3, device=imgs.device)
        for i in range(0, H * W, R):
            rays = ops.generate_rays(idx[i:i + R].contiguous(), data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"],
                                     aabb=trainer.aabb, near_plane=trainer.cfg.near_plane, training=False)
            out[i:i + R] = trainer.forward(rays, None, anneal, training=False)
        gt = imgs[m].reshape(-1, 3).float() / 255.0
        mse = torch.mean((out - gt) ** 2)
        psnrs.append(float(10.0 * torch.log10(1.0 / mse)))
        chw = lambda t: t.view(H, W, 3).permute(2, 0, 1)[None]
        ssims.append(float(ssim_fn(chw(gt), chw(out))))  # as get_image_metrics_and_images (kplanes.py:469-473)
    eval_psnr.last_ssim = ssims
    return psnrs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30000)
    ap.add_argument("--eval-images", type=int, default=8)
    ap.add_argument("--eval-every", type=int, default=10000)
    ap.add_argument("--out", default="gpurun_out/psnr.json")
    ap.add_argument("--seed", type=int, default=20231029)
    ap.add_argument("--mlp-operands", default="fp32", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--also-eval-fp32", action="store_true", help="16-bit operands: evaluate the final model a second time with the exact fp32 MLP kernels (paired comparison)")
    ap.add_argument("--no-overlap", action="store_true", help="single-stream step (A/B against stream-ordering effects)")
    ap.add_argument("--sync-adam", action="store_true", help="field-plane optimiser sweep on the main stream (A/B)")
    ap.add_argument("--prop-on-main", action="store_true", help="proposal backward on the main stream, before the field chain (A/B)")
    ap.add_argument("--no-defer", action="store_true", help="join the proposal chain at the end of backward (A/B)")
    ap.add_argument("--check-finite", type=int, default=0, help="every N steps: stop at the first non-finite parameter / Adam state and say where")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    cfg = KPlanesTrainConfig(max_steps=30000, mlp_operands=args.mlp_operands)
    R = 4096
    trainer = KPlanesTrainer(cfg, R, dev)
    if args.no_overlap:
        trainer.overlap, trainer.async_field_adam = False, False
    if args.sync_adam:
        trainer.async_field_adam = False
    trainer.prop_on_main, trainer.defer_prop = args.prop_on_main, not args.no_defer
    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 3)
    train = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
    held = synthetic.render_dataset(cams, times, [19], dev, chunk_rows=540)
    M, H, W = train["images"].shape[:3]
    t0 = time.time()
    ist = compute_ist(train["images"], train["cam_id"], train["times"], ist_range=1.0)  # method_configs.py:503
    batch = {"image": train["images"], "image_idx": torch.arange(M, device=dev), "ist_weights": ist, "iter_steps": 0}
    sampler = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
    DynamicBasedPixelSampler.prepare(batch)
    torch.cuda.synchronize()
    log = {"config": "k-planes preset, synthetic Broadcast-style (19 train cams x 33 frames 960x540, camera 19 held out)", "steps": args.steps,
           "ist_precompute_s": time.time() - t0, "ist_nonzero_fraction": float((ist > 0).float().mean()), "evals": []}
    t_train = 0.0
    for step in range(args.steps):
        if step % 500 == 0:
            torch.cuda.synchronize()
            t1 = time.time()
        batch["iter_steps"] = step
        idx = sampler.sample_method(R, M, H, W, batch=batch, device=dev)
        target = train["images"][idx[:, 0], idx[:, 1], idx[:, 2]].float() / 255.0
        rays = ops.generate_rays(idx, train["fx"], train["fy"], train["cx"], train["cy"], train["c2w"], train["times"], aabb=trainer.aabb,
                                 near_plane=cfg.near_plane, training=True)
        trainer.train_step(rays, target)
        if args.check_finite and step % args.check_finite == args.check_finite - 1:
            trainer.synchronize()
            bad = [(name, what) for name, _, _, o, n in trainer.segments for what, buf in (("param", trainer.params), ("exp_avg", trainer.exp_avg),
                   ("exp_avg_sq", trainer.exp_avg_sq), ("grad", trainer.grads)) if not bool(torch.isfinite(buf[o:o + n]).all())]
            if bad:
                print(f"non-finite values first seen at step {step + 1}: {bad}", flush=True)
                for name, _, _, o, n in trainer.segments:
                    p_ = trainer.params[o:o + n]
                    fin = p_[torch.isfinite(p_)]
                    print(f"  {name}: non-finite params {int((~torch.isfinite(p_)).sum())} of {n}, max |p| {float(fin.abs().max()) if fin.numel() else float('nan'):.3e}")
                return
        if step % 500 == 499:
            torch.cuda.synchronize()
            dt = time.time() - t1
            t_train += dt
            ld = {k: float(v) for k, v in trainer.loss_dict().items()}
            print(f"step {step + 1}: {R * 500 / dt:,.0f} rays/s  rgb_loss {ld['rgb_loss']:.5f} (psnr~{-10 * torch.log10(torch.tensor(ld['rgb_loss'])).item():.2f})"
                  f" interlevel {ld['interlevel_loss']:.2e} distortion {ld['distortion_loss']:.2e}", flush=True)
        if (step + 1) % args.eval_every == 0 or step + 1 == args.steps:
            from soccernerfs_amd.trainer import anneal_value
            ps = eval_psnr(trainer, held, args.eval_images, anneal_value(step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope))
            ss = eval_psnr.last_ssim
            ps_tr = eval_psnr(trainer, train, 4, 1.0)
            log["evals"].append({"step": step + 1, "psnr_heldout_mean": sum(ps) / len(ps), "psnr_heldout": ps, "ssim_heldout_mean": sum(ss) / len(ss),
                                 "psnr_train_views_mean": sum(ps_tr) / len(ps_tr)})
            print(f"== step {step + 1}: held-out camera PSNR {sum(ps) / len(ps):.2f} dB, SSIM {sum(ss) / len(ss):.4f} over {len(ps)} frames; train views {sum(ps_tr) / len(ps_tr):.2f} dB", flush=True)
    if args.also_eval_fp32 and args.mlp_operands != "fp32":
        nets = [trainer.sigma_net] + list(trainer.prop_nets)
        saved = [n.desc.operands for n in nets]
        for n in nets:
            n.desc.operands = 0
        ps32 = eval_psnr(trainer, held, args.eval_images, 1.0)
        ss32 = eval_psnr.last_ssim
        tr32 = eval_psnr(trainer, train, 4, 1.0)
        for n, o in zip(nets, saved):
            n.desc.operands = o
        log["evals"][-1].update({"fp32_eval_psnr_heldout_mean": sum(ps32) / len(ps32), "fp32_eval_ssim_heldout_mean": sum(ss32) / len(ss32),
                                 "fp32_eval_psnr_train_views_mean": sum(tr32) / len(tr32)})
        print(f"== same weights evaluated with fp32 MLP kernels: held-out {sum(ps32) / len(ps32):.2f} dB, SSIM {sum(ss32) / len(ss32):.4f}; train views {sum(tr32) / len(tr32):.2f} dB", flush=True)
    log["train_seconds"] = t_train
    log["train_rays_per_s_mean"] = R * args.steps / max(t_train, 1e-9)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(log, open(args.out, "w"), indent=1)
    print(json.dumps(log["evals"][-1]))


if __name__ == "__main__":
    main()

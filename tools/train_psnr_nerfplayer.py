#!/usr/bin/env python3
"""PSNR@N-iterations run of the nerfplayer-nerfacto preset with the fused trainer on the synthetic multi-view clip (stadium-style split:
30 training cameras + 6 held out, every frame; NS/data/dataparsers/stadiumwide_dataparser.py:94-112), uniform pixels, then full-image
renders of held-out cameras in eval mode.  Convergence check of soccernerfs_amd.nerfplayer_trainer, not the bench line.

    python tools/train_psnr_nerfplayer.py --steps 30000 --out gpurun_out/psnr_nerfplayer.json
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic
from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModelConfig
from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer
from soccernerfs_amd.trainer import anneal_value


@torch.no_grad()
def eval_psnr(tr, data, n_images, anneal):
    imgs = data["images"]; M, H, W = imgs.shape[:3]; R = tr.R
    ys, xs = torch.meshgrid(torch.arange(H, device=imgs.device), torch.arange(W, device=imgs.device), indexing="ij")
    out_ps = []
    for m in torch.linspace(0, M - 1, n_images).long().tolist():
        idx = torch.stack([torch.full_like(ys, m), ys, xs], -1).reshape(-1, 3)
        n = idx.shape[0] // R * R  # whole chunks of R rays (the trainer's buffers have a fixed batch size)
        out = torch.empty(n, 3, device=imgs.device)
        for i in range(0, n, R):
            rays = ops.generate_rays(idx[i:i + R].contiguous(), data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"])
            out[i:i + R] = tr.forward(rays, None, {"bg": torch.rand(R, 3, device=imgs.device)}, anneal, training=False)
        gt = imgs[m].reshape(-1, 3)[:n].float() / 255.0
        out_ps.append(float(10.0 * torch.log10(1.0 / torch.mean((out - gt) ** 2))))
    return out_ps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30000)
    ap.add_argument("--eval-every", type=int, default=10000)
    ap.add_argument("--out", default="gpurun_out/psnr_nerfplayer.json")
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--scene", default="clip", choices=["clip", "stadium"], help="stadium: synthetic.make_stadium_cameras / shade_stadium -- 30 wide-angle cameras high in "
                    "the bleachers + 6 evaluation cameras near the players, aabb [-1,1]^3 (BASELINE.json configs[3]; stadiumwide_dataparser.py:94-112)")
    ap.add_argument("--seed", type=int, default=20231029)
    ap.add_argument("--mlp-operands", default="fp32", choices=["fp32", "bf16"], help="NerfplayerTrainer(mlp_operands=...): bf16 MFMA operands in the decode net and the colour head")
    ap.add_argument("--async-sweep", action="store_true", help="NerfplayerTrainer(async_field_sweep=True): the main table's optimiser sweep on a side stream")
    ap.add_argument("--standin", action="store_true", help="train oracle/nerfplayer_standin.NerfplayerStandinTrainer instead (the reference's algorithm in stock PyTorch, started "
                    "from the HIP trainer's initial parameters): the reference-algorithm arm of config 4's PSNR comparison")
    ap.add_argument("--train-budget-s", type=float, default=0.0, help="stop training when this much wall-clock is spent (a gpurun call is capped at one hour), evaluate there")
    ap.add_argument("--tiled", action="store_true", help="NerfplayerTrainer(tiled_field_backward=True): the main table's scatter + TV + Adam as one owner-computes pass (round 6)")
    args = ap.parse_args()
    dev = torch.device("cuda:0"); torch.manual_seed(args.seed)
    R = 4096
    Wd, Hd = args.width, args.width * 9 // 16
    stadium = args.scene == "stadium"
    cams = synthetic.make_stadium_cameras(30, 6, Wd, Hd) if stadium else synthetic.make_cameras(36, Wd, Hd)
    times = synthetic.frame_times(100, 100 // args.frames)
    variant = "stadium" if stadium else "default"
    train = synthetic.render_dataset(cams, times, list(range(30)), dev, chunk_rows=Hd, variant=variant)
    held = synthetic.render_dataset(cams, times, list(range(30, 36)), dev, chunk_rows=Hd, variant=variant)
    M, H, W = train["images"].shape[:3]
    tr = NerfplayerTrainer(NerfplayerNerfactoModelConfig(), R, M, aabb_scale=1.0 if stadium else 1.5, device=dev, max_steps=args.steps,
                           mlp_operands=args.mlp_operands, async_field_sweep=args.async_sweep, tiled_field_backward=args.tiled)
    if args.standin:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle.nerfplayer_standin import NerfplayerStandinTrainer  # checker code: this dev tool trains it for the PSNR anchor

        hip = tr
        tr = NerfplayerStandinTrainer(hip, dev, max_steps=args.steps)
        del hip
        torch.cuda.empty_cache()
    log = {"config": f"nerfplayer-nerfacto preset, {'stock-PyTorch stand-in (oracle/nerfplayer_standin.py)' if args.standin else 'fused trainer'}, synthetic {'stadium-players scene' if stadium else 'clip'} ({M} training images {W}x{H}, "
                     f"{len(times)} frames per camera, 6 cameras held out), seed {args.seed}", "scene": args.scene, "mlp_operands": args.mlp_operands,
           "async_field_sweep": bool(args.async_sweep), "tiled_field_backward": bool(args.tiled), "standin": bool(args.standin), "evals": []}
    t_begin = time.time()
    t_train = 0.0
    for step in range(args.steps):
        if step % 500 == 0:
            torch.cuda.synchronize(); t1 = time.time()
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, train["images"])
        rays = ops.generate_rays(idx, train["fx"], train["fy"], train["cx"], train["cy"], train["c2w"], train["times"])
        tr.train_step(rays, idx[:, 0].contiguous(), target)
        if step % 500 == 499:
            torch.cuda.synchronize(); dt = time.time() - t1; t_train += dt
            ld = {k: float(v) for k, v in tr.loss_dict().items()}
            print(f"step {step + 1}: {R * 500 / dt:,.0f} rays/s  " + "  ".join(f"{k} {v:.3e}" for k, v in ld.items()), flush=True)
        out_of_time = bool(args.train_budget_s and step % 100 == 0 and time.time() - t_begin > args.train_budget_s)
        if out_of_time:
            log["stopped_early_at_step"] = step + 1
        if (step + 1) % args.eval_every == 0 or step + 1 == args.steps or out_of_time:
            ps = eval_psnr(tr, held, 6, anneal_value(step, 1000, 10.0))
            ps_tr = eval_psnr(tr, train, 4, 1.0)
            log["evals"].append({"step": step + 1, "psnr_heldout_mean": sum(ps) / len(ps), "psnr_heldout": ps, "psnr_train_views_mean": sum(ps_tr) / len(ps_tr)})
            print(f"== step {step + 1}: held-out cameras PSNR {sum(ps) / len(ps):.2f} dB; train views {sum(ps_tr) / len(ps_tr):.2f} dB", flush=True)
        if out_of_time:
            break
    log["train_rays_per_s_mean"] = R * (log.get("stopped_early_at_step", args.steps) // 500 * 500) / max(t_train, 1e-9)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(log, open(args.out, "w"), indent=1)
    print(json.dumps(log["evals"][-1]))


if __name__ == "__main__":
    main()

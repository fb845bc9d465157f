#!/usr/bin/env python3
"""PSNR@N-iterations runs of the k-planes preset on the synthetic Broadcast-style scene (BASELINE.json metric, 2nd half).

Trains with the fused HIP trainer exactly as the reference schedules it (4096 rays/step, uniform pixels for the first
`iters_to_start_is` = 2000 steps, then 15 % IST importance rays; proposal-weight annealing; proposal updates every n steps;
Adam + cosine schedule), then renders held-out views (full images, eval-mode sampler, 'last_sample' background, clamp) and
reports PSNR = 10 log10(1/MSE) per image and averaged (NS/models/kplanes.py:291,472; NS/pipelines/base_pipeline.py:323-362).

Evaluation sets (all frames of every camera unless --eval-frames):
  * "camera_20": the 20th camera of the arc, the reference's "all" split eval camera (broadcaststyle_dataparser.py:166-190).  It sits at
    the END of the arc, i.e. it is an EXTRAPOLATED view;
  * "novel": three evaluation-only cameras half-way between training cameras (synthetic.make_novel_cameras): interpolated views;
  * "train": four training images (sanity: the fit itself).
Several seeds run in one process on the same dataset; the JSON holds every run and the mean / spread per set.

    python tools/train_psnr.py --steps 30000 --seeds 1,2,3 --out gpurun_out/psnr_r02_fp32.json
"""
import argparse
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.metrics import structural_similarity_index_measure as ssim_fn  # noqa: E402
from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler, compute_ist  # noqa: E402
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer, anneal_value  # noqa: E402


@torch.no_grad()
def eval_set(trainer, data, image_ids, anneal, with_ssim=True):
    """PSNR (and SSIM) of full-image renders of data["images"][image_ids]."""
    imgs = data["images"]
    H, W = imgs.shape[1:3]
    R = getattr(trainer, "eval_chunk", trainer.R)
    ys, xs = torch.meshgrid(torch.arange(H, device=imgs.device), torch.arange(W, device=imgs.device), indexing="ij")
    psnrs, ssims = [], []
    for m in image_ids:
        idx = torch.stack([torch.full_like(ys, m), ys, xs], -1).reshape(-1, 3)
        out = torch.empty(H * W, 3, device=imgs.device)
        for i in range(0, H * W, R):
            rays = ops.generate_rays(idx[i:i + R].contiguous(), data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"],
                                     aabb=trainer.aabb, near_plane=trainer.cfg.near_plane, training=False)
            out[i:i + R] = trainer.forward(rays, None, anneal, training=False)
        gt = imgs[m].reshape(-1, 3).float() / 255.0
        mse = torch.mean((out - gt) ** 2)
        psnrs.append(float(10.0 * torch.log10(1.0 / mse)))
        if with_ssim:
            chw = lambda t: t.view(H, W, 3).permute(2, 0, 1)[None]
            ssims.append(float(ssim_fn(chw(gt), chw(out))))  # as get_image_metrics_and_images (kplanes.py:469-473)
    return psnrs, ssims


def stats(xs):
    n = len(xs)
    mean = sum(xs) / n
    return {"mean": mean, "min": min(xs), "max": max(xs), "std": (sum((x - mean) ** 2 for x in xs) / max(n - 1, 1)) ** 0.5, "n": n}


def run_one(args, seed, train, sets, ist, dev, log_steps=True):
    torch.manual_seed(seed)
    random.seed(seed)
    cfg = KPlanesTrainConfig(max_steps=args.schedule_steps, mlp_operands=args.mlp_operands, seed=seed, deterministic=args.deterministic,
                             nonfinite_policy=args.nonfinite_policy, fused_field=not args.no_fused_field, quotient_scatter=not args.no_quotient_scatter, gvec_dtype=args.gvec_dtype,
                             emulate_transports=args.emulate_transports, quotient_epilogue=not args.no_quotient_epilogue, fused_proposal=not args.no_fused_proposal,
                             sigma_operands=args.sigma_operands, color_operands=args.color_operands, proposal_operands=args.proposal_operands)
    R = 4096
    if args.standin:
        # the reference's algorithm in stock PyTorch (oracle/torch_standin.py: checker / baseline code, imported ONLY for this mode) on the same
        # pixel draws (same sampler, same seed), the same rays and the same evaluation; F.grid_sample per plane as the reference calls it
        from oracle import kplanes_oracle as KO, torch_standin as TS
        KO.USE_GRID_SAMPLE = args.standin_layout != "hwc"
        trainer = TS.StandinTrainer(dev, R, seed=seed, max_steps=args.schedule_steps, plane_layout=args.standin_layout)
    else:
        trainer = KPlanesTrainer(cfg, R, dev)
        if args.oracle_init:  # the stand-in's initial parameters (same generator, same seed) instead of the trainer's own draw
            from oracle import kplanes_oracle as KO, torch_standin as TS
            trainer.load_oracle_params(KO.make_kplanes_params(seed=seed, **TS.PRESET))
        if args.no_overlap:
            trainer.overlap, trainer.async_field_adam = False, False
    M, H, W = train["images"].shape[:3]
    batch = {"image": train["images"], "image_idx": torch.arange(M, device=dev), "ist_weights": ist, "iter_steps": 0}
    sampler = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
    DynamicBasedPixelSampler.prepare(batch)
    time_key, n_time_keys = ops.image_time_keys(train["times"])
    run = {"seed": seed, "evals": []}
    t_train, t_eval = 0.0, 0.0
    t_start, last_step = time.time(), args.steps - 1
    eval_at = {int(x) for x in args.eval_at.split(",") if x}

    def evaluate(step):
        nonlocal t_eval
        an = anneal_value(step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope)
        ev = {"step": step + 1}
        t2 = time.time()
        for name, (data, ids) in sets.items():
            ps, ss = eval_set(trainer, data, ids, an, with_ssim=name != "train")
            ev[name] = {"psnr_mean": sum(ps) / len(ps), "psnr_min": min(ps), "images": len(ps), "psnr_per_image": [round(p, 3) for p in ps]}
            if ss:
                ev[name]["ssim_mean"] = sum(ss) / len(ss)
        run["evals"].append(ev)
        torch.cuda.synchronize()
        t_eval += time.time() - t2
        print(f"== [seed {seed}] step {step + 1}: " + "  ".join(f"{k} {v['psnr_mean']:.2f} dB" for k, v in ev.items() if k != "step") + f"  (eval {time.time() - t2:.0f} s)", flush=True)

    for step in range(args.steps):
        if args.train_budget_s and step % 100 == 0 and time.time() - t_start > args.train_budget_s:
            # out of wall-clock (a gpurun call is capped at one hour): stop here, evaluate, and SAY how far the run got
            last_step = step - 1
            run["stopped_early_at_step"] = step
            print(f"[seed {seed}] training budget of {args.train_budget_s} s spent at step {step}: evaluating there", flush=True)
            break
        if step % 1000 == 0:
            torch.cuda.synchronize()
            t1 = time.time()
        batch["iter_steps"] = step
        idx = sampler.sample_method(R, M, H, W, batch=batch, device=dev)
        if args.time_sorted_rays:  # the batch in order of frame time (ops.sort_rays_by_time: a batch is a set), as bench.py runs it
            idx = ops.sort_rays_by_time(idx, time_key, n_time_keys)
        target = train["images"][idx[:, 0], idx[:, 1], idx[:, 2]].float() / 255.0
        rays = ops.generate_rays(idx, train["fx"], train["fy"], train["cx"], train["cy"], train["c2w"], train["times"], aabb=trainer.aabb,
                                 near_plane=cfg.near_plane, training=True)
        trainer.train_step(rays, target)
        if args.standin and step % 100 == 99:
            print(f"[standin seed {seed}] step {step + 1} {time.time() - t1:.1f}s into this thousand", flush=True)
        if step % 1000 == 999:
            torch.cuda.synchronize()
            dt = time.time() - t1
            t_train += dt
            if log_steps:
                ld = {k: float(v) for k, v in trainer.loss_dict().items()}
                print(f"[seed {seed}] step {step + 1}: {R * 1000 / dt:,.0f} rays/s  rgb_loss {ld['rgb_loss']:.5f} "
                      f"(psnr~{-10 * torch.log10(torch.tensor(ld['rgb_loss'])).item():.2f}) interlevel {ld['interlevel_loss']:.2e} "
                      f"skipped steps {trainer.skipped_steps()}", flush=True)
        if ((step + 1) % args.eval_every == 0 or (step + 1) in eval_at) and step + 1 < args.steps:
            evaluate(step)
    evaluate(last_step)
    run["train_seconds"] = t_train
    run["train_rays_per_s_mean"] = R * ((last_step + 1) // 1000 * 1000) / max(t_train, 1e-9)
    run["skipped_steps"] = trainer.skipped_steps()
    trainer.synchronize()
    run["param_checksum"] = float(trainer.params.double().sum())
    run["eval_seconds"] = t_eval
    return run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30000)
    ap.add_argument("--schedule-steps", type=int, default=30000, help="max_steps of the cosine schedule")
    ap.add_argument("--seeds", default="20231029")
    ap.add_argument("--eval-frames", type=int, default=0, help="frames per held-out camera (0 = all)")
    ap.add_argument("--eval-every", type=int, default=10000)
    ap.add_argument("--out", default="gpurun_out/psnr.json")
    ap.add_argument("--mlp-operands", default="bf16", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--gvec-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--sigma-operands", default=None)
    ap.add_argument("--color-operands", default=None)
    ap.add_argument("--proposal-operands", default=None)
    ap.add_argument("--deterministic", action="store_true", help="fixed-point gradient accumulation: bit-identical reruns")
    ap.add_argument("--nonfinite-policy", default="skip_step", choices=["skip_step", "drop_elements"])
    ap.add_argument("--no-fused-field", action="store_true", help="unfused forward kernels")
    ap.add_argument("--no-quotient-epilogue", action="store_true", help="round 3's flow: G = gfeat .* feat from the separate pass over fp32 features (A-B)")
    ap.add_argument("--no-fused-proposal", action="store_true", help="proposal levels as gather + net kernels (A-B; bit-identical densities)")
    ap.add_argument("--no-quotient-scatter", action="store_true", help="product form of the field's sorted scatter")
    ap.add_argument("--emulate-transports", default="", choices=["", "grad", "param", "both"],
                    help="single-GPU emulation of the bf16 gradient / parameter-update transports of the sharded multi-GPU step (KPlanesTrainConfig.emulate_transports)")
    ap.add_argument("--time-sorted-rays", action="store_true", help="every batch in order of frame time, as bench.py runs it")
    ap.add_argument("--standin", action="store_true",
                    help="train the REFERENCE'S ALGORITHM in stock PyTorch (oracle/torch_standin.StandinTrainer: F.grid_sample per plane, Linear stacks, autograd, "
                         "two torch.optim.Adam, fp32) instead of the HIP trainer -- same pixel draws, rays, schedule and evaluation; ~95 ms / step")
    ap.add_argument("--standin-layout", default="chw", choices=["chw", "hwc"], help="--standin: chw = planes [1,C,H,W] through F.grid_sample, the reference's own call; "
                    "hwc = the same algorithm on channel-last planes gathered as rows (oracle KO._bilinear_plane_rows, checked against F.grid_sample by "
                    "tests/test_standin_cpu.py): about twice the rate, so that 30 000 steps fit one GPU call")
    ap.add_argument("--eval-at", default="", help="extra evaluation steps, comma separated (e.g. the step a stand-in run got to)")
    ap.add_argument("--train-budget-s", type=float, default=0.0, help="stop training when this much wall-clock is spent (a gpurun call is capped at one hour), "
                    "evaluate there and record `stopped_early_at_step`")
    ap.add_argument("--oracle-init", action="store_true", help="HIP trainer from the stand-in's initial parameters of the same seed")
    ap.add_argument("--scene", default="default", choices=["default", "textured"], help="synthetic.shade variant (textured: grass grain, board, crowd, ten players)")
    ap.add_argument("--no-overlap", action="store_true", help="single-stream step (A/B against stream-ordering effects)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 3)
    train = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540, variant=args.scene)
    held = synthetic.render_dataset(cams, times, [19], dev, chunk_rows=540, variant=args.scene)
    novel = synthetic.render_dataset(synthetic.make_novel_cameras(3, 960, 540), times, [0, 1, 2], dev, chunk_rows=540, variant=args.scene)
    t0 = time.time()
    ist = compute_ist(train["images"], train["cam_id"], train["times"], ist_range=1.0)  # method_configs.py:503
    torch.cuda.synchronize()
    pick = lambda data: (list(range(data["images"].shape[0])) if not args.eval_frames else
                         [c * len(times) + int(f) for c in range(data["images"].shape[0] // len(times))
                          for f in torch.linspace(0, len(times) - 1, args.eval_frames).long().tolist()])
    sets = {"camera_20": (held, pick(held)), "novel": (novel, pick(novel)),
            "train": (train, torch.linspace(0, train["images"].shape[0] - 1, 4).long().tolist())}
    log = {"config": "k-planes preset, synthetic Broadcast-style (19 train cams x 33 frames 960x540)", "steps": args.steps, "scene": args.scene, "eval_frames": args.eval_frames or "all",
           "trainer": ("oracle/torch_standin.StandinTrainer: the reference's algorithm in stock PyTorch-ROCm, fp32"
                       + (", planes stored channel-last and gathered as rows (index_select) instead of F.grid_sample" if args.standin_layout == "hwc" else ", F.grid_sample per plane"))
           if args.standin else "soccernerfs_amd KPlanesTrainer (HIP)",
           "oracle_init": bool(args.oracle_init or args.standin),
           "mlp_operands": args.mlp_operands, "gvec_dtype": args.gvec_dtype, "per_net_operands": [args.sigma_operands, args.color_operands, args.proposal_operands],
           "deterministic": args.deterministic, "nonfinite_policy": args.nonfinite_policy,
           "emulate_transports": args.emulate_transports, "time_sorted_rays": args.time_sorted_rays, "fused_field": not args.no_fused_field, "quotient_scatter": not args.no_quotient_scatter,
           "quotient_epilogue": not args.no_quotient_epilogue, "fused_proposal": not args.no_fused_proposal,
           "eval_sets": {"camera_20": "20th arc camera (reference 'all' split eval camera; extrapolated view), %d frames" % len(sets["camera_20"][1]),
                         "novel": "3 evaluation-only cameras between training cameras (interpolated views), %d images" % len(sets["novel"][1]),
                         "train": "4 training images"},
           "ist_precompute_s": time.time() - t0, "ist_nonzero_fraction": float((ist > 0).float().mean()), "runs": []}
    for seed in [int(s) for s in args.seeds.split(",")]:
        log["runs"].append(run_one(args, seed, train, sets, ist, dev))
        final = [r["evals"][-1] for r in log["runs"]]
        log["summary"] = {name: stats([f[name]["psnr_mean"] for f in final]) for name in sets}
        log["summary"]["novel_ssim"] = stats([f["novel"]["ssim_mean"] for f in final])
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(log, open(args.out, "w"), indent=1)
    print(json.dumps(log["summary"]))


if __name__ == "__main__":
    main()

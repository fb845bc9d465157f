"""Config 4 (nerfplayer-nerfacto preset, R = 4096) through the nerfstudio-shaped model on the HIP ops: rays/s of a full train step
(forward, loss, autograd backward through the HIP kernels, Adam).  Dev tool / profile note, not the bench line."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModel, NerfplayerNerfactoModelConfig
from soccernerfs_amd.rays import RayBundle
from soccernerfs_amd.scene_colliders import SceneBox

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--frames", type=int, default=0, help="quantise the ray times to this many frame times (0: continuous)")
ap.add_argument("--sort-times", action="store_true", help="with --frames: the batch in order of frame time")
ap.add_argument("--optim", default="snerf", choices=["snerf", "fused", "foreach", "single"])
ap.add_argument("--no-fuse-tv", action="store_true")
ap.add_argument("--fused", action="store_true", help="soccernerfs_amd.nerfplayer_trainer.NerfplayerTrainer instead of the autograd model")
ap.add_argument("--full", action="store_true", help="full NeRFPlayer (`nerfplayer` preset, soccernerfs_amd.nerfplayer.NerfplayerModel) instead of the nerfacto variant")
ap.add_argument("--profile", action="store_true")
ap.add_argument("--mlp-operands", default="bf16", choices=["fp32", "bf16"], help="--fused: fp32 = every net exact (the parity path); bf16 = 16-bit MFMA operands where the fused kernels have them")
ap.add_argument("--sync-sweep", action="store_true", help="A/B (--fused): the field table's optimiser sweep in order on the main stream instead of on a side stream behind its gradient scatter")
ap.add_argument("--tiled", action="store_true", help="with --fused: the field table through the owner-computes backward + fused Adam (csrc/tgrid_tiles.hip)")
ap.add_argument("--tiled-first-level", type=int, default=0)
ap.add_argument("--late-bin", action="store_true", help="A/B (--tiled): the binning pass behind the decode net's backward on the caller's stream instead of beside the field forward")
ap.add_argument("--stadium", action="store_true", help="with --fused: camera rays of the synthetic stadium-players scene (30 cameras in the bleachers, aabb [-1,1]^3, uniform "
                "pixels) instead of random rays through the box -- what bench.py's config-4 leg times")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
if args.fused:
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R = args.rays
    tr = NerfplayerTrainer(NerfplayerNerfactoModelConfig(), R, 36 * 100, device=dev, async_field_sweep=not args.sync_sweep, mlp_operands=args.mlp_operands, tiled_field_backward=args.tiled, tiled_first_level=args.tiled_first_level)
    tr.step = 600  # past the learning-rate warm-up

    if args.stadium:
        from soccernerfs_amd import ops, synthetic

        cams = synthetic.make_stadium_cameras(30, 6, 960, 540)
        frame_ids = torch.linspace(0, 99, 8).long()
        data = synthetic.render_dataset(cams, frame_ids.float() / 99, list(range(30)), dev, chunk_rows=540, variant="stadium")
        M, H, W = data["images"].shape[:3]
        full_index = (data["cam_id"] * 100 + frame_ids.to(dev).repeat(30)).contiguous()
        tr = NerfplayerTrainer(NerfplayerNerfactoModelConfig(), R, 3000, aabb_scale=1.0, device=dev, async_field_sweep=not args.sync_sweep, mlp_operands=args.mlp_operands, tiled_field_backward=args.tiled, tiled_first_level=args.tiled_first_level)
        tr.step = 600
    tr.early_bin = not args.late_bin

    def fstep():
        if args.stadium:
            idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
            rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"])
            tr.train_step(rays, full_index[idx[:, 0]].contiguous(), target)
            return
        o = (torch.rand(R, 3, device=dev) * 2 - 1) * 0.6
        d = torch.nn.functional.normalize(torch.rand(R, 3, device=dev) * 2 - 1, dim=-1)
        t = torch.rand(R, 1, device=dev)
        if args.frames:  # frame times as a dataset has them (stadium-players: 100 frames), optionally the batch in order of time
            t = torch.floor(t * args.frames) / max(args.frames - 1, 1)
            if args.sort_times:
                t = torch.sort(t, dim=0).values
        tr.train_step({"origins": o, "directions": d, "times": t}, torch.randint(0, 3600, (R,), device=dev), torch.rand(R, 3, device=dev))

    for _ in range(args.warmup):
        fstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fstep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "nerfplayer-nerfacto preset (fused trainer)" + (", stadium-players camera rays" if args.stadium else ", random rays through the box"), "rays": R, "params": int(tr.n_params), "ms_per_step": dt / args.steps * 1e3,
                      "rays_per_s": R * args.steps / dt}))
    sys.exit(0)
if args.full:
    from soccernerfs_amd.nerfplayer import NerfplayerModel, NerfplayerModelConfig
    model = NerfplayerModel(NerfplayerModelConfig(), SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=36 * 100).to(dev).train()
    main_encoders = [model.field.newness_field, model.field.decomposition_field]
else:
    model = NerfplayerNerfactoModel(NerfplayerNerfactoModelConfig(), SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=36 * 100).to(dev).train()
    main_encoders = [model.field.mlp_base]
model.scene_box.aabb = model.scene_box.aabb.to(dev)
params = [p for g in model.get_param_groups().values() for p in g if p.requires_grad]
encoders = main_encoders + [p.encoding for p in model.proposal_networks]
for e in encoders:
    e.fuse_tv = not args.no_fuse_tv
if args.optim == "snerf":
    from soccernerfs_amd.optimizers import FusedAdam
    opt = FusedAdam(params, lr=1e-2, eps=1e-15, encoders=encoders)
else:
    kw = {"fused": True} if args.optim == "fused" else ({"foreach": True} if args.optim == "foreach" else {"foreach": False})
    opt = torch.optim.Adam(params, lr=1e-2, eps=1e-15, **kw)
R = args.rays
cbs = model.get_training_callbacks()

def step(i):
    o = (torch.rand(R, 3, device=dev) * 2 - 1) * 0.6
    d = torch.nn.functional.normalize(torch.rand(R, 3, device=dev) * 2 - 1, dim=-1)
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1, device=dev), camera_indices=torch.randint(0, 3600, (R, 1), device=dev),
                   times=torch.rand(R, 1, device=dev))
    target = torch.rand(R, 3, device=dev)
    for where, fn in cbs:
        if where == "before":
            fn(i)
    out = model(rb)
    ld = model.get_loss_dict(out, {"image": target}, model.get_metrics_dict(out, {"image": target}))
    loss = sum(ld.values())
    opt.zero_grad(set_to_none=True)  # torch optimisers: incoming gradient tensors are adopted; FusedAdam: no-op (cleared by its sweep)
    loss.backward()
    opt.step()
    for where, fn in cbs:
        if where == "after":
            fn(i)

for i in range(args.warmup):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    step(args.warmup + i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"config": ("nerfplayer preset, full NeRFPlayer" if args.full else "nerfplayer-nerfacto preset") + " (autograd face on HIP ops)", "rays": R, "params": sum(p.numel() for p in params), "optim": args.optim,
                  "ms_per_step": dt / args.steps * 1e3, "rays_per_s": R * args.steps / dt}))

"""Find where the first non-finite value of a long K-Planes run appears (dev tool): checks work buffers, gradients and parameters
every step with device-side flags (no host sync), reports every 250 steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer, anneal_value, update_schedule

dev = torch.device("cuda:0"); torch.manual_seed(int(os.environ.get("SEED", "20231029")))
cfg = KPlanesTrainConfig(max_steps=30000); R = 4096
tr = KPlanesTrainer(cfg, R, dev)
cams = synthetic.make_cameras(20, 960, 540); times = synthetic.frame_times(100, 3)
data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
M, H, W = data["images"].shape[:3]
names = ["feat", "h", "dens2", "w2", "rgb", "rgb_out", "gw2", "grgb", "gh", "gfeat", "gvec", "grads.field.planes", "grads.field.sigma", "grads.field.color",
         "grads.prop", "params"]
first = torch.full((len(names),), 1 << 30, dtype=torch.int64, device=dev)
def chk(i, t, step):
    bad = ~torch.isfinite(t).all()
    first[i] = torch.where(bad & (first[i] == (1 << 30)), torch.tensor(step, device=dev), first[i])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
for step in range(steps):
    idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
    rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=0.0, training=True)
    anneal = anneal_value(tr.step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope)
    sstep = max(tr.step - 1, 0)
    updated = tr._steps_since_update > update_schedule(sstep, cfg.proposal_warmup, cfg.proposal_update_every) or sstep < 10
    rng = tr.random_draws()
    tr.forward(rays, rng, anneal, training=True)
    b = tr.buf
    tr.backward(target, rng, proposal_grads=updated, include_reg=False)
    for i, t in enumerate([b["feat"], b["h"], b["dens"][2], b["w"][2], b["rgb"], b["rgb_out"], b["gw"][2], b["grgb"], b["gh"], b["gfeat"], tr._ss.G if tr._ss.quotient else tr._ss.gvec,
                           tr.gviews["field.planes"], tr.gviews["field.sigma"], tr.gviews["field.color"], tr.grads[:tr.n_proposal_params]]):
        chk(i, t, step)
    tr.allreduce_grads()
    tr.optimizer_step(fused_reg=True)
    chk(len(names) - 1, tr.params, step)
    if updated:
        tr._steps_since_update = 0
    tr._steps_since_update += 1
    if step % 250 == 249:
        f = first.cpu().tolist()
        if min(v for n, v in zip(names, f) if n != 'dens2') < (1 << 30):
            print(f"step {step + 1}: first non-finite step per buffer:")
            for n, v in sorted(zip(names, f), key=lambda kv: kv[1]):
                print(f"   {n:22s} {v if v < (1 << 30) else '-'}")
            # statistics of the step's inputs to help: extremes of sigma-net output / density
            print("   max dens2", float(torch.nan_to_num(b['dens'][2]).max()), "max |h|", float(torch.nan_to_num(b['h']).abs().max()), "max |feat|", float(torch.nan_to_num(b['feat']).abs().max()))
            break
        if step % 2500 == 2499:
            print(f"step {step + 1} ok; max dens {float(b['dens'][2].max()):.3e} max |h| {float(b['h'].abs().max()):.3e} max|feat| {float(b['feat'].abs().max()):.3e} max |gfeat| {float(b['gfeat'].abs().max()):.3e}", flush=True)
else:
    print("no non-finite value in", steps, "steps")

"""Config 3 (K-Planes multiscale-res 1-32, IST range 0.75, fps-downsample 4; 546 M parameters) through the fused trainer: rays/s of the
full train step incl. the IST importance sampler.  Dev tool / profile note, not the bench line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic
from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler, compute_ist
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

dev = torch.device("cuda:0"); torch.manual_seed(0)
steps, warmup, R = 100, 20, 4096
cfg = KPlanesTrainConfig(mlp_operands=(sys.argv[1] if len(sys.argv) > 1 else "bf16"), multiscale_res=(1, 2, 4, 8, 16, 32), spacetime_resolution=(64, 64, 64, 25),
                         proposal_resolutions=((128, 128, 128, 25), (256, 256, 256, 25)))
tr = KPlanesTrainer(cfg, R, dev)
tr.step = 6000  # steady-state schedule, IST active (iters_to_start_ist = 2000)
cams = synthetic.make_cameras(20, 960, 540); times = synthetic.frame_times(100, 4)
data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
M, H, W = data["images"].shape[:3]
ist = compute_ist(data["images"], data["cam_id"], data["times"], ist_range=0.75)
batch = {"image": data["images"], "image_idx": torch.arange(M, device=dev), "ist_weights": ist, "iter_steps": 6000}
sampler = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
DynamicBasedPixelSampler.prepare(batch)

def step():
    idx = sampler.sample_method(R, M, H, W, batch=batch, device=dev)
    target = data["images"][idx[:, 0], idx[:, 1], idx[:, 2]].float() / 255.0
    rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=0.0, training=True)
    tr.train_step(rays, target)

for _ in range(warmup):
    step()
tr.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step()
tr.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"config": "K-Planes multiscale 1-32 (6 scales), C=32, 25 frames, IST range 0.75, 15 % importance rays", "mlp_operands": cfg.mlp_operands, "params": int(tr.n_params),
                  "images": int(M), "ms_per_step": dt / steps * 1e3, "rays_per_s": R * steps / dt}))

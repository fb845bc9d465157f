"""How many distinct (row, x0) texel keys do the field samples of one step touch per plane and scale?  (dev tool:
upper bound on what sorting the samples per plane could remove from the scatter's atomic requests.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = KPlanesTrainConfig(); R = 4096
tr = KPlanesTrainer(cfg, R, dev)
cams = synthetic.make_cameras(20, 960, 540); times = synthetic.frame_times(100, 3)[:4]
data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
M, H, W = data["images"].shape[:3]; scale = torch.tensor([M, H, W], dtype=torch.float32, device=dev)
def step():
    idx = torch.floor(torch.rand(R, 3, device=dev) * scale).long()
    target = data["images"][idx[:, 0], idx[:, 1], idx[:, 2]].float() / 255.0
    rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=0.0, training=True)
    tr.train_step(rays, target); return rays
for target_step in (20, 500, 3000):
    while tr.step < target_step: rays = step()
    eb = tr.buf["eb"][2]; mid = (eb[:, :-1] + eb[:, 1:]) / 2
    pos = rays["origins"][:, None, :] + rays["directions"][:, None, :] * mid[..., None]
    p = ((pos + 1.5) / 3.0).clamp(0, 1).reshape(-1, 3)
    t = rays["times"].expand(R, 64).reshape(-1, 1)
    pts = torch.cat([p, t], -1)  # [N,4] in [0,1]
    N = pts.shape[0]
    print(f"step {tr.step}: N={N}")
    tot_now = tot_sorted = 0
    for s, m in enumerate((1, 2, 4, 8, 16)):
        res = [64 * m, 64 * m, 64 * m, 100]
        line = f"  scale {m:2d}:"
        for (a, b) in [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]:
            x0 = torch.floor(pts[:, a] * (res[a] - 1)).long(); y0 = torch.floor(pts[:, b] * (res[b] - 1)).long()
            key = y0 * res[a] + x0
            uniq = torch.unique(key).numel()
            k2 = key.view(R, 64); runs = int((k2[:, 1:] != k2[:, :-1]).sum()) + R  # flushes with in-ray run-length combining (one row)
            tot_now += runs; tot_sorted += uniq
            line += f" {runs/N:5.2f}/{uniq/N:5.3f}"
        print(line + "   (flushes per sample now / if sorted per plane)")
    print(f"  total row-flushes: now {tot_now/1e6:.2f} M, sorted {tot_sorted/1e6:.2f} M  (x ~2 rows x 4 requests)")

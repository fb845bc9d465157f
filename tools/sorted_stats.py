"""Flush counts of the sorted scatter on a real training batch (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = KPlanesTrainConfig(); R = 4096
tr = KPlanesTrainer(cfg, R, dev)
cams = synthetic.make_cameras(20, 960, 540); times = synthetic.frame_times(100, 3)[:4]
data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
M, H, W = data["images"].shape[:3]; scale = torch.tensor([M, H, W], dtype=torch.float32, device=dev)
def step():
    idx = torch.floor(torch.rand(R, 3, device=dev) * scale).long()
    target = data["images"][idx[:, 0], idx[:, 1], idx[:, 2]].float() / 255.0
    rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=0.0, training=True)
    tr.train_step(rays, target)
for _ in range(30): step()
torch.cuda.synchronize()
N = R * 64
rec = tr._ss.sorted_rec.view(6, N, 4)
pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
tot = 0
for si, m in enumerate((1, 2, 4, 8, 16)):
    res = [64 * m, 64 * m, 64 * m, 100]
    line = f"scale {m:2d}:"
    for q, (a, b) in enumerate(pairs):
        xa, xb = rec[q, :, 1], rec[q, :, 2]
        pa = (((xa + 1) / 2) * (res[a] - 1)).clamp(0, res[a] - 1).floor().long()
        pb = (((xb + 1) / 2) * (res[b] - 1)).clamp(0, res[b] - 1).floor().long()
        key = pb * res[a] + pa
        chg = (key[1:] != key[:-1])
        starts = torch.zeros(N, dtype=torch.bool, device=dev); starts[::64] = True
        fl = int((chg | starts[1:]).sum()) + 1
        uniq = torch.unique(key).numel()
        tot += fl
        line += f" {fl/N:5.3f}/{uniq/N:5.3f}"
    print(line + "   (flushes per entry in sorted order with 64-entry chunks / distinct keys per entry)")
print(f"total row-0 flushes {tot/1e6:.2f} M of {30*N/1e6:.2f} M entries")

# ---- what an LDS texel cache per chunk of sorted entries could save: distinct (chunk, texel) pairs vs today's texel-row flushes ----
print("texel-row updates per entry: today's run-length flushes (x2 texels x2 rows) vs distinct texels per chunk of K sorted entries")
for K in (64, 256, 1024):
    tot_now, tot_distinct = 0, 0
    for si, m in enumerate((1, 2, 4, 8, 16)):
        res = [64 * m, 64 * m, 64 * m, 100]
        for q, (a, b) in enumerate(pairs):
            xa, xb = rec[q, :, 1], rec[q, :, 2]
            pa = (((xa + 1) / 2) * (res[a] - 1)).clamp(0, res[a] - 1).floor().long()
            pb = (((xb + 1) / 2) * (res[b] - 1)).clamp(0, res[b] - 1).floor().long()
            pa1, pb1 = (pa + 1).clamp(max=res[a] - 1), (pb + 1).clamp(max=res[b] - 1)
            chunk = torch.arange(N, device=dev) // K
            keys = torch.cat([(chunk * res[b] + y) * res[a] + x for x in (pa, pa1) for y in (pb, pb1)])
            tot_distinct += torch.unique(keys).numel()
            if K == 64:
                key = pb * res[a] + pa
                chg = (key[1:] != key[:-1])
                starts = torch.zeros(N, dtype=torch.bool, device=dev); starts[::64] = True
                tot_now += (int((chg | starts[1:]).sum()) + 1) * 4
    if K == 64:
        print(f"  today: {tot_now / 1e6:.2f} M texel-row flushes")
    print(f"  chunk {K:5d}: {tot_distinct / 1e6:.2f} M distinct (chunk, texel) rows")

# ---- round 4: what a larger re-sort window of pass B would save.  A wave re-sorts ITS window of consecutive sorted entries by this scale's key, so
# the cell flushes it issues = the distinct keys inside the window, summed over the windows (x-carry aside).
print("distinct (cell) keys per window, summed over the windows, per entry -- windows of 256 (pass B today), 1024, 4096 entries; 'all' = distinct keys of the segment")
tot = {256: 0, 1024: 0, 4096: 0, "all": 0}
for si, m in enumerate((1, 2, 4, 8, 16)):
    res = [64 * m, 64 * m, 64 * m, 100]
    line = f"scale {m:2d}:"
    for q, (a, b) in enumerate(pairs):
        xa, xb = rec[q, :, 1], rec[q, :, 2]
        pa = (((xa + 1) / 2) * (res[a] - 1)).clamp(0, res[a] - 1).floor().long()
        pb = (((xb + 1) / 2) * (res[b] - 1)).clamp(0, res[b] - 1).floor().long()
        key = pb * res[a] + pa
        cell = []
        for wnd in (256, 1024, 4096):
            k2 = key.view(-1, wnd) + (torch.arange(N // wnd, device=dev)[:, None] << 40)  # keys made unique per window
            n = torch.unique(k2).numel()
            tot[wnd] += n
            cell.append(n / N)
        u = torch.unique(key).numel()
        tot["all"] += u
        line += " " + "/".join(f"{c:.3f}" for c in cell) + f"/{u / N:.3f}"
    print(line)
print({k: round(v / 1e6, 3) for k, v in tot.items()}, "M cell flushes per step")

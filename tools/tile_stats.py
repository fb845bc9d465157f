#!/usr/bin/env python3
"""Distribution of a training batch's samples over the texel tiles of the finest scale (what the owner-computes kernel sees): entries per tile
and per cell row, per plane.  python tools/tile_stats.py [--train-steps 300] [--tw 16 --th 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=300)
    ap.add_argument("--tw", type=int, default=16)
    ap.add_argument("--th", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    cfg = KPlanesTrainConfig()
    R = 4096
    tr = KPlanesTrainer(cfg, R, dev)
    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 3)
    data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
    images = data["images"]
    M, H, W = images.shape[:3]

    def batch():
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, images)
        return ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=cfg.near_plane,
                                 training=True), target

    for _ in range(args.train_steps):
        tr.train_step(*batch())
    tr.synchronize()
    rays, target = batch()
    tr.forward(rays, tr.random_draws(), 1.0, training=True)
    torch.cuda.synchronize()
    eb = tr.buf["eb"][2]
    mid = (eb[:, :-1] + eb[:, 1:]) / 2
    pos = tr.rays["origins"][:, None, :] + tr.rays["directions"][:, None, :] * mid[..., None]
    a = cfg.aabb_scale
    xyz = (pos + a) / (2 * a) * 2 - 1
    t = (tr.rays["times"].reshape(-1, 1, 1) * 2 - 1).expand(-1, mid.shape[1], 1)
    pts = torch.cat([xyz, t], -1).reshape(-1, 4)
    res = [r * cfg.multiscale_res[-1] for r in cfg.spacetime_resolution[:3]] + [cfg.spacetime_resolution[3]]
    names = ["XY", "XZ", "XT", "YZ", "YT", "ZT"]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    for name, (ax, bx) in zip(names, pairs):
        Wp, Hp = res[ax], res[bx]
        ix = ((pts[:, ax] + 1) / 2 * (Wp - 1)).clamp(0, Wp - 1).floor().long()
        iy = ((pts[:, bx] + 1) / 2 * (Hp - 1)).clamp(0, Hp - 1).floor().long()
        ntx, nty = (Wp + args.tw - 1) // args.tw, (Hp + args.th - 1) // args.th
        tile = (iy // args.th) * ntx + ix // args.tw
        cnt = torch.bincount(tile, minlength=ntx * nty).float()
        rows = torch.bincount(iy * ntx + ix // args.tw, minlength=Hp * ntx).float()  # entries per (cell row, tile column)
        cells = torch.bincount(iy * Wp + ix, minlength=Hp * Wp).float()
        srt = cnt.sort(descending=True).values
        print(f"{name}: tiles {ntx * nty}, non-empty {int((cnt > 0).sum())} ({float((cnt > 0).float().mean()):.2f}), max/tile {int(cnt.max())}, "
              f"99.9% {int(srt[int(0.001 * srt.numel())])}, 99% {int(srt[int(0.01 * srt.numel())])}, median non-empty {int(cnt[cnt > 0].median())}; "
              f"max per cell row of a tile {int(rows.max())}; non-empty cells {int((cells > 0).sum())}, max/cell {int(cells.max())}; "
              f"tiles with < 64 entries {int(((cnt > 0) & (cnt < 64)).sum())}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Step time of the FULL NeRFPlayer preset (458.6 M parameters) through the fused flat-buffer trainer on random rays (frame times as a
100-frame clip has them).  `--profile-steps N`: run only N steps after warm-up (for rocprofv3 --kernel-trace --stats).

    python tools/bench_nerfplayer_full.py [--steps 30] [--warmup 5]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd.nerfplayer import NerfplayerModelConfig  # noqa: E402
from soccernerfs_amd.nerfplayer_full_trainer import NerfplayerFullTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--mlp-operands", default="bf16", choices=["fp32", "bf16"], help="fp32: every net on the fp32 matrix instructions (the parity path); bf16: 16-bit MFMA operands everywhere")
ap.add_argument("--sync-sweeps", action="store_true", help="A/B: the newness / decomposition tables' optimiser sweeps in order on the main stream (round 4) instead of on a side stream")
ap.add_argument("--tiled", action="store_true", help="the newness / decomposition tables through the owner-computes backward + fused Adam (round 6, csrc/tgrid_tiles.hip)")
ap.add_argument("--late-bin", action="store_true", help="A/B (--tiled): binning passes on the caller's stream in front of the tile passes instead of beside the forward")
ap.add_argument("--tiled-hash", action="store_true", help="with --tiled: the static hash grid's table through its owner-computes pass too (csrc/hashgrid_tiles.hip)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = args.rays
tr = NerfplayerFullTrainer(NerfplayerModelConfig(), R, aabb_scale=1.0, device=dev, max_steps=30000, seed=0, async_table_sweeps=not args.sync_sweeps, mlp_operands=args.mlp_operands, tiled_table_backward=args.tiled, tiled_hash_backward=args.tiled_hash)
tr.step = 600
tr.early_bin = not args.late_bin


def step():
    o = (torch.rand(R, 3, device=dev) * 2 - 1) * 0.6
    d = torch.nn.functional.normalize(torch.rand(R, 3, device=dev) * 2 - 1, dim=-1)
    t = torch.floor(torch.rand(R, 1, device=dev) * 100) / 99
    tr.train_step({"origins": o, "directions": d, "times": t}, torch.rand(R, 3, device=dev))


for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"config": "nerfplayer preset, full NeRFPlayer (fused flat-buffer trainer)", "rays": R, "params": int(tr.n_params), "ms_per_step": dt / args.steps * 1e3,
                  "rays_per_s": R * args.steps / dt, "launches_per_step": tr.launches,
                  "async_table_sweeps": not args.sync_sweeps, "tiled_table_backward": bool(args.tiled), "mlp_operands": args.mlp_operands}))

#!/usr/bin/env python3
"""Pass B of the field's quotient scatter ALONE on a real training batch (dev tool).

Trains the preset for --train-steps steps on the synthetic scene (so that the sample distribution is a real one), then repeats the LAST step's
quotient scatter (prepare excluded) with HIP events, for the round-2 kernel (scatter_grouped_kernel, selected through the dev switch
SNERF_PASSB_GROUPED=1, which the launcher reads per call) and the round-4 kernel (scatter_halfwave_kernel): ms per launch for all scales, the
finest scale and the coarser ones, and the distance between the two kernels' gradients.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=300)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--six-scales", action="store_true", help="config 3's plane set")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = KPlanesTrainConfig(multiscale_res=(1, 2, 4, 8, 16, 32) if args.six_scales else (1, 2, 4, 8, 16))
    R = 4096
    tr = KPlanesTrainer(cfg, R, dev)
    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 3)[:4]
    data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
    M, H, W = data["images"].shape[:3]
    for _ in range(args.train_steps):
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=0.0, training=True)
        tr.train_step(rays, target)
    tr.synchronize()
    # the last step's state: sorted records, G, coords are still in the trainer's buffers
    b, ss = tr.buf, tr._ss
    co = ops.coords_from_rays(tr.rays["origins"], tr.rays["directions"], tr.rays["times"].reshape(-1), b["eb"][2], tr.aabb, True)
    ss.sort(co)
    if not tr._qg_step:  # round 3's flow: G from the separate pass; with the quotient epilogue the last step's G is still in ss.G
        ss.quotient_prepare(b["gfeat"], b["feat"])
    assert float(ss.G.abs().max()) > 0
    planes = tr.field_planes.planes
    g = torch.zeros_like(planes)
    ns = len(cfg.multiscale_res)

    def timed(lo, hi):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ss.quotient_scatter_scales(planes, co, b["gfeat"], g, lo, hi)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            ss.quotient_scatter_scales(planes, co, b["gfeat"], g, lo, hi)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    res, grads = {}, {}
    for name, env in (("scatter_grouped (round 2)", "1"), ("scatter_halfwave (round 4)", "0")):
        os.environ["SNERF_PASSB_GROUPED"] = env
        res[name] = {"ms_all_scales": timed(0, ns), "ms_finest": timed(ns - 1, ns), "ms_coarser": timed(0, ns - 1)}
        g.zero_()
        ss.quotient_scatter_scales(planes, co, b["gfeat"], g, 0, ns)
        torch.cuda.synchronize()
        grads[name] = g.clone()
        res[name]["grad_norm"] = float(g.double().norm())
    a, c = list(grads.values())[:2]
    out = {"entries": int(ss.N) * ns * 6, "kernels": res, "rel_l2_between_kernels": float((a - c).double().norm() / a.double().norm()),
           "max_abs_diff": float((a - c).abs().max()), "grad_abs_max": float(a.abs().max()), "fix_rows": int(ss.fix_count.item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

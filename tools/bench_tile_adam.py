#!/usr/bin/env python3
"""Stand-alone timing of the tail of the K-Planes step on real training batches (k-planes preset, synthetic Broadcast-style scene, after
`--train-steps` steps so that the proposal sampler concentrates the samples as it does in training):

  pass B of the quotient scatter per scale | the optimiser sweep (whole / coarse scales) | the owner-computes scatter + Adam kernel per tile shape

Every kernel alone on the GPU (HIP events, 20 launches each); parameters are restored between launches.  Run on the GPU box:
    python tools/bench_tile_adam.py [--train-steps 300] [--start-step 0]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer, cosine_lr_factor  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=300)
    ap.add_argument("--start-step", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/tile_adam_bench.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    cfg = KPlanesTrainConfig()
    R = 4096
    tr = KPlanesTrainer(cfg, R, dev)
    tr.step = args.start_step
    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 3)
    data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
    images = data["images"]
    M, H, W = images.shape[:3]

    def batch():
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, images)
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=cfg.near_plane,
                                 training=True)
        return rays, target

    for _ in range(args.train_steps):
        tr.train_step(*batch())
    tr.synchronize()
    # one more forward + MLP backward so that G, the sort and the fix list describe a real batch; then time the tail's kernels in isolation
    rays, target = batch()
    rng = tr.random_draws()
    tr.forward(rays, rng, 1.0, training=True, defer_render=True)
    tr.tile_adam = False
    tr.backward(target, rng, proposal_grads=False, include_reg=False)
    tr.synchronize()
    ss, ps = tr._ss, tr.field_planes
    co = tr._coords[2]
    ns = len(cfg.multiscale_res)
    o, n = next((o, n) for name, _, _, o, n in tr.segments if name == "field.planes")
    lc = cfg.loss_coefficients
    coefs = tuple(lc[k] for k in ("space_tv_loss", "time_smoothness_loss", "sparse_transients_loss"))
    p_in, p_alt = tr.params[o:o + n], tr._params_alt[o:o + n]
    g, m, v = tr.gviews["field.planes"], tr.mviews["field.planes"], tr.vviews["field.planes"]
    lr = cfg.lr * cosine_lr_factor(tr.step, cfg.warm_up_end, cfg.max_steps, cfg.lr_alpha)
    fin = ns - 1
    lo = tr._finest_offset()
    res = {"train_steps": args.train_steps, "params_finest": n - lo, "params_all": n, "N": ss.N}
    # occupancy of the finest scale: distinct cells per plane
    st = torch.cuda.current_stream().cuda_stream
    g.zero_()
    for s0, s1, name in [(0, ns, "passB_all")] + [(s, s + 1, f"passB_scale{s}") for s in range(ns)] + [(0, fin, "passB_coarse")]:
        res[name + "_ms"] = timeit(lambda: ss.quotient_scatter_scales(ps.planes, co, tr.buf["gfeat"], g, s0, s1))
        g.zero_()
    losses = tr.buf["reg"][0]
    m0, v0 = m.clone(), v.clone()
    res["sweep_all_ms"] = timeit(lambda: ops.adam_planes_step(ps, p_in, p_alt, g, m, v, coefs, losses, tr.step + 1, lr, eps=cfg.adam_eps))
    res["sweep_coarse_ms"] = timeit(lambda: ops.adam_planes_step(ps, p_in, p_alt, g, m, v, coefs, losses, tr.step + 1, lr, eps=cfg.adam_eps, shard_range=(0, lo)))
    res["sweep_finest_ms"] = timeit(lambda: ops.adam_planes_step(ps, p_in, p_alt, g, m, v, coefs, losses, tr.step + 1, lr, eps=cfg.adam_eps, shard_range=(lo, n + 3 & ~3)))
    for shape in (0, 1, 2, 3, 4):
        m.copy_(m0), v.copy_(v0)
        res[f"tile_adam_shape{shape}_ms"] = timeit(lambda: ss.scatter_adam_scale(fin, p_in, p_alt, g, m, v, coefs, losses, tr.step + 1, lr, eps=cfg.adam_eps, tile_shape=shape))
    for dbg in (1, 2, 3):  # timing bisection of the default shape (results are wrong in these modes): no walk / records only / walk without LDS adds
        m.copy_(m0), v.copy_(v0)
        res[f"tile_adam_shape0_debug{dbg}_ms"] = timeit(lambda: ss.scatter_adam_scale(fin, p_in, p_alt, g, m, v, coefs, losses, tr.step + 1, lr, eps=cfg.adam_eps, tile_shape=dbg << 8))
    # the same kernel on an EMPTY batch (every tile takes the streaming path): the kernel's floor
    hist0 = ss.hist.clone()
    ss.hist.fill_(ss.N * 6)  # every cell starts at the end of the records: all counts are zero
    for shape in (0, 1, 2):
        res[f"tile_adam_shape{shape}_empty_ms"] = timeit(lambda: _empty(ss, fin, p_in, p_alt, g, m, v, coefs, losses, tr, lr, cfg, shape))
    ss.hist.copy_(hist0)
    gb = lambda ms, b: round(b / (ms * 1e-3) / 1e12, 2)
    res["sweep_all_TBps_32B"] = gb(res["sweep_all_ms"], 32 * n)
    res["tile_adam_shape0_TBps_24B"] = gb(res["tile_adam_shape0_ms"], 24 * (n - lo))
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


def _empty(ss, fin, p_in, p_alt, g, m, v, coefs, losses, tr, lr, cfg, shape):
    ss.scatter_adam_scale(fin, p_in, p_alt, g, m, v, coefs, losses, tr.step + 1, lr, eps=cfg.adam_eps, tile_shape=shape)


if __name__ == "__main__":
    main()

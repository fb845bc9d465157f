"""Dev tool: where the temporal-grid backward / forward of config 4 spends its time, level by level, and what the gradient scatter's touches look like
(per-ray runs, distinct cells, distinct (cell, time row) pairs, the busiest cell) -- on the stadium scene's camera rays and on random rays.
Uses the dev switch SNERF_TGRID_LEVELS=lo:hi of csrc/tgrid.hip.  Writes one JSON object to stdout."""
import argparse, ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import _lib, ops, synthetic
from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModelConfig
from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--train-steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = args.rays


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def touches(tr, lvl_net, S):
    """per level: samples, per-ray runs (what tgrid_bwd_runs sends / 8), distinct cells, distinct (cell, time row), busiest cell"""
    enc = tr.enc if lvl_net == 2 else tr.prop_enc[lvl_net]
    o, d, t = tr.rays["origins"], tr.rays["directions"], tr.rays["times"].reshape(-1)
    eb = tr.buf["eb"][lvl_net]
    mid = (eb[:, :-1] + eb[:, 1:]) / 2
    x = (o[:, None, :] + d[:, None, :] * mid[..., None] + 1.0) / 2.0  # aabb [-1,1]^3
    inb = ((x >= 0) & (x <= 1)).all(-1)
    n_rows = enc.desc.grid_C - enc.desc.C - 1
    trow = torch.clamp((t * (n_rows - 1)).long(), max=n_rows - 1)
    out = []
    segs = (S + 31) // 32
    run = (S + segs - 1) // segs
    for l in range(enc.desc.L):
        scale = 2.0 ** (l * enc.desc.S) * enc.desc.H - 1.0
        pg = torch.floor(x * scale + 0.5).long()
        cell = pg[..., 0] + pg[..., 1] * 4096 + pg[..., 2] * 4096 * 4096
        cell = torch.where(inb, cell, torch.full_like(cell, -1))
        brk = torch.ones_like(cell, dtype=torch.bool)
        brk[:, 1:] = cell[:, 1:] != cell[:, :-1]
        brk[:, ::run] = True
        runs = int((brk & inb).sum())
        flat = cell[inb]
        uc, cnt = torch.unique(flat, return_counts=True)
        key2 = (cell * 64 + trow[:, None])[inb]
        out.append({"level": l, "res": int(scale) + 2, "samples_in_box": int(inb.sum()), "runs": runs, "cells": int(uc.numel()),
                    "cell_x_timerow": int(torch.unique(key2).numel()), "busiest_cell": int(cnt.max())})
    return out


def measure(tag, make_batch):
    tr = NerfplayerTrainer(NerfplayerNerfactoModelConfig(), R, 3000, aabb_scale=1.0, device=dev, async_field_sweep=False, mlp_operands="bf16")
    tr.step = 600
    for _ in range(args.train_steps):
        tr.train_step(*make_batch())
    torch.cuda.synchronize()
    res = {"batch": tag}
    t = tr.rays["times"].reshape(-1)
    for name, k in (("field", 2), ("prop0", 0), ("prop1", 1)):
        enc = tr.enc if k == 2 else tr.prop_enc[k]
        S, N = tr.S[k], R * tr.S[k]
        gout = tr.buf["gfeat"] if k == 2 else tr.buf["gpfeat"][k]
        if float(gout.abs().sum()) == 0:
            gout.normal_()
        gtab = tr.gviews["field.table" if k == 2 else f"prop{k}.table"]
        table = enc.embeddings
        out = tr.buf["feat"] if k == 2 else tr.buf["pfeat"][k]
        tr._st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        bwd = lambda: tr._tgrid_bwd(enc, tr._coords[k], t, S, N, gout, gtab)
        fwd = lambda: tr._tgrid_fwd(enc, table, tr._coords[k], t, S, N, out)
        r = {"S": S, "bwd_ms": timed(bwd, args.reps), "fwd_ms": timed(fwd, args.reps), "levels": touches(tr, k, S)}
        for l in range(enc.desc.L):
            os.environ["SNERF_TGRID_LEVELS"] = f"{l}:{l + 1}"
            r["levels"][l]["bwd_ms"] = timed(bwd, args.reps)
            r["levels"][l]["fwd_ms"] = timed(fwd, args.reps)
        os.environ.pop("SNERF_TGRID_LEVELS")
        os.environ["SNERF_TGRID_RUNS"] = "0"
        r["bwd_per_sample_ms"] = timed(bwd, args.reps)
        os.environ.pop("SNERF_TGRID_RUNS")
        gtab.zero_()
        res[name] = r
    return res


cams = synthetic.make_stadium_cameras(30, 6, 960, 540)
frame_ids = torch.linspace(0, 99, args.frames).long()
data = synthetic.render_dataset(cams, frame_ids.float() / 99, list(range(30)), dev, chunk_rows=540, variant="stadium")
M, H, W = data["images"].shape[:3]
full_index = (data["cam_id"] * 100 + frame_ids.to(dev).repeat(30)).contiguous()


def stadium_batch():
    idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
    rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"])
    return rays, full_index[idx[:, 0]].contiguous(), target


def random_batch():
    o = (torch.rand(R, 3, device=dev) * 2 - 1) * 0.6
    d = torch.nn.functional.normalize(torch.rand(R, 3, device=dev) * 2 - 1, dim=-1)
    return {"origins": o, "directions": d, "times": torch.rand(R, 1, device=dev)}, torch.randint(0, 3000, (R,), device=dev), torch.rand(R, 3, device=dev)


print(json.dumps({"stadium": measure("stadium camera rays", stadium_batch), "random": measure("random rays", random_batch)}))

"""Dev tool: the tiled (owner-computes) temporal-grid backward against the run-length atomic kernel + dense Adam sweep, on config 4's tables with the
stadium scene's camera rays.  One JSON object on stdout."""
import argparse, ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import _lib, ops, synthetic
from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModelConfig
from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer
from soccernerfs_amd.temporal_grid import TiledTableBackward

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--train-steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = args.rays


def timed(fn, reps=args.reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


cams = synthetic.make_stadium_cameras(30, 6, 960, 540)
frame_ids = torch.linspace(0, 99, args.frames).long()
data = synthetic.render_dataset(cams, frame_ids.float() / 99, list(range(30)), dev, chunk_rows=540, variant="stadium")
M, H, W = data["images"].shape[:3]
full_index = (data["cam_id"] * 100 + frame_ids.to(dev).repeat(30)).contiguous()
tr = NerfplayerTrainer(NerfplayerNerfactoModelConfig(), R, 3000, aabb_scale=1.0, device=dev, async_field_sweep=False, mlp_operands="bf16")
tr.step = 600
for _ in range(args.train_steps):
    idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
    rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"])
    tr.train_step(rays, full_index[idx[:, 0]].contiguous(), target)
torch.cuda.synchronize()
t = tr.rays["times"].reshape(-1)
tr._st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
for name, k in (("field", 2), ("prop0", 0), ("prop1", 1)):
    enc = tr.enc if k == 2 else tr.prop_enc[k]
    S, N = tr.S[k], R * tr.S[k]
    gout = tr.buf["gfeat"] if k == 2 else tr.buf["gpfeat"][k]
    gname = "field.table" if k == 2 else f"prop{k}.table"
    gtab = tr.gviews[gname]
    co = tr._coords[k]
    o, n = next((o, n) for nm, _, _, o, n in tr.segments if nm == gname)
    sl = slice(o, o + n)
    p, m, v = tr.params[sl].clone(), tr.exp_avg[sl].clone(), tr.exp_avg_sq[sl].clone()
    rows_, gc = enc.embeddings.shape
    srow = tr._srow[{2: 0, 0: 1, 1: 2}[k]]
    r = {"S": S, "rows": rows_, "grid_C": gc}
    r["old.bwd_runs_ms"] = timed(lambda: tr._tgrid_bwd(enc, co, t, S, N, gout, gtab))
    gtab.zero_()
    adam_tv = lambda: _lib.check(tr.lib.snerf_adam_step_tv(ops._ptr(p), ops._ptr(gtab), ops._ptr(m), ops._ptr(v), C.c_int64(rows_), gc, 0, 1, ops._ptr(srow), 1e-3, 0.9, 0.999,
                                                           1e-12, 700, 1.0, 1, None, tr._st))
    r["old.adam_tv_ms"] = timed(adam_tv)
    r["old.bwd_then_adam_ms"] = timed(lambda: (tr._tgrid_bwd(enc, co, t, S, N, gout, gtab), adam_tv()))
    for sh, lc in ((0, -1), (0, 0), (7, 0), (0, 2)):
        if sh == 9 and gc > 40:
            continue
        tb = TiledTableBackward(enc, N, tile_rows_log2=sh, first_tiled_level=lc)
        tag = f"tiles[sh={tb.plan.tile_rows_log2},lc={tb.plan.first_tiled_level}]"
        q = {"n_tiles": tb.plan.n_tiles, "lds_bytes": tb.plan.lds_bytes}
        q["bin_ms"] = timed(lambda: tb.bin(co, t, S, gout))
        q["records"] = int(tb.tile_base[tb.plan.n_tiles])
        q["coarse_ms"] = timed(lambda: tb.coarse_levels(co, t, S, gout, gtab)) if tb.plan.first_tiled_level > 0 else 0.0
        gtab.zero_()
        tiles0 = lambda: tb.scatter(gout, gtab)
        q["tiles_accumulate_ms"] = timed(tiles0)
        gtab.zero_()
        fused = lambda: tb.scatter_adam(gout, gtab, p, m, v, 1e-3, 700, 1e-12, tv_cols=(0, 1), srow=srow)
        q["tiles_adam_ms"] = timed(fused)
        q["all_fused_ms"] = timed(lambda: (tb.coarse_levels(co, t, S, gout, gtab), tb.bin(co, t, S, gout), fused()))
        r[tag] = q
        del tb
    res[name] = r
print(json.dumps(res))

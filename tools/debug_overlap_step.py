"""One full train_step from the SAME state / rays / draws: overlapped (default) against single-stream, compared per parameter segment
(parameters and Adam moments after the step).  Dev tool for a suspected stream race with 16-bit MLP operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

dev = torch.device("cuda:0")
op = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(0)
cfg = KPlanesTrainConfig(mlp_operands=op)
R = 4096
tr = KPlanesTrainer(cfg, R, dev)
cams = synthetic.make_cameras(20, 960, 540)
data = synthetic.render_dataset(cams, synthetic.frame_times(100, 3)[:int(os.environ.get('FRAMES', '4'))], list(range(19)), dev, chunk_rows=540)
M, H, W = data["images"].shape[:3]
def batch():
    idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
    return ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=cfg.near_plane), target
for _ in range(int(os.environ.get('TRAIN', '200'))):
    tr.train_step(*batch())
tr.synchronize()

def snap():
    tr.synchronize()
    return dict(p=tr.params.clone(), m=tr.exp_avg.clone(), v=tr.exp_avg_sq.clone(), step=tr.step, ssu=tr._steps_since_update)
def restore(s):
    tr.synchronize()
    tr.params.copy_(s["p"]); tr._params_alt.copy_(s["p"]); tr.exp_avg.copy_(s["m"]); tr.exp_avg_sq.copy_(s["v"]); tr.grads.zero_()
    tr.step, tr._steps_since_update = s["step"], s["ssu"]
    torch.cuda.synchronize()

names = [(n, o, k) for n, _, _, o, k in tr.segments]
worst = {}
for trial in range(int(os.environ.get("TRIALS", "60"))):
    s0 = snap()
    K = int(os.environ.get("CHAIN", "1"))  # consecutive steps without a host synchronisation in between
    batches = [batch() for _ in range(K)]
    rngs = [tr.random_draws() for _ in range(K)]
    out = {}
    for mode in ("serial", "overlap", "serial2"):
        restore(s0)
        tr.overlap = mode == "overlap"
        tr.async_field_adam = mode == "overlap" and not os.environ.get("SYNC_ADAM")
        for (rays, target), rng in zip(batches, rngs):
            tr.train_step(rays, target, rng)
        out[mode] = snap()
    for n, o, k in names:
        for key in ("p", "m", "v"):
            a, b, c = out["serial"][key][o:o + k], out["overlap"][key][o:o + k], out["serial2"][key][o:o + k]
            d_ov = float((a - b).norm() / (a.norm() + 1e-30)); d_fl = float((a - c).norm() / (a.norm() + 1e-30))
            worst[(n, key)] = max(worst.get((n, key), (0.0, 0.0))[0], d_ov), max(worst.get((n, key), (0.0, 0.0))[1], d_fl)
            if d_ov > 50 * max(d_fl, 1e-7):
                print(f"trial {trial} step {s0['step']}: {n}.{key} overlap-vs-serial {d_ov:.3e}  (serial-vs-serial {d_fl:.3e})", flush=True)
    restore(out["serial"]); tr.overlap = True; tr.async_field_adam = True
print(op, {f"{n}.{k}": f"{v[0]:.1e}/{v[1]:.1e}" for (n, k), v in worst.items()})

"""Same state, same rays, same draws: gradient buffer of a backward with the proposal chain on its own stream against the single-stream
backward, per parameter segment.  Dev tool for a suspected stream race (16-bit MLP operands)."""
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
data = synthetic.render_dataset(cams, synthetic.frame_times(100, 3)[:4], list(range(19)), dev, chunk_rows=540)
M, H, W = data["images"].shape[:3]
def batch():
    idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
    return ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=cfg.near_plane), target
for _ in range(300):
    rays, target = batch()
    tr.train_step(rays, target)
tr.synchronize()
names = [(n, o, k) for n, _, _, o, k in tr.segments]
worst = {}
for trial in range(int(os.environ.get('TRIALS', '12'))):
    rays, target = batch()
    rng = tr.random_draws()
    res = {}
    for mode in (False, True):
        tr.overlap = mode
        tr.grads.zero_()
        tr.forward(rays, rng, 1.0, training=True)
        tr.backward(target, rng, proposal_grads=True, include_reg=False, defer_prop_join=False)
        tr.synchronize()
        res[mode] = tr.grads.clone()
    for n, o, k in names:
        a, b = res[False][o:o + k], res[True][o:o + k]
        rel = float((a - b).norm() / (a.norm() + 1e-30))
        if rel > 1e-4 or not bool(torch.isfinite(b).all()):
            print(f"trial {trial}: segment {n} overlapped vs serial rel {rel:.3e} finite {bool(torch.isfinite(b).all())}", flush=True)
        worst[n] = max(worst.get(n, 0.0), rel)
    # and the single-stream backward against itself (atomic-order noise floor)
    tr.overlap = False
    tr.grads.zero_(); tr.forward(rays, rng, 1.0, training=True); tr.backward(target, rng, proposal_grads=True, include_reg=False); tr.synchronize()
    for n, o, k in names:
        a, b = res[False][o:o + k], tr.grads[o:o + k]
        worst["floor:" + n] = max(worst.get("floor:" + n, 0.0), float((a - b).norm() / (a.norm() + 1e-30)))
tr.grads.zero_()
print(op, {k: f"{v:.2e}" for k, v in worst.items()})

"""One training step as the GPU ran it WITHOUT a profiler attached: start / end offsets of the trainer's kernel groups from HIP events (the spans of
KPlanesTrainer.enable_kernel_timing), relative to the step's first kernel.  rocprofv3's kernel trace serialises launches enough to change what overlaps
what (profiles/r05_timeline_step.txt shows the sweep at 0.77 ms where the untraced step measures 0.85); this is the untraced view.  Dev tool.

    python tools/step_offsets.py [--steps 12] [--start-step 0]      prints two consecutive steps (one with, one without proposal update in the early schedule)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--start-step", type=int, default=20)
    ap.add_argument("--interleave-prop-levels", action="store_true")
    ap.add_argument("--sort-before-field-fwd", action="store_true")
    ap.add_argument("--pipeline-sweep", default="off", choices=["coarse_first", "fine_first", "off"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    R = 4096
    cfg = KPlanesTrainConfig(interleave_proposal_levels=args.interleave_prop_levels, sort_before_field_fwd=args.sort_before_field_fwd, pipeline_sweep="" if args.pipeline_sweep == "off" else args.pipeline_sweep)
    tr = KPlanesTrainer(cfg, R, dev)
    tr.step = args.start_step
    cams = synthetic.make_cameras(20, 960, 540)
    data = synthetic.render_dataset(cams, synthetic.frame_times(100, 3)[:2], list(range(19)), dev, chunk_rows=540)
    M, H, W = data["images"].shape[:3]

    def step():
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=cfg.near_plane, training=True)
        tr.train_step(rays, target)

    for _ in range(20):
        step()
    tr.enable_kernel_timing(None)
    marks = []
    for _ in range(args.steps):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        step()
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
    spans = [(name, a, b) for name, evs in tr._timing.items() for a, b in evs]
    t0 = marks[0]
    rows = sorted(((t0.elapsed_time(a), t0.elapsed_time(b), name) for name, a, b in spans), key=lambda r: r[0])
    lo, hi = t0.elapsed_time(marks[-4]), t0.elapsed_time(marks[-2])  # two consecutive steps near the end
    print(f"{args.steps} steps in {t0.elapsed_time(end):.3f} ms = {t0.elapsed_time(end) / args.steps:.3f} ms / step (timing events on); two consecutive steps, offsets in us from the first one's start:")
    for s, e, name in rows:
        if lo <= s < hi + 1.2:
            print(f"  {1e3 * (s - lo):9.1f} {1e3 * (e - lo):9.1f} {1e3 * (e - s):8.1f} us  {name}")
    print("  step boundaries (host enqueue points reached on the main stream):", [round(1e3 * (t0.elapsed_time(m) - lo), 1) for m in marks[-4:]])


if __name__ == "__main__":
    main()

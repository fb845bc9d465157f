#!/usr/bin/env python3
"""How far do repeated runs of the fused full-NeRFPlayer trainer spread around the reference's own 50-step run (G13b)?  The run is chaotic
beyond ~5 steps (Adam moves every parameter by ~lr per step whatever the gradient's size; atomics order differs run to run; now and then a run takes
a visibly different branch from step 4 on), so per-step values are not comparable -- means over windows of steps are.  Prints, per loss term and
window, the reference's mean and the ratio of each run's mean to it: the numbers behind the tolerances of
tests/test_gpu_nerfplayer_full_trainer.py::test_fifty_training_steps_track_the_reference_models_own_run.

    python tools/g13b_spread.py [runs]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_golden  # noqa: E402
from tests.test_gpu_hashgrid import _full_model  # noqa: E402
from tests.test_gpu_nerfplayer_full_trainer import WINDOWS, _pairs  # noqa: E402
from soccernerfs_amd.nerfplayer_full_trainer import NerfplayerFullTrainer  # noqa: E402

DEV = "cuda:0"
g, gb = load_golden("g13_nerfplayer_full"), load_golden("g13b_nerfplayer_dynamics")
steps = int(gb["steps"])
KEYS = ["rgb_loss", "interlevel_loss", "distortion_loss", "temporal_tv_loss", "prob_loss"]


def run():
    model, _ = _full_model(g)
    tr = NerfplayerFullTrainer(model.config, int(g["R"]), aabb_scale=1.0, device=DEV, lr=float(gb["lr0"]), adam_eps=float(gb["eps"]),
                               warm_up_end=int(gb["warm_up_end"]), max_steps=int(gb["max_steps"]), seed=0)
    with torch.no_grad():
        for name, p in _pairs(model).items():
            tr.views[name].copy_(p.detach().reshape(tr.views[name].shape))
    t_ = lambda k: g[k].to(DEV).contiguous()
    rays = {"origins": t_("origins"), "directions": t_("directions"), "times": t_("times")}
    hist = []
    for step in range(steps):
        rng = {"t_rand": gb["t_rand"][step].to(DEV), "u": [gb["u0"][step].to(DEV), gb["u1"][step].to(DEV)], "bg": gb["bg"][step].to(DEV)}
        tr.tv_rows = [int(x) for x in gb["tv_rows"][step]]
        tr.train_step(rays, t_("target"), rng)
        d = {k: float(v) for k, v in tr.loss_dict().items()}
        d["probs0"] = float(tr.rendered_probs().mean(0)[0])
        hist.append(d)
    return hist


runs = [run() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8)]
for k in KEYS:
    for lo, hi in WINDOWS:
        ref = float(gb["loss_" + k][lo:hi].mean())
        ratios = [sum(h[s][k] for s in range(lo, hi)) / (hi - lo) / ref for h in runs]
        print(f"{k:18s} steps {lo:2d}-{hi - 1:2d}  reference mean {ref:.4e}  runs / reference: " + " ".join(f"{r:5.2f}" for r in ratios))
print("static share at the last step: reference", float(gb["probs_mean"][steps - 1][0]), "runs", [round(h[-1]["probs0"], 3) for h in runs])

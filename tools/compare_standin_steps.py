#!/usr/bin/env python3
"""Trajectory parity at the PRESET's size: the HIP trainer and the reference's algorithm in stock PyTorch (oracle/torch_standin.StandinTrainer; checker
code, imported only by this dev tool) start from the same parameters and are fed the SAME pixel batches, rays and uniform draws for K steps.  Per step:
every loss term of both, and the relative distance of the parameter vectors.  What the 3-step oracle tests do on a small model, here on 156 M parameters over
hundreds of steps: a coefficient or a schedule that differs only at scale would show as a drift that grows faster than the float-atomic noise floor, which
the tool measures too (two HIP runs against each other).

    python tools/compare_standin_steps.py --steps 300 --operands fp32 --out gpurun_out/r04_trajectory_fp32.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kplanes_oracle as KO, torch_standin as TS  # noqa: E402  (checker code: this tool compares against it)
from soccernerfs_amd import ops, synthetic  # noqa: E402
from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer  # noqa: E402


def flat_like_standin(tr):
    """The HIP trainer's parameters in the stand-in's order and layout (reference NCHW planes, [out,in] weights)."""
    prop = [t for i in range(2) for t in tr.prop_planes[i].to_reference()[0]] + [w for i in range(2) for w in tr.prop_nets[i].linear_weights()]
    fld = [t for sc in tr.field_planes.to_reference() for t in sc] + list(tr.sigma_net.linear_weights()) + list(tr.color_net.linear_weights())
    return torch.cat([x.reshape(-1) for x in prop + fld])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--operands", default="fp32", choices=["fp32", "bf16"], help="HIP trainer's MLP operands (fp32 = the exact parity path, product-form scatter)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="gpurun_out/trajectory.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    KO.USE_GRID_SAMPLE = True
    R = 4096
    exact = args.operands == "fp32"
    mk = lambda: KPlanesTrainer(KPlanesTrainConfig(mlp_operands=args.operands, seed=args.seed, quotient_scatter=not exact, fused_field=not exact), R, dev)
    hip, hip2 = mk(), mk()
    P0 = KO.make_kplanes_params(seed=args.seed, **TS.PRESET)
    hip.load_oracle_params(P0)
    hip2.load_oracle_params(P0)
    ref = TS.StandinTrainer(dev, R, seed=args.seed)
    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 3)
    data = synthetic.render_dataset(cams, times, list(range(19)), dev, chunk_rows=540)
    M, H, W = data["images"].shape[:3]
    gen = torch.Generator(device=dev).manual_seed(1234 + args.seed)
    S0, S1, S2 = hip.S
    log = {"steps": args.steps, "operands": args.operands, "rows": []}
    rel = lambda a, b: float((a - b).double().norm() / b.double().norm())
    for step in range(args.steps):
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev, generator=gen), M, H, W, data["images"])
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=hip.aabb, near_plane=0.0, training=True)
        rnd = lambda *s: torch.rand(*s, device=dev, generator=gen)
        rng = {"t_rand": rnd(R, S0 + 1), "u": [rnd(R, S1 + 1), rnd(R, S2 + 1)], "bg": rnd(R, 3)}
        for tr in (hip, hip2):
            tr.train_step(rays, target, rng)
        ref.train_step(rays, target, rng)
        if step < 10 or step % 10 == 9:
            a, b, c = ({k: float(v) for k, v in t.loss_dict().items()} for t in (hip, hip2, ref))
            hip.synchronize(); hip2.synchronize()
            pa, pb, pc = flat_like_standin(hip), flat_like_standin(hip2), ref.params
            row = {"step": step + 1, "hip": a, "standin": c, "param_rel_hip_vs_standin": rel(pa, pc), "param_rel_hip_vs_hip": rel(pa, pb),
                   "moved_rel_standin": rel(pc, torch.cat([x.reshape(-1) for x in [t for lv in P0["prop_grids"] for t in lv] + [w for lv in P0["prop_sigma"] for w in lv]
                                                            + [t for sc in P0["field_grids"] for t in sc] + list(P0["field_sigma"]) + list(P0["field_color"])]).to(dev))}
            log["rows"].append(row)
            print(f"step {step + 1:4d}  rgb {a['rgb_loss']:.6f} / {c['rgb_loss']:.6f}  interlevel {a['interlevel_loss']:.3e} / {c['interlevel_loss']:.3e}  "
                  f"dist {a['distortion_loss']:.3e} / {c['distortion_loss']:.3e}  tv {a['space_tv_loss']:.4e} / {c['space_tv_loss']:.4e}  "
                  f"|p_hip - p_ref| / |p_ref| {row['param_rel_hip_vs_standin']:.3e}  (hip vs hip {row['param_rel_hip_vs_hip']:.3e}; moved {row['moved_rel_standin']:.3e})", flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(log, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

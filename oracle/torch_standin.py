"""Stock-PyTorch K-Planes train step -- the "reference rate" stand-in of BASELINE.md section 2.  BASELINE / TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference publishes no rays/s and its CUDA stack (tiny-cuda-nn, nerfacc, a CUDA extension) cannot run on an MI355X, so BASELINE.md
prescribes the denominator of north_star's ">= 10x": "the same algorithm expressed in stock PyTorch-ROCm ops (`F.grid_sample` per plane +
`nn.Linear` MLPs) on the same MI355X, reported next to the HIP path".  This file is that: the oracle's restatement of
KPlanesModel.forward / get_loss_dict (oracle/kplanes_oracle.py, pinned against the reference's own outputs G1-G11) with

* `torch.nn.functional.grid_sample` per plane on the reference's NCHW planes, as NS/utils/interpolation.py:5-33 calls it,
* bias-free `x @ W.t()` Linear stacks (what the tcnn shim defines; optionally under torch.autocast, as the reference trains with
  mixed_precision=True, NS/configs/method_configs.py:489),
* autograd for the backward, `torch.optim.Adam` (lr 1e-2, eps 1e-12; one optimiser, foreach kernels) and the cosine schedule,

run on whatever device it is handed (bench.py: the MI355X, a handful of steps outside the HIP path's timed region; also the CPU).
Only bench.py's baseline leg and tests/ import it."""
import time
from typing import Dict, Optional

import torch

from . import kplanes_oracle as KO

PRESET = dict(base_res=(64, 64, 64, 100), multiscale=(1, 2, 4, 8, 16), prop_res=((128, 128, 128, 100), (256, 256, 256, 100)))
CONFIG1 = dict(base_res=(64, 64, 64, 8), multiscale=(1,), prop_res=((128, 128, 128, 8), (256, 256, 256, 8)))  # BASELINE.json configs[0]


def params_to(P: Dict, device) -> Dict:
    mv = lambda t: t.to(device)
    return {"aabb": mv(P["aabb"]), "field_grids": [[mv(t) for t in sc] for sc in P["field_grids"]], "field_sigma": [mv(w) for w in P["field_sigma"]],
            "field_color": [mv(w) for w in P["field_color"]], "prop_grids": [[mv(t) for t in lv] for lv in P["prop_grids"]],
            "prop_sigma": [[mv(w) for w in lv] for lv in P["prop_sigma"]]}


def time_train_steps(device, rays_per_step: int = 4096, steps: int = 8, warmup: int = 3, model: Dict = PRESET, autocast: Optional[torch.dtype] = None,
                     grid_sample: bool = True, samples=(256, 128, 64)) -> Dict:
    """Seconds per full train step (forward, losses incl. the plane regularisers, autograd backward, Adam) on random rays through the box."""
    device = torch.device(device)
    P = params_to(KO.make_kplanes_params(**model), device)
    leaves = KO.all_param_tensors(P)
    for x in leaves:
        x.requires_grad_(True)
    opt = torch.optim.Adam(leaves, lr=1e-2, eps=1e-12)
    R = rays_per_step
    S0, S1, S2 = samples
    gen = torch.Generator(device=device).manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device=device, generator=gen)
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)
    prev = KO.USE_GRID_SAMPLE
    KO.USE_GRID_SAMPLE = grid_sample
    try:
        t0 = None
        for step in range(warmup + steps):
            if step == warmup:
                sync()
                t0 = time.perf_counter()
            rays = {"origins": (rnd(R, 3) * 2 - 1) * 0.9, "directions": torch.nn.functional.normalize(rnd(R, 3) * 2 - 1, dim=-1), "times": rnd(R, 1)}
            rng = {"t_rand": rnd(R, S0 + 1), "u": [rnd(R, S1 + 1), rnd(R, S2 + 1)], "bg": rnd(R, 3)}
            target = rnd(R, 3)
            for g in opt.param_groups:
                g["lr"] = 1e-2 * KO.cosine_lr_factor(step)
            opt.zero_grad(set_to_none=True)
            with torch.autocast(device_type=device.type, dtype=autocast, enabled=autocast is not None):
                out = KO.kplanes_forward(P, rays, rng, (S0, S1), S2, anneal=KO.anneal_value(step))
                loss = sum(KO.kplanes_loss_dict(P, out, target).values())
            loss.backward()
            opt.step()
        sync()
        dt = (time.perf_counter() - t0) / steps
    finally:
        KO.USE_GRID_SAMPLE = prev
    return {"seconds_per_step": dt, "rays_per_s": R / dt, "steps": steps, "warmup": warmup, "rays_per_step": R, "loss": float(loss),
            "autocast": str(autocast).replace("torch.", "") if autocast is not None else None}

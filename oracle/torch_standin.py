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


def to_channel_last_planes(P: Dict):
    """Re-stores every plane of P channel-last: returns (leaves, rebuild) -- `leaves` = the [H,W,C] tensors that now own the values (what an optimiser
    steps), `rebuild()` = puts fresh [1,C,H,W] VIEWS of them into P (call it before every forward: the views carry the autograd link to the leaves).
    The reference's code sees the same shapes and values; only the memory order behind them changes (KO._bilinear_plane_rows)."""
    slots = [(P["field_grids"], s, i) for s in range(len(P["field_grids"])) for i in range(len(P["field_grids"][s]))]
    slots += [(P["prop_grids"], l, i) for l in range(len(P["prop_grids"])) for i in range(len(P["prop_grids"][l]))]
    leaves = []
    for holder, a, b in slots:
        leaves.append(holder[a][b].detach()[0].permute(1, 2, 0).contiguous())

    def rebuild():
        for (holder, a, b), leaf in zip(slots, leaves):
            holder[a][b] = leaf.permute(2, 0, 1).unsqueeze(0)

    rebuild()
    return leaves, rebuild


def time_train_steps(device, rays_per_step: int = 4096, steps: int = 8, warmup: int = 3, model: Dict = PRESET, autocast: Optional[torch.dtype] = None,
                     grid_sample: bool = True, samples=(256, 128, 64), plane_layout: str = "chw") -> Dict:
    """Seconds per full train step (forward, losses incl. the plane regularisers, autograd backward, Adam) on random rays through the box.
    plane_layout "hwc": planes stored channel-last and gathered as rows (KO._bilinear_plane_rows) instead of F.grid_sample."""
    device = torch.device(device)
    P = params_to(KO.make_kplanes_params(**model), device)
    rebuild = lambda: None
    if plane_layout == "hwc":
        grid_sample = False
        plane_leaves, rebuild = to_channel_last_planes(P)
        leaves = plane_leaves + list(P["field_sigma"]) + list(P["field_color"]) + [w for lv in P["prop_sigma"] for w in lv]
    else:
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
            rebuild()
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


class StandinTrainer:
    """The reference's K-Planes training loop in stock PyTorch, for PSNR@30k runs next to the HIP trainer (tools/train_psnr.py --standin).

    What one `train_step` does follows Trainer.train_iteration (NS/engine/trainer.py:383-412) for the `k-planes` preset
    (NS/configs/method_configs.py:481-560): callbacks (proposal-weight annealing kplanes.py:326-331, the proposal sampler's update
    schedule ray_samplers.py:544-557 with its one-step lag, kplanes.py:340-346) -> forward -> loss dict -> ONE backward ->
    the two optimisers ("proposal_networks", "fields": Adam lr 1e-2 eps 1e-12, cosine schedule with 512 warm-up steps), each skipped
    when one of its gradients is non-finite as GradScaler.step does per optimiser (trainer.py:394-408; no loss scaling is emulated: fp32).
    Interface = what tools/train_psnr.py uses of KPlanesTrainer: R, aabb, cfg.near_plane, step, train_step, forward(training=False),
    loss_dict, skipped_steps, synchronize, params."""

    class _Cfg:
        near_plane = 0.0
        proposal_weights_anneal_max_num_iters = 1000
        proposal_weights_anneal_slope = 10.0

    def __init__(self, device, num_rays: int = 4096, seed: int = 0, max_steps: int = 30000, model: Dict = PRESET, samples=(256, 128, 64),
                 eval_chunk: int = 32768, plane_layout: str = "chw"):
        """plane_layout "hwc": planes stored channel-last, texels fetched as rows (KO._bilinear_plane_rows; KO.USE_GRID_SAMPLE must be False) -- the same
        algorithm at roughly twice the rate, so that 30 000 steps fit one GPU call."""
        self.dev = torch.device(device)
        self.R, self.S, self.max_steps, self.eval_chunk = num_rays, tuple(samples), max_steps, eval_chunk
        self.cfg = self._Cfg()
        self.P = params_to(KO.make_kplanes_params(seed=seed, **model), self.dev)
        self.aabb = self.P["aabb"]
        self._rebuild = lambda: None
        if plane_layout == "hwc":
            n_field = sum(len(sc) for sc in self.P["field_grids"])
            plane_leaves, self._rebuild = to_channel_last_planes(self.P)
            fld_planes, prop_planes = plane_leaves[:n_field], plane_leaves[n_field:]
        else:
            fld_planes, prop_planes = [t for sc in self.P["field_grids"] for t in sc], [t for lv in self.P["prop_grids"] for t in lv]
        prop = prop_planes + [w for lv in self.P["prop_sigma"] for w in lv]
        fld = fld_planes + list(self.P["field_sigma"]) + list(self.P["field_color"])
        for x in prop + fld:
            x.requires_grad_(True)
        self.groups = {"proposal_networks": prop, "fields": fld}
        # fused = one multi-tensor kernel per optimiser (same update rule as the default foreach path)
        self.opts = {k: torch.optim.Adam(v, lr=1e-2, eps=1e-12, **({"fused": True} if self.dev.type == "cuda" else {})) for k, v in self.groups.items()}
        self._one = torch.ones(1, device=self.dev)
        self.step = 0
        self._since = 0
        self._skipped = {k: 0 for k in self.groups}
        self._ld: Dict[str, torch.Tensor] = {}
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)

    @property
    def params(self) -> torch.Tensor:
        return torch.cat([x.detach().reshape(-1) for g in self.groups.values() for x in g])

    def synchronize(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def skipped_steps(self) -> Dict[str, int]:
        return dict(self._skipped)

    def loss_dict(self) -> Dict[str, torch.Tensor]:
        return self._ld

    def train_step(self, rays: Dict[str, torch.Tensor], target: torch.Tensor, rng: Optional[Dict] = None):
        """rng: the step's uniform draws {"t_rand", "u": [.,.], "bg"} (tools/compare_standin_steps.py feeds both trainers the same ones); None = own draws."""
        R, (S0, S1, S2) = self.R, self.S
        step = self.step
        anneal = KO.anneal_value(step)
        sstep = max(step - 1, 0)  # the sampler's counter is set AFTER the iteration (kplanes.py:340-346)
        updated = self._since > KO.update_schedule(sstep) or sstep < 10
        rnd = lambda *s: torch.rand(*s, device=self.dev, generator=self.gen)
        if rng is None:
            rng = {"t_rand": rnd(R, S0 + 1), "u": [rnd(R, S1 + 1), rnd(R, S2 + 1)], "bg": rnd(R, 3)}
        lr = 1e-2 * KO.cosine_lr_factor(step, max_steps=self.max_steps)
        for opt in self.opts.values():
            for g in opt.param_groups:
                g["lr"] = lr
            opt.zero_grad(set_to_none=True)
        self._rebuild()  # channel-last layout: fresh [1,C,H,W] views of the leaves (they carry this step's autograd link)
        out = KO.kplanes_forward(self.P, rays, rng, (S0, S1), S2, anneal=anneal, training=True, proposal_requires_grad=updated)
        ld = KO.kplanes_loss_dict(self.P, out, target)
        sum(ld.values()).backward()
        self._ld = {k: v.detach() for k, v in ld.items()}
        # found_inf per optimiser with GradScaler's own kernel (unscale by 1), then the one host read of the step, as GradScaler.step does
        found = []
        for k, v in self.groups.items():
            f = torch.zeros(1, device=self.dev)
            torch._amp_foreach_non_finite_check_and_unscale_([x.grad for x in v if x.grad is not None], f, self._one)
            found.append(f)
        found = torch.cat(found).tolist()
        for bad, k in zip(found, self.groups):
            if not bad:
                self.opts[k].step()
            else:
                self._skipped[k] += 1
        if updated:
            self._since = 0
        self._since += 1
        self.step += 1
        return out["rgb"].detach()

    @torch.no_grad()
    def forward(self, rays: Dict[str, torch.Tensor], rng, anneal: float, training: bool = False) -> torch.Tensor:
        """Eval render (base_model.py:162-186: eval-mode samplers, 'last_sample' background, clamped rgb)."""
        assert not training
        S0, S1, S2 = self.S
        out = KO.kplanes_forward(self.P, rays, None, (S0, S1), S2, anneal=anneal, training=False, proposal_requires_grad=False)
        return out["rgb"]

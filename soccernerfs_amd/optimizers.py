"""Optimiser for the nerfstudio-shaped (autograd) models: torch.optim.Adam's update through libsnerf's streaming kernel.

The reference builds `torch.optim.Adam(lr, eps=1e-15)` per parameter group (NS/engine/optimizers.py:35-60,
NS/configs/method_configs.py:648-657 for nerfplayer-nerfacto) and calls `zero_grad()` every iteration (trainer.py:386).  On the
1.54 GB temporal hash table that is a 12 GB multi-tensor sweep plus a gradient clear plus a dense gradient allocation per step.
`FusedAdam` does the sweep with `snerf_adam_step` (p, g, m, v read; p, m, v written; g CLEARED in the same pass), so
`zero_grad()` has nothing left to do, and a `TemporalGridEncoder` registered with it scatters its gradient straight into the
persistent `.grad` buffer (`accumulate_into_grad`) instead of into a fresh zero-filled tensor."""
from typing import Iterable

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, encoders: Iterable = ()):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        for group in self.param_groups:
            for p in group["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 HIP tensors")
                if p.requires_grad and p.grad is None:
                    p.grad = torch.zeros_like(p)  # persistent: cleared by the sweep, never reallocated
        for enc in encoders:
            enc.accumulate_into_grad = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, betas, eps = group["lr"], group["betas"], group["eps"]
            for p in group["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                st = self.state[p]
                if not st:
                    st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(p).view(-1), torch.zeros_like(p).view(-1)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p.view(-1), g.view(-1), st["exp_avg"], st["exp_avg_sq"], st["step"], lr, betas=betas, eps=eps, zero_grad=True)
                if g is not p.grad:
                    p.grad.zero_()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        """The sweep already cleared every gradient it consumed; parameters it skipped (no gradient yet) have nothing to clear."""
        return None

"""Renderers with the interface of NS/model_components/renderers.py (RGBRenderer :58-140, AccumulationRenderer
:197-223, DepthRenderer :226-287, MedianRGBRenderer :290-362).  All four are views of ONE fused per-ray kernel
(`ops.render`); `render_all` exposes the fused call for callers that want every output from a single pass."""
from typing import Optional, Union

import torch
from torch import nn

from . import ops
from .rays import RaySamples


def _ebins(ray_samples: RaySamples):
    if ray_samples._compact is not None:
        return ray_samples._compact["ebins"]
    return torch.cat([ray_samples.frustums.starts[..., 0], ray_samples.frustums.ends[..., -1:, 0]], dim=-1).contiguous()


def render_all(rgb, weights, ray_samples: RaySamples, background_color, training: bool, rand_fn=None):
    """rgb [R,S,3], weights [R,S,1] -> dict from ops.render.  background_color: 'random' | 'last_sample' | 'black' | 'white' | tensor."""
    if isinstance(background_color, str) and background_color == "random":
        R = weights.shape[0]
        background_color = rand_fn((R, 3), weights.device) if rand_fn is not None else torch.rand(R, 3, device=weights.device)
    return ops.render(weights[..., 0], rgb, _ebins(ray_samples), background_color, training)


class RGBRenderer(nn.Module):
    def __init__(self, background_color: Union[str, torch.Tensor] = "random") -> None:
        super().__init__()
        self.background_color = background_color
        self.rand_fn = None

    def forward(self, rgb, weights, ray_indices=None, num_rays=None) -> torch.Tensor:
        if ray_indices is not None:
            raise NotImplementedError("packed samples (nerfacc) are not on the proposal-sampler path")
        R, S = weights.shape[:2]
        dummy = torch.zeros(R, S + 1, device=weights.device)
        bg = self.background_color
        if isinstance(bg, str) and bg == "random":
            bg = self.rand_fn((R, 3), weights.device) if self.rand_fn is not None else torch.rand(R, 3, device=weights.device)
        return ops.render(weights[..., 0], rgb, dummy, bg, self.training)["rgb"]


class AccumulationRenderer(nn.Module):
    def forward(self, weights, ray_indices=None, num_rays=None) -> torch.Tensor:
        if ray_indices is not None:
            raise NotImplementedError("packed samples (nerfacc) are not on the proposal-sampler path")
        return torch.sum(weights, dim=-2)


class DepthRenderer(nn.Module):
    def __init__(self, method: str = "median") -> None:
        super().__init__()
        if method not in ("median", "expected"):
            raise NotImplementedError(f"Method {method} not implemented")
        self.method = method

    def forward(self, weights, ray_samples: RaySamples, ray_indices=None, num_rays=None) -> torch.Tensor:
        if ray_indices is not None:
            raise NotImplementedError("packed samples (nerfacc) are not on the proposal-sampler path")
        R, S = weights.shape[:2]
        eb = _ebins(ray_samples)
        out = ops.render(weights[..., 0].detach(), torch.zeros(R, S, 3, device=weights.device), eb, "black", True)
        if self.method == "median":
            return out["depth_median"][:, None]
        steps = (eb[:, :-1] + eb[:, 1:]) / 2
        return torch.clip(out["depth_expected"][:, None], steps.min(), steps.max())


class MedianRGBRenderer(nn.Module):
    def __init__(self, background_color: Union[str, torch.Tensor] = "random") -> None:
        super().__init__()
        self.background_color = background_color

    def forward(self, rgb, weights, ray_indices=None, num_rays=None) -> torch.Tensor:
        R, S = weights.shape[:2]
        out = ops.render(weights[..., 0].detach(), rgb.detach(), torch.zeros(R, S + 1, device=weights.device), "black", self.training)
        return out["median_rgb"][:, None, :]  # [R,1,3]: the reference's shape (renderers.py:319-320)

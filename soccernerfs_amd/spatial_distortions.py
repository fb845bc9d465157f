"""SceneContraction of NS/field_components/spatial_distortions.py:42-89 (positions only; the Gaussian branch belongs to mip-NeRF-style
fields, which this path does not have): f(x) = x for ||x|| < 1, (2 - 1/||x||) x/||x|| otherwise; order = inf contracts onto the cube
[-2, 2]^3, which the K-Planes fields then halve into grid_sample's [-1, 1] (NS/fields/kplanes_field.py:278-280, :438-440).
Elementwise device arithmetic on the sample positions in front of the plane gather (the unbounded option of KPlanesModel, config.bounded =
False; the soccer presets are bounded and derive their sample coordinates inside the gather kernel instead)."""
from typing import Optional, Union

import torch
from torch import nn


class SpatialDistortion(nn.Module):
    """spatial_distortions.py:27-39."""

    def forward(self, positions: torch.Tensor) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError


class SceneContraction(SpatialDistortion):
    def __init__(self, order: Optional[Union[float, int]] = None) -> None:
        super().__init__()
        self.order = order

    def forward(self, positions: torch.Tensor) -> torch.Tensor:
        mag = torch.linalg.norm(positions, ord=self.order, dim=-1)[..., None]
        return torch.where(mag < 1, positions, (2 - (1 / mag)) * (positions / mag))

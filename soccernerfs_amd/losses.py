"""Losses with the function names of NS/model_components/losses.py (the K-Planes / Mip-NeRF-360 subset)."""
from typing import List

import torch
from torch import nn

from . import ops
from .plane_set import PlaneSet

MSELoss = nn.MSELoss
EPS = 1.0e-7


def ray_samples_to_sdist(ray_samples) -> torch.Tensor:
    """losses.py:98-103."""
    if getattr(ray_samples, "_compact", None) is not None:
        return ray_samples._compact["sbins"]
    return torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], dim=-1)


def interlevel_loss(weights_list, ray_samples_list) -> torch.Tensor:
    """losses.py:106-121."""
    return ops.interlevel_loss([w[..., 0] for w in weights_list], [ray_samples_to_sdist(rs) for rs in ray_samples_list])


def distortion_loss(weights_list, ray_samples_list) -> torch.Tensor:
    """losses.py:139-144."""
    return ops.distortion_loss(weights_list[-1][..., 0], ray_samples_to_sdist(ray_samples_list[-1]))


def _as_sets(multi_res_grids) -> List[PlaneSet]:
    return [multi_res_grids] if isinstance(multi_res_grids, PlaneSet) else list(multi_res_grids)


def space_tv_loss(multi_res_grids) -> torch.Tensor:
    """losses.py:383-406.  Argument: a PlaneSet (all scales of one field) or a list of PlaneSets (the proposal fields)."""
    return sum(ops.plane_regularizers(ps)[0] for ps in _as_sets(multi_res_grids))


def time_smoothness_loss(multi_res_grids) -> torch.Tensor:
    """losses.py:409-428."""
    return sum(ops.plane_regularizers(ps)[1] for ps in _as_sets(multi_res_grids))


def sparse_transients_loss(multi_res_grids) -> torch.Tensor:
    """losses.py:431-452."""
    return sum(ops.plane_regularizers(ps)[2] for ps in _as_sets(multi_res_grids))

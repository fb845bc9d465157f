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


class DepthLossType:
    """losses.py:35-40."""
    DS_NERF = 1
    URF = 2


def _ebins_of(ray_samples) -> torch.Tensor:
    if getattr(ray_samples, "_compact", None) is not None:
        return ray_samples._compact["ebins"]
    return torch.cat([ray_samples.frustums.starts[..., 0], ray_samples.frustums.ends[..., -1:, 0]], dim=-1)


def depth_loss(weights, ray_samples, termination_depth, predicted_depth, sigma, directions_norm, is_euclidean: bool,
               depth_loss_type=DepthLossType.DS_NERF) -> torch.Tensor:
    """losses.py:261-311 with the argument list of the reference: DS-NeRF (the K-Planes default, kplanes.py:172) or Urban Radiance Fields."""
    dn = None if is_euclidean else directions_norm.reshape(-1)
    if depth_loss_type == DepthLossType.URF:
        return ops.urf_depth_loss(weights[..., 0], _ebins_of(ray_samples), termination_depth.reshape(-1), predicted_depth, float(sigma), dn)
    if depth_loss_type != DepthLossType.DS_NERF:
        raise NotImplementedError("Provided depth loss type not implemented.")
    return ops.ds_nerf_depth_loss(weights[..., 0], _ebins_of(ray_samples), termination_depth.reshape(-1), float(sigma),
                                  None if is_euclidean else directions_norm.reshape(-1))


def _as_sets(multi_res_grids) -> List[PlaneSet]:
    return [multi_res_grids] if isinstance(multi_res_grids, PlaneSet) else list(multi_res_grids)


def plane_regularizer_terms(multi_res_grids) -> torch.Tensor:
    """[space_tv, time_smoothness, sparse_transients] of a field's plane set(s) from ONE sweep per set (csrc/optim.hip::plane_reg_kernel);
    pass the result as `terms=` to the three functions below when more than one of them is needed in a step."""
    return sum(ops.plane_regularizers(ps) for ps in _as_sets(multi_res_grids))


def space_tv_loss(multi_res_grids, terms: torch.Tensor = None) -> torch.Tensor:
    """losses.py:383-406.  Argument: a PlaneSet (all scales of one field) or a list of PlaneSets (the proposal fields)."""
    return (plane_regularizer_terms(multi_res_grids) if terms is None else terms)[0]


def time_smoothness_loss(multi_res_grids, terms: torch.Tensor = None) -> torch.Tensor:
    """losses.py:409-428."""
    return (plane_regularizer_terms(multi_res_grids) if terms is None else terms)[1]


def sparse_transients_loss(multi_res_grids, terms: torch.Tensor = None) -> torch.Tensor:
    """losses.py:431-452."""
    return (plane_regularizer_terms(multi_res_grids) if terms is None else terms)[2]

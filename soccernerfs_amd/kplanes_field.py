"""K-Planes fields with the interface of NS/fields/kplanes_field.py (KPlanesField :129-370,
KPlanesDensityField :373-463): the MLP decoder (view-independent -- the `k-planes` preset -- or view-dependent) and the linear decoder
(a learned colour basis of the direction contracted with the plane features, one linear layer for the density); bounded or contracted scenes;
no appearance embedding (that branch of the reference cannot run: oracle/gen_golden_field_options.py)."""
import enum
from typing import Optional, Sequence

import torch
from torch import nn

from . import ops
from .plane_set import PlaneSet
from .rays import Frustums, RaySamples
from .scene_colliders import SceneBox
from .tcnn_compat import Network


class FieldHeadNames(enum.Enum):
    """NS/field_components/field_heads.py:28-43 (the heads this path emits; PROBS only from the full NeRFPlayer field)."""

    RGB = "rgb"
    DENSITY = "density"
    PROBS = "probs"


def interpolate_kplanes(pts: torch.Tensor, ms_grids: PlaneSet, concat_features: bool = None, freeze_time_planes: bool = False,
                        freeze_space_planes: bool = False) -> torch.Tensor:
    """interpolate_kplanes (kplanes_field.py:77-126) on a PlaneSet (which already knows concat-vs-sum)."""
    return ops.interpolate_kplanes(pts, ms_grids, freeze_time_planes, freeze_space_planes)


def _pts_from_positions(positions, times, aabb, rescale: bool, spatial_distortion=None):
    if spatial_distortion is not None:  # unbounded scene: contraction onto [-2, 2]^3, halved (kplanes_field.py:278-280, :438-440)
        p = spatial_distortion(positions) / 2.0
    else:
        p = SceneBox.get_normalized_positions(positions, aabb)
        if rescale:
            p = p * 2.0 - 1.0
    t = (times * 2) - 1
    if t.dim() == 2:
        t = t[:, None, :]
    return torch.cat((p, t.expand(*p.shape[:-1], 1)), dim=-1).reshape(-1, 4)


class KPlanesField(nn.Module):
    def __init__(self, aabb, spacetime_resolution: Sequence[int] = (256, 256, 256, 150), feat_dim: int = 16,
                 multiscale_res: Optional[Sequence[int]] = None, concat_features_across_scales: bool = False,
                 linear_decoder: bool = False, linear_decoder_layers: Optional[int] = None, disable_viewing_dependent: bool = True,
                 sigma_net_layers: int = 1,
                 sigma_net_hidden_dim: int = 64, rgb_net_layers: int = 2, rgb_net_hidden_dim: int = 64, use_appearance_embedding: bool = False,
                 spatial_distortion=None, freeze_time_planes: bool = False, freeze_space_planes: bool = False, **_unused) -> None:
        super().__init__()
        self.freeze_time_planes, self.freeze_space_planes = freeze_time_planes, freeze_space_planes
        if use_appearance_embedding:
            # (the reference's appearance-embedding branch, kplanes_field.py:325-346, cannot run with per-sample camera indices: see
            # oracle/gen_golden_field_options.py)
            raise NotImplementedError("built: the MLP decoder (view-independent or view-dependent) and the linear decoder, bounded or contracted scenes")
        self.linear_decoder = linear_decoder
        self.spatial_distortion = spatial_distortion
        self.disable_viewing_dependent = disable_viewing_dependent
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        mult = list(multiscale_res or [1])
        base = list(spacetime_resolution)
        self.grids = PlaneSet(feat_dim, [[r * m for r in base[:3]] + base[3:] for m in mult], concat=concat_features_across_scales)
        self.feature_dim = self.grids.out_dim
        cfg = {"otype": "FullyFusedMLP", "activation": "ReLU"}
        if linear_decoder:
            # kplanes_field.py:219-246: the net learns a basis (in place of spherical harmonics) from the RAW direction, 3 F weights that combine
            # the plane features into RGB; the density is one linear layer on the features.  Both run on libsnerf's dense-layer kernels.
            assert linear_decoder_layers is not None
            self.color_basis = Network(3, 3 * self.feature_dim, {**cfg, "output_activation": "None", "n_neurons": 128,
                                                                 "n_hidden_layers": linear_decoder_layers})
            self.sigma_net = Network(self.feature_dim, 1, {"otype": "CutlassMLP", "activation": "None", "output_activation": "None",
                                                           "n_neurons": 128, "n_hidden_layers": 0})
            return
        self.geo_feat_dim = 15
        self.sigma_net = Network(self.feature_dim, self.geo_feat_dim + 1,
                                 {**cfg, "output_activation": "None", "n_neurons": sigma_net_hidden_dim, "n_hidden_layers": sigma_net_layers})
        self.in_dim_color = self.geo_feat_dim
        if not disable_viewing_dependent:
            # kplanes_field.py:206-216: tcnn SphericalHarmonics degree 4 on the directions shifted to [0, 1]; 16 values in FRONT of the features
            from .tcnn_compat import Encoding
            self.direction_encoder = Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})
            self.in_dim_color += self.direction_encoder.n_output_dims
        self.color_net = Network(self.in_dim_color, 3,
                                 {**cfg, "output_activation": "Sigmoid", "n_neurons": rgb_net_hidden_dim, "n_hidden_layers": rgb_net_layers})

    def _features(self, ray_samples: RaySamples):
        c = ray_samples._compact
        if c is not None and c["times"] is not None and self.spatial_distortion is None:
            return ops.interpolate_kplanes_rays(self.grids, c["origins"], c["directions"], c["times"], c["ebins"], self.aabb, rescale=True,
                                                freeze_time_planes=self.freeze_time_planes, freeze_space_planes=self.freeze_space_planes)
        pts = _pts_from_positions(ray_samples.frustums.get_positions(), ray_samples.times, self.aabb, True, self.spatial_distortion)
        return ops.interpolate_kplanes(pts, self.grids, self.freeze_time_planes, self.freeze_space_planes)

    def get_density(self, ray_samples: RaySamples):
        """kplanes_field.py:275-312 -> (density [R,S,1], geo features [N,15])."""
        n_rays, n_samples = ray_samples.frustums.shape[:2]
        if self.linear_decoder:  # :305-311: the features themselves go on to the colour head
            features = self._features(ray_samples)
            return ops.trunc_exp(self.sigma_net(features)).view(n_rays, n_samples, 1), features
        h, dens = self.sigma_net.forward_with_exp_head(self._features(ray_samples), self.geo_feat_dim)
        return dens.view(n_rays, n_samples, 1), h[:, : self.geo_feat_dim]

    def get_outputs(self, ray_samples: RaySamples, density_embedding=None):
        """kplanes_field.py:314-358."""
        assert density_embedding is not None
        n_rays, n_samples = ray_samples.frustums.shape[:2]
        if self.linear_decoder:  # :349-354
            basis = self.color_basis(ray_samples.frustums.directions.expand(n_rays, n_samples, 3).reshape(-1, 3).contiguous())
            return ops.basis_rgb(density_embedding, basis).view(n_rays, n_samples, 3)
        if self.disable_viewing_dependent:
            return self.color_net(density_embedding).view(n_rays, n_samples, 3)
        # kplanes_field.py:318-323: get_normalized_directions = (d + 1) / 2 (base_field.py:131-137), SH encoding, [encoded directions | features]
        directions = ray_samples.frustums.directions.expand(n_rays, n_samples, 3).reshape(-1, 3)
        enc = self.direction_encoder((directions + 1.0) / 2.0)
        return self.color_net(torch.cat([enc, density_embedding], dim=-1).contiguous()).view(n_rays, n_samples, 3)

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False, mask=None, bg_color=None):
        density, feats = self.get_density(ray_samples)
        return {FieldHeadNames.DENSITY: density, FieldHeadNames.RGB: self.get_outputs(ray_samples, feats)}


class KPlanesDensityField(nn.Module):
    def __init__(self, aabb, resolution, feature_dim, spatial_distortion=None, linear_decoder: bool = False, freeze_time_planes: bool = False,
                 freeze_space_planes: bool = False, **_unused) -> None:
        super().__init__()
        self.freeze_time_planes, self.freeze_space_planes = freeze_time_planes, freeze_space_planes
        self.spatial_distortion = spatial_distortion
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.grids = PlaneSet(feature_dim, [list(resolution)], concat=False, a=0.1, b=0.15)
        # kplanes_field.py:391-407: with the linear decoder the hidden layer loses its ReLU
        self.sigma_net = Network(feature_dim, 1, {"otype": "FullyFusedMLP", "activation": "None" if linear_decoder else "ReLU",
                                                  "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1})

    def density_fn(self, positions, times):
        """kplanes_field.py:410-432: positions [R,S,3] (or [N,3]), times [R,1]."""
        shape = positions.shape[:-1]
        if positions.dim() == 2:
            positions, times = positions[:, None, :], times
        pts = _pts_from_positions(positions, times, self.aabb, rescale=False, spatial_distortion=self.spatial_distortion)  # bounded: [0,1] coordinates, reference quirk (:440)
        _, dens = self.sigma_net.forward_with_exp_head(ops.interpolate_kplanes(pts, self.grids, self.freeze_time_planes, self.freeze_space_planes), 0)
        return dens.view(*shape, 1)

    def density_from_ray_samples(self, ray_samples: RaySamples):
        """Same values as density_fn(ray_samples.frustums.get_positions(), times) with the coordinates derived in-kernel."""
        c = ray_samples._compact
        n_rays, n_samples = ray_samples.frustums.shape[:2]
        f = ops.interpolate_kplanes_rays(self.grids, c["origins"], c["directions"], c["times"], c["ebins"], self.aabb, rescale=False,
                                         freeze_time_planes=self.freeze_time_planes, freeze_space_planes=self.freeze_space_planes)
        _, dens = self.sigma_net.forward_with_exp_head(f, 0)
        return dens.view(n_rays, n_samples, 1)

    def get_density(self, ray_samples: RaySamples):
        if ray_samples._compact is not None and self.spatial_distortion is None:
            return self.density_from_ray_samples(ray_samples), None
        return self.density_fn(ray_samples.frustums.get_positions(), ray_samples.times), None

    def get_outputs(self, ray_samples, density_embedding=None):
        return {}

"""NeRFPlayer-nerfacto fields with the interface of NS/fields/nerfplayer_nerfacto_field.py
(TemporalHashMLPDensityField :50-149, NerfplayerNerfactoField :152-409; no transient / semantic / normal heads -- the
`nerfplayer-nerfacto` preset, NS/configs/method_configs.py:616-660)."""
from typing import Optional

import numpy as np
import torch
from torch import nn

from .kplanes_field import FieldHeadNames
from .rays import RaySamples
from .scene_colliders import SceneBox
from .sh import sh4_from_unit_dirs
from .tcnn_compat import Network
from .temporal_grid import TemporalGridEncoder


class TemporalHashMLPDensityField(nn.Module):
    def __init__(self, aabb, temporal_dim: int = 64, num_layers: int = 2, hidden_dim: int = 64, spatial_distortion=None, num_levels: int = 8,
                 max_res: int = 1024, base_res: int = 16, log2_hashmap_size: int = 18, features_per_level: int = 2) -> None:
        super().__init__()
        if spatial_distortion is not None:
            raise NotImplementedError("scene contraction is disabled in the nerfplayer-nerfacto preset")
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        growth = float(np.exp((np.log(max_res) - np.log(base_res)) / (num_levels - 1)))  # :83
        self.encoding = TemporalGridEncoder(input_dim=3, temporal_dim=temporal_dim, num_levels=num_levels, level_dim=features_per_level,
                                            per_level_scale=growth, base_resolution=base_res, log2_hashmap_size=log2_hashmap_size)
        self.linear = Network(num_levels * features_per_level, 1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                                                    "n_neurons": hidden_dim, "n_hidden_layers": num_layers - 1})

    def density_fn(self, positions, times):
        """:108-131: positions [R,S,3], times [R,1]."""
        shape = positions.shape[:-1]
        p = SceneBox.get_normalized_positions(positions, self.aabb).reshape(-1, 3)
        t = times[:, None].expand(*shape, 1).reshape(-1, 1) if positions.dim() == 3 else times
        _, dens = self.linear.forward_with_exp_head(self.encoding(p.contiguous(), t.contiguous()), 0)
        return dens.view(*shape, 1)

    def density_from_ray_samples(self, ray_samples: RaySamples):
        c = ray_samples._compact
        n_rays, n_samples = ray_samples.frustums.shape[:2]
        x = self.encoding.forward_rays(c["origins"], c["directions"], c["times"], c["ebins"], self.aabb)
        _, dens = self.linear.forward_with_exp_head(x, 0)
        return dens.view(n_rays, n_samples, 1)

    def get_density(self, ray_samples: RaySamples):
        if ray_samples._compact is not None:
            return self.density_from_ray_samples(ray_samples), None
        return self.density_fn(ray_samples.frustums.get_positions(), ray_samples.times[:, 0]), None

    def get_outputs(self, ray_samples, density_embedding=None):
        return {}


class NerfplayerNerfactoField(nn.Module):
    def __init__(self, aabb, num_images: int, num_layers: int = 2, hidden_dim: int = 64, geo_feat_dim: int = 15, temporal_dim: int = 64,
                 num_levels: int = 16, features_per_level: int = 2, log2_hashmap_size: int = 19, num_layers_color: int = 3,
                 hidden_dim_color: int = 64, appearance_embedding_dim: int = 32, use_average_appearance_embedding: bool = False,
                 use_transient_embedding: bool = False, use_semantics: bool = False, use_pred_normals: bool = False,
                 spatial_distortion=None, **_unused) -> None:
        super().__init__()
        if use_transient_embedding or use_semantics or use_pred_normals or spatial_distortion is not None:
            raise NotImplementedError("only the nerfplayer-nerfacto preset heads (density + rgb) are built")
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.geo_feat_dim, self.num_images, self.appearance_embedding_dim = geo_feat_dim, num_images, appearance_embedding_dim
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.embedding_appearance = nn.Embedding(num_images, appearance_embedding_dim)
        self.mlp_base = TemporalGridEncoder(input_dim=3, temporal_dim=temporal_dim, num_levels=num_levels, level_dim=features_per_level,
                                            log2_hashmap_size=log2_hashmap_size, desired_resolution=1024 * float(aabb.max() - aabb.min()))
        cfg = {"otype": "FullyFusedMLP", "activation": "ReLU"}
        self.mlp_base_decode = Network(num_levels * features_per_level, 1 + geo_feat_dim,
                                       {**cfg, "output_activation": "None", "n_neurons": hidden_dim, "n_hidden_layers": num_layers - 1})
        self.mlp_head = Network(16 + geo_feat_dim + appearance_embedding_dim, 3,
                                {**cfg, "output_activation": "Sigmoid", "n_neurons": hidden_dim_color, "n_hidden_layers": num_layers_color - 1})

    def get_density(self, ray_samples: RaySamples):
        """:313-332: h = decode(grid(xyz_normalised, t)); split [1 | geo]; density = trunc_exp(h[..., 0])."""
        n_rays, n_samples = ray_samples.frustums.shape[:2]
        c = ray_samples._compact
        assert ray_samples.times is not None, "Time should be included in the input for NeRFPlayer"
        if c is not None:
            x = self.mlp_base.forward_rays(c["origins"], c["directions"], c["times"], c["ebins"], self.aabb)
        else:
            p = SceneBox.get_normalized_positions(ray_samples.frustums.get_positions(), self.aabb).reshape(-1, 3)
            x = self.mlp_base(p.contiguous(), ray_samples.times.expand(n_rays, n_samples, 1).reshape(-1, 1).contiguous())
        h, dens = self.mlp_base_decode.forward_with_exp_head(x, 0)
        return dens.view(n_rays, n_samples, 1), h[:, 1:]

    def get_outputs(self, ray_samples: RaySamples, density_embedding=None):
        """:334-409: rgb = mlp_head([SH16(dir) | geo | appearance])."""
        assert density_embedding is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        n_rays, n_samples = ray_samples.frustums.shape[:2]
        dirs = ray_samples.frustums.directions[:, 0, :]  # per ray
        d = sh4_from_unit_dirs(dirs)
        cam = ray_samples.camera_indices.reshape(n_rays, -1)[:, 0]
        if self.training:
            app = self.embedding_appearance(cam)
        elif self.use_average_appearance_embedding:
            app = self.embedding_appearance.weight.mean(dim=0)[None, :].expand(n_rays, -1)
        else:
            app = torch.zeros(n_rays, self.appearance_embedding_dim, device=dirs.device)
        expand = lambda v: v[:, None, :].expand(n_rays, n_samples, v.shape[-1]).reshape(n_rays * n_samples, -1)
        h = torch.cat([expand(d), density_embedding.reshape(-1, self.geo_feat_dim), expand(app)], dim=-1)
        return {FieldHeadNames.RGB: self.mlp_head(h).view(n_rays, n_samples, 3)}

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False):
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=emb)
        out[FieldHeadNames.DENSITY] = density
        return out

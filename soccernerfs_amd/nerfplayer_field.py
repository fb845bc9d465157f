"""Full NeRFPlayer field with the interface of NS/fields/nerfplayer_field.py:142-414 (NerfplayerField): a deformation MLP, a static
multiresolution hash grid evaluated at the original and at the deformed positions, two temporal hash grids (newness,
decomposition) and a 3-way softmax that mixes the stationary / deformed / new features before the decode MLPs.

Kernels: static hash grid -> csrc/hashgrid.hip (with the coordinate gradient the deformation MLP trains on); temporal grids ->
csrc/tgrid.hip; MLPs -> csrc/mlp.hip: fused where the shape is instantiated (decomposition 32->64->3, decode 32->64->64->16), chained
dense-layer kernels otherwise (tcnn_compat.Network).  Heads built: density + rgb + probs (no transient / semantic / normal heads, as the
`nerfplayer` preset, NS/configs/method_configs.py:562-614)."""
import torch
from torch import nn

from .kplanes_field import FieldHeadNames
from .rays import RaySamples
from .scene_colliders import SceneBox
from .tcnn_compat import Encoding, Network
from .temporal_grid import TemporalGridEncoder


def shift_directions_for_tcnn(directions):
    """NS/fields/base_field.py:131-137."""
    return (directions + 1.0) / 2.0


class NerfplayerField(nn.Module):
    def __init__(self, aabb, num_images: int, num_layers: int = 3, hidden_dim: int = 64, geo_feat_dim: int = 15, temporal_dim: int = 64,
                 num_levels: int = 16, features_per_level: int = 2, base_resolution: int = 16, log2_hashmap_size: int = 19,
                 num_layers_color: int = 4, hidden_dim_color: int = 64, appearance_embedding_dim: int = 32,
                 use_transient_embedding: bool = False, use_semantics: bool = False, use_pred_normals: bool = False,
                 use_average_appearance_embedding: bool = False, disable_viewing_dependent: bool = False, spatial_distortion=None,
                 **_unused) -> None:
        super().__init__()
        if use_transient_embedding or use_semantics or use_pred_normals or spatial_distortion is not None:
            raise NotImplementedError("only the nerfplayer preset heads (density + rgb + probs, no scene contraction) are built")
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.geo_feat_dim, self.num_images = geo_feat_dim, num_images
        self.appearance_embedding_dim = appearance_embedding_dim
        self.embedding_appearance = nn.Embedding(num_images, appearance_embedding_dim)  # :205 (constructed, not read by get_outputs)
        self.use_average_appearance_embedding = use_average_appearance_embedding
        cfg = {"otype": "FullyFusedMLP", "activation": "ReLU"}
        feature_dim = num_levels * features_per_level
        self.direction_encoding = None if disable_viewing_dependent else Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})
        self.position_encoding = Encoding(3, {"otype": "Frequency", "n_frequencies": 2})  # :223 (unused by the forward pass)
        self.deformation_field = Network(3, 3, {**cfg, "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 3})  # :231
        self.stationary_field = Encoding(3, {"otype": "HashGrid", "n_levels": num_levels, "n_features_per_level": features_per_level,
                                             "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
                                             "per_level_scale": 1.4472692012786865})  # :243
        self.stationary_field_mlp = Network(feature_dim + 1, feature_dim, {**cfg, "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1})
        span = 1024 * float(aabb.max() - aabb.min())
        grid = dict(input_dim=3, temporal_dim=temporal_dim, num_levels=num_levels, level_dim=features_per_level, base_resolution=base_resolution,
                    log2_hashmap_size=log2_hashmap_size, desired_resolution=span)
        self.newness_field = TemporalGridEncoder(**grid)        # :270
        self.decomposition_field = TemporalGridEncoder(**grid)  # :280
        self.decomposition_mlp = Network(feature_dim, 3, {**cfg, "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1})
        self._probs = None
        self.mlp_base_decode = Network(feature_dim, 1 + geo_feat_dim, {**cfg, "output_activation": "None", "n_neurons": hidden_dim,
                                                                      "n_hidden_layers": num_layers - 1})
        in_dim = geo_feat_dim if disable_viewing_dependent else 16 + geo_feat_dim
        self.mlp_head = Network(in_dim, 3, {**cfg, "output_activation": "Sigmoid", "n_neurons": hidden_dim_color,
                                            "n_hidden_layers": num_layers_color - 1})

    def get_density(self, ray_samples: RaySamples):
        """:330-380."""
        positions = SceneBox.get_normalized_positions(ray_samples.frustums.get_positions(), self.aabb)
        shape = positions.shape[:-1]
        p = positions.reshape(-1, 3).contiguous()
        assert ray_samples.times is not None, "Time should be included in the input for NeRFPlayer"
        t = ray_samples.times.expand(*shape, 1).reshape(-1, 1).contiguous()
        # 1. deformation, 2. stationary grid at both positions, decoded with the time appended
        deformed = p + self.deformation_field(p)
        v_stat = self.stationary_field_mlp(torch.cat([self.stationary_field(p), t], dim=-1))
        v_deform = self.stationary_field_mlp(torch.cat([self.stationary_field(deformed), t], dim=-1))
        # 3. newness grid, 4. decomposition grid -> probabilities
        v_new = self.newness_field(p, t)
        probs = torch.softmax(self.decomposition_mlp(self.decomposition_field(p, t)), dim=-1)
        self._probs = probs
        v = probs[:, 0:1] * v_stat + probs[:, 1:2] * v_deform + probs[:, 2:3] * v_new
        h, dens = self.mlp_base_decode.forward_with_exp_head(v, 0)  # trunc_exp (activations.py:25-41)
        return dens.view(*shape, 1), h[:, 1:].reshape(*shape, self.geo_feat_dim)

    def get_outputs(self, ray_samples: RaySamples, density_embedding=None):
        """:382-414."""
        assert density_embedding is not None
        shape = density_embedding.shape[:-1]
        geo = density_embedding.reshape(-1, self.geo_feat_dim)
        if self.direction_encoding is not None:
            directions = ray_samples.frustums.directions.expand(*shape, 3)  # stored per ray ([R,1,3]) on the compact sample layout
            d = self.direction_encoding(shift_directions_for_tcnn(directions).reshape(-1, 3))
            h = torch.cat([d, geo], dim=-1)
        else:
            h = geo
        outputs = {FieldHeadNames.RGB: self.mlp_head(h.contiguous()).view(*shape, 3)}
        if self._probs is not None:
            outputs[FieldHeadNames.PROBS] = self._probs.view(*shape, 3)
            self._probs = None
        return outputs

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False):
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=emb)
        out[FieldHeadNames.DENSITY] = density
        return out

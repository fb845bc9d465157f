"""NeRFPlayer-nerfacto model with the plugin surface of NS/models/nerfplayer_nerfacto.py:108-344 (+ the parts of
NS/models/nerfacto.py it inherits: callbacks :235-264, metrics :309-315)."""
import functools
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import nn

from .kplanes_field import FieldHeadNames
from .losses import MSELoss, distortion_loss, interlevel_loss
from .nerfplayer_nerfacto_field import NerfplayerNerfactoField, TemporalHashMLPDensityField
from .ray_samplers import ProposalNetworkSampler
from .rays import RayBundle
from .renderers import render_all
from .scene_colliders import AABBBoxCollider, SceneBox


@dataclass
class NerfplayerNerfactoModelConfig:
    """nerfplayer_nerfacto.py:62-105 + inherited nerfacto defaults, with the preset's overrides (method_configs.py:641-647)."""

    near_plane: float = 0.05
    far_plane: float = 1000.0
    background_color: str = "random"
    num_levels: int = 16
    features_per_level: int = 2
    log2_hashmap_size: int = 19
    temporal_dim: int = 64
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"hidden_dim": 16, "temporal_dim": 32, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 64},
        {"hidden_dim": 16, "temporal_dim": 32, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256}])
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    use_proposal_weight_anneal: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    use_average_appearance_embedding: bool = True
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 1e-3
    temporal_tv_weight: float = 1.0
    disable_scene_contraction: bool = True
    eval_num_rays_per_chunk: int = 1 << 15


class NerfplayerNerfactoModel(nn.Module):
    def __init__(self, config: NerfplayerNerfactoModelConfig, scene_box: SceneBox, num_train_data: int, **kwargs) -> None:
        super().__init__()
        self.config, self.scene_box, self.num_train_data = config, scene_box, num_train_data
        self.device_indicator_param = nn.Parameter(torch.empty(0))
        self.populate_modules()

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        cfg = self.config
        if not cfg.disable_scene_contraction:
            raise NotImplementedError("the nerfplayer-nerfacto preset disables scene contraction")
        self.field = NerfplayerNerfactoField(self.scene_box.aabb, temporal_dim=cfg.temporal_dim, num_levels=cfg.num_levels,
                                             features_per_level=cfg.features_per_level, log2_hashmap_size=cfg.log2_hashmap_size,
                                             num_images=self.num_train_data,
                                             use_average_appearance_embedding=cfg.use_average_appearance_embedding)
        self.proposal_networks = nn.ModuleList()
        n = cfg.num_proposal_iterations
        for i in range(1 if cfg.use_same_proposal_network else n):
            args = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
            self.proposal_networks.append(TemporalHashMLPDensityField(self.scene_box.aabb, **args))
        nets = [self.proposal_networks[0]] * n if cfg.use_same_proposal_network else list(self.proposal_networks)
        self.density_fns = [net.density_fn for net in nets]
        sched = lambda step: np.clip(np.interp(step, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]), 1, cfg.proposal_update_every)
        self.proposal_sampler = ProposalNetworkSampler(num_nerf_samples_per_ray=cfg.num_nerf_samples_per_ray,
                                                       num_proposal_samples_per_ray=cfg.num_proposal_samples_per_ray,
                                                       num_proposal_network_iterations=n, single_jitter=cfg.use_single_jitter, update_sched=sched)
        self.collider = AABBBoxCollider(self.scene_box)
        self.rgb_loss = MSELoss()
        self.rand_fn = None
        self.tv_row_fn = None  # parity hook: encoder -> table row for get_temporal_tv_loss; None = random (as the reference)

    def set_rand_fn(self, fn):
        self.rand_fn = fn
        for s in (self.proposal_sampler.initial_sampler, self.proposal_sampler.pdf_sampler):
            s.rand_fn = fn

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        return {"proposal_networks": list(self.proposal_networks.parameters()), "fields": list(self.field.parameters())}

    def get_training_callbacks(self, training_callback_attributes=None):
        cfg, cbs = self.config, []
        if cfg.use_proposal_weight_anneal:
            N, b = cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope

            def set_anneal(step):
                frac = np.clip(step / N, 0, 1)
                self.proposal_sampler.set_anneal((b * frac) / ((b - 1) * frac + 1))

            cbs += [("before", set_anneal), ("after", self.proposal_sampler.step_cb)]
        return cbs

    def forward(self, ray_bundle: RayBundle):
        return self.get_outputs(self.collider(ray_bundle))

    def _background_color(self) -> str:
        return self.config.background_color

    def get_outputs(self, ray_bundle: RayBundle):
        """nerfplayer_nerfacto.py:206-256."""
        return self._outputs_and_field(ray_bundle)[0]

    def _outputs_and_field(self, ray_bundle: RayBundle):
        """-> (outputs, field outputs, weights of the final samples)."""
        assert ray_bundle.times is not None, "Time not provided."
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(
            ray_bundle, density_fns=[functools.partial(f, times=ray_bundle.times) for f in self.density_fns])
        fo = self.field(ray_samples)
        weights = ray_samples.get_weights(fo[FieldHeadNames.DENSITY])
        weights_list.append(weights)
        ray_samples_list.append(ray_samples)
        r = render_all(fo[FieldHeadNames.RGB], weights, ray_samples, self._background_color(), self.training, self.rand_fn)
        eb = ray_samples._compact["ebins"]
        steps = (eb[:, :-1] + eb[:, 1:]) / 2
        outputs = {"rgb": r["rgb"], "accumulation": r["accumulation"][:, None],
                   "depth": torch.clip(r["depth_expected"][:, None], steps.min(), steps.max())}  # DepthRenderer("expected") :190
        if self.training:
            outputs["weights_list"], outputs["ray_samples_list"] = weights_list, ray_samples_list
        for i in range(self.config.num_proposal_iterations):
            rs, w = ray_samples_list[i], weights_list[i]
            R, S = w.shape[:2]
            pr = render_all(torch.zeros(R, S, 3, device=w.device), w.detach(), rs, "black", True)
            e = rs._compact["ebins"]
            st = (e[:, :-1] + e[:, 1:]) / 2
            outputs[f"prop_depth_{i}"] = torch.clip(pr["depth_expected"][:, None], st.min(), st.max())
        if ray_bundle.metadata is not None and "directions_norm" in ray_bundle.metadata:
            outputs["directions_norm"] = ray_bundle.metadata["directions_norm"]
        return outputs, fo, weights

    def get_metrics_dict(self, outputs, batch):
        """nerfacto.py:309-315."""
        image = batch["image"].to(self.device)
        m = {"psnr": 10.0 * torch.log10(1.0 / torch.mean((outputs["rgb"] - image) ** 2))}
        if self.training:
            m["distortion"] = distortion_loss(outputs["weights_list"], outputs["ray_samples_list"])
        return m

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        """nerfplayer_nerfacto.py:289-318."""
        cfg = self.config
        image = batch["image"].to(self.device)
        ld = {"rgb_loss": self.rgb_loss(image, outputs["rgb"])}
        if self.training:
            ld["interlevel_loss"] = cfg.interlevel_loss_mult * interlevel_loss(outputs["weights_list"], outputs["ray_samples_list"])
            assert metrics_dict is not None and "distortion" in metrics_dict
            ld["distortion_loss"] = cfg.distortion_loss_mult * metrics_dict["distortion"]
            if cfg.temporal_tv_weight > 0:
                row = (lambda e: None) if self.tv_row_fn is None else self.tv_row_fn
                tv = self.field.mlp_base.get_temporal_tv_loss(row(self.field.mlp_base))
                for net in self.proposal_networks:
                    tv = tv + net.encoding.get_temporal_tv_loss(row(net.encoding))
                ld["temporal_tv_loss"] = tv * cfg.temporal_tv_weight
        return ld

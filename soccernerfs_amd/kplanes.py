"""K-Planes model with the plugin surface of NS/models/kplanes.py (KPlanesModelConfig :67-177, KPlanesModel :180-515):
same config fields, method names, output / loss / metric dict keys and parameter-group names, running on libsnerf."""
import functools
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from .kplanes_field import FieldHeadNames, KPlanesDensityField, KPlanesField
from .losses import (MSELoss, depth_loss, distortion_loss, interlevel_loss, plane_regularizer_terms, space_tv_loss, sparse_transients_loss,
                     time_smoothness_loss)
from .ray_samplers import ProposalNetworkSampler, UniformLinDispPiecewiseSampler, UniformSampler
from .rays import RayBundle
from .renderers import render_all
from .scene_colliders import AABBBoxCollider, NearFarCollider, SceneBox
from .spatial_distortions import SceneContraction


@dataclass
class KPlanesModelConfig:
    """Field-for-field the reference's config (kplanes.py:67-177); defaults are the reference class defaults,
    `k_planes_preset()` applies the `k-planes` method preset (NS/configs/method_configs.py:511-543)."""

    near_plane: float = 0.05
    far_plane: float = 1000.0
    bounded: bool = True
    spacetime_resolution: Sequence[int] = (64, 64, 64, 50)
    feature_dim: int = 32
    multiscale_res: Sequence[int] = (1, 2, 4, 8)
    concat_features_across_scales: bool = True
    linear_decoder: bool = False
    linear_decoder_layers: Optional[int] = 1
    freeze_time_planes: bool = False
    freeze_space_planes: bool = False
    sigma_net_layers: int = 1
    sigma_net_hidden_dim: int = 64
    rgb_net_layers: int = 2
    rgb_net_hidden_dim: int = 64
    background_color_train: str = "random"
    background_color_eval: str = "last_sample"
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"feature_dim": 8, "resolution": [128, 128, 128, 150]}, {"feature_dim": 8, "resolution": [256, 256, 256, 150]}])
    num_nerf_samples_per_ray: int = 48
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 128)
    use_single_jitter: bool = False
    proposal_warmup: int = 5000
    proposal_update_every: int = 5
    use_proposal_weight_anneal: bool = True
    proposal_weights_anneal_max_num_iters: int = 1000
    proposal_weights_anneal_slope: float = 10.0
    disable_viewing_dependent: bool = True
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {
        "rgb_loss": 1.0, "interlevel_loss": 1.0, "distortion_loss": 0.001, "space_tv_loss": 0.0002, "time_smoothness_loss": 0.001,
        "sparse_transients_loss": 0.0001, "space_tv_proposal_loss": 0.0002, "time_smoothness_proposal_loss": 0.00001,
        "sparse_transients_proposal_loss": 0.0001, "depth_loss": 0.05})
    eval_num_rays_per_chunk: int = 32768
    # depth supervision (kplanes.py:162-172): used only when the batch carries "depth_image" (the Broadcast-style default is depth_maps="none")
    is_euclidean_depth: bool = True
    depth_sigma: float = 0.01
    should_decay_sigma: bool = False
    starting_depth_sigma: float = 0.2
    sigma_decay_rate: float = 0.99985
    depth_loss_type: int = 1  # losses.DepthLossType.DS_NERF

    @staticmethod
    def k_planes_preset() -> "KPlanesModelConfig":
        return KPlanesModelConfig(
            multiscale_res=(1, 2, 4, 8, 16), spacetime_resolution=(64, 64, 64, 100), feature_dim=32, concat_features_across_scales=True,
            disable_viewing_dependent=True,
            proposal_net_args_list=[{"feature_dim": 8, "resolution": (128, 128, 128, 100)}, {"feature_dim": 8, "resolution": (256, 256, 256, 100)}],
            sigma_net_layers=1, sigma_net_hidden_dim=128, rgb_net_layers=2, rgb_net_hidden_dim=64,
            num_proposal_samples_per_ray=(256, 128), num_nerf_samples_per_ray=64)


class KPlanesModel(nn.Module):
    def __init__(self, config: KPlanesModelConfig, scene_box: SceneBox, num_train_data: int = 0, **kwargs) -> None:
        super().__init__()
        self.config, self.scene_box, self.num_train_data = config, scene_box, num_train_data
        self.device_indicator_param = nn.Parameter(torch.empty(0))
        self.populate_modules()

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        """kplanes.py:188-309."""
        cfg = self.config
        # unbounded scenes: L-inf scene contraction in every field, near / far collider, piecewise initial sampler (kplanes.py:194,260-281)
        scene_contraction = None if cfg.bounded else SceneContraction(order=float("inf"))
        frozen = dict(freeze_time_planes=cfg.freeze_time_planes, freeze_space_planes=cfg.freeze_space_planes)
        self.field = KPlanesField(self.scene_box.aabb, spatial_distortion=scene_contraction, feat_dim=cfg.feature_dim, spacetime_resolution=cfg.spacetime_resolution,
                                  concat_features_across_scales=cfg.concat_features_across_scales, multiscale_res=cfg.multiscale_res,
                                  linear_decoder=cfg.linear_decoder, linear_decoder_layers=cfg.linear_decoder_layers,
                                  disable_viewing_dependent=cfg.disable_viewing_dependent,
                                  sigma_net_layers=cfg.sigma_net_layers, sigma_net_hidden_dim=cfg.sigma_net_hidden_dim,
                                  rgb_net_layers=cfg.rgb_net_layers, rgb_net_hidden_dim=cfg.rgb_net_hidden_dim, **frozen)
        self.proposal_networks = nn.ModuleList()
        n = cfg.num_proposal_iterations
        if cfg.use_same_proposal_network:
            assert len(cfg.proposal_net_args_list) == 1, "Only one proposal network is allowed."
            net = KPlanesDensityField(self.scene_box.aabb, spatial_distortion=scene_contraction, linear_decoder=cfg.linear_decoder, **frozen,
                                      **cfg.proposal_net_args_list[0])
            self.proposal_networks.append(net)
            self.density_fns = [net.density_fn for _ in range(n)]
        else:
            for i in range(n):
                args = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
                self.proposal_networks.append(KPlanesDensityField(self.scene_box.aabb, spatial_distortion=scene_contraction,
                                                                  linear_decoder=cfg.linear_decoder, **frozen, **args))
            self.density_fns = [net.density_fn for net in self.proposal_networks]

        def update_schedule(step):
            return np.clip(np.interp(step, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]), 1, cfg.proposal_update_every)

        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=cfg.num_nerf_samples_per_ray, num_proposal_samples_per_ray=cfg.num_proposal_samples_per_ray,
            num_proposal_network_iterations=cfg.num_proposal_iterations, single_jitter=cfg.use_single_jitter, update_sched=update_schedule,
            initial_sampler=(UniformSampler if cfg.bounded else UniformLinDispPiecewiseSampler)(single_jitter=cfg.use_single_jitter))
        self.collider = AABBBoxCollider(scene_box=self.scene_box) if cfg.bounded else NearFarCollider(near_plane=cfg.near_plane, far_plane=cfg.far_plane)
        self.rgb_loss = MSELoss()
        self.rand_fn = None  # parity hook: (shape, device) -> uniform draws; None = torch.rand

    def set_rand_fn(self, fn: Callable):
        self.rand_fn = fn
        for s in (self.proposal_sampler.initial_sampler, self.proposal_sampler.pdf_sampler):
            s.rand_fn = fn

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """kplanes.py:311-316."""
        return {"proposal_networks": list(self.proposal_networks.parameters()), "fields": list(self.field.parameters())}

    def get_training_callbacks(self, training_callback_attributes=None):
        """kplanes.py:318-347: returns [(where, fn)] with where in {'before', 'after'} train iteration."""
        cfg = self.config
        cbs = []
        if cfg.use_proposal_weight_anneal:
            N = cfg.proposal_weights_anneal_max_num_iters

            def set_anneal(step):
                frac = np.clip(step / N, 0, 1)
                b = cfg.proposal_weights_anneal_slope
                self.proposal_sampler.set_anneal((b * frac) / ((b - 1) * frac + 1))

            cbs.append(("before", set_anneal))
            cbs.append(("after", self.proposal_sampler.step_cb))
        return cbs

    def forward(self, ray_bundle: RayBundle) -> Dict[str, torch.Tensor]:
        """Model.forward (NS/models/base_model.py:128-139): collider, then get_outputs."""
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        return self.get_outputs(ray_bundle)

    def get_outputs(self, ray_bundle: RayBundle):
        """kplanes.py:349-388."""
        cfg = self.config
        density_fns = self.density_fns
        if ray_bundle.times is not None:
            density_fns = [functools.partial(f, times=ray_bundle.times) for f in density_fns]
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=density_fns)
        field_out = self.field(ray_samples)
        weights = ray_samples.get_weights(field_out[FieldHeadNames.DENSITY])
        weights_list.append(weights)
        ray_samples_list.append(ray_samples)
        bg = cfg.background_color_train if self.training else cfg.background_color_eval
        r = render_all(field_out[FieldHeadNames.RGB], weights, ray_samples, bg, self.training, self.rand_fn)
        outputs = {"rgb": r["rgb"], "accumulation": r["accumulation"][:, None], "depth": r["depth_median"][:, None],
                   "median_rgb": r["median_rgb"][:, None, :]}
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list
        for i in range(cfg.num_proposal_iterations):
            rs, w = ray_samples_list[i], weights_list[i]
            R, S = w.shape[:2]
            pr = render_all(torch.zeros(R, S, 3, device=w.device), w.detach(), rs, "black", True)
            outputs[f"prop_depth_{i}"] = pr["depth_median"][:, None]
        if ray_bundle.metadata is not None and "directions_norm" in ray_bundle.metadata:
            outputs["directions_norm"] = ray_bundle.metadata["directions_norm"]
        return outputs

    def _get_sigma(self) -> float:
        """kplanes.py:508-515: the depth uncertainty, optionally decayed towards config.depth_sigma once per call."""
        if not hasattr(self, "depth_sigma"):
            self.depth_sigma = self.config.starting_depth_sigma if self.config.should_decay_sigma else self.config.depth_sigma
        if self.config.should_decay_sigma:
            self.depth_sigma = max(self.config.sigma_decay_rate * self.depth_sigma, self.config.depth_sigma)
        return self.depth_sigma

    def get_metrics_dict(self, outputs, batch):
        """kplanes.py:390-412: PSNR and, when the batch carries depth maps, the depth loss averaged over the sampling levels."""
        image = batch["image"].to(self.device)
        mse = torch.mean((outputs["rgb"] - image) ** 2)
        metrics_dict = {"psnr": 10.0 * torch.log10(1.0 / mse)}
        if "depth_image" in batch and self.training and self.config.loss_coefficients.get("depth_loss", 0) > 0:
            sigma = self._get_sigma()
            termination_depth = batch["depth_image"].to(self.device)
            n = len(outputs["weights_list"])
            metrics_dict["depth_loss"] = sum(
                depth_loss(weights=outputs["weights_list"][i], ray_samples=outputs["ray_samples_list"][i], termination_depth=termination_depth,
                           predicted_depth=outputs["depth"], sigma=sigma, directions_norm=outputs.get("directions_norm"),
                           is_euclidean=self.config.is_euclidean_depth, depth_loss_type=self.config.depth_loss_type) / n for i in range(n))
        return metrics_dict

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """kplanes.py:414-452, scaled by misc.scale_dict (only keys present in loss_coefficients)."""
        image = batch["image"].to(outputs["rgb"].device)
        coef = self.config.loss_coefficients
        loss_dict = {"rgb_loss": self.rgb_loss(image, outputs["rgb"])}
        if self.training:
            if "distortion_loss" in coef:
                loss_dict["distortion_loss"] = distortion_loss(outputs["weights_list"], outputs["ray_samples_list"])
            if "interlevel_loss" in coef:
                loss_dict["interlevel_loss"] = interlevel_loss(outputs["weights_list"], outputs["ray_samples_list"])
            nerf, prop = self.field.grids, [p.grids for p in self.proposal_networks]
            # one sweep per plane set yields all three regulariser terms (the reference walks every plane once per loss name)
            t_nerf, t_prop = plane_regularizer_terms(nerf), plane_regularizer_terms(prop)
            if "space_tv_loss" in coef:
                loss_dict["space_tv_loss"] = space_tv_loss(nerf, t_nerf)
            if "space_tv_proposal_loss" in coef:
                loss_dict["space_tv_proposal_loss"] = space_tv_loss(prop, t_prop)
            if len(self.config.spacetime_resolution) > 3:
                if "sparse_transients_loss" in coef:
                    loss_dict["sparse_transients_loss"] = sparse_transients_loss(nerf, t_nerf)
                if "sparse_transients_proposal_loss" in coef:
                    loss_dict["sparse_transients_proposal_loss"] = sparse_transients_loss(prop, t_prop)
                if "time_smoothness_loss" in coef:
                    loss_dict["time_smoothness_loss"] = time_smoothness_loss(nerf, t_nerf)
                if "time_smoothness_proposal_loss" in coef:
                    loss_dict["time_smoothness_proposal_loss"] = time_smoothness_loss(prop, t_prop)
            if "depth_image" in batch and coef.get("depth_loss", 0) > 0:
                loss_dict["depth_loss"] = metrics_dict["depth_loss"]  # kplanes.py:448-449
        return {k: v * coef[k] if k in coef else v for k, v in loss_dict.items()}

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, torch.Tensor]:
        """Model.get_outputs_for_camera_ray_bundle (base_model.py:162-186): chunked full-image inference."""
        h, w = camera_ray_bundle.origins.shape[:2]
        n = h * w
        chunks: Dict[str, List[torch.Tensor]] = {}
        for i in range(0, n, self.config.eval_num_rays_per_chunk):
            rb = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + self.config.eval_num_rays_per_chunk)
            for k, v in self.forward(rb).items():
                if isinstance(v, torch.Tensor):
                    chunks.setdefault(k, []).append(v)
        return {k: torch.cat(v).view(h, w, *v[0].shape[1:]) for k, v in chunks.items()}

    def get_image_metrics_and_images(self, outputs: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor]):
        """kplanes.py:454-498: PSNR / SSIM of a full rendered image against the ground truth, plus the images a viewer would log
        (side-by-side rgb, raw accumulation / depth maps -- the reference colour-maps them for display, which is viewer code).
        LPIPS and the RetinaNet-box metrics (DynMetric) need pretrained networks and are not computed."""
        from .metrics import psnr, structural_similarity_index_measure

        image = batch["image"].to(outputs["rgb"].device)
        rgb = outputs["rgb"]
        combined_rgb = torch.cat([image, rgb], dim=1)
        im, pr = torch.moveaxis(image, -1, 0)[None, ...], torch.moveaxis(rgb, -1, 0)[None, ...]
        metrics_dict = {"psnr": float(psnr(im, pr)), "ssim": float(structural_similarity_index_measure(im, pr))}
        images_dict = {"img": combined_rgb, "accumulation": outputs["accumulation"], "depth": outputs["depth"]}
        for i in range(self.config.num_proposal_iterations):
            images_dict[f"prop_depth_{i}"] = outputs[f"prop_depth_{i}"]
        return metrics_dict, images_dict

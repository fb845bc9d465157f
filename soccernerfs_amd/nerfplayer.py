"""Full NeRFPlayer model with the plugin surface of NS/models/nerfplayer.py:64-343 (NerfplayerModelConfig / NerfplayerModel): the
nerfacto proposal sampler over TemporalHashMLPDensityField proposals, NerfplayerField, the decomposition renderer (`probs` output)
and the probability regulariser, on top of what it shares with the nerfacto variant (nerfplayer_nerfacto.py here)."""
from dataclasses import dataclass
from typing import Dict

import torch
from torch import nn

from .kplanes_field import FieldHeadNames
from .losses import interlevel_loss
from .nerfplayer_field import NerfplayerField
from .nerfplayer_nerfacto import NerfplayerNerfactoModel, NerfplayerNerfactoModelConfig
from .rays import RayBundle


class DecompositionRenderer(nn.Module):
    """NS/model_components/renderers.py:422-444: sum_s weights * probs."""

    @classmethod
    def forward(cls, probs, weights, ray_indices=None, num_rays=None):
        if ray_indices is not None or num_rays is not None:
            raise NotImplementedError("packed samples (nerfacc) are not on this path")
        return torch.sum(weights * probs, dim=-2)


@dataclass
class NerfplayerModelConfig(NerfplayerNerfactoModelConfig):
    """nerfplayer.py:64-104 with the `nerfplayer` preset's overrides (method_configs.py:589-599)."""

    log2_hashmap_size: int = 18
    train_background_color: str = "random"
    eval_background_color: str = "white"
    disable_viewing_dependent: bool = True
    use_average_appearance_embedding: bool = True
    prob_reg_loss_mult: float = 0.1


class NerfplayerModel(NerfplayerNerfactoModel):
    config: NerfplayerModelConfig

    def populate_modules(self):
        super().populate_modules()
        cfg = self.config
        self.field = NerfplayerField(self.scene_box.aabb, temporal_dim=cfg.temporal_dim, num_levels=cfg.num_levels,
                                     features_per_level=cfg.features_per_level, log2_hashmap_size=cfg.log2_hashmap_size,
                                     num_images=self.num_train_data, use_average_appearance_embedding=cfg.use_average_appearance_embedding,
                                     disable_viewing_dependent=cfg.disable_viewing_dependent)
        self.renderer_probs = DecompositionRenderer()

    def _background_color(self) -> str:
        return self.config.train_background_color if self.training else self.config.eval_background_color  # :228-231

    def get_outputs(self, ray_bundle: RayBundle):
        """nerfplayer.py:218-283."""
        outputs, fo, weights = self._outputs_and_field(ray_bundle)
        if FieldHeadNames.PROBS in fo:
            outputs["probs"] = self.renderer_probs(probs=fo[FieldHeadNames.PROBS], weights=weights)
        return outputs

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """nerfplayer.py:309-343."""
        cfg = self.config
        image = batch["image"].to(self.device)
        ld = {"rgb_loss": self.rgb_loss(image, outputs["rgb"])}
        if self.training:
            ld["interlevel_loss"] = cfg.interlevel_loss_mult * interlevel_loss(outputs["weights_list"], outputs["ray_samples_list"])
            assert metrics_dict is not None and "distortion" in metrics_dict
            ld["distortion_loss"] = cfg.distortion_loss_mult * metrics_dict["distortion"]
            if cfg.temporal_tv_weight > 0:
                row = (lambda e: None) if self.tv_row_fn is None else self.tv_row_fn
                encs = [self.field.newness_field, self.field.decomposition_field] + [net.encoding for net in self.proposal_networks]
                tv = encs[0].get_temporal_tv_loss(row(encs[0]))
                for e in encs[1:]:
                    tv = tv + e.get_temporal_tv_loss(row(e))
                ld["temporal_tv_loss"] = tv * cfg.temporal_tv_weight / (len(self.proposal_networks) + 2)  # :329-333
            if "probs" in outputs:
                pm = outputs["probs"].view(-1, 3).mean(dim=0)  # 0 = static, 1 = deform, 2 = new
                ld["prob_loss"] = (0.01 * pm[1] + pm[2]) * cfg.prob_reg_loss_mult
        return ld

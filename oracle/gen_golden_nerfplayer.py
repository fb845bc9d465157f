"""G12: the reference's own NeRFPlayer-nerfacto model (NS/models/nerfplayer_nerfacto.py, NS/fields/nerfplayer_nerfacto_field.py) run
end to end on the CPU -- forward, metrics, loss dict, backward -- with explicit random draws, on a small configuration.

Everything executed is the reference's code except its two native dependencies: tiny-cuda-nn (shim: bias-free fp32 Linear stacks +
SH-4, oracle/shims/tinycudann) and the CUDA extension behind `nerfstudio.field_components.cuda` (its two entry points are served by
the oracle's restatement of temporal_gridencoder.cu, oracle/tgrid_oracle.py).  So this fixture pins the MODEL WIRING of §8a P16
(sampler choice, field composition, appearance embedding, renderers, loss terms) against the reference itself; the grid kernel's
arithmetic stays pinned by G9 / G9b / the reference's KAT.

    python oracle/gen_golden_nerfplayer.py        # build container only; writes tests/golden/g12_nerfplayer.npz

TEST INFRASTRUCTURE ONLY (header as oracle/_refimport.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle._refimport import import_reference  # noqa: E402
from oracle import tgrid_oracle as TO  # noqa: E402
from oracle.gen_golden import RandQueue, npy  # noqa: E402

import_reference()
import nerfstudio.field_components.temporal_grid as TG  # noqa: E402


class _OracleBackend:
    """Stand-in for nerfstudio.field_components.cuda (the two functions TemporalGridEncodeFunc calls, temporal_grid.py:82-150)."""

    @staticmethod
    def temporal_grid_encode_forward(inputs, trow, emb, offsets, outputs, B, D, grid_c, C, L, S, H, dy_dx, gridtype, align):
        out = TO.encode(inputs, trow, emb, offsets.tolist(), float(S), int(H), int(gridtype), int(C), bool(align))  # [B, L*C]
        outputs.copy_(out.detach().view(B, L, C).permute(1, 0, 2))

    @staticmethod
    def temporal_grid_encode_backward(grad, inputs, trow, emb, offsets, grad_emb, B, D, grid_c, C, L, S, H, dy_dx, grad_inputs, gridtype, align):
        e = emb.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            out = TO.encode(inputs, trow, e, offsets.tolist(), float(S), int(H), int(gridtype), int(C), bool(align))
        out.backward(grad.permute(1, 0, 2).reshape(B, L * C))  # grad arrives as [L, B, C]
        grad_emb.add_(e.grad)


TG._C = _OracleBackend

import nerfstudio.models.nerfplayer_nerfacto as NPM  # noqa: E402
from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.data.scene_box import SceneBox  # noqa: E402


class _NoMetric:
    def __init__(self, *a, **k):
        pass


NPM.DynMetric = _NoMetric

CFG = dict(disable_scene_contraction=True,  # the nerfplayer-nerfacto preset (method_configs.py:640-647): AABB collider, no contraction
           num_levels=4, features_per_level=2, log2_hashmap_size=10, temporal_dim=8,
           proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 32},
                                   {"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 64}],
           num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8)
TV_ROW = 2
NUM_IMAGES = 5


def main():
    torch.manual_seed(7)
    gen = torch.Generator().manual_seed(7)
    cfg = NPM.NerfplayerNerfactoModelConfig(**CFG)
    model = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=NUM_IMAGES)
    model.train()
    with torch.no_grad():  # the 1e-4 table init gives a featureless field: O(1) tables make the comparison meaningful
        for enc in [model.field.mlp_base] + [p.encoding for p in model.proposal_networks]:
            enc.embeddings.copy_(torch.rand(enc.embeddings.shape, generator=gen) * 2 - 1)
    R = 20
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.4
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    times = torch.rand(R, 1, generator=gen)
    cams = torch.randint(0, NUM_IMAGES, (R, 1), generator=gen)
    target = torch.rand(R, 3, generator=gen)
    S0, S1 = CFG["num_proposal_samples_per_ray"]
    S2 = CFG["num_nerf_samples_per_ray"]
    draws = {"t_rand": torch.rand(R, 1, generator=gen), "u0": torch.rand(R, 1, generator=gen), "u1": torch.rand(R, 1, generator=gen),
             "bg": torch.rand(R, 3, generator=gen)}
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1), camera_indices=cams, times=times)
    anneal = 0.37
    model.proposal_sampler.set_anneal(anneal)
    # fixed table row for the temporal-TV terms (the reference draws it with torch.randint inside get_temporal_tv_loss)
    orig_randint = torch.randint
    torch.randint = lambda *a, **k: torch.tensor([TV_ROW])
    try:
        with RandQueue([draws["t_rand"], draws["u0"], draws["u1"], draws["bg"]]):
            out = model(rb)
        metrics = model.get_metrics_dict(out, {"image": target})
        loss_dict = model.get_loss_dict(out, {"image": target}, metrics)
    finally:
        torch.randint = orig_randint
    loss = sum(loss_dict.values())
    loss.backward()
    g = {"R": R, "anneal": anneal, "tv_row": TV_ROW, "num_images": NUM_IMAGES, "origins": o, "directions": d, "times": times, "cams": cams,
         "target": target, **draws, "rgb": out["rgb"], "accumulation": out["accumulation"], "depth": out["depth"],
         "prop_depth_0": out["prop_depth_0"], "prop_depth_1": out["prop_depth_1"], "loss_total": loss, "psnr": metrics["psnr"],
         "distortion": metrics["distortion"]}
    for i, (w, rs) in enumerate(zip(out["weights_list"], out["ray_samples_list"])):
        g[f"weights_{i}"] = w[..., 0]
        g[f"ebins_{i}"] = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1)
    for k, v in loss_dict.items():
        g["loss_" + k] = v
    names = []
    for name, p in model.named_parameters():
        if not p.requires_grad or p.numel() == 0:
            continue
        names.append(name)
        g["param_" + name] = p
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        g["gsum_" + name] = gr.double().sum()
        g["gabs_" + name] = gr.double().abs().sum()
        g["gprobe_" + name] = gr.flatten()[:: max(1, gr.numel() // 64)][:64]
    g["param_names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "g12_nerfplayer.npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in g.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(names), "parameter tensors; losses",
          {k: float(v) for k, v in loss_dict.items()})


if __name__ == "__main__":
    main()

"""G13: the reference's own full NeRFPlayer model (NS/models/nerfplayer.py, NS/fields/nerfplayer_field.py) run end to end on the CPU --
forward, metrics, loss dict (incl. prob_loss and the averaged temporal TV), backward -- with explicit random draws, on a small
configuration that has both a dense and hashed levels in the static grid.

Everything executed is the reference's code except its native dependencies: tiny-cuda-nn (shim: bias-free fp32 Linear stacks, and the
HashGrid restated in oracle/hashgrid_oracle.py -- PARITY UNPINNED for that encoding, see its header) and the CUDA extension behind
`nerfstudio.field_components.cuda` (served by oracle/tgrid_oracle.py, pinned by G9/G9b/KAT).  The fixture therefore pins the MODEL
WIRING of SURVEY §8f rank 3 (deformation -> static grid at both positions -> time-conditioned MLP, newness / decomposition grids,
softmax mixing, decode + colour heads, DecompositionRenderer, prob_loss, TV averaging) against the reference itself.

    python oracle/gen_golden_nerfplayer_full.py        # build container only; writes tests/golden/g13_nerfplayer_full.npz

TEST INFRASTRUCTURE ONLY (header as oracle/_refimport.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_nerfplayer as G12  # noqa: E402  (imports the reference, installs the temporal-grid backend)
from oracle.gen_golden import RandQueue, npy  # noqa: E402

import nerfstudio.models.nerfplayer as NP  # noqa: E402
from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.data.scene_box import SceneBox  # noqa: E402

NP.DynMetric = G12._NoMetric

CFG = dict(disable_scene_contraction=True, num_levels=4, features_per_level=2, log2_hashmap_size=12, temporal_dim=8,
           proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 32},
                                   {"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 64}],
           num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8, prob_reg_loss_mult=0.1, depth_weight=0.0)
TV_ROW = 2
NUM_IMAGES = 5


def main():
    torch.manual_seed(11)
    gen = torch.Generator().manual_seed(11)
    cfg = NP.NerfplayerModelConfig(**CFG)
    model = cfg.setup(scene_box=SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=NUM_IMAGES)
    model.train()
    f = model.field
    with torch.no_grad():  # O(1) tables so that every branch contributes measurably (the 1e-4 init gives a featureless field)
        for enc in [f.newness_field, f.decomposition_field] + [p.encoding for p in model.proposal_networks]:
            enc.embeddings.copy_(torch.rand(enc.embeddings.shape, generator=gen) * 2 - 1)
        f.stationary_field.params.copy_(torch.rand(f.stationary_field.params.shape, generator=gen) * 2 - 1)
        for l in f.deformation_field.layers:  # a visible deformation (some deformed points leave [0,1]: tcnn wraps, no bounds check)
            l.weight.mul_(1.5)
    R = 20
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.4
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    times = torch.rand(R, 1, generator=gen)
    cams = torch.randint(0, NUM_IMAGES, (R, 1), generator=gen)
    target = torch.rand(R, 3, generator=gen)
    draws = {"t_rand": torch.rand(R, 1, generator=gen), "u0": torch.rand(R, 1, generator=gen), "u1": torch.rand(R, 1, generator=gen),
             "bg": torch.rand(R, 3, generator=gen)}
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1), camera_indices=cams, times=times)
    anneal = 0.41
    model.proposal_sampler.set_anneal(anneal)
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
         "target": target, **draws, "rgb": out["rgb"], "accumulation": out["accumulation"], "depth": out["depth"], "probs": out["probs"],
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
    path = os.path.join(ROOT, "tests", "golden", "g13_nerfplayer_full.npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in g.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(names), "parameter tensors; losses",
          {k: float(v) for k, v in loss_dict.items()})
    print({n: float(g["gabs_" + n]) for n in names})


if __name__ == "__main__":
    main()

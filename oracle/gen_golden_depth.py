"""G8b: depth supervision vectors captured from the REFERENCE itself (TEST INFRASTRUCTURE; runs only where /root/reference exists).

Calls nerfstudio.model_components.losses.depth_loss (DS_NERF branch -> ds_nerf_depth_loss, losses.py:213-235,261-311, and the URF branch ->
urban_radiance_field_depth_loss, :238-274) on explicit
weights / bins / termination depths, for Euclidean and z-distance depth maps, and stores values and gradients w.r.t. the weights.

    python -m oracle.gen_golden_depth        ->  tests/golden/g8b_depth.npz
"""
import os

import numpy as np
import torch

from oracle._refimport import import_reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g8b_depth.npz")


def main():
    import_reference()
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from nerfstudio.model_components.losses import DepthLossType, depth_loss

    gen = torch.Generator().manual_seed(808)
    R, S = 24, 17
    nears = torch.rand(R, 1, generator=gen) * 0.3 + 0.05
    bins = nears + torch.cumsum(torch.rand(R, S + 1, generator=gen) * 0.2 + 0.01, dim=1)
    starts, ends = bins[:, :-1, None], bins[:, 1:, None]
    o = torch.zeros(R, S, 3)
    rs = RaySamples(frustums=Frustums(origins=o, directions=torch.ones_like(o), starts=starts, ends=ends, pixel_area=torch.ones(R, S, 1)))
    weights = (torch.rand(R, S, 1, generator=gen) ** 3)
    weights[3] = 0.0            # -log(0 + EPS)
    weights[5, 4:9] = 1e-9
    term = torch.rand(R, 1, generator=gen) * 2.5 + 0.2
    term[[2, 11, 19]] = 0.0     # masked rays (no depth)
    term[7] = -1.0
    dnorm = torch.rand(R, 1, generator=gen) * 0.4 + 0.9
    g = {"bins": bins, "weights": weights[..., 0], "termination_depth": term[:, 0], "directions_norm": dnorm[:, 0]}
    for tag, sigma, eucl in (("eucl_s001", 0.01, True), ("eucl_s02", 0.2, True), ("z_s02", 0.2, False)):
        w = weights.clone().requires_grad_(True)
        val = depth_loss(weights=w, ray_samples=rs, termination_depth=term, predicted_depth=torch.zeros(R, 1), sigma=torch.tensor([sigma]),
                         directions_norm=dnorm, is_euclidean=eucl, depth_loss_type=DepthLossType.DS_NERF)
        val.backward()
        g[f"loss_{tag}"], g[f"grad_{tag}"], g[f"sigma_{tag}"] = val.detach(), w.grad[..., 0], torch.tensor(sigma)
    # Urban Radiance Fields variant (losses.py:238-274): values + gradients w.r.t. the weights AND the predicted depth
    pred = term + (torch.rand(R, 1, generator=gen) - 0.5) * 0.3
    g["predicted_depth"] = pred[:, 0]
    for tag, sigma, eucl in (("urf_eucl_s02", 0.2, True), ("urf_z_s05", 0.5, False), ("urf_eucl_s001", 0.01, True)):
        w = weights.clone().requires_grad_(True)
        pd = pred.clone().requires_grad_(True)
        val = depth_loss(weights=w, ray_samples=rs, termination_depth=term, predicted_depth=pd, sigma=torch.tensor([sigma]), directions_norm=dnorm,
                         is_euclidean=eucl, depth_loss_type=DepthLossType.URF)
        val.backward()
        g[f"loss_{tag}"], g[f"grad_{tag}"], g[f"gpred_{tag}"], g[f"sigma_{tag}"] = val.detach(), w.grad[..., 0], pd.grad[:, 0], torch.tensor(sigma)
    np.savez_compressed(OUT, **{k: v.numpy() for k, v in g.items()})
    print("wrote", OUT, {k: tuple(v.shape) for k, v in g.items()})


if __name__ == "__main__":
    main()

"""G6d: KPlanesField / KPlanesDensityField with linear_decoder=True (NS/fields/kplanes_field.py:219-246 constructor, :305-311 density,
:349-354 colour, :391-407 proposal field) evaluated by the REFERENCE's own classes (imported through oracle/_refimport.py with the shims of
SURVEY Appendix A): outputs AND the gradient of a fixed weighted sum of them with respect to every parameter.  Two field shapes: F = 64
(2 scales) with a one-hidden-layer basis net, and F = 160 (5 scales: the density layer has more than 128 inputs, the basis net 480 outputs)
with two hidden layers.

TEST INFRASTRUCTURE.  Run in the container that holds /root/reference:   python -m oracle.gen_golden_linear_decoder
Writes tests/golden/g6d_linear_decoder.npz (inputs, every parameter tensor, outputs, gradients)."""
import os

import numpy as np
import torch

from oracle._refimport import import_reference

CASES = (("a", dict(multiscale_res=[1, 2], linear_decoder_layers=1)), ("b", dict(multiscale_res=[1, 2, 3, 4, 5], linear_decoder_layers=2)))


def main():
    import_reference()
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from nerfstudio.field_components.field_heads import FieldHeadNames
    from nerfstudio.fields.kplanes_field import KPlanesDensityField, KPlanesField

    torch.manual_seed(11)
    gen = torch.Generator().manual_seed(11)
    aabb = torch.tensor([[-1.2, -1.0, -0.8], [1.2, 1.0, 0.8]])
    out = {"aabb": aabb.numpy()}
    R, S = 7, 9
    pos = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.05
    dirs = torch.nn.functional.normalize(torch.rand(R, 1, 3, generator=gen) * 2 - 1, dim=-1).expand(R, S, 3).contiguous()
    tms = torch.rand(R, 1, generator=gen)
    w_rgb, w_den = torch.rand(R, S, 3, generator=gen) * 2 - 1, torch.rand(R, S, generator=gen) * 2 - 1
    out.update(positions=pos.numpy(), directions=dirs.numpy(), times=tms.numpy(), w_rgb=w_rgb.numpy(), w_density=w_den.numpy())
    rs = RaySamples(frustums=Frustums(origins=pos, directions=dirs, starts=torch.zeros(R, S, 1), ends=torch.zeros(R, S, 1), pixel_area=torch.ones(R, S, 1)),
                    camera_indices=torch.zeros(R, S, 1, dtype=torch.long), times=tms[:, None])
    for tag, kw in CASES:
        f = KPlanesField(aabb, spacetime_resolution=[6, 5, 4, 3], feat_dim=32, concat_features_across_scales=True, linear_decoder=True,
                         disable_viewing_dependent=True, use_appearance_embedding=False, **kw)
        with torch.no_grad():
            for p in f.parameters():
                if p.dim() == 4:  # planes: away from the uniform(0.1, 0.5) / ones initialisation
                    p.copy_(torch.rand(p.shape, generator=gen) * 1.2 - 0.1)
        o = f(rs)
        den, rgb = o[FieldHeadNames.DENSITY][..., 0], o[FieldHeadNames.RGB]
        ((w_rgb * rgb).sum() + (w_den * den).sum()).backward()
        out[f"{tag}_density"], out[f"{tag}_rgb"] = den.detach().numpy(), rgb.detach().numpy()
        for s, pl in enumerate(f.grids):
            for q, g in enumerate(pl):
                out[f"{tag}_plane_{s}_{q}"], out[f"{tag}_g_plane_{s}_{q}"] = g.detach().numpy(), g.grad.numpy()
        for i, l in enumerate(f.sigma_net.layers):
            out[f"{tag}_sigma_{i}"], out[f"{tag}_g_sigma_{i}"] = l.weight.detach().numpy(), l.weight.grad.numpy()
        for i, l in enumerate(f.color_basis.layers):
            out[f"{tag}_basis_{i}"], out[f"{tag}_g_basis_{i}"] = l.weight.detach().numpy(), l.weight.grad.numpy()
    # the proposal field: a 8 -> 64 -> 1 net whose hidden layer has no activation
    df = KPlanesDensityField(aabb, resolution=[8, 7, 6, 3], feature_dim=8, linear_decoder=True)
    with torch.no_grad():
        for p in df.parameters():
            if p.dim() == 4:
                p.copy_(torch.rand(p.shape, generator=gen) * 0.9 - 0.2)
    den = df.density_fn(pos, tms)[..., 0]
    (w_den * den).sum().backward()
    out["prop_density"] = den.detach().numpy()
    for q, g in enumerate(df.grids):
        out[f"prop_plane_{q}"], out[f"prop_g_plane_{q}"] = g.detach().numpy(), g.grad.numpy()
    for i, l in enumerate(df.sigma_net.layers):
        out[f"prop_sigma_{i}"], out[f"prop_g_sigma_{i}"] = l.weight.detach().numpy(), l.weight.grad.numpy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6d_linear_decoder.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: v.shape for k, v in out.items() if "rgb" in k or "sigma" in k or "basis" in k})


if __name__ == "__main__":
    main()

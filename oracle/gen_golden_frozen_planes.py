"""G6e: KPlanesField / KPlanesDensityField with freeze_time_planes / freeze_space_planes (NS/fields/kplanes_field.py:95-107: the time planes
are SKIPPED -- the feature is the product of the three space planes alone --, resp. the space planes are interpolated with autograd off and
receive no gradient from the samples) evaluated by the REFERENCE's own classes (imported through oracle/_refimport.py with the shims of
SURVEY Appendix A): outputs and the gradient of a fixed weighted sum with respect to every plane.

TEST INFRASTRUCTURE.  Run in the container that holds /root/reference:   python -m oracle.gen_golden_frozen_planes
Writes tests/golden/g6e_frozen_planes.npz."""
import os

import numpy as np
import torch

from oracle._refimport import import_reference


def main():
    import_reference()
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from nerfstudio.field_components.field_heads import FieldHeadNames
    from nerfstudio.fields.kplanes_field import KPlanesDensityField, KPlanesField

    torch.manual_seed(13)
    gen = torch.Generator().manual_seed(13)
    aabb = torch.tensor([[-1.2, -1.0, -0.8], [1.2, 1.0, 0.8]])
    out = {"aabb": aabb.numpy()}
    R, S = 6, 8
    pos = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.05
    dirs = torch.nn.functional.normalize(torch.rand(R, 1, 3, generator=gen) * 2 - 1, dim=-1).expand(R, S, 3).contiguous()
    tms = torch.rand(R, 1, generator=gen)
    w_rgb, w_den = torch.rand(R, S, 3, generator=gen) * 2 - 1, torch.rand(R, S, generator=gen) * 2 - 1
    out.update(positions=pos.numpy(), directions=dirs.numpy(), times=tms.numpy(), w_rgb=w_rgb.numpy(), w_density=w_den.numpy())
    rs = RaySamples(frustums=Frustums(origins=pos, directions=dirs, starts=torch.zeros(R, S, 1), ends=torch.zeros(R, S, 1), pixel_area=torch.ones(R, S, 1)),
                    camera_indices=torch.zeros(R, S, 1, dtype=torch.long), times=tms[:, None])
    f = KPlanesField(aabb, spacetime_resolution=[6, 5, 4, 3], feat_dim=32, multiscale_res=[1, 2], concat_features_across_scales=True, linear_decoder=False,
                     disable_viewing_dependent=True, use_appearance_embedding=False, sigma_net_layers=1, sigma_net_hidden_dim=128, rgb_net_layers=2,
                     rgb_net_hidden_dim=64)
    df = KPlanesDensityField(aabb, resolution=[8, 7, 6, 3], feature_dim=8, linear_decoder=False)
    with torch.no_grad():
        for m in (f, df):
            for p in m.parameters():
                if p.dim() == 4:
                    p.copy_(torch.rand(p.shape, generator=gen) * 1.2 - 0.1)
    for s, pl in enumerate(f.grids):
        for q, g in enumerate(pl):
            out[f"plane_{s}_{q}"] = g.detach().numpy()
    for i, l in enumerate(f.sigma_net.layers):
        out[f"sigma_{i}"] = l.weight.detach().numpy()
    for i, l in enumerate(f.color_net.layers):
        out[f"color_{i}"] = l.weight.detach().numpy()
    for q, g in enumerate(df.grids):
        out[f"prop_plane_{q}"] = g.detach().numpy()
    for i, l in enumerate(df.sigma_net.layers):
        out[f"prop_sigma_{i}"] = l.weight.detach().numpy()
    for tag, ft, fs in (("time", True, False), ("space", False, True)):
        f.freeze_time_planes = df.freeze_time_planes = ft
        f.freeze_space_planes = df.freeze_space_planes = fs
        for m in (f, df):
            m.zero_grad(set_to_none=True)
        o = f(rs)
        den, rgb = o[FieldHeadNames.DENSITY][..., 0], o[FieldHeadNames.RGB]
        ((w_rgb * rgb).sum() + (w_den * den).sum()).backward()
        out[f"{tag}_density"], out[f"{tag}_rgb"] = den.detach().numpy(), rgb.detach().numpy()
        for s, pl in enumerate(f.grids):
            for q, g in enumerate(pl):
                out[f"{tag}_g_plane_{s}_{q}"] = (g.grad if g.grad is not None else torch.zeros_like(g)).numpy()
        pd = df.density_fn(pos, tms)[..., 0]
        (w_den * pd).sum().backward()
        out[f"{tag}_prop_density"] = pd.detach().numpy()
        for q, g in enumerate(df.grids):
            out[f"{tag}_prop_g_plane_{q}"] = (g.grad if g.grad is not None else torch.zeros_like(g)).numpy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6e_frozen_planes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))
    for tag in ("time", "space"):
        print(tag, [float(np.abs(out[f"{tag}_g_plane_0_{q}"]).sum()) for q in range(6)], [float(np.abs(out[f"{tag}_prop_g_plane_{q}"]).sum()) for q in range(6)])


if __name__ == "__main__":
    main()

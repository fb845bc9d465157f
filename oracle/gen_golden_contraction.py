"""G6c: the unbounded-scene option of the K-Planes path (KPlanesModelConfig.bounded = False, NS/models/kplanes.py:194,260-281): L-inf
SceneContraction (NS/field_components/spatial_distortions.py:42-89) in front of KPlanesField and KPlanesDensityField
(NS/fields/kplanes_field.py:278-280, :438-440), evaluated by the REFERENCE's own classes on sample positions inside and far outside the
unit cube.  TEST INFRASTRUCTURE.  Run where /root/reference exists:   python -m oracle.gen_golden_contraction
Writes tests/golden/g6c_contraction.npz (inputs, every parameter tensor, outputs)."""
import os

import numpy as np
import torch

from oracle._refimport import import_reference


def main():
    import_reference()
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from nerfstudio.field_components.field_heads import FieldHeadNames
    from nerfstudio.field_components.spatial_distortions import SceneContraction
    from nerfstudio.fields.kplanes_field import KPlanesDensityField, KPlanesField

    torch.manual_seed(17)
    gen = torch.Generator().manual_seed(17)
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    sc = SceneContraction(order=float("inf"))
    f = KPlanesField(aabb, spacetime_resolution=[6, 5, 4, 3], feat_dim=32, multiscale_res=[1, 2], concat_features_across_scales=True,
                     spatial_distortion=sc, linear_decoder=False, linear_decoder_layers=None, disable_viewing_dependent=True, sigma_net_layers=1,
                     sigma_net_hidden_dim=128, rgb_net_layers=2, rgb_net_hidden_dim=64)
    df = KPlanesDensityField(aabb, resolution=[8, 7, 6, 3], feature_dim=8, spatial_distortion=sc, linear_decoder=False)
    with torch.no_grad():
        for m in (f, df):
            for p in m.parameters():
                if p.dim() == 4:
                    p.copy_(torch.rand(p.shape, generator=gen) * 1.2 - 0.1)
                elif p.requires_grad:  # (not the aabb buffer)
                    p.mul_(4.0)  # visible densities / colours (the nets' small initialisation maps everything to density ~ 1)
    R, S = 8, 7
    pos = (torch.rand(R, S, 3, generator=gen) * 2 - 1)
    pos[2:5] *= 3.0       # outside the unit cube: contracted
    pos[5:] *= 40.0       # far away: close to the cube [-2, 2]^3's faces
    pos[0, 0] = torch.tensor([1.0, -1.0, 0.5])  # on the boundary ||x||_inf = 1
    dirs = torch.nn.functional.normalize(torch.rand(R, 1, 3, generator=gen) * 2 - 1, dim=-1).expand(R, S, 3).contiguous()
    tms = torch.rand(R, 1, generator=gen)
    rs = RaySamples(frustums=Frustums(origins=pos, directions=dirs, starts=torch.zeros(R, S, 1), ends=torch.zeros(R, S, 1), pixel_area=torch.ones(R, S, 1)),
                    times=tms[:, None])
    out = {"aabb": aabb.numpy(), "positions": pos.numpy(), "directions": dirs.numpy(), "times": tms.numpy(), "contracted": sc(pos).numpy()}
    with torch.no_grad():
        o = f(rs)
        out["density"], out["rgb"] = o[FieldHeadNames.DENSITY][..., 0].numpy(), o[FieldHeadNames.RGB].numpy()
        out["prop_density"] = df.density_fn(pos, tms)[..., 0].numpy()
    for s, pl in enumerate(f.grids):
        for q, g in enumerate(pl):
            out[f"plane_{s}_{q}"] = g.detach().numpy()
    for q, g in enumerate(df.grids):
        out[f"prop_plane_{q}"] = g.detach().numpy()
    for name, net in (("sigma", f.sigma_net), ("color", f.color_net), ("prop_sigma", df.sigma_net)):
        for i, l in enumerate(net.layers):
            out[f"{name}_{i}"] = l.weight.detach().numpy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6c_contraction.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; density range", float(out["density"].min()), float(out["density"].max()))


if __name__ == "__main__":
    main()

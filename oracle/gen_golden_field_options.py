"""G6b: KPlanesField with the decoder option the k-planes preset switches off -- view-dependent colour: spherical harmonics of the direction
concatenated in front of the geometry features as color_net's input (NS/fields/kplanes_field.py:206-216 constructor, :314-323 get_outputs)
-- evaluated by the REFERENCE's own class (imported through oracle/_refimport.py with the shims of SURVEY Appendix A), training and eval mode.
(The appearance-embedding branch of the reference, :325-346, cannot run with per-sample camera indices: its view(-1, 1, D).expand(n_rays,
n_samples, -1) needs n_rays rows and gets n_rays * n_samples -- RuntimeError for every S > 1.  Not mirrored.)

TEST INFRASTRUCTURE.  Run in the container that holds /root/reference:   python -m oracle.gen_golden_field_options
Writes tests/golden/g6b_field_options.npz (inputs, every parameter tensor, outputs)."""
import os

import numpy as np
import torch

from oracle._refimport import import_reference


def main():
    import_reference()
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from nerfstudio.field_components.field_heads import FieldHeadNames
    from nerfstudio.fields.kplanes_field import KPlanesField

    torch.manual_seed(7)
    gen = torch.Generator().manual_seed(7)
    aabb = torch.tensor([[-1.2, -1.0, -0.8], [1.2, 1.0, 0.8]])
    out = {"aabb": aabb.numpy()}
    for tag, kw in (("vd", dict(disable_viewing_dependent=False, use_appearance_embedding=False)),):
        f = KPlanesField(aabb, spacetime_resolution=[6, 5, 4, 3], feat_dim=32, multiscale_res=[1, 2], concat_features_across_scales=True,
                         linear_decoder=False, linear_decoder_layers=None, sigma_net_layers=1, sigma_net_hidden_dim=128, rgb_net_layers=2,
                         rgb_net_hidden_dim=64, **kw)
        with torch.no_grad():
            for p in f.parameters():
                if p.dim() == 4:  # planes: away from the uniform(0.1, 0.5) / ones initialisation
                    p.copy_(torch.rand(p.shape, generator=gen) * 1.2 - 0.1)
        R, S = 6, 9
        pos = (torch.rand(R, S, 3, generator=gen) * 2 - 1) * 1.1
        dirs = torch.nn.functional.normalize(torch.rand(R, 1, 3, generator=gen) * 2 - 1, dim=-1).expand(R, S, 3).contiguous()
        tms = torch.rand(R, 1, generator=gen)
        cam = torch.randint(0, 5, (R, 1), generator=gen)
        rs = RaySamples(frustums=Frustums(origins=pos, directions=dirs, starts=torch.zeros(R, S, 1), ends=torch.zeros(R, S, 1),
                                          pixel_area=torch.ones(R, S, 1)),
                        camera_indices=cam[:, None, :].expand(R, S, 1).contiguous(), times=tms[:, None])
        out[f"{tag}_positions"], out[f"{tag}_directions"], out[f"{tag}_times"], out[f"{tag}_camera_indices"] = pos.numpy(), dirs.numpy(), tms.numpy(), cam.numpy()
        for mode in ("train", "eval"):
            f.train(mode == "train")
            with torch.no_grad():
                o = f(rs)
            out[f"{tag}_{mode}_density"] = o[FieldHeadNames.DENSITY][..., 0].numpy()
            out[f"{tag}_{mode}_rgb"] = o[FieldHeadNames.RGB].numpy()
        for s, pl in enumerate(f.grids):
            for q, g in enumerate(pl):
                out[f"{tag}_plane_{s}_{q}"] = g.detach().numpy()
        for i, l in enumerate(f.sigma_net.layers):
            out[f"{tag}_sigma_{i}"] = l.weight.detach().numpy()
        for i, l in enumerate(f.color_net.layers):
            out[f"{tag}_color_{i}"] = l.weight.detach().numpy()
        if f.appearance_embedding is not None:
            out[f"{tag}_appearance"] = f.appearance_embedding.embedding.weight.detach().numpy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6b_field_options.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if "rgb" in k})


if __name__ == "__main__":
    main()

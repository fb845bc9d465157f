"""G4e: the reference's PDFSampler at the `k-planes` preset's sizes -- TEST INFRASTRUCTURE (runs only where /root/reference exists).

512 rays through the preset's two resampling levels (256 -> 128 -> 64 samples, histogram_padding 0.01, train-mode stratified u,
NS/model_components/ray_samplers.py:274-369 called exactly as ProposalNetworkSampler does at :573-600): weights come from
surface-like densities through the reference's own RaySamples.get_weights and are annealed (`weights ** anneal`, :584) before each
level.  `torch.searchsorted` is wrapped to capture the reference's `inds`, stored as int16 (the bit-exact target of SURVEY 8a P6):
~99 k indices instead of G4's ~12 k, at the sizes the training run evaluates them.

    python oracle/gen_golden_pdf_preset.py        # writes tests/golden/g4e_pdf_preset.npz
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle._refimport import import_reference  # noqa: E402
from oracle.gen_golden import RandQueue, save  # noqa: E402


def main():
    import_reference()
    import nerfstudio.model_components.ray_samplers as RS
    from nerfstudio.cameras.rays import RayBundle

    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(20231029)
    R, S0, S1, S2 = 512, 256, 128, 64
    nears = torch.rand(R, 1, generator=gen) * 0.5 + 0.05
    fars = nears + torch.rand(R, 1, generator=gen) * 3 + 0.5
    bundle = RayBundle(origins=torch.zeros(R, 3), directions=torch.ones(R, 3), pixel_area=torch.ones(R, 1), nears=nears, fars=fars)
    us = RS.UniformSampler()
    us.train()
    with RandQueue([torch.rand(R, S0 + 1, generator=gen)]):
        level0 = us(bundle, num_samples=S0)

    def density(rs, sharp):
        """a ground surface (step) + up to two thin objects in front of it, as a trained proposal network sees a soccer pitch"""
        mid = (rs.frustums.starts + rs.frustums.ends)[..., 0] / 2  # [R, S]
        t_surf = nears + (fars - nears) * (0.35 + 0.6 * torch.rand(R, 1, generator=gen))
        d = torch.sigmoid((mid - t_surf) * sharp) * (20 + 200 * torch.rand(R, 1, generator=gen))
        for _ in range(2):
            c = nears + (t_surf - nears) * torch.rand(R, 1, generator=gen)
            on = (torch.rand(R, 1, generator=gen) < 0.3).float()
            d = d + on * 150 * torch.exp(-((mid - c) / 0.02) ** 2)
        d = d + 0.05 * torch.rand(R, mid.shape[1], generator=gen)  # floaters
        d[:4] = 0.0  # empty rays: the padding alone decides
        return d[..., None]

    captured = {}
    orig_ss = torch.searchsorted

    def ss(cdf, u, side="left", **kw):
        r = orig_ss(cdf, u, side=side, **kw)
        captured["inds"], captured["cdf"], captured["u"] = r, cdf, u
        return r

    out = {"nears": nears, "fars": fars}
    prev = level0
    pdf = RS.PDFSampler(include_original=False, single_jitter=False)
    pdf.train()
    for tag, S, sharp, anneal in (("a", S1, 30.0, 0.526), ("b", S2, 60.0, 1.0)):  # anneal: step ~100 of 1000 (slope 10) / fully annealed
        w = prev.get_weights(density(prev, sharp))  # reference code (rays.py:127-149)
        annealed = torch.pow(w, anneal)  # ray_samplers.py:584
        rand = torch.rand(R, S + 1, generator=gen)
        rand[5] = 0.0  # u exactly on the stratum edges
        torch.searchsorted = ss
        try:
            with RandQueue([rand]):
                new = pdf(bundle, prev, annealed, num_samples=S)
        finally:
            torch.searchsorted = orig_ss
        assert int(captured["inds"].max()) <= 32767
        out[f"{tag}_weights"] = annealed[..., 0]
        out[f"{tag}_prev_sbins"] = torch.cat([prev.spacing_starts[..., 0], prev.spacing_ends[:, -1:, 0]], -1)
        out[f"{tag}_rand"] = rand
        out[f"{tag}_inds"] = captured["inds"].to(torch.int16)
        out[f"{tag}_new_sbins"] = torch.cat([new.spacing_starts[..., 0], new.spacing_ends[:, -1:, 0]], -1)
        print(tag, "weights max", float(annealed.max()), "inds", tuple(captured["inds"].shape))
        prev = new
    save("g4e_pdf_preset", **out)


if __name__ == "__main__":
    main()

"""Golden-vector generator -- TEST INFRASTRUCTURE.

Imports the read-only reference (/root/reference, via `oracle/_refimport.py` + throw-away shims),
runs ITS code on seeded inputs with every random draw made explicit, and stores inputs + expected
outputs as small .npz fixtures under tests/golden/.  Run only in the build container:

    python -m oracle.gen_golden

The fixtures are data (inputs and expected outputs); no reference source text is stored.
Parameters for the model-level fixtures are regenerated from a seed by
`oracle.kplanes_oracle.make_kplanes_params` and LOADED INTO the reference modules, so the
fixtures stay small.  SURVEY.md §8c lists the vectors (G1..G11).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle._refimport import import_reference  # noqa: E402
from oracle import kplanes_oracle as KO  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def npy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


class RandQueue:
    """Replace torch.rand / torch.rand_like by pre-drawn tensors (checked by shape)."""

    def __init__(self, tensors):
        self.q = list(tensors)
        self.orig = (torch.rand, torch.rand_like)

    def __enter__(self):
        def rand(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            t = self.q.pop(0)
            assert tuple(t.shape) == tuple(size), (t.shape, size)
            return t.clone()

        def rand_like(x, **kw):
            t = self.q.pop(0)
            assert t.shape == x.shape, (t.shape, x.shape)
            return t.clone()

        torch.rand, torch.rand_like = rand, rand_like
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like = self.orig
        assert not self.q, f"{len(self.q)} unused random tensors"


def make_cameras(gen, n_cam=3, H=54, W=96):
    """Pinhole cameras looking at the origin from a ring (Broadcast-style-like geometry)."""
    c2w = []
    for i in range(n_cam):
        ang = 2 * np.pi * i / n_cam + 0.3
        pos = np.array([np.cos(ang) * 1.0, np.sin(ang) * 1.0, 0.35 + 0.1 * i])
        fwd = -pos / np.linalg.norm(pos)
        up = np.array([0, 0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        up2 = np.cross(right, fwd)
        rot = np.stack([right, up2, -fwd], axis=1)  # camera looks along -z
        c2w.append(np.concatenate([rot, pos[:, None]], axis=1))
    c2w = torch.tensor(np.stack(c2w), dtype=torch.float32)
    fx = torch.tensor([100.0 + 7 * i for i in range(n_cam)])
    fy = torch.tensor([101.0 + 5 * i for i in range(n_cam)])
    cx = torch.full((n_cam,), W / 2.0) + 0.25
    cy = torch.full((n_cam,), H / 2.0) - 0.5
    times = torch.linspace(0, 1, n_cam)
    return c2w, fx, fy, cx, cy, times, H, W


def load_params_into_reference(model, params):
    """Copy oracle-format params into the reference KPlanesModel (shim tcnn = Linear stacks)."""
    with torch.no_grad():
        for s, grids in enumerate(params["field_grids"]):
            for p, g in enumerate(grids):
                model.field.grids[s][p].copy_(g)
        for l, w in zip(model.field.sigma_net.layers, params["field_sigma"]):
            l.weight.copy_(w)
        for l, w in zip(model.field.color_net.layers, params["field_color"]):
            l.weight.copy_(w)
        for i, pn in enumerate(model.proposal_networks):
            for p, g in enumerate(params["prop_grids"][i]):
                pn.grids[p].copy_(g)
            for l, w in zip(pn.sigma_net.layers, params["prop_sigma"][i]):
                l.weight.copy_(w)


E2E_CFG = dict(
    base_res=(32, 32, 32, 6),
    multiscale=(1, 2),
    feat_dim=32,
    prop_res=((48, 48, 48, 6), (96, 96, 96, 6)),
    prop_feat=8,
    sigma_hidden=128,
    color_hidden=64,
    aabb_scale=1.5,
    seed=1234,
)


def build_reference_model(ns, cfg=E2E_CFG, samples=(256, 128), nerf_samples=64):
    import nerfstudio.models.kplanes as km
    from nerfstudio.data.scene_box import SceneBox

    mc = km.KPlanesModelConfig(
        multiscale_res=tuple(cfg["multiscale"]),
        spacetime_resolution=tuple(cfg["base_res"]),
        feature_dim=cfg["feat_dim"],
        proposal_net_args_list=[{"feature_dim": cfg["prop_feat"], "resolution": list(r)} for r in cfg["prop_res"]],
        num_proposal_samples_per_ray=tuple(samples),
        num_nerf_samples_per_ray=nerf_samples,
        disable_viewing_dependent=True,
        sigma_net_hidden_dim=cfg["sigma_hidden"],
        rgb_net_hidden_dim=cfg["color_hidden"],
        loss_coefficients=dict(KO.DEFAULT_LOSS_COEF),
    )
    a = cfg["aabb_scale"]
    model = km.KPlanesModel(mc, scene_box=SceneBox(aabb=torch.tensor([[-a] * 3, [a] * 3])), num_train_data=4)
    params = KO.make_kplanes_params(**cfg)
    load_params_into_reference(model, params)
    return model, params


def main():
    ns = import_reference()
    torch.manual_seed(20231029)
    gen = torch.Generator().manual_seed(20231029)
    from nerfstudio.cameras.cameras import Cameras
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox
    from nerfstudio.model_components import losses as RL
    from nerfstudio.model_components import ray_samplers as RS
    from nerfstudio.model_components import renderers as RR
    from nerfstudio.model_components.scene_colliders import AABBBoxCollider
    import nerfstudio.fields.kplanes_field as KF

    # ---------------- G1: ray generation (pinhole) ----------------
    print("G1 raygen")
    c2w, fx, fy, cx, cy, times, H, W = make_cameras(gen)
    cams = Cameras(camera_to_worlds=c2w, fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=H, times=times)
    n = 64
    idx = torch.stack(
        [
            torch.randint(0, 3, (n,), generator=gen),
            torch.randint(0, H, (n,), generator=gen),
            torch.randint(0, W, (n,), generator=gen),
        ],
        dim=-1,
    )
    idx[0] = torch.tensor([0, 0, 0])
    idx[1] = torch.tensor([2, H - 1, W - 1])
    coords = cams.get_image_coords()[idx[:, 1], idx[:, 2]]
    rb = cams.generate_rays(camera_indices=idx[:, 0:1], coords=coords)
    save(
        "g1_raygen",
        indices=idx, c2w=c2w, fx=fx, fy=fy, cx=cx, cy=cy, times=times,
        origins=rb.origins, directions=rb.directions, pixel_area=rb.pixel_area,
        directions_norm=rb.metadata["directions_norm"], ray_times=rb.times, camera_indices=rb.camera_indices,
    )

    # ---------------- G2: AABB collider ----------------
    print("G2 collider")
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    o = torch.cat([rb.origins, torch.tensor([[3.0, 3.0, 3.0], [0.0, 0.0, 0.0], [2.0, 0.1, 0.2]])])
    d = torch.cat([rb.directions, torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])])
    col = AABBBoxCollider(SceneBox(aabb=aabb), near_plane=0.05)
    col.train()
    nt, ft = col._intersect_with_aabb(o, d, aabb)
    col.eval()
    ne, fe = col._intersect_with_aabb(o, d, aabb)
    save("g2_collider", origins=o, directions=d, aabb=aabb, near_plane=0.05,
         nears_train=nt, fars_train=ft, nears_eval=ne, fars_eval=fe)

    # ---------------- G3: spaced samplers ----------------
    print("G3 spaced samplers")
    R = 16
    nears = torch.rand(R, 1, generator=gen) * 0.5 + 0.05
    fars = nears + torch.rand(R, 1, generator=gen) * 3 + 0.2
    bundle = RayBundle(origins=torch.zeros(R, 3), directions=torch.ones(R, 3), pixel_area=torch.ones(R, 1),
                       nears=nears, fars=fars)
    g3 = dict(nears=nears, fars=fars)
    for S in (256, 48, 7):
        for sj in (False, True):
            t_rand = torch.rand((R, 1) if sj else (R, S + 1), generator=gen)
            for kind, cls in (("uniform", RS.UniformSampler), ("piecewise", RS.UniformLinDispPiecewiseSampler)):
                smp = cls(single_jitter=sj)
                smp.train()
                with RandQueue([t_rand]):
                    rs = smp(bundle, num_samples=S)
                key = f"{kind}_S{S}_sj{int(sj)}"
                g3[key + "_trand"] = t_rand
                g3[key + "_sbins"] = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1)
                g3[key + "_ebins"] = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1)
                smp.eval()
                rs = smp(bundle, num_samples=S)
                g3[key + "_eval_sbins"] = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1)
                g3[key + "_eval_ebins"] = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1)
    save("g3_spaced", **g3)

    # ---------------- G4: PDF sampler (inds exact) ----------------
    print("G4 pdf sampler")
    g4 = {}
    for tag, (Sp, S) in {"a": (256, 128), "b": (128, 64), "c": (96, 48), "d": (13, 7)}.items():
        R = 48
        nears = torch.rand(R, 1, generator=gen) * 0.5 + 0.05
        fars = nears + torch.rand(R, 1, generator=gen) * 3 + 0.2
        bundle = RayBundle(origins=torch.zeros(R, 3), directions=torch.ones(R, 3), pixel_area=torch.ones(R, 1),
                           nears=nears, fars=fars)
        us = RS.UniformSampler()
        us.train()
        with RandQueue([torch.rand(R, Sp + 1, generator=gen)]):
            prev = us(bundle, num_samples=Sp)
        w = torch.rand(R, Sp, generator=gen) ** 8  # peaky
        w[0] = 0.0  # zero-weight ray
        w[1] = 0.0
        w[1, Sp // 2] = 1.0  # single spike
        w[2] = 1e-12  # tiny weights (eps padding branch irrelevant w/ 0.01 padding, still)
        rand = torch.rand(R, S + 1, generator=gen)
        rand[3] = 0.0  # u exactly on stratum edges
        pdf = RS.PDFSampler(include_original=False)
        pdf.train()
        captured = {}
        orig_ss = torch.searchsorted

        def ss(cdf, u, side="left", **kw):
            r = orig_ss(cdf, u, side=side, **kw)
            captured["inds"], captured["cdf"], captured["u"] = r, cdf, u
            return r

        torch.searchsorted = ss
        try:
            with RandQueue([rand]):
                new = pdf(bundle, prev, w[..., None], num_samples=S)
        finally:
            torch.searchsorted = orig_ss
        prev_s = torch.cat([prev.spacing_starts[..., 0], prev.spacing_ends[:, -1:, 0]], -1)
        new_s = torch.cat([new.spacing_starts[..., 0], new.spacing_ends[:, -1:, 0]], -1)
        new_e = torch.cat([new.frustums.starts[..., 0], new.frustums.ends[:, -1:, 0]], -1)
        # eval-mode too
        pdf.eval()
        torch.searchsorted = ss
        try:
            new_ev = pdf(bundle, prev, w[..., None], num_samples=S)
        finally:
            torch.searchsorted = orig_ss
        g4[f"{tag}_nears"], g4[f"{tag}_fars"] = nears, fars
        g4[f"{tag}_weights"], g4[f"{tag}_prev_sbins"], g4[f"{tag}_rand"] = w, prev_s, rand
        g4[f"{tag}_new_sbins"], g4[f"{tag}_new_ebins"] = new_s, new_e
        g4[f"{tag}_eval_inds"] = captured["inds"]
        g4[f"{tag}_eval_sbins"] = torch.cat([new_ev.spacing_starts[..., 0], new_ev.spacing_ends[:, -1:, 0]], -1)
        # train inds: recompute capture (second call overwrote) -> run train again
        pdf.train()
        torch.searchsorted = ss
        try:
            with RandQueue([rand]):
                pdf(bundle, prev, w[..., None], num_samples=S)
        finally:
            torch.searchsorted = orig_ss
        g4[f"{tag}_inds"], g4[f"{tag}_cdf"], g4[f"{tag}_u"] = captured["inds"], captured["cdf"], captured["u"]
    save("g4_pdf", **g4)

    # ---------------- G5: interpolate_kplanes fwd + plane grads ----------------
    print("G5 interpolate_kplanes")
    g5 = {}
    cases = {
        "single": dict(C=32, base=(16, 12, 10, 5), ms=(1,), concat=True, rng01=False),
        "multi": dict(C=32, base=(8, 8, 8, 4), ms=(1, 2, 4), concat=True, rng01=False),
        "prop": dict(C=8, base=(24, 20, 16, 6), ms=(1,), concat=False, rng01=True),
    }
    for tag, cs in cases.items():
        grids = []
        for m in cs["ms"]:
            reso = [r * m for r in cs["base"][:3]] + [cs["base"][3]]
            pl = KF.init_kplanes_field(cs["C"], reso)
            with torch.no_grad():
                for p in pl:
                    p.copy_(torch.rand(p.shape, generator=gen) * 0.9 + 0.1)
            grids.append(pl)
        N = 200
        pts = torch.rand(N, 4, generator=gen) * (1.0 if cs["rng01"] else 2.4) - (0.0 if cs["rng01"] else 1.2)
        pts[0] = torch.tensor([-1.0, -1.0, -1.0, -1.0])
        pts[1] = torch.tensor([1.0, 1.0, 1.0, 1.0])
        pts[2] = torch.tensor([0.0, 0.0, 0.0, 0.0])
        pts[3] = torch.tensor([1.0, -1.0, 0.5, 1.0])
        feats = KF.interpolate_kplanes(pts, grids, cs["concat"], False, False)
        gout = torch.rand(feats.shape, generator=gen) - 0.5
        feats.backward(gout)
        g5[f"{tag}_pts"], g5[f"{tag}_feats"], g5[f"{tag}_gout"] = pts, feats, gout
        g5[f"{tag}_meta"] = np.array([cs["C"], len(cs["ms"]), int(cs["concat"])])
        for s, pl in enumerate(grids):
            for p, g in enumerate(pl):
                g5[f"{tag}_plane_{s}_{p}"] = g
                g5[f"{tag}_grad_{s}_{p}"] = g.grad
    save("g5_interp", **g5)

    # ---------------- G6 / G7 / G8 / G11: model-level ----------------
    print("G6-G11 model")
    model, params = build_reference_model(ns)
    model.train()
    R = 24
    c2w, fx, fy, cx, cy, times, H, W = make_cameras(gen, n_cam=4)
    cams = Cameras(camera_to_worlds=c2w, fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=H, times=times)
    idx = torch.stack([torch.randint(0, 4, (R,), generator=gen), torch.randint(0, H, (R,), generator=gen),
                       torch.randint(0, W, (R,), generator=gen)], -1)
    rb = cams.generate_rays(camera_indices=idx[:, 0:1], coords=cams.get_image_coords()[idx[:, 1], idx[:, 2]])
    target = torch.rand(R, 3, generator=gen)
    rng = {
        "t_rand": torch.rand(R, 257, generator=gen),
        "u": [torch.rand(R, 129, generator=gen), torch.rand(R, 65, generator=gen)],
        "bg": torch.rand(R, 3, generator=gen),
    }
    anneal = KO.anneal_value(300)
    model.proposal_sampler.set_anneal(anneal)
    with RandQueue([rng["t_rand"], rng["u"][0], rng["u"][1], rng["bg"]]):
        outputs = model(rb)
    loss_dict = model.get_loss_dict(outputs, {"image": target}, {})
    loss = sum(loss_dict.values())
    loss.backward()
    g11 = dict(
        cfg_base_res=np.array(E2E_CFG["base_res"]), cfg_multiscale=np.array(E2E_CFG["multiscale"]),
        cfg_prop_res=np.array(E2E_CFG["prop_res"]), cfg_seed=E2E_CFG["seed"], anneal=anneal,
        origins=rb.origins, directions=rb.directions, times=rb.times, target=target,
        t_rand=rng["t_rand"], u0=rng["u"][0], u1=rng["u"][1], bg=rng["bg"],
        rgb=outputs["rgb"], accumulation=outputs["accumulation"], depth=outputs["depth"],
        median_rgb=outputs["median_rgb"], prop_depth_0=outputs["prop_depth_0"], prop_depth_1=outputs["prop_depth_1"],
        loss_total=loss,
    )
    for i, (w, rs) in enumerate(zip(outputs["weights_list"], outputs["ray_samples_list"])):
        g11[f"weights_{i}"] = w[..., 0]
        g11[f"sbins_{i}"] = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1)
        g11[f"ebins_{i}"] = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1)
    for k, v in loss_dict.items():
        g11["loss_" + k] = v
    # gradient checksums per parameter tensor (sum, abs-sum, and a strided probe)
    names = []
    for name, p in list(model.field.named_parameters()) + [("prop." + n, q) for n, q in model.proposal_networks.named_parameters()]:
        if p.grad is None:
            continue
        names.append(name)
        g = p.grad.double()
        g11["gsum_" + name] = g.sum()
        g11["gabs_" + name] = g.abs().sum()
        g11["gprobe_" + name] = p.grad.flatten()[:: max(1, p.grad.numel() // 64)][:64]
    g11["grad_names"] = np.array(names)
    save("g11_model", **g11)

    # G6: field / density-field values on explicit positions
    print("G6 fields")
    from nerfstudio.cameras.rays import Frustums, RaySamples
    Rf, Sf = 8, 16
    pos = (torch.rand(Rf, Sf, 3, generator=gen) * 2 - 1) * 1.6  # some outside the aabb
    tms = torch.rand(Rf, 1, generator=gen)
    rs = RaySamples(frustums=Frustums(origins=pos, directions=torch.ones_like(pos), starts=torch.zeros_like(pos[..., :1]),
                                      ends=torch.zeros_like(pos[..., :1]), pixel_area=torch.ones_like(pos[..., :1])),
                    times=tms[:, None])
    fo = model.field(rs)
    from nerfstudio.field_components.field_heads import FieldHeadNames
    d0 = model.proposal_networks[0].density_fn(pos, times=tms)
    d1 = model.proposal_networks[1].density_fn(pos, times=tms)
    save("g6_fields", positions=pos, times=tms, density=fo[FieldHeadNames.DENSITY][..., 0], rgb=fo[FieldHeadNames.RGB],
         prop_density_0=d0[..., 0], prop_density_1=d1[..., 0])

    # ---------------- G7: weights + renderers ----------------
    print("G7 renderers")
    R, S = 20, 64
    dens = torch.rand(R, S, generator=gen) ** 6 * 40
    dens[0] = 0.0
    dens[1] = 1e4
    starts = torch.cumsum(torch.rand(R, S + 1, generator=gen) * 0.05 + 1e-3, -1)
    fr = Frustums(origins=torch.zeros(R, S, 3), directions=torch.ones(R, S, 3), starts=starts[:, :-1, None],
                  ends=starts[:, 1:, None], pixel_area=torch.ones(R, S, 1))
    rs = RaySamples(frustums=fr, deltas=(starts[:, 1:] - starts[:, :-1])[..., None])
    w = rs.get_weights(dens[..., None])
    rgb = torch.rand(R, S, 3, generator=gen)
    bg = torch.rand(R, 3, generator=gen)
    g7 = dict(density=dens, ebins=starts, rgb=rgb, bg=bg, weights=w[..., 0])
    ren = RR.RGBRenderer(background_color="random")
    ren.train()
    with RandQueue([bg]):
        g7["rgb_random_train"] = ren(rgb, w)
    for name in ("black", "white", "last_sample"):
        ren.background_color = name
        ren.train()
        g7[f"rgb_{name}_train"] = ren(rgb, w)
        ren.eval()
        g7[f"rgb_{name}_eval"] = ren(rgb, w)
    g7["accumulation"] = RR.AccumulationRenderer()(w)
    g7["depth_median"] = RR.DepthRenderer("median")(w, rs)
    g7["depth_expected"] = RR.DepthRenderer("expected")(w, rs)
    mr = RR.MedianRGBRenderer()
    mr.train()
    g7["median_rgb"] = mr(rgb, w)
    g7["median_index"] = KO.median_index(w[..., 0])
    save("g7_render", **g7)

    # ---------------- G8: losses (values + grads) ----------------
    print("G8 losses")
    g8 = {}
    R = 12
    sb = []
    ws = []
    for S in (256, 128, 64):
        b = torch.sort(torch.rand(R, S + 1, generator=gen), -1).values
        b[:, 0], b[:, -1] = 0.0, 1.0
        wt = torch.rand(R, S, generator=gen) ** 3
        wt = (wt / wt.sum(-1, keepdim=True) * torch.rand(R, 1, generator=gen)).requires_grad_(True)
        sb.append(b)
        ws.append(wt)

    class _RS:  # minimal stand-in exposing what ray_samples_to_sdist reads (losses.py:98-103)
        def __init__(self, b):
            self.spacing_starts = b[:, :-1, None]
            self.spacing_ends = b[:, 1:, None]

    rsl = [_RS(b) for b in sb]
    wl = [w_[..., None] for w_ in ws]
    li = RL.interlevel_loss(wl, rsl)
    li.backward()
    ld = RL.distortion_loss(wl, rsl)
    ld.backward()
    for i in range(3):
        g8[f"sbins_{i}"], g8[f"w_{i}"] = sb[i], ws[i]
    g8["interlevel"], g8["distortion"] = li, ld
    g8["grad_w0"], g8["grad_w1"] = ws[0].grad, ws[1].grad
    g8["grad_w2_distortion"] = ws[2].grad  # interlevel detaches the nerf level
    planes = KF.init_kplanes_field(8, [12, 10, 9, 7])
    with torch.no_grad():
        for p in planes:
            p.copy_(torch.rand(p.shape, generator=gen))
    for nm, fn in (("space_tv", RL.space_tv_loss), ("time_smooth", RL.time_smoothness_loss),
                   ("sparse_transients", RL.sparse_transients_loss)):
        for p in planes:
            p.grad = None
        v = fn([planes])
        v.backward()
        g8[nm] = v
        for pi, p in enumerate(planes):
            g8[f"{nm}_grad_{pi}"] = p.grad if p.grad is not None else torch.zeros_like(p)
    for pi, p in enumerate(planes):
        g8[f"reg_plane_{pi}"] = p
    save("g8_losses", **g8)
    # ---------------- G9: temporal grid (the importable Python half of the reference) ----------------
    print("G9 temporal grid tables")
    from nerfstudio.field_components.temporal_grid import TemporalGridEncoder
    g9 = {}
    cases = {  # name: ctor kwargs (config-4 main + proposal encoders, the KAT encoder, odd shapes)
        "main": dict(temporal_dim=64, level_dim=2, num_levels=16, log2_hashmap_size=19, base_resolution=16, desired_resolution=2048),
        # proposal encoders as TemporalHashMLPDensityField builds them (nerfplayer_nerfacto_field.py:83-93): growth factor via exp/log
        "prop0": dict(temporal_dim=32, level_dim=2, num_levels=5, log2_hashmap_size=17, base_resolution=16,
                      per_level_scale=np.exp((np.log(64) - np.log(16)) / 4)),
        "prop1": dict(temporal_dim=32, level_dim=2, num_levels=5, log2_hashmap_size=17, base_resolution=16,
                      per_level_scale=np.exp((np.log(256) - np.log(16)) / 4)),
        "kat": dict(temporal_dim=2, input_dim=1, num_levels=1, level_dim=1, per_level_scale=1, base_resolution=1, log2_hashmap_size=2,
                    gridtype="tiled"),
        "c4": dict(temporal_dim=8, level_dim=4, num_levels=3, log2_hashmap_size=10, base_resolution=4),
        "c3": dict(temporal_dim=5, level_dim=3, num_levels=2, log2_hashmap_size=8, base_resolution=4, input_dim=2),
    }
    tt = torch.tensor([0.0, 0.3, 0.5, 0.999, 1.0, 0.25, 0.0163])
    for name, kw in cases.items():
        enc = TemporalGridEncoder(**kw)
        g9[f"{name}_offsets"] = enc.offsets
        g9[f"{name}_sampling_index"] = enc.sampling_index
        g9[f"{name}_index_list"] = enc.index_list
        g9[f"{name}_mask_a"] = enc.index_a_mask
        g9[f"{name}_mask_b"] = enc.index_b_mask
        g9[f"{name}_trow"] = enc.get_temporal_index(tt)
        g9[f"{name}_per_level_scale"] = float(enc.per_level_scale)
        g9[f"{name}_embed_shape"] = np.array(enc.embeddings.shape)
    g9["times"] = tt
    save("g9_tgrid", **g9)
    # ---------------- G10: IST weight maps (the reference's own compute_ist on an 8-frame synthetic clip) ----------------
    print("G10 IST maps")
    import types
    import cv2  # shim
    import nerfstudio.data.datasets.dynamic_dataset as DD
    Mc, Hc, Wc = 4, 27, 48  # 4 cameras x 8 frames
    ids = torch.arange(Mc).repeat_interleave(8)
    tms = torch.linspace(0, 1, 8).repeat(Mc)
    imgs = torch.rand(Mc, 1, Hc, Wc, 3, generator=gen).expand(Mc, 8, Hc, Wc, 3).clone()  # static background per camera
    for c in range(Mc):
        for f in range(8):  # a moving bright blob + mild sensor noise
            y0, x0 = 3 + f * 2, 5 + f * 4 + c
            imgs[c, f, y0:y0 + 4, x0:x0 + 5] = torch.rand(3, generator=gen)
    imgs = (imgs.reshape(-1, Hc, Wc, 3) + 0.02 * torch.rand(Mc * 8, Hc, Wc, 3, generator=gen)).clamp(0, 1)
    u8 = (imgs * 255).round().to(torch.uint8)
    imgs = u8.float() / 255.0  # what base_dataset.py:82 hands over
    ids[-1] = 99  # a camera with a single image: no neighbour -> all-ones map
    g10 = {"images_u8": u8, "cam_ids": ids, "cam_times": tms}
    for rng_ in (1.0, 0.3):
        fake = types.SimpleNamespace(ist_range=rng_, eval_dataset=False, cameras=types.SimpleNamespace(times=tms[:, None], ids=ids[:, None]),
                                     _dataparser_outputs=None)
        wmap = DD.DynamicDataset.compute_ist(fake, {"image": imgs, "image_idx": torch.arange(Mc * 8)}, "cpu", offline=False)
        g10[f"ist_{str(rng_).replace('.', '_')}"] = wmap.float()
    save("g10_ist", **g10)
    print("done")


if __name__ == "__main__":
    main()

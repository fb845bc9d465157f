"""GPU parity: per-ray sampling ops (HIP, through the C ABI) vs golden vectors and the CPU oracle.
Sample indices must be bit-exact; bins <= 1e-6 abs; weights rtol 1e-5; gradients rtol 1e-4."""
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, rtol=1e-5, atol=1e-6):
    torch.testing.assert_close(a.detach().cpu(), torch.as_tensor(b), rtol=rtol, atol=atol)


@pytest.mark.parametrize("kind", ["uniform", "piecewise"])
@pytest.mark.parametrize("S", [256, 48, 7])
@pytest.mark.parametrize("sj", [0, 1])
def test_spaced_bins_golden(kind, S, sj):
    from soccernerfs_amd import ops

    g = load_golden("g3_spaced")
    key = f"{kind}_S{S}_sj{sj}"
    nears, fars = g["nears"].to(DEV), g["fars"].to(DEV)
    sb, eb = ops.spaced_bins(nears, fars, S, g[key + "_trand"].to(DEV), kind)
    close(sb, g[key + "_sbins"], rtol=0, atol=1e-6)
    close(eb, g[key + "_ebins"], rtol=1e-6, atol=2e-6)
    sb, eb = ops.spaced_bins(nears, fars, S, None, kind)
    close(sb, g[key + "_eval_sbins"], rtol=0, atol=1e-6)
    close(eb, g[key + "_eval_ebins"], rtol=1e-6, atol=2e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_pdf_resample_golden_indices_bit_exact(tag):
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    g = load_golden("g4_pdf")
    w, prev, rand = g[f"{tag}_weights"], g[f"{tag}_prev_sbins"], g[f"{tag}_rand"]
    nears, fars = g[f"{tag}_nears"].to(DEV), g[f"{tag}_fars"].to(DEV)
    S = rand.shape[1] - 1
    R = w.shape[0]
    # (1) explicit u: indices identical to the oracle's (sequential-order CDF), every element
    u = KO.pdf_u(R, S, rand)
    o_bins, o_inds, _ = KO.pdf_sample(w, prev, u)
    sb, eb, inds = ops.pdf_resample(prev.to(DEV), nears, fars, S, weights=w.to(DEV), u=u.to(DEV), return_inds=True)
    assert torch.equal(inds.cpu(), o_inds)
    assert torch.equal(inds.cpu(), g[f"{tag}_inds"])  # and to the reference's on this fixture
    close(sb, o_bins, rtol=0, atol=1e-7)  # same arithmetic as the oracle (double-accumulated CDF)
    # vs the reference's own output: its CDF used torch.sum's association order; the inverse CDF amplifies that ulp
    close(sb, g[f"{tag}_new_sbins"], rtol=0, atol=5e-6)
    close(eb, g[f"{tag}_new_ebins"], rtol=1e-6, atol=2e-5)
    # (2) u built in-kernel from the random draws
    sb2, eb2, inds2 = ops.pdf_resample(prev.to(DEV), nears, fars, S, weights=w.to(DEV), rand=rand.to(DEV), return_inds=True)
    bad = inds2.cpu() != o_inds
    assert int(bad.sum()) <= 2, int(bad.sum())  # only fp ties of u itself may differ
    close(sb2[~bad.to(DEV)], o_bins[~bad], rtol=0, atol=5e-6)  # in-kernel u differs from torch.linspace in the last ulp
    # (3) eval mode (u at bin centres): equal up to exact CDF ties (oracle docstring)
    sb3, eb3, inds3 = ops.pdf_resample(prev.to(DEV), nears, fars, S, weights=w.to(DEV), return_inds=True)
    ue = KO.pdf_u(R, S, None)
    e_bins, e_inds, e_cdf = KO.pdf_sample(w, prev, ue)
    bad = inds3.cpu() != e_inds
    if bad.any():
        dist = (ue[..., :, None] - e_cdf[..., None, :]).abs().min(dim=-1).values
        assert bool((dist[bad] <= 2e-7).all())
    close(sb3[~bad.to(DEV)], e_bins[~bad], rtol=0, atol=5e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_pdf_resample_indices_bit_exact_at_preset_sizes(tag):
    """G4e: 512 rays x (129 | 65) indices of the reference's own PDFSampler at the preset's sizes (int16 fixture).  The kernel's wavefront
    scan with double accumulators and the cumsum-normalised CDF flip NONE of the 99 328 indices (the count is printed and asserted)."""
    from soccernerfs_amd import ops

    g = load_golden("g4e_pdf_preset")
    w, prev, rand = g[f"{tag}_weights"], g[f"{tag}_prev_sbins"], g[f"{tag}_rand"]
    nears, fars = g["nears"].to(DEV), g["fars"].to(DEV)
    S, R = rand.shape[1] - 1, w.shape[0]
    ref = g[f"{tag}_inds"].long()
    from oracle import kplanes_oracle as KO

    u = KO.pdf_u(R, S, rand)
    sb, _, inds = ops.pdf_resample(prev.to(DEV), nears, fars, S, weights=w.to(DEV), u=u.to(DEV), return_inds=True)
    flips = int((inds.cpu() != ref).sum())
    print(f"G4e level {tag}: {flips} of {ref.numel()} indices differ from the reference")
    assert flips == 0
    close(sb, g[f"{tag}_new_sbins"], rtol=0, atol=5e-6)
    # u formed in-kernel from the draws (the training path): only fp ties of u itself may move an index
    _, _, inds2 = ops.pdf_resample(prev.to(DEV), nears, fars, S, weights=w.to(DEV), rand=rand.to(DEV), return_inds=True)
    assert int((inds2.cpu() != ref).sum()) <= 4, int((inds2.cpu() != ref).sum())


def test_pdf_resample_anneal_and_fused_weights():
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(3)
    R, Sp, S = 130, 128, 64  # ragged vs 4 rays per workgroup
    nears = torch.rand(R, 1, generator=gen) * 0.3
    fars = nears + 1 + torch.rand(R, 1, generator=gen)
    sb_prev = KO.spaced_bins(R, Sp, torch.rand(R, Sp + 1, generator=gen))
    eb_prev = KO.spacing_to_euclidean(sb_prev, nears, fars)
    # densities bounded away from 0: for delta*sigma < ~1e-6 the reference's own alpha = 1 - exp(-x) is pure rounding
    # noise (one ulp of exp), which weights**anneal then amplifies by orders of magnitude in ANY implementation
    dens = torch.rand(R, Sp, generator=gen) ** 4 * 30 + 0.5
    w = KO.get_weights(eb_prev[:, 1:] - eb_prev[:, :-1], dens)
    anneal = 0.37
    u = KO.pdf_u(R, S, torch.rand(R, S + 1, generator=gen))
    o_bins, o_inds, _ = KO.pdf_sample(torch.pow(w, anneal), sb_prev, u)
    sb, eb, inds, wout = ops.pdf_resample(sb_prev.to(DEV), nears.to(DEV), fars.to(DEV), S, density=dens.to(DEV), ebins_prev=eb_prev.to(DEV),
                                          u=u.to(DEV), anneal=anneal, return_inds=True, return_weights=True)
    close(wout, w, rtol=1e-5, atol=1e-7)
    frac = float((inds.cpu() == o_inds).float().mean())
    assert frac > 0.999, frac  # expf/powf differ from the CPU libm in the last ulp => rare tie flips
    # The inverse CDF is ill-conditioned in low-density bins, so compare in the well-conditioned direction:
    # the oracle's piecewise-linear CDF evaluated at the kernel's bins must reproduce u.
    _, _, cdf = KO.pdf_sample(torch.pow(w, anneal), sb_prev, u)
    nb = sb.cpu()
    j = torch.clamp(torch.searchsorted(sb_prev, nb.contiguous(), side="right") - 1, 0, Sp - 1)
    b0, b1 = torch.gather(sb_prev, -1, j), torch.gather(sb_prev, -1, j + 1)
    c0, c1 = torch.gather(cdf, -1, j), torch.gather(cdf, -1, j + 1)
    u_back = c0 + (nb - b0) / (b1 - b0) * (c1 - c0)
    torch.testing.assert_close(u_back, u, rtol=0, atol=1e-5)
    close(eb, KO.spacing_to_euclidean(nb, nears, fars), rtol=1e-6, atol=2e-6)
    # with the kernel's own weights as the oracle's input only powf's last ulp remains
    o2_bins, o2_inds, _ = KO.pdf_sample(torch.pow(wout.cpu(), anneal), sb_prev, u)
    assert float((inds.cpu() == o2_inds).float().mean()) > 0.9995


def test_weights_fwd_bwd_golden_and_oracle():
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    g = load_golden("g7_render")
    eb = g["ebins"].to(DEV)
    w = ops.get_weights(g["density"].to(DEV), eb)
    close(w, g["weights"], rtol=1e-5, atol=1e-7)
    gen = torch.Generator().manual_seed(5)
    for S in (64, 256, 37):
        R = 41
        dens = (torch.rand(R, S, generator=gen) ** 3 * 20).requires_grad_(True)
        ebins = torch.cumsum(torch.rand(R, S + 1, generator=gen) * 0.05 + 1e-3, -1)
        ref = KO.get_weights(ebins[:, 1:] - ebins[:, :-1], dens)
        gw = torch.rand(R, S, generator=gen) - 0.3
        ref.backward(gw)
        d2 = dens.detach().to(DEV).requires_grad_(True)
        out = ops.get_weights(d2, ebins.to(DEV))
        close(out, ref, rtol=1e-5, atol=1e-7)
        out.backward(gw.to(DEV))
        close(d2.grad, dens.grad, rtol=1e-4, atol=1e-7)


def test_overflowed_densities_behave_as_the_sequential_cumsum():
    """density = exp(x) overflows to +inf during training (trunc_exp, activations.py:32).  torch.cumsum walks left to right: the samples in
    FRONT of an infinite one keep finite transmittance and weights, the ones behind get T = 0; a zero-width bin under an infinite density
    (0 * inf) poisons what follows with NaN, which nan_to_num turns into 0.  The wavefront scan must reproduce exactly that -- inf at the
    start, in the middle and at the end of a lane's block, next to each other, and in every level's sample count; weights, the resampled bins
    (indices) and the backward included."""
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(17)
    for S in (64, 128, 256, 37):
        R = 24
        dens = torch.rand(R, S, generator=gen) ** 3 * 20
        ebins = torch.cumsum(torch.rand(R, S + 1, generator=gen) * 0.05 + 1e-3, -1)
        per = (S + 63) // 64
        for r, pos in enumerate([0, 1, per - 1, per, S // 2, S // 2 + 1, S - 2, S - 1]):
            dens[r, pos] = float("inf")
        dens[8, 5] = dens[8, 6] = float("inf")
        dens[9, 3 * per + 1] = float("inf")
        ebins[9, 3 * per + 2] = ebins[9, 3 * per + 1]  # zero-width bin: delta * sigma = 0 * inf = NaN from there on
        dens[10, :] = float("inf")
        ref = KO.get_weights(ebins[:, 1:] - ebins[:, :-1], dens)
        got = ops.get_weights(dens.to(DEV), ebins.to(DEV)).cpu()
        assert bool(torch.isfinite(got).all())
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-7)
        assert float(got[4, :S // 2].sum()) > 0  # weights in front of the overflow survive
        # PDF resampling driven by those weights: same bins as the oracle, hence the same indices
        nears, fars = torch.zeros(R), torch.ones(R) * 3
        sb = torch.linspace(0, 1, S + 1).expand(R, S + 1).contiguous()
        u = torch.rand(R, 33, generator=gen)
        want = KO.pdf_sample(ref, sb, KO.pdf_u(R, 32, u))[0]
        got_sb, _ = ops.pdf_resample(sb.to(DEV), nears.to(DEV), fars.to(DEV), 32, density=dens.to(DEV), ebins_prev=ebins.to(DEV), rand=u.to(DEV))
        torch.testing.assert_close(got_sb.cpu(), want, rtol=0, atol=2e-6)
        # backward: finite everywhere, equal to autograd where autograd is finite
        d2 = dens.clone().requires_grad_(True)
        gw = torch.rand(R, S, generator=gen) - 0.3
        KO.get_weights(ebins[:, 1:] - ebins[:, :-1], d2).backward(gw)
        d3 = dens.to(DEV).requires_grad_(True)
        ops.get_weights(d3, ebins.to(DEV)).backward(gw.to(DEV))
        g3 = d3.grad.cpu()
        assert bool(torch.isfinite(g3).all())
        ok = torch.isfinite(d2.grad)
        rows = ok.all(dim=1)  # rays whose reference gradient is finite throughout
        torch.testing.assert_close(g3[rows], d2.grad[rows], rtol=1e-4, atol=1e-7)


def test_ray_ops_argument_errors():
    from soccernerfs_amd import ops

    nears = torch.zeros(4, device=DEV)
    with pytest.raises(RuntimeError, match="t_rand"):
        ops.spaced_bins(nears, nears + 1, 8, torch.zeros(4, 5, device=DEV))
    with pytest.raises(RuntimeError, match="S"):
        ops.get_weights(torch.zeros(2, 400, device=DEV), torch.zeros(2, 401, device=DEV))


def test_sample_pixels_uniform_matches_torch_expression():
    """PixelSampler.sample_method (pixel_samplers.py:74-77) + the image gather of collate_image_dataset_batch (:111-123), fused."""
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(8)
    M, H, W, R = 7, 33, 50, 5000
    u = torch.rand(R, 3, generator=gen)
    u[0], u[1] = 0.0, 1.0 - 2.0**-24  # both ends of [0, 1)
    images = torch.randint(0, 256, (M, H, W, 3), generator=gen, dtype=torch.uint8)
    want = torch.floor(u * torch.tensor([M, H, W])).long()
    want_t = images[want[:, 0], want[:, 1], want[:, 2]].float() / 255.0
    idx, tgt = ops.sample_pixels_uniform(u.to("cuda:0"), M, H, W, images.to("cuda:0"))
    assert idx.dtype == torch.int64 and torch.equal(idx.cpu(), want)
    assert torch.equal(tgt.cpu(), want_t)
    idx2, none = ops.sample_pixels_uniform(u.to("cuda:0"), M, H, W)
    assert none is None and torch.equal(idx2, idx)
    with pytest.raises(RuntimeError):
        ops.sample_pixels_uniform(u.to("cuda:0"), M, H, W, images.to("cuda:0").float())


def test_weights_bwd_with_overflowed_density_stays_finite():
    """density = exp(x) is unbounded (trunc_exp forward, activations.py:32): inf densities, also on zero-width bins (0 * inf),
    must give finite gradients -- nan_to_num semantics; the reference relies on GradScaler skipping such steps."""
    import ctypes as C
    from soccernerfs_amd import _lib, ops

    gen = torch.Generator().manual_seed(2)
    R, S = 8, 64
    eb = torch.sort(torch.rand(R, S + 1, generator=gen), dim=1).values
    eb[:, 10] = eb[:, 11]          # a zero-width bin
    dens = torch.rand(R, S, generator=gen) * 5
    dens[0, 10] = float("inf")     # 0 * inf
    dens[1, 20] = float("inf")     # delta > 0
    dens[2, 10] = float("inf"); dens[2, 30] = float("inf")
    gw = torch.rand(R, S, generator=gen) - 0.5
    d, e, g = dens.to("cuda:0"), eb.to("cuda:0").contiguous(), gw.to("cuda:0")
    out = torch.empty_like(d)
    _lib.check(_lib.lib().snerf_weights_bwd(ops._ptr(d), ops._ptr(e), ops._ptr(g), R, S, ops._ptr(out), 0, None, ops._stream()))
    assert bool(torch.isfinite(out).all())
    # rays without overflow are unaffected: compare with autograd of the oracle formula
    from oracle import kplanes_oracle as KO
    dr = dens[3:].clone().requires_grad_(True)
    KO.get_weights(eb[3:, 1:] - eb[3:, :-1], dr).backward(gw[3:])
    torch.testing.assert_close(out[3:].cpu(), dr.grad, rtol=1e-4, atol=1e-6)


def test_adam_drops_non_finite_gradient_elements():
    from soccernerfs_amd import ops

    n = 1000
    p = torch.ones(n, device="cuda:0"); g = torch.full((n,), 0.5, device="cuda:0"); m = torch.zeros(n, device="cuda:0"); v = torch.zeros(n, device="cuda:0")
    g[3], g[7], g[11] = float("nan"), float("inf"), float("-inf")
    ops.adam_step(p, g, m, v, 1, 1e-2, zero_grad=True)
    assert bool(torch.isfinite(p).all()) and bool(torch.isfinite(m).all()) and bool(torch.isfinite(v).all())
    assert float(p[3]) == 1.0 and float(p[7]) == 1.0 and float(p[0]) < 1.0 and float(g.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("R", [1, 2, 777, 4096, 5000, 16384])
def test_sort_rays_by_time_is_a_stable_permutation(R):
    """snerf_sort_rays_by_key: the batch's pixel indices (and the target colours with them) in order of the images' frame time; rays of the same
    time keep the order they were drawn in (the result is a pure function of the batch); nothing but a permutation happens."""
    from soccernerfs_amd import ops

    dev = "cuda:0"
    gen = torch.Generator().manual_seed(R)
    M = 60
    times = (torch.randint(0, 25, (M,), generator=gen).float() / 24.0).to(dev)  # several images share a frame time
    key, n_keys = ops.image_time_keys(times)
    assert n_keys == int(torch.unique(times).numel()) and key.dtype == torch.int32
    idx = torch.stack([torch.randint(0, M, (R,), generator=gen), torch.randint(0, 540, (R,), generator=gen), torch.randint(0, 960, (R,), generator=gen)], -1).to(dev)
    tgt = torch.rand(R, 3, generator=gen).to(dev)
    out, tout = ops.sort_rays_by_time(idx, key, n_keys, tgt)
    order = torch.argsort(key[idx[:, 0]].long(), stable=True)
    assert torch.equal(out, idx[order]) and torch.equal(tout, tgt[order])
    assert bool((torch.diff(times[out[:, 0]]) >= 0).all())
    only = ops.sort_rays_by_time(idx, key, n_keys)
    assert torch.equal(only, out)


@pytest.mark.gpu
def test_sort_rays_by_time_leaves_oversized_batches_alone():
    from soccernerfs_amd import ops

    dev = "cuda:0"
    key, n_keys = ops.image_time_keys(torch.tensor([0.5, 0.0], device=dev))
    idx = torch.zeros(20000, 3, dtype=torch.int64, device=dev)
    idx[::2, 0] = 1
    assert ops.sort_rays_by_time(idx, key, n_keys) is idx


@pytest.mark.gpu
def test_pixel_sampler_batches_in_order_of_frame_time():
    """PixelSampler.collate_image_dataset_batch with batch["time_key"]: the batch comes out in order of the images' frame time, every per-pixel
    entry gathered at the permuted indices (so image, depth etc. stay attached to their rays) and image_idx remapped as without the option."""
    from soccernerfs_amd import ops
    from soccernerfs_amd.pixel_samplers import PixelSampler

    dev = "cuda:0"
    gen = torch.Generator().manual_seed(4)
    M, H, W = 12, 20, 30
    images = torch.randint(0, 256, (M, H, W, 3), generator=gen, dtype=torch.uint8).to(dev)
    depth = torch.rand(M, H, W, 1, generator=gen).to(dev)
    times = torch.tensor([0.5, 0.0, 1.0, 0.25] * 3, device=dev)
    key, n = ops.image_time_keys(times)
    batch = {"image": images, "depth_image": depth, "image_idx": torch.arange(100, 100 + M, device=dev), "time_key": key, "n_time_keys": n}
    torch.manual_seed(9)
    out = PixelSampler(512).sample(batch)
    c = out["indices"][:, 0] - 100
    assert bool((torch.diff(times[c]) >= 0).all()) and len(torch.unique(c)) > 4
    y, x = out["indices"][:, 1], out["indices"][:, 2]
    assert torch.equal(out["image"], images[c, y, x].float() / 255.0) and torch.equal(out["depth_image"], depth[c, y, x])
    assert "time_key" not in out and "n_time_keys" not in out
    torch.manual_seed(9)
    plain = PixelSampler(512).sample({k: v for k, v in batch.items() if k not in ("time_key", "n_time_keys")})
    assert torch.equal(torch.sort(plain["indices"].view(-1, 3)[:, 0]).values, torch.sort(out["indices"][:, 0]).values)  # the same draw, permuted

"""GPU: IST maps (bit-exact vs the reference golden / oracle) and the device-side importance pixel sampler (counts exact,
distribution statistical -- the reference's draws are RNG-specific)."""
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rng", ["1_0", "0_3"])
@pytest.mark.parametrize("as_float", [False, True])
def test_ist_maps_match_reference_golden(rng, as_float):
    from soccernerfs_amd.pixel_samplers import compute_ist

    g = load_golden("g10_ist")
    imgs = g["images_u8"].to(DEV)
    if as_float:
        imgs = imgs.float() / 255.0
    out = compute_ist(imgs, g["cam_ids"].to(DEV), g["cam_times"].to(DEV), float(rng.replace("_", ".")))
    ref = g[f"ist_{rng}"]
    got = out.float().cpu()
    bad = got != ref
    # identical except (possibly) pixels whose mean difference sits within an ulp of the 0.15 threshold
    assert int(bad.sum()) <= 2, int(bad.sum())
    assert out.dtype == torch.float16 and torch.all(out[-1] == 1.0)


def test_ist_maps_larger_random_vs_oracle():
    from oracle import ist_oracle as IO
    from soccernerfs_amd.pixel_samplers import compute_ist

    gen = torch.Generator().manual_seed(3)
    M, H, W = 12, 33, 50
    u8 = torch.randint(0, 256, (M, H, W, 3), generator=gen, dtype=torch.uint8)
    u8[6:] = (u8[:6].float() * 0.9).to(torch.uint8)  # correlated pairs -> differences around the threshold
    ids = torch.tensor([0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 2])
    t = torch.tensor([0.0, 0.2, 0.4, 0.0, 0.005, 0.5, 0.6, 0.8, 1.0, 0.7, 0.9, 0.3])
    ref = IO.compute_ist(u8.float() / 255.0, ids, t, 0.45).float()
    got = compute_ist(u8.to(DEV), ids.to(DEV), t.to(DEV), 0.45).float().cpu()
    assert int((got != ref).sum()) <= 3


def test_dynamic_sampler_counts_and_distribution():
    from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler

    torch.manual_seed(0)
    M, H, W, R = 40, 20, 30, 4096
    w = torch.zeros(M, H, W)
    w[:, 5:9, 10:16] = torch.rand(M, 4, 6) + 0.2  # a small bright region per image
    w[7] = 0.0                                    # an empty map must never be chosen
    images = torch.randint(0, 256, (M, H, W, 3), dtype=torch.uint8)
    batch = {"image": images.to(DEV), "image_idx": torch.arange(M, device=DEV) + 100, "ist_weights": w.to(DEV).half(), "iter_steps": 5000}
    smp = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
    num_ist = int(0.15 * R)  # 614
    per_image = 10 * (-(-num_ist // M))  # 160
    hits = torch.zeros(M, H, W)
    for _ in range(30):
        idx = smp.sample_method(R, M, H, W, batch=batch, device=DEV).cpu()
        assert idx.shape == (R, 3) and idx.dtype == torch.int64
        ist = idx[:num_ist]
        assert torch.all(w[ist[:, 0], ist[:, 1], ist[:, 2]] > 0)  # every IST draw lands on a non-zero weight
        counts = torch.bincount(ist[:, 0], minlength=M)
        assert counts[7] == 0
        used = counts[counts > 0]
        assert int((used == per_image).sum()) >= len(used) - 1  # 10*ceil(num_ist/M) per chosen image, the last one clipped
        hits.index_put_((ist[:, 0], ist[:, 1], ist[:, 2]), torch.ones(num_ist), accumulate=True)
        rest = idx[num_ist:]
        assert int(rest[:, 0].max()) < M and int(rest[:, 1].max()) < H and int(rest[:, 2].max()) < W
        # torch.multinomial(..., replacement=False) (pixel_samplers.py:400-402): 24 non-zero pixels < 160 draws here -> WITH replacement
    # within one image the empirical pixel distribution follows the weights
    m = int(hits.sum((1, 2)).argmax())
    p_emp = hits[m, 5:9, 10:16].flatten() / hits[m].sum()
    p_ref = w[m, 5:9, 10:16].flatten() / w[m].sum()
    n = float(hits[m].sum())
    assert float(((p_emp - p_ref) ** 2 / p_ref).sum() * n) < 60.0  # chi-square, 23 dof: P(>60) ~ 4e-5
    # before iters_to_start_ist: uniform sampling only; collate returns the reference's keys
    batch["iter_steps"] = 10
    out = smp.collate_image_dataset_batch(batch, R)
    assert out["image"].shape == (R, 3) and out["image"].dtype == torch.float32 and float(out["image"].max()) <= 1.0
    assert out["indices"].shape == (R, 3) and int(out["indices"][:, 0].min()) >= 100  # remapped through image_idx
    assert out["ist_weights"].shape == (R,)


def test_ist_draws_match_oracle_and_do_not_repeat_pixels():
    """snerf_ist_sample == oracle/ist_oracle.py::sample on the same (cdf, chosen images, uniforms): identical indices; no pixel twice inside
    a slot whenever the map has enough non-zero pixels (torch.multinomial(replacement=False), pixel_samplers.py:400-402), replacement
    otherwise; the per-slot counts of :393-397; and the sequential-removal distribution (first draw ~ w, second ~ w without the first)."""
    import ctypes as C

    from oracle import ist_oracle as IO
    from soccernerfs_amd import _lib, ops

    gen = torch.Generator().manual_seed(3)
    M, H, W = 12, 16, 24
    w = torch.zeros(M, H, W)
    w[:, 3:9, 4:14] = (torch.rand(M, 6, 10, generator=gen) * 0.85 + 0.15)
    w[2] = 0.0
    w[2, 5, 5:8] = torch.tensor([0.9, 0.3, 0.6])  # 3 non-zero pixels < 10 draws: replacement
    w[4] = 0.0
    w[4, 1, 1:13] = 0.5                            # exactly 12 >= 10: without replacement, heavy depletion
    w = w.half().float()
    cdf = torch.cumsum(w.reshape(M, -1), dim=1).contiguous()
    nnz = (w.reshape(M, -1) > 0).sum(1).to(torch.int32)
    chosen = torch.tensor([4, 2, 0, 7, 11, 5], dtype=torch.int64)
    per_image, n = 10, 57  # the last slot is clipped to 7 draws
    u = torch.rand(n, generator=gen)
    u[3], u[17] = 0.0, 0.99999994  # the ends of the interval
    idx = torch.empty(n, 3, dtype=torch.int64, device=DEV)
    keep = [cdf.to(DEV), chosen.to(DEV), nnz.to(DEV), u.to(DEV)]
    _lib.check(_lib.lib().snerf_ist_sample(ops._ptr(keep[0]), H, W, ops._ptr(keep[1]), ops._ptr(keep[2]), per_image, ops._ptr(keep[3]), n, ops._ptr(idx),
                                           ops._stream()))
    idx = idx.cpu()
    pix, img = IO.sample(cdf.numpy(), chosen.tolist(), nnz.numpy(), per_image, u.numpy())
    assert torch.equal(idx[:, 0], torch.from_numpy(img)) and torch.equal(idx[:, 1] * W + idx[:, 2], torch.from_numpy(pix))
    assert torch.all(w[idx[:, 0], idx[:, 1], idx[:, 2]] > 0)
    for slot in range(6):
        p = (idx[slot * 10:(slot + 1) * 10, 1] * W + idx[slot * 10:(slot + 1) * 10, 2]).tolist()
        if slot == 1:
            assert len(set(p)) <= 3 and len(p) == 10  # image 2: with replacement
        else:
            assert len(set(p)) == len(p) == (7 if slot == 5 else 10)
    # distribution of the second draw of a slot: P(j second) = sum_i p_i p_j / (1 - p_i)
    wv = torch.tensor([0.5, 0.2, 0.2, 0.1])
    cdf2 = torch.cumsum(wv, 0)[None].contiguous().to(DEV)
    T = 20000
    uu = torch.rand(2 * T, generator=gen).to(DEV)
    ch, nz = torch.zeros(T, dtype=torch.int64, device=DEV), torch.tensor([4], dtype=torch.int32, device=DEV)
    out = torch.empty(2 * T, 3, dtype=torch.int64, device=DEV)
    _lib.check(_lib.lib().snerf_ist_sample(ops._ptr(cdf2), 1, 4, ops._ptr(ch), ops._ptr(nz), 2, ops._ptr(uu), 2 * T, ops._ptr(out), ops._stream()))
    first, second = out[0::2, 2].cpu(), out[1::2, 2].cpu()
    assert bool((first != second).all())
    p = wv / wv.sum()
    p2 = torch.stack([sum(p[i] * p[j] / (1 - p[i]) for i in range(4) if i != j) for j in range(4)])
    for emp, ref in ((torch.bincount(first, minlength=4) / T, p), (torch.bincount(second, minlength=4) / T, p2)):
        assert float(((emp - ref) ** 2 / ref).sum() * T) < 25.0  # chi-square, 3 dof: P(> 25) ~ 1.5e-5


def test_ist_sampler_with_few_images_and_more_than_1024_draws_per_slot():
    """per_image = 10 * ceil(0.15 R / M) (pixel_samplers.py:369) exceeds 1024 with few images or large batches (R = 4096 with M <= 5): the
    slot's removed-pixel list is sized for the draws it really makes.  Kernel == oracle index for index beyond 1024 draws in one slot, no
    repeats; and the sampler itself with M = 3."""
    import ctypes as C

    from oracle import ist_oracle as IO
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler

    gen = torch.Generator().manual_seed(8)
    M, H, W = 2, 36, 40
    w = (torch.rand(M, H, W, generator=gen) * 0.85 + 0.15).half().float()
    w[1, :, ::3] = 0.0
    cdf = torch.cumsum(w.reshape(M, -1), dim=1).contiguous()
    nnz = (w.reshape(M, -1) > 0).sum(1).to(torch.int32)
    chosen = torch.tensor([0, 1], dtype=torch.int64)
    per_image, n = 1100, 1100 + 37  # slot 0: 1100 draws of 1440 pixels without replacement; slot 1: 37
    u = torch.rand(n, generator=gen)
    idx = torch.empty(n, 3, dtype=torch.int64, device=DEV)
    keep = [cdf.to(DEV), chosen.to(DEV), nnz.to(DEV), u.to(DEV)]
    _lib.check(_lib.lib().snerf_ist_sample(ops._ptr(keep[0]), H, W, ops._ptr(keep[1]), ops._ptr(keep[2]), per_image, ops._ptr(keep[3]), n, ops._ptr(idx),
                                           ops._stream()))
    idx = idx.cpu()
    pix, img = IO.sample(cdf.numpy(), chosen.tolist(), nnz.numpy(), per_image, u.numpy())
    assert torch.equal(idx[:, 0], torch.from_numpy(img)) and torch.equal(idx[:, 1] * W + idx[:, 2], torch.from_numpy(pix))
    p0 = (idx[:per_image, 1] * W + idx[:per_image, 2]).tolist()
    assert len(set(p0)) == per_image and torch.all(w[idx[:, 0], idx[:, 1], idx[:, 2]] > 0)
    # the sampler with three images and the preset's batch: per_image = 2050, one slot of 614 draws
    M, H, W, R = 3, 54, 96, 4096
    maps = (torch.rand(M, H, W, generator=gen) > 0.5).half().to(DEV)
    batch = {"image": torch.zeros(M, H, W, 3, dtype=torch.uint8, device=DEV), "image_idx": torch.arange(M, device=DEV), "ist_weights": maps,
             "iter_steps": 5000}
    smp = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
    ind = smp.sample_method(R, M, H, W, batch=batch, device=DEV).cpu()
    assert ind.shape == (R, 3)
    ist = ind[:614]
    assert len(set(ist[:, 0].tolist())) == 1 and bool((maps.cpu()[ist[:, 0], ist[:, 1], ist[:, 2]] > 0).all())
    assert len(set((ist[:, 1] * W + ist[:, 2]).tolist())) == 614  # without replacement inside the image


@pytest.mark.parametrize("tag,gamma", [("0_05", 5e-2), ("0_2", 2e-1)])
@pytest.mark.parametrize("as_float", [False, True])
def test_isg_maps_match_reference_golden(tag, gamma, as_float):
    """ISG maps (compute_isg: per-camera lower median + psi) vs the reference's own output (G10b)."""
    from tests.conftest import load_golden
    from soccernerfs_amd.pixel_samplers import compute_isg

    g, gb = load_golden("g10_ist"), load_golden("g10b_isg")
    imgs = g["images_u8"].to(DEV)
    if as_float:
        imgs = imgs.float() / 255.0
    got = compute_isg(imgs, g["cam_ids"].to(DEV), gamma).float().cpu()
    want = gb["isg_" + tag]
    # fp16 rounding of values computed with a different association order of the three-term mean: at most one fp16 ulp apart
    assert float((got - want).abs().max()) <= 1e-3 and float((got != want).float().mean()) < 0.02


def test_isg_median_is_the_lower_median_with_ties_and_even_counts():
    from oracle import ist_oracle as IO
    from soccernerfs_amd.pixel_samplers import compute_isg

    gen = torch.Generator().manual_seed(6)
    M, H, W = 14, 9, 11
    u8 = torch.randint(0, 6, (M, H, W, 3), generator=gen, dtype=torch.uint8) * 40  # few distinct values: many ties
    ids = torch.tensor([0] * 6 + [5] * 5 + [2] * 2 + [9])  # even, odd, two and one image per camera
    ref = IO.compute_isg(u8.float() / 255.0, ids, 0.1).float()
    got = compute_isg(u8.to(DEV), ids.to(DEV), 0.1).float().cpu()
    assert float((got - ref).abs().max()) <= 1e-3
    assert float(got[13].abs().max()) == 0.0  # a single image is its own median

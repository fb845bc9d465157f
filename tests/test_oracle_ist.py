"""CPU: IST oracle vs golden maps captured from the reference's DynamicDataset.compute_ist."""
import pytest
import torch

from oracle import ist_oracle as IO
from tests.conftest import load_golden


@pytest.mark.parametrize("rng", ["1_0", "0_3"])
def test_ist_oracle_matches_reference(rng):
    g = load_golden("g10_ist")
    imgs = g["images_u8"].float() / 255.0
    out = IO.compute_ist(imgs, g["cam_ids"], g["cam_times"], float(rng.replace("_", ".")))
    assert out.dtype == torch.float16
    assert torch.equal(out.float(), g[f"ist_{rng}"])
    assert torch.all(out[-1] == 1.0)  # the single-image camera: uniform map
    frac = float((out[:-1] > 0).float().mean())
    assert 0.0 < frac < 0.2  # sparse maps: only the moving blob


@pytest.mark.parametrize("tag,gamma", [("0_05", 5e-2), ("0_2", 2e-1)])
def test_isg_oracle_matches_reference_golden(tag, gamma):
    """G10b (oracle/gen_golden_isg.py): the reference's own compute_isg on the G10 clip."""
    from oracle import ist_oracle as IO

    g, gb = load_golden("g10_ist"), load_golden("g10b_isg")
    got = IO.compute_isg(g["images_u8"].float() / 255.0, g["cam_ids"], gamma).float()
    assert torch.equal(got, gb["isg_" + tag])


@pytest.mark.parametrize("mode,k", [("normal", 20), ("randsteps", 20), ("randsteps", 33), ("lowfps", 30), ("lowfps", 20)])
@pytest.mark.parametrize("seed", [0, 1])
def test_image_cache_pick_modes_match_reference(mode, k, seed):
    """G10c (oracle/gen_golden_pick.py): the reference's CacheDataloader._get_batch_list with `random.seed(seed)` picks exactly these images."""
    import random

    from soccernerfs_amd.pixel_samplers import pick_cached_images, weights_cache_name

    g = load_golden("g10c_pick")
    got = pick_cached_images(g["times"], g["ids"], k, mode, rng=random.Random(seed))
    assert got == g[f"{mode}_{k}_{seed}"].tolist()
    assert weights_cache_name("ist", 0.75, 475, 540) == "ist-weights-0_75-train-475-540p.pt"
    assert weights_cache_name("isg", 0.05, 627, 540, eval_split=True) == "isg-weights-0.05-eval-627-540p.pt"


def test_oracle_sampler_follows_torch_multinomial_without_replacement():
    """oracle/ist_oracle.py::sample draws a slot's pixels by sequential removal; torch.multinomial(weights, k, replacement=False) -- what the
    reference calls (NS/data/pixel_samplers.py:400-402) -- samples from the same distribution: compare the joint distribution of ordered
    pairs statistically, and the replacement switch / per-slot counts exactly."""
    import numpy as np
    import torch

    from oracle import ist_oracle as IO

    w = torch.tensor([[0.0, 0.5, 0.0, 0.2, 0.2, 0.1]])
    cdf = torch.cumsum(w, 1).numpy()
    T = 6000
    rng = np.random.default_rng(0)
    pix, _ = IO.sample(cdf, [0] * T, np.array([4]), 2, rng.random(2 * T, dtype=np.float32))
    mine = np.zeros((6, 6))
    np.add.at(mine, (pix[0::2], pix[1::2]), 1)
    torch.manual_seed(0)
    ref = np.zeros((6, 6))
    for _ in range(T):
        a, b = torch.multinomial(w[0], 2, replacement=False).tolist()
        ref[a, b] += 1
    assert np.all(np.diag(mine) == 0) and mine[:, [0, 2]].sum() == 0 and mine[[0, 2]].sum() == 0
    mask = ref + mine > 0
    chi2 = (((mine - ref) ** 2) / (mine + ref))[mask].sum()  # two-sample chi-square, 11 dof
    assert chi2 < 45.0
    # fewer non-zero pixels than draws -> with replacement (repeats allowed); the last slot takes what is left of n
    pix, img = IO.sample(cdf, [0, 0, 0], np.array([4]), 5, rng.random(13, dtype=np.float32))
    assert len(pix) == 13 and set(pix.tolist()) <= {1, 3, 4, 5} and all(len(set(pix[s:s + 5].tolist())) < 5 for s in (0, 5))

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

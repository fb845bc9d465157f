"""PSNR / SSIM of the full-image evaluation path: properties of the oracle (CPU) and the device implementation against it (GPU)."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO


def _pair(seed=0, B=2, C=3, H=40, W=56):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = 0.5 + 0.3 * np.sin(xx / 7.0)[None, None] * np.cos(yy / 5.0)[None, None] + 0.05 * rng.standard_normal((B, C, H, W))
    base = np.clip(base, 0, 1).astype(np.float32)
    noisy = np.clip(base + 0.1 * rng.standard_normal(base.shape), 0, 1).astype(np.float32)
    return base, noisy


def test_ssim_oracle_properties():
    a, b = _pair()
    assert abs(MO.ssim(a, a) - 1.0) < 1e-12
    s = MO.ssim(a, b)
    assert 0.0 < s < 1.0 and abs(s - MO.ssim(b, a)) < 1e-12  # symmetric
    assert MO.ssim(a, np.clip(a + 0.05 * np.random.default_rng(1).standard_normal(a.shape), 0, 1).astype(np.float32)) > s  # less noise, higher SSIM
    # luminance shift lowers only the luminance term: constant images, data_range 1 -> (2 mu1 mu2 + c1) / (mu1^2 + mu2^2 + c1)
    x, y = np.full((1, 1, 32, 32), 0.4, np.float32), np.full((1, 1, 32, 32), 0.6, np.float32)
    c1 = 0.01 ** 2
    assert abs(MO.ssim(x, y, data_range=1.0) - (2 * 0.4 * 0.6 + c1) / (0.4 ** 2 + 0.6 ** 2 + c1)) < 1e-6


def test_metrics_reject_bad_shapes():
    from soccernerfs_amd.metrics import structural_similarity_index_measure

    with pytest.raises(ValueError):
        structural_similarity_index_measure(torch.rand(3, 8, 8), torch.rand(3, 8, 8))


@pytest.mark.gpu
def test_device_ssim_and_psnr_match_oracle():
    from soccernerfs_amd.metrics import psnr, structural_similarity_index_measure

    a, b = _pair(3)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert abs(float(structural_similarity_index_measure(ta, tb)) - MO.ssim(a, b)) < 2e-5
    assert abs(float(structural_similarity_index_measure(ta, tb, data_range=1.0)) - MO.ssim(a, b, data_range=1.0)) < 2e-5
    assert abs(float(structural_similarity_index_measure(ta, ta)) - 1.0) < 1e-5
    mse = float(np.mean((a.astype(np.float64) - b) ** 2))
    assert abs(float(psnr(ta, tb)) - 10 * np.log10(1.0 / mse)) < 1e-3


@pytest.mark.gpu
def test_model_image_metrics():
    """KPlanesModel.get_image_metrics_and_images (kplanes.py:454-498): [H,W,3] image + outputs -> psnr / ssim floats and the combined image."""
    from tests.test_formats_cpu import _small_kplanes

    model = _small_kplanes().cuda().eval()
    H, W = 24, 32
    img = torch.rand(H, W, 3, device="cuda")
    outputs = {"rgb": (img + 0.05 * torch.randn_like(img)).clamp(0, 1), "accumulation": torch.rand(H, W, 1, device="cuda"),
               "depth": torch.rand(H, W, 1, device="cuda"), "prop_depth_0": torch.rand(H, W, 1, device="cuda"), "prop_depth_1": torch.rand(H, W, 1, device="cuda")}
    metrics, images = model.get_image_metrics_and_images(outputs, {"image": img.cpu()})
    a = img.permute(2, 0, 1)[None].cpu().numpy()
    b = outputs["rgb"].permute(2, 0, 1)[None].cpu().numpy()
    assert abs(metrics["ssim"] - MO.ssim(a, b)) < 2e-5
    assert abs(metrics["psnr"] - 10 * np.log10(1.0 / float(np.mean((a.astype(np.float64) - b) ** 2)))) < 1e-3
    assert images["img"].shape == (H, 2 * W, 3) and set(images) >= {"img", "accumulation", "depth", "prop_depth_0", "prop_depth_1"}

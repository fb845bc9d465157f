"""The reference's own unit tests for this path, re-run on the HIP classes with the same constructor arguments and the same assertions:
NSR/tests/model_components/test_ray_sampler.py (uniform :18-35, pdf :89-111), NSR/tests/model_components/test_renderers.py
(rgb :11-26, accumulation :47-58, depth :61-87) and the temporal-grid KAT (tests/test_gpu_tgrid.py::test_reference_known_answer).
The samplers the K-Planes / NeRFPlayer-nerfacto presets never construct (LinearDisparity, Sqrt, Log) and the SH renderer are not built."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bundle():
    from soccernerfs_amd.rays import RayBundle
    from soccernerfs_amd.scene_colliders import NearFarCollider

    origins = torch.zeros((10, 3), device=DEV)
    directions = torch.ones_like(origins)
    radius = torch.ones((10, 1), device=DEV)
    ray_bundle = RayBundle(origins=origins, directions=directions, pixel_area=radius)
    return NearFarCollider(near_plane=2, far_plane=4)(ray_bundle)


def test_uniform_sampler():
    from soccernerfs_amd.ray_samplers import UniformSampler

    num_samples = 15
    sampler = UniformSampler(num_samples=num_samples)
    ray_samples = sampler(_bundle())
    assert ray_samples.frustums.get_positions().shape[-2] == num_samples
    # more precise than the reference's test: samples lie inside [near, far] along the ray and are ordered
    starts, ends = ray_samples.frustums.starts[..., 0], ray_samples.frustums.ends[..., 0]
    assert float(starts.min()) >= 2.0 - 1e-6 and float(ends.max()) <= 4.0 + 1e-6
    assert bool((ends >= starts).all()) and bool((starts[:, 1:] >= starts[:, :-1]).all())


def test_pdf_sampler():
    from soccernerfs_amd.ray_samplers import PDFSampler, UniformSampler

    num_samples = 15
    ray_bundle = _bundle()
    coarse_ray_samples = UniformSampler(num_samples=num_samples)(ray_bundle)
    weights = torch.ones((10, num_samples, 1), device=DEV)
    pdf_sampler = PDFSampler(num_samples)
    fine = pdf_sampler(ray_bundle, coarse_ray_samples, weights, num_samples)
    # include_original (the class default): the 16 existing bin edges are merged into the 16 new ones -> 31 samples (ray_samplers.py:353-354)
    assert fine.frustums.get_positions().shape[-2] == 2 * num_samples + 1
    # uniform weights: the resampled bins stay inside [near, far]
    assert float(fine.frustums.starts.min()) >= 2.0 - 1e-5 and float(fine.frustums.ends.max()) <= 4.0 + 1e-5


def test_rgb_renderer():
    from soccernerfs_amd import renderers

    num_samples = 10
    rgb_samples = torch.ones((3, num_samples, 3), device=DEV)
    weights = torch.ones((3, num_samples, 1), device=DEV)
    weights /= torch.sum(weights, dim=-2, keepdim=True)
    rgb_renderer = renderers.RGBRenderer()
    rgb = rgb_renderer(rgb=rgb_samples, weights=weights)
    assert torch.max(rgb) > 0.9
    rgb = rgb_renderer(rgb=rgb_samples * 0, weights=weights)
    assert float(torch.max(rgb)) == pytest.approx(0, abs=1e-6)


def test_acc_renderer():
    from soccernerfs_amd import renderers

    num_samples = 10
    weights = torch.ones((3, num_samples, 1), device=DEV)
    weights /= torch.sum(weights, dim=-2, keepdim=True)
    accumulation = renderers.AccumulationRenderer()(weights=weights)
    assert torch.max(accumulation) > 0.9


def test_depth_renderer():
    from soccernerfs_amd import renderers
    from soccernerfs_amd.ray_samplers import UniformSampler

    num_samples = 10
    ray_samples = UniformSampler(num_samples=num_samples)(_bundle())
    weights = torch.ones((10, num_samples, 1), device=DEV)
    weights /= torch.sum(weights, dim=-2, keepdim=True)
    for method in ("median", "expected"):
        depth = renderers.DepthRenderer(method=method)(weights=weights, ray_samples=ray_samples)
        assert torch.min(depth) > 0
        assert float(depth.min()) >= 2.0 - 1e-5 and float(depth.max()) <= 4.0 + 1e-5

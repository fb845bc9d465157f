"""The reference's ray-container unit tests (NSR/tests/cameras/test_rays.py) on soccernerfs_amd.rays -- plain tensor containers, so they run
on the CPU.  `get_gaussian_blob` (mip-NeRF conical frustums, rays.py:67-85) is not on the K-Planes / NeRFPlayer path and is not built."""
import pytest
import torch

from soccernerfs_amd.rays import Frustums


def test_frustum_get_position():
    origin = torch.Tensor([0, 1, 2])[None, ...]
    direction = torch.Tensor([0, 1, 0])[None, ...]
    frustum_start = torch.Tensor([2])[None, ...]
    frustum_end = torch.Tensor([3])[None, ...]
    target_position = torch.Tensor([0, 3.5, 2])[None, ...]
    frustum = Frustums(origins=origin, directions=direction, starts=frustum_start, ends=frustum_end, pixel_area=torch.ones((1, 1)))
    positions = frustum.get_positions()
    assert positions == pytest.approx(target_position, abs=1e-6)


def test_frustum_apply_masks():
    frustum = Frustums(origins=torch.ones((5, 3)), directions=torch.ones((5, 3)), starts=torch.ones((5, 1)), ends=torch.ones((5, 1)),
                       pixel_area=torch.ones((5, 1)))
    mask = torch.tensor([False, True, False, True, True], dtype=torch.bool)
    frustum = frustum[mask]
    assert frustum.origins.shape == (3, 3)
    assert frustum.directions.shape == (3, 3)
    assert frustum.starts.shape == (3, 1)
    assert frustum.ends.shape == (3, 1)
    assert frustum.pixel_area.shape == (3, 1)


def test_get_mock_frustum():
    f = Frustums.get_mock_frustum()
    assert f.shape == (1,) and f.get_positions().shape == (1, 3)

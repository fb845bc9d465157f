"""Pinhole camera table + RayGenerator with the interface of NS/cameras/cameras.py (perspective, no distortion slice)
and NS/model_components/ray_generators.py."""
from typing import Optional

import torch
from torch import nn

from . import ops
from .rays import RayBundle


class Cameras:
    """camera_to_worlds [M,3,4]; fx, fy, cx, cy [M] (or scalars); width/height ints; times [M]."""

    def __init__(self, camera_to_worlds, fx, fy, cx, cy, width: int, height: int, times: Optional[torch.Tensor] = None, **kwargs):
        M = camera_to_worlds.shape[0]
        dev = camera_to_worlds.device
        ex = lambda v: (v.reshape(-1).float() if isinstance(v, torch.Tensor) else torch.tensor([float(v)])).to(dev).expand(M).contiguous()
        self.camera_to_worlds = camera_to_worlds.float().contiguous()
        self.fx, self.fy, self.cx, self.cy = ex(fx), ex(fy), ex(cx), ex(cy)
        self.width, self.height = int(width), int(height)
        self.times = None if times is None else times.reshape(-1).float().to(dev).contiguous()
        self.ids = kwargs.get("ids")  # camera uid per image (Broadcast-style parser), carried for the samplers / metrics
        self.distortion_params = kwargs.get("distortion_params")  # carried only: rays are generated for the pinhole model

    def rescale_output_resolution(self, scaling_factor: float) -> None:
        """NS/cameras/cameras.py:792-816."""
        s = float(scaling_factor)
        self.fx, self.fy, self.cx, self.cy = self.fx * s, self.fy * s, self.cx * s, self.cy * s
        self.height = int(torch.tensor(float(self.height)).mul(torch.tensor(s)).to(torch.int64))
        self.width = int(torch.tensor(float(self.width)).mul(torch.tensor(s)).to(torch.int64))

    def __len__(self):
        return self.camera_to_worlds.shape[0]

    def to(self, device):
        return Cameras(self.camera_to_worlds.to(device), self.fx.to(device), self.fy.to(device), self.cx.to(device), self.cy.to(device),
                       self.width, self.height, None if self.times is None else self.times.to(device), ids=self.ids,
                       distortion_params=self.distortion_params)

    def generate_rays(self, camera_indices: torch.Tensor, coords: Optional[torch.Tensor] = None, aabb=None, near_plane=0.0,
                      training=True, **kwargs) -> RayBundle:
        """camera_indices [R,1] (or an int for a full image); coords [R,2] = (y+0.5, x+0.5) pixel centres."""
        if isinstance(camera_indices, int):
            ys, xs = torch.meshgrid(torch.arange(self.height), torch.arange(self.width), indexing="ij")
            idx = torch.stack([torch.full_like(ys, camera_indices), ys, xs], -1).reshape(-1, 3).to(self.camera_to_worlds.device)
            shape = (self.height, self.width)
        else:
            yx = torch.floor(coords).long()
            idx = torch.cat([camera_indices.reshape(-1, 1).long(), yx], dim=-1)
            shape = None
        out = ops.generate_rays(idx.contiguous(), self.fx, self.fy, self.cx, self.cy, self.camera_to_worlds, self.times, aabb, near_plane, training)
        rb = RayBundle(origins=out["origins"], directions=out["directions"], pixel_area=out["pixel_area"], camera_indices=out["camera_indices"],
                       nears=out.get("nears"), fars=out.get("fars"), metadata={"directions_norm": out["directions_norm"]},
                       times=out["times"] if self.times is not None else None)
        if shape is not None:
            rb = rb._map(lambda t: t.view(*shape, t.shape[-1]))
        return rb


class RayGenerator(nn.Module):
    """ray_generators.py:27-59 (camera optimiser 'off')."""

    def __init__(self, cameras: Cameras, pose_optimizer=None) -> None:
        super().__init__()
        self.cameras = cameras

    def forward(self, ray_indices: torch.Tensor) -> RayBundle:
        idx = ray_indices.long()
        coords = idx[:, 1:3].float() + 0.5
        return self.cameras.generate_rays(camera_indices=idx[:, 0:1], coords=coords)

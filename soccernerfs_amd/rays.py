"""Ray / sample containers with the attribute names and shapes of NS/cameras/rays.py (Frustums :31-102,
RaySamples :105-170, RayBundle :173-277) -- the data-layout contract of nerfstudio's plugin surface.

Storage is compact (per-ray origins/directions + bin edges [R,S+1]); the reference's broadcast views
([R,S,3] origins, [R,S,1] starts ...) are produced on demand as torch views, so a caller written against
nerfstudio sees the same tensors while the HIP kernels read the compact form.
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch

from . import ops


class Frustums:
    """origins/directions [..,3], starts/ends/pixel_area [..,1] (NS/cameras/rays.py:31-46)."""

    def __init__(self, origins, directions, starts, ends, pixel_area, offsets=None):
        self.origins, self.directions, self.starts, self.ends, self.pixel_area, self.offsets = origins, directions, starts, ends, pixel_area, offsets

    @property
    def shape(self):
        return self.starts.shape[:-1]

    def get_positions(self) -> torch.Tensor:
        """origins + directions * (starts + ends) / 2 (rays.py:48-57)."""
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        if self.offsets is not None:
            pos = pos + self.offsets
        return pos

    def get_start_positions(self) -> torch.Tensor:
        return self.origins + self.directions * self.starts

    def set_offsets(self, offsets):
        """rays.py:63-65."""
        self.offsets = offsets

    def __getitem__(self, idx) -> "Frustums":
        """Index / mask every field alike, as TensorDataclass does (NS/utils/tensor_dataclass.py:131-140)."""
        return Frustums(self.origins[idx], self.directions[idx], self.starts[idx], self.ends[idx], self.pixel_area[idx],
                        None if self.offsets is None else self.offsets[idx])

    @classmethod
    def get_mock_frustum(cls, device="cpu") -> "Frustums":
        """rays.py:87-102: a single frustum at the origin looking along +z."""
        return cls(origins=torch.ones((1, 3), device=device), directions=torch.ones((1, 3), device=device), starts=torch.ones((1, 1), device=device),
                   ends=torch.ones((1, 1), device=device), pixel_area=torch.ones((1, 1), device=device))


class RaySamples:
    """Samples along rays (rays.py:105-170).  Built by RayBundle.get_ray_samples or the samplers."""

    def __init__(self, frustums: Frustums, camera_indices=None, deltas=None, spacing_starts=None, spacing_ends=None,
                 spacing_to_euclidean_fn: Optional[Callable] = None, metadata=None, times=None, _compact=None):
        self.frustums, self.camera_indices, self.deltas = frustums, camera_indices, deltas
        self.spacing_starts, self.spacing_ends, self.spacing_to_euclidean_fn = spacing_starts, spacing_ends, spacing_to_euclidean_fn
        self.metadata, self.times = metadata, times
        self._compact = _compact  # dict(origins [R,3], directions [R,3], times [R,1], ebins [R,S+1], sbins [R,S+1], nears, fars, kind)

    @property
    def shape(self):
        return self.frustums.shape

    def get_weights(self, densities: torch.Tensor) -> torch.Tensor:
        """RaySamples.get_weights (rays.py:127-149): densities [R,S,1] -> weights [R,S,1]."""
        if self._compact is not None:
            ebins = self._compact["ebins"]
        else:
            ebins = torch.cat([self.frustums.starts[..., 0], self.frustums.ends[..., -1:, 0]], dim=-1).contiguous()
        return ops.get_weights(densities[..., 0], ebins)[..., None]


@dataclass
class RayBundle:
    """rays.py:173-277."""

    origins: torch.Tensor
    directions: torch.Tensor
    pixel_area: torch.Tensor
    camera_indices: Optional[torch.Tensor] = None
    nears: Optional[torch.Tensor] = None
    fars: Optional[torch.Tensor] = None
    metadata: Optional[Dict[str, torch.Tensor]] = None
    times: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return self.origins.numel() // self.origins.shape[-1]

    def set_camera_indices(self, camera_index: int) -> None:
        self.camera_indices = torch.ones_like(self.origins[..., 0:1]).long() * camera_index

    def _map(self, fn) -> "RayBundle":
        g = lambda t: None if t is None else fn(t)
        md = None if self.metadata is None else {k: fn(v) for k, v in self.metadata.items()}
        return RayBundle(fn(self.origins), fn(self.directions), fn(self.pixel_area), g(self.camera_indices), g(self.nears), g(self.fars), md,
                         g(self.times))

    def flatten(self) -> "RayBundle":
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        return self.flatten()._map(lambda t: t[start_idx:end_idx])

    def to(self, device) -> "RayBundle":
        return self._map(lambda t: t.to(device))

    def get_ray_samples(self, bin_starts, bin_ends, spacing_starts=None, spacing_ends=None, spacing_to_euclidean_fn=None,
                        _compact=None) -> RaySamples:
        """rays.py:233-277: bin_starts/ends [R,S,1]."""
        fr = Frustums(origins=self.origins[..., None, :], directions=self.directions[..., None, :], starts=bin_starts, ends=bin_ends,
                      pixel_area=self.pixel_area[..., None, :])
        md = None if self.metadata is None else {k: v[..., None, :] for k, v in self.metadata.items()}
        return RaySamples(frustums=fr, camera_indices=None if self.camera_indices is None else self.camera_indices[..., None, :],
                          deltas=bin_ends - bin_starts, spacing_starts=spacing_starts, spacing_ends=spacing_ends,
                          spacing_to_euclidean_fn=spacing_to_euclidean_fn, metadata=md,
                          times=None if self.times is None else self.times[..., None, :], _compact=_compact)

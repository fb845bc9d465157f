"""Scene colliders with the interface of NS/model_components/scene_colliders.py."""
import torch
from torch import nn

from . import ops
from .rays import RayBundle


class SceneBox:
    """NS/data/scene_box.py: aabb [2,3]."""

    def __init__(self, aabb: torch.Tensor):
        self.aabb = aabb

    @staticmethod
    def get_normalized_positions(positions, aabb):
        return (positions - aabb[0]) / (aabb[1] - aabb[0])


class SceneCollider(nn.Module):
    def set_nears_and_fars(self, ray_bundle) -> RayBundle:
        raise NotImplementedError

    def forward(self, ray_bundle: RayBundle) -> RayBundle:
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            return ray_bundle
        return self.set_nears_and_fars(ray_bundle)


class AABBBoxCollider(SceneCollider):
    """scene_colliders.py:47-108; near_plane applies in training mode only (:91)."""

    def __init__(self, scene_box: SceneBox, near_plane: float = 0.0, **kwargs) -> None:
        super().__init__()
        self.scene_box, self.near_plane = scene_box, near_plane

    def _intersect_with_aabb(self, rays_o, rays_d, aabb):
        n, f = ops.aabb_collide(rays_o, rays_d, aabb, self.near_plane, self.training)
        return n[:, 0], f[:, 0]

    def set_nears_and_fars(self, ray_bundle: RayBundle) -> RayBundle:
        nears, fars = self._intersect_with_aabb(ray_bundle.origins, ray_bundle.directions, self.scene_box.aabb)
        ray_bundle.nears, ray_bundle.fars = nears[..., None], fars[..., None]
        return ray_bundle


class NearFarCollider(SceneCollider):
    """scene_colliders.py:170-188."""

    def __init__(self, near_plane: float, far_plane: float, **kwargs) -> None:
        super().__init__()
        self.near_plane, self.far_plane = near_plane, far_plane

    def set_nears_and_fars(self, ray_bundle: RayBundle) -> RayBundle:
        ones = torch.ones_like(ray_bundle.origins[..., 0:1])
        ray_bundle.nears = ones * (self.near_plane if self.training else 0)
        ray_bundle.fars = ones * self.far_plane
        return ray_bundle

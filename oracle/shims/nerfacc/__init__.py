"""Stub for nerfacc 0.3.5: the proposal-sampler path never calls into it."""
import enum
class ContractionType(enum.Enum):
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2
class OccupancyGrid:
    def __init__(self, *a, **k):
        raise RuntimeError("nerfacc stub: OccupancyGrid is not on the K-Planes path")
def _raise(*a, **k):
    raise RuntimeError("nerfacc stub: packed-sample path is out of scope")
ray_marching = accumulate_along_rays = pack_info = render_weight_from_density = _raise
ray_aabb_intersect = unpack_info = render_visibility = _raise
def cuda_toolkit_available():
    return False
class _Cuda:
    pass
cuda = _Cuda()

"""soccernerfs_amd -- MI355X-native K-Planes / NeRFPlayer hot path behind nerfstudio's plugin surface.

Python host on PyTorch-ROCm (device memory, streams, torch.distributed) calling hand-written
gfx950 HIP kernels through the C ABI in include/snerf.h.  See DESIGN.md.
"""
__version__ = "0.1.0"

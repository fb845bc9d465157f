"""Micro-benchmarks of individual libsnerf kernels at BASELINE config-2 sizes (GPU only; dev tool)."""
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops  # noqa: E402
from soccernerfs_amd.plane_set import PlaneSet  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def ray_like_points(R, S, dev, lo=-1.0, hi=1.0):
    """Samples along R random rays through the box (spatially coherent like the real workload)."""
    o = torch.rand(R, 1, 3, device=dev) * 2 - 1
    d = torch.nn.functional.normalize(torch.rand(R, 1, 3, device=dev) * 2 - 1, dim=-1)
    t = torch.sort(torch.rand(R, S, 1, device=dev) * 0.5, dim=1).values
    p = (o * 0.3 + d * t).clamp(-1, 1)
    p = (p + 1) / 2 * (hi - lo) + lo
    tm = (torch.rand(R, 1, 1, device=dev) * 2 - 1).expand(R, S, 1)
    return torch.cat([p, tm], -1).reshape(-1, 4).contiguous()


def main():
    dev = torch.device("cuda:0")
    R = 4096
    res = {}
    # main field: 5 scales, C=32
    ps = PlaneSet(32, [[64 * m] * 3 + [100] for m in (1, 2, 4, 8, 16)], concat=True, device=dev)
    for name, pts in (("random", torch.rand(R * 64, 4, device=dev) * 2 - 1), ("raylike", ray_like_points(R, 64, dev))):
        N = pts.shape[0]
        out = ops.interpolate_kplanes(pts, ps)
        gout = torch.rand_like(out)
        ms_f = timeit(lambda: ops.interpolate_kplanes(pts, ps.requires_grad_(False)))
        ps.requires_grad_(True)
        import ctypes as C
        from soccernerfs_amd import _lib
        desc, co = ps.desc(), ops.coords_from_points(pts)
        gpl = torch.zeros_like(ps.planes)
        def bwd():
            _lib.check(_lib.lib().snerf_kplanes_gather_bwd(C.byref(desc), C.c_void_p(ps.planes.data_ptr()), C.byref(co), C.c_int64(N),
                                                           C.c_void_p(gout.data_ptr()), C.c_void_p(gpl.data_ptr()), ops._stream()))
        ms_b = timeit(bwd)
        bytes_f = N * 5 * 6 * 4 * 32 * 4
        print(f"field gather {name}: fwd {ms_f:.3f} ms ({bytes_f / ms_f / 1e9:.2f} TB/s alg), bwd {ms_b:.3f} ms ({2 * bytes_f / ms_b / 1e9:.2f} TB/s alg rmw)")
    for lvl, (S, r) in enumerate(((256, 128), (128, 256))):
        pp = PlaneSet(8, [[r, r, r, 100]], concat=False, a=0.1, b=0.15, device=dev)
        pts = ray_like_points(R, S, dev, 0.0, 1.0)
        N = pts.shape[0]
        pp.requires_grad_(False)
        ms_f = timeit(lambda: ops.interpolate_kplanes(pts, pp))
        bytes_f = N * 6 * 4 * 8 * 4
        print(f"prop{lvl} gather: fwd {ms_f:.3f} ms ({bytes_f / ms_f / 1e9:.2f} TB/s alg)")
    z = torch.zeros(156_000_000, device=dev)
    ms = timeit(lambda: z.zero_())
    print(f"memset 624MB: {ms:.3f} ms ({z.numel() * 4 / ms / 1e9:.2f} TB/s)")
    y = torch.empty_like(z)
    ms = timeit(lambda: y.copy_(z))
    print(f"copy 624MB: {ms:.3f} ms ({2 * z.numel() * 4 / ms / 1e9:.2f} TB/s)")


if __name__ == "__main__":
    main()

"""What a kernel with the optimiser sweep's byte mix reaches on this GPU when it has NO stencil: the plain Adam kernel (adam_kernel: p, g, m, v
read, p_out, m, v written, one float4 per lane, nontemporal) over a buffer of the field planes' size, beside the fused sweep
(plane_reg_kernel<32,true>: the same streams + the regularisers' neighbour reads of p) on the k-planes preset's plane set.  Gradient density as
in training: a fraction `--touched` of the texels holds a non-zero gradient (the kernels clear only those).  HIP events around 20 launches each.

    python tools/stream_ceiling.py [--touched 0.2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from soccernerfs_amd import ops
from soccernerfs_amd.plane_set import PlaneSet


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--touched", type=float, default=0.2)
    args = ap.parse_args()
    dev = "cuda:0"
    ps = PlaneSet(32, [[64 * m, 64 * m, 64 * m, 100] for m in (1, 2, 4, 8, 16)], concat=True, device=dev)
    n = ps.numel
    p, p2 = ps.planes.detach(), torch.empty(n, device=dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    g = torch.zeros(n, device=dev)
    mask = (torch.rand(n // 32, device=dev) < args.touched)
    gsrc = (torch.randn(n // 32, 32, device=dev) * 1e-3 * mask[:, None]).reshape(-1)
    del mask
    losses = torch.zeros(ops.REG_SLOTS * 16, device=dev)
    state = {"flip": False}

    def plain():
        g.copy_(gsrc)

    t_copy = timed(plain)

    def adam_plain():
        g.copy_(gsrc)
        a, b = (p, p2) if not state["flip"] else (p2, p)
        ops.adam_step(a, g, m, v, 5, 1e-2, zero_grad=True, p_out=b)
        state["flip"] = not state["flip"]

    def sweep():
        g.copy_(gsrc)
        a, b = (p, p2) if not state["flip"] else (p2, p)
        ops.adam_planes_step(ps, a, b, g, m, v, (2e-4, 1e-3, 1e-4), losses, 5, 1e-2, zero_grad=True)
        state["flip"] = not state["flip"]

    t_adam = timed(adam_plain) - t_copy
    t_sweep = timed(sweep) - t_copy
    real = n * (28 + 8 * args.touched)  # p, g, m, v read + p, m, v written; g re-written where it was non-zero (and that line read back: 64-B sectors)
    out = {"params": n, "touched": args.touched, "gradient_refill_ms": round(t_copy, 4),
           "plain_adam_ms": round(t_adam, 4), "sweep_ms": round(t_sweep, 4),
           "plain_adam_TBps_32B": round(n * 32 / t_adam / 1e9, 3), "sweep_TBps_32B": round(n * 32 / t_sweep / 1e9, 3),
           "plain_adam_TBps_moved": round(real / t_adam / 1e9, 3), "sweep_TBps_moved": round(real / t_sweep / 1e9, 3),
           "sweep_over_plain": round(t_sweep / t_adam, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

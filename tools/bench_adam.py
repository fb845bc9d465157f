"""The dominant sweep alone: fused Adam + plane regularisers (plane_reg_kernel<32,true>) against the plain Adam kernel over the same
153 M field-plane parameters (k-planes preset).  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import ops
from soccernerfs_amd.plane_set import PlaneSet

dev = "cuda:0"
ps = PlaneSet(32, [[64 * m, 64 * m, 64 * m, 100] for m in (1, 2, 4, 8, 16)], concat=True, device=dev)
n = ps.numel
p = ps.planes.detach().clone(); po = torch.zeros_like(p)
g = torch.rand(n, device=dev) - 0.5; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
losses = torch.zeros(ops.REG_SLOTS, 16, device=dev)
def timed(fn, k=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
t_f = timed(lambda: ops.adam_planes_step(ps, p, po, g, m, v, (2e-4, 1e-3, 1e-4), losses, 3, 1e-2, zero_grad=True))
t_p = timed(lambda: ops.adam_step(p, g, m, v, 3, 1e-2, zero_grad=True, p_out=po))
B = 32.0 * n
print(f"params {n/1e6:.1f} M: fused Adam+regularisers {t_f:.3f} ms = {B/t_f/1e9:.2f} TB/s ({B/t_f/8e9:.3f} of 8 TB/s);  plain Adam {t_p:.3f} ms = {B/t_p/1e9:.2f} TB/s ({B/t_p/8e9:.3f})")

"""Fixed vs per-sample cost of the 16-bit MLP backward kernels: time at N and 4 N (dev tool)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import _lib, ops
from soccernerfs_amd.tcnn_compat import Network

dev = "cuda:0"
L = _lib.lib()
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for d_in, ldx, d_out, hidden, nh, act, N0 in ((15, 16, 3, 64, 2, "Sigmoid", 4096 * 48), (8, 8, 1, 64, 1, "None", 4096 * 256), (160, 160, 16, 128, 1, "None", 4096 * 48)):
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": hidden, "n_hidden_layers": nh}
    net = Network(d_in, d_out, cfg, operands="bf16").to(dev)
    gw = torch.zeros_like(net.params)
    for N in (N0 // 4, N0, N0 * 4):
        x, gy, gx = torch.rand(N, ldx, device=dev), torch.rand(N, d_out, device=dev) - 0.5, torch.empty(N, ldx, device=dev)
        y = torch.empty(N, d_out, device=dev)
        f = lambda: _lib.check(L.snerf_mlp_fwd(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x), ldx, C.c_int64(N), ops._ptr(y), d_out, -1, None, ops._stream()))
        b = lambda: _lib.check(L.snerf_mlp_bwd(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x), ldx, C.c_int64(N), ops._ptr(gy), d_out, -1, None, ops._ptr(gx), ldx, ops._ptr(gw), ops._stream()))
        bnw = lambda: _lib.check(L.snerf_mlp_bwd(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x), ldx, C.c_int64(N), ops._ptr(gy), d_out, -1, None, ops._ptr(gx), ldx, None, ops._stream()))
        print(f"{d_in}->{hidden}x{nh}->{d_out}  N={N:8d}: fwd {timed(f):7.1f} us   bwd {timed(b):7.1f} us   bwd without gW {timed(bnw):7.1f} us")

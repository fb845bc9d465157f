"""Micro-benchmark of the MLP kernels at config-2 sizes (dev tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd.tcnn_compat import Network
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
cfg = lambda h, nh, act="None": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": h, "n_hidden_layers": nh}
for name, (din, dout, h, nh, N) in {"sigma 160x128": (160, 16, 128, 1, 64 * 4096), "color 15x64x2": (15, 3, 64, 2, 64 * 4096),
                                    "prop 8x64 (256)": (8, 1, 64, 1, 256 * 4096), "prop 8x64 (128)": (8, 1, 64, 1, 128 * 4096)}.items():
    net = Network(din, dout, cfg(h, nh, "Sigmoid" if dout == 3 else "None")).to(dev)
    x = torch.rand(N, din, device=dev).requires_grad_(True)
    y = net(x)
    gy = torch.rand_like(y)
    fwd = timeit(lambda: net(x.detach()))
    def fb():
        x.grad = None; net.params.grad = None
        net(x).backward(gy)
    tot = timeit(fb)
    flops = 2 * N * sum(a * b for a, b in zip(net.dims[:-1], net.dims[1:]))
    print(f"{name:18s} fwd {fwd:.3f} ms ({flops / fwd / 1e9:.1f} TF)  fwd+bwd {tot:.3f} ms  (bwd ~{tot - fwd:.3f} ms, {3 * flops / (tot - fwd) / 1e9:.1f} TF)")

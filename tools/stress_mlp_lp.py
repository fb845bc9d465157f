"""Stress: 16-bit-operand MLP backward kernels of two different shapes running CONCURRENTLY on two streams must give the results they give
alone (gX is atomics-free, hence bit-reproducible; gW up to atomic order).  Dev tool for a suspected scheduling-dependent race."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd.tcnn_compat import Network

dev = "cuda:0"
op = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(0)
cfgs = [(160, 16, 128, 4096 * 64), (8, 1, 64, 4096 * 256)]
nets, xs, gos = [], [], []
for d_in, d_out, h, N in cfgs:
    nets.append(Network(d_in, d_out, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": h, "n_hidden_layers": 1},
                        operands=op).to(dev))
    xs.append(torch.rand(N, d_in, device=dev) - 0.3)
    gos.append(torch.rand(N, d_out, device=dev) - 0.5)

def run(i):
    x = xs[i].clone().requires_grad_(True)
    nets[i].params.grad = None
    y = nets[i](x)
    y.backward(gos[i])
    return y.detach().clone(), x.grad.clone(), nets[i].params.grad.clone()

ref = [run(0), run(1)]
torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for it in range(30):
    outs = [None, None]
    for i in (0, 1):
        s[i].wait_stream(torch.cuda.current_stream())
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(s[i]):
                outs[i] = run(i)
    torch.cuda.synchronize()
    for i in (0, 1):
        dy = float((outs[i][0] - ref[i][0]).abs().max()); dx = float((outs[i][1] - ref[i][1]).abs().max())
        dw = float((outs[i][2] - ref[i][2]).abs().max() / ref[i][2].abs().max())
        if dy != 0.0 or dx != 0.0 or dw > 1e-4:
            bad += 1
            print(f"iter {it} net {i}: max|dy| {dy:.3e} max|dx| {dx:.3e} rel dW {dw:.3e}")
print(op, "mismatching (iteration, net) pairs:", bad, "of 60")

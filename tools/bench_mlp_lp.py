"""fp32-operand vs bf16-operand MLP kernels at the K-Planes preset's shapes (sigma_net 160->128->16 over 262 144 samples, proposal nets
8->64->1 over 1 048 576 / 524 288 samples).  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd.tcnn_compat import Network

dev = "cuda:0"
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for d_in, d_out, hidden, N in ((160, 16, 128, 4096 * 64), (8, 1, 64, 4096 * 256), (8, 1, 64, 4096 * 128)):
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": hidden, "n_hidden_layers": 1}
    x = torch.rand(N, d_in, device=dev)
    go = torch.rand(N, d_out, device=dev)
    row = {}
    for op in ("fp32", "bf16", "fp16"):
        net = Network(d_in, d_out, cfg, operands=op).to(dev)
        xg = x.clone().requires_grad_(True)
        with torch.no_grad():
            row[op + " fwd"] = timed(lambda: net(x))
        def fb():
            net.params.grad = None; xg.grad = None
            net(xg).backward(go)
        row[op + " fwd+bwd"] = timed(fb)
    print((d_in, hidden, d_out, N), {k: round(v, 3) for k, v in row.items()})

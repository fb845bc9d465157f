"""Static hash grid kernels alone at the full NeRFPlayer's size (196 608 samples, 16 levels x 2 features, 2^18 rows per level): forward,
table-gradient backward, coordinate-gradient backward, both -- for coordinates inside [0,1] and for 'deformed' ones that leave it.
Dev tool."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import _lib
from soccernerfs_amd.tcnn_compat import Encoding

dev = "cuda:0"
enc = Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": int(os.environ.get("LOG2", "18")),
                   "base_resolution": 16, "per_level_scale": 1.4472692012786865}).to(dev)
B = 4096 * 48
torch.manual_seed(0)
xs = {"inside": torch.rand(B, 3, device=dev), "deformed": torch.rand(B, 3, device=dev) + 0.5 * torch.randn(B, 3, device=dev),
      "ray-ordered": None}
# ray-ordered: 48 consecutive samples along each of 4096 rays
o = torch.rand(4096, 1, 3, device=dev) * 0.5 + 0.25
d = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev), dim=-1)
t = torch.linspace(-0.25, 0.25, 48, device=dev)[None, :, None]
xs["ray-ordered"] = (o + d * t).reshape(-1, 3).contiguous()
g = torch.randn(B, 32, device=dev)
out = torch.empty(B, 32, device=dev)
gt = torch.zeros_like(enc.params)
gx = torch.zeros(B, 3, device=dev)
l = _lib.lib()
P = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for name, x in xs.items():
    x = x.contiguous()
    r = {"fwd": timed(lambda: l.snerf_hashgrid_encode_fwd(C.byref(enc.desc), P(enc.params), P(x), C.c_int64(B), P(out), st())),
         "bwd table": timed(lambda: l.snerf_hashgrid_encode_bwd(C.byref(enc.desc), P(enc.params), P(x), C.c_int64(B), P(g), P(gt), None, st())),
         "bwd x": timed(lambda: l.snerf_hashgrid_encode_bwd(C.byref(enc.desc), P(enc.params), P(x), C.c_int64(B), P(g), None, P(gx), st())),
         "bwd both": timed(lambda: l.snerf_hashgrid_encode_bwd(C.byref(enc.desc), P(enc.params), P(x), C.c_int64(B), P(g), P(gt), P(gx), st()))}
    print(name, {k: round(v, 3) for k, v in r.items()})

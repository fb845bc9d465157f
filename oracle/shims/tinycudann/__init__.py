"""Stub for tiny-cuda-nn: fp32 pure-PyTorch semantics that the build targets.

Network  = bias-free Linear stack, ReLU/None hidden activation, None/Sigmoid output.
Encoding = real spherical harmonics degree 4 on (2x-1) (tcnn convention); HashGrid = oracle/hashgrid_oracle.py (restatement of the
           published algorithm, one flat `params` vector as tcnn exposes it); Frequency is construct-only (never called on the path).
"""
import os
import sys
import torch
from torch import nn

_ACT = {"ReLU": nn.ReLU, "None": nn.Identity, "Sigmoid": nn.Sigmoid}

class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        h = network_config["n_neurons"]
        nh = network_config["n_hidden_layers"]
        dims = [n_input_dims] + [h] * nh + [n_output_dims]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(len(dims) - 1)])
        self.act = _ACT[network_config["activation"]]()
        self.out_act = _ACT[network_config["output_activation"]]()
    def forward(self, x):
        x = x.float()
        for i, l in enumerate(self.layers):
            x = l(x)
            x = self.act(x) if i < len(self.layers) - 1 else self.out_act(x)
        return x

def sh4(d):
    """Real SH basis, degree 4 (16 values), d = unit vectors; tcnn ordering/constants."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = [
        0.28209479177387814 * torch.ones_like(x),
        -0.48860251190291987 * y,
        0.48860251190291987 * z,
        -0.48860251190291987 * x,
        1.0925484305920792 * xy,
        -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ]
    return torch.stack(o, dim=-1)

class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, seed=1337):
        super().__init__()
        self.cfg = dict(encoding_config)
        ot = self.cfg["otype"]
        if ot == "SphericalHarmonics":
            assert self.cfg["degree"] == 4
            self.n_output_dims = 16
        elif ot == "Frequency":
            self.n_output_dims = n_input_dims * 2 * self.cfg["n_frequencies"]
        elif ot == "HashGrid":
            self.n_output_dims = self.cfg["n_levels"] * self.cfg["n_features_per_level"]
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
            from oracle import hashgrid_oracle as HG
            self._hg = HG
            c = self.cfg
            self._geo = (c["n_levels"], c["n_features_per_level"], c["base_resolution"], c["per_level_scale"], c["log2_hashmap_size"])
            rows = HG.level_geometry(c["n_levels"], c["base_resolution"], c["per_level_scale"], c["log2_hashmap_size"], n_input_dims)[2][-1]
            gen = torch.Generator().manual_seed(seed)
            self.params = nn.Parameter((torch.rand(rows * c["n_features_per_level"], generator=gen) * 2 - 1) * 1e-4)
        else:
            raise NotImplementedError(ot)
        self.n_input_dims = n_input_dims
    def forward(self, x):
        ot = self.cfg["otype"]
        if ot == "SphericalHarmonics":
            return sh4(x.float() * 2.0 - 1.0)
        if ot == "HashGrid":
            return self._hg.encode(x.float(), self.params.view(-1, self._geo[1]), *self._geo)
        raise NotImplementedError(ot)

class NetworkWithInputEncoding(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError

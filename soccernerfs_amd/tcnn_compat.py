"""`tcnn.Network`-shaped module backed by libsnerf's fp32-MFMA MLP kernels.

Mirrors tinycudann.Network(n_input_dims, n_output_dims, network_config) as the reference constructs it
(NS/fields/kplanes_field.py:249-273,397-407): bias-free, ReLU/None hidden activation, None/Sigmoid output,
one flat `params` vector (tcnn also exposes a single flat parameter).  Layer l is stored row-major
[d_l][d_{l+1}]; `load_linear_weights` imports torch.nn.Linear-style [out,in] matrices (oracle / checkpoints).
"""
import math
from typing import Dict, Sequence

import torch
from torch import nn

from . import _lib, ops

_ACT = {"ReLU": 1, "None": 0}
_OUT = {"None": 0, "Sigmoid": 1}


class Network(nn.Module):
    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: Dict, seed: int = 1337, device=None):
        super().__init__()
        otype = network_config.get("otype", "FullyFusedMLP")
        if otype not in ("FullyFusedMLP", "CutlassMLP"):
            raise ValueError(f"unsupported network otype {otype}")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        d = _lib.MlpDesc()
        d.d_in, d.d_out = n_input_dims, n_output_dims
        d.hidden, d.n_hidden = network_config["n_neurons"], network_config["n_hidden_layers"]
        d.hidden_act = _ACT[network_config["activation"]]
        d.out_act = _OUT[network_config["output_activation"]]
        self.desc = d
        self.dims = [n_input_dims] + [d.hidden] * d.n_hidden + [n_output_dims]
        gen = torch.Generator().manual_seed(seed)
        chunks = []
        for i in range(len(self.dims) - 1):
            bound = math.sqrt(6.0 / (self.dims[i] + self.dims[i + 1]))  # xavier-uniform, as tcnn initialises FullyFusedMLP
            chunks.append(((torch.rand(self.dims[i], self.dims[i + 1], generator=gen) * 2 - 1) * bound).reshape(-1))
        self.params = nn.Parameter(torch.cat(chunks).to(device) if device is not None else torch.cat(chunks))

    def layer_slices(self):
        off = 0
        for i in range(len(self.dims) - 1):
            n = self.dims[i] * self.dims[i + 1]
            yield i, off, off + n
            off += n

    @torch.no_grad()
    def load_linear_weights(self, weights: Sequence[torch.Tensor]):
        """weights[l]: [out,in] (torch.nn.Linear layout)."""
        for (i, a, b), w in zip(self.layer_slices(), weights):
            assert tuple(w.shape) == (self.dims[i + 1], self.dims[i])
            self.params[a:b].copy_(w.t().reshape(-1))

    def linear_weights(self, buf: torch.Tensor = None):
        buf = self.params if buf is None else buf
        return [buf[a:b].detach().view(self.dims[i], self.dims[i + 1]).t().contiguous() for i, a, b in self.layer_slices()]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.mlp_forward(x, self.params, self.desc)

    def forward_with_exp_head(self, x: torch.Tensor, col: int):
        """Returns (y, exp(raw y[:, col])) -- the fused trunc_exp density head (kplanes_field.py:308-311)."""
        return ops.mlp_forward(x, self.params, self.desc, aux_col=col)

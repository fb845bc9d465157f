"""`tcnn.Network` / `tcnn.Encoding`-shaped modules backed by libsnerf.

Mirrors tinycudann.Network(n_input_dims, n_output_dims, network_config) as the reference constructs it
(NS/fields/kplanes_field.py:249-273,397-407): bias-free, ReLU/None hidden activation, None/Sigmoid output,
one flat `params` vector (tcnn also exposes a single flat parameter).  Layer l is stored row-major
[d_l][d_{l+1}]; `load_linear_weights` imports torch.nn.Linear-style [out,in] matrices (oracle / checkpoints).

Shapes the fused MFMA kernels are instantiated for (snerf_mlp_supported: one or two hidden layers, <= 16 outputs -- every net of the
K-Planes and NeRFPlayer-nerfacto presets) run fused.  The full NeRFPlayer field (NS/fields/nerfplayer_field.py:228-316) also has
three-hidden-layer nets (deformation 3->128x3->3, colour head 15->64x3->3) and a 32-output one (33->64->32): those are chained
layer by layer from libsnerf's single dense-layer kernels (snerf_dense_fwd/bwd: same fp32 MFMA arithmetic, weight matrix resident in
LDS, activations through HBM between layers; widths <= 128), same flat parameter layout.

Encoding mirrors tcnn.Encoding(n_input_dims, encoding_config) for the otypes the reference constructs: "HashGrid" (libsnerf
hashgrid kernels, one flat `params` vector as tcnn exposes it), "SphericalHarmonics" degree 4, and "Frequency" (constructed at
nerfplayer_field.py:223-226 but never called on the path: construct-only here).
"""
import ctypes as C
import math
from typing import Dict, Optional, Sequence

import torch
from torch import nn

from . import _lib, ops

_ACT = {"ReLU": 1, "None": 0}
_OUT = {"None": 0, "Sigmoid": 1}


class Network(nn.Module):
    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: Dict, seed: int = 1337, device=None, operands: str = "fp32",
                 chained_16bit: bool = False):
        """operands: "fp32" (exact, the parity path), "bf16" or "fp16" (16-bit MFMA operands, fp32 accumulation; fp16 is tcnn's own).
        chained_16bit: a shape outside the fused 16-bit kernels' table is an error by default (the fused trainers rely on that); True accepts it as a
        layer-chained net on csrc/dense_lp.hip's bf16 single layers (widths <= 128)."""
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
        if operands not in ("fp32", "bf16", "fp16"):
            raise ValueError(f"operands must be 'fp32', 'bf16' or 'fp16', got {operands!r}")
        d.operands = {"fp32": 0, "bf16": 1, "fp16": 2}[operands]
        self.dims = [n_input_dims] + [d.hidden] * d.n_hidden + [n_output_dims]
        fused16 = d.operands == 0 or bool(_lib.lib().snerf_mlp_supported(C.byref(d)))
        # shapes outside the fused 16-bit table run layer by layer: csrc/dense_lp.hip has bf16 single layers up to 128 wide (round 5)
        self.dense_operands = 0
        if not fused16:
            if not chained_16bit or not all(_lib.lib().snerf_dense_lp_supported(k, m, d.operands) for k, m in zip(self.dims[:-1], self.dims[1:])):
                raise ValueError(f"16-bit operands: fused kernels for d_in <= 160 -> 128 | d_in <= 32 -> 64 (one hidden layer) and d_in <= 64 -> 64 -> 64, bf16 single "
                                 f"layers up to 128 wide for everything else; got {operands} for {n_input_dims} -> {d.hidden} x {d.n_hidden} -> {n_output_dims}")
            self.dense_operands = d.operands
            d.operands = 0  # the descriptor describes the fused kernels; this net never reaches them
        self.desc = d
        self.operands = operands
        self.hidden_act, self.out_act = network_config["activation"], network_config["output_activation"]
        self.fused = bool(_lib.lib().snerf_mlp_supported(C.byref(d))) and self.dense_operands == 0
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
        if self.fused:
            return ops.mlp_forward(x, self.params, self.desc)
        return ops.dense_net_forward(x, self.params, self.dims, self.hidden_act, self.out_act, self.dense_operands)

    def forward_with_exp_head(self, x: torch.Tensor, col: int):
        """Returns (y, exp(raw y[:, col])) -- the fused trunc_exp density head (kplanes_field.py:308-311)."""
        if not self.fused:
            raise RuntimeError("forward_with_exp_head needs a shape the fused kernels are instantiated for")
        return ops.mlp_forward(x, self.params, self.desc, aux_col=col)


class Encoding(nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: Dict, seed: int = 1337, device=None):
        super().__init__()
        self.n_input_dims, self.cfg = n_input_dims, dict(encoding_config)
        self.otype = self.cfg["otype"]
        if self.otype == "HashGrid":
            c = self.cfg
            self.desc, rows = ops.hashgrid_desc(n_input_dims, c["n_levels"], c["n_features_per_level"], c["base_resolution"], c["per_level_scale"],
                                                c["log2_hashmap_size"])
            self.n_output_dims = c["n_levels"] * c["n_features_per_level"]
            gen = torch.Generator().manual_seed(seed)
            init = (torch.rand(rows * c["n_features_per_level"], generator=gen) * 2 - 1) * 1e-4  # tcnn: U(-1e-4, 1e-4)
            self.params = nn.Parameter(init.to(device) if device is not None else init)
        elif self.otype == "SphericalHarmonics":
            if self.cfg["degree"] != 4:
                raise ValueError("only degree-4 spherical harmonics are built")
            self.n_output_dims = 16
        elif self.otype == "Frequency":
            self.n_output_dims = n_input_dims * 2 * self.cfg["n_frequencies"]
        else:
            raise ValueError(f"unsupported encoding otype {self.otype}")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.otype == "HashGrid":
            return ops.hashgrid_encode(x.reshape(-1, self.n_input_dims), self.params, self.desc)
        if self.otype == "SphericalHarmonics":
            from .sh import sh4_from_unit_dirs
            return sh4_from_unit_dirs(x * 2.0 - 1.0)
        raise NotImplementedError("Frequency encoding is constructed by the reference but never evaluated on the path")


class TiledHashTableBackward:
    """Owner-computes backward of a HashGrid Encoding's table for a fixed batch size (csrc/hashgrid_tiles.hip, ABI 14): `bin` files the batch's (point, level,
    corner pair) touches under tiles of consecutive table rows; `scatter` adds the tiles into a dense gradient, `scatter_adam` runs torch.optim.Adam over the
    table straight from the tiles' LDS images (no dense gradient).  Levels below plan.first_tiled_level (few rows: every point of the batch lands in a handful of
    tiles) go through the atomic kernel (`coarse_levels`) into the dense gradient, which `scatter_adam` then reads and clears for those rows.  x [B,3] and
    gout [B, L*F] must stay valid until the tile pass has run."""

    def __init__(self, enc: Encoding, B: int, tile_rows_log2: int = 0, first_tiled_level: int = -1):
        import ctypes as C

        self.enc, self.B = enc, int(B)
        self.plan = _lib.HashgridTilePlan()
        _lib.check(_lib.lib().snerf_hashgrid_tile_plan_make(C.byref(enc.desc), C.c_int64(B), tile_rows_log2, first_tiled_level, C.byref(self.plan)),
                   "hashgrid_tile_plan_make")
        dev = enc.params.device
        self.counts = torch.empty(max(int(self.plan.count_ints), 1), dtype=torch.int32, device=dev)
        self.tile_base = torch.empty(self.plan.n_tiles + 1, dtype=torch.int32, device=dev)
        self.records = torch.empty(max(int(self.plan.record_capacity), 1), dtype=torch.int32, device=dev)

    def bin(self, x: torch.Tensor, gout: Optional[torch.Tensor], stream=None):
        """gout = None: file every point (the pass then needs the positions only and can run beside the forward)."""
        import ctypes as C

        st = stream if stream is not None else ops._stream()
        _lib.check(_lib.lib().snerf_hashgrid_bwd_bin(C.byref(self.enc.desc), C.byref(self.plan), ops._ptr(x), C.c_int64(self.B),
                                                     ops._ptr(gout) if gout is not None else None, ops._ptr(self.counts),
                                                     ops._ptr(self.tile_base), ops._ptr(self.records), st), "hashgrid_bwd_bin")

    def coarse_levels(self, x: torch.Tensor, gout: torch.Tensor, gtable: torch.Tensor, stream=None):
        import ctypes as C

        lc = self.plan.first_tiled_level
        if lc > 0:
            st = stream if stream is not None else ops._stream()
            _lib.check(_lib.lib().snerf_hashgrid_encode_bwd_levels(C.byref(self.enc.desc), None, ops._ptr(x), C.c_int64(self.B), ops._ptr(gout), ops._ptr(gtable), None, 0, lc,
                                                                   st), "hashgrid_encode_bwd_levels")

    def scatter(self, x: torch.Tensor, gout: torch.Tensor, gtable: torch.Tensor, stream=None):
        """gtable += the tiled levels' share (after `bin`; `coarse_levels` adds the rest)."""
        import ctypes as C

        st = stream if stream is not None else ops._stream()
        _lib.check(_lib.lib().snerf_hashgrid_bwd_tiles(C.byref(self.enc.desc), C.byref(self.plan), ops._ptr(x), C.c_int64(self.B), ops._ptr(gout), ops._ptr(self.tile_base),
                                                       ops._ptr(self.records), ops._ptr(gtable), st), "hashgrid_bwd_tiles")

    def scatter_adam(self, x: torch.Tensor, gout: torch.Tensor, gtable: Optional[torch.Tensor], p: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, step: int,
                     eps: float, betas=(0.9, 0.999), stream=None):
        import ctypes as C

        st = stream if stream is not None else ops._stream()
        _lib.check(_lib.lib().snerf_hashgrid_bwd_tiles_adam(C.byref(self.enc.desc), C.byref(self.plan), ops._ptr(x), C.c_int64(self.B), ops._ptr(gout),
                                                            ops._ptr(self.tile_base), ops._ptr(self.records), ops._ptr(gtable) if gtable is not None else None, ops._ptr(p),
                                                            ops._ptr(m), ops._ptr(v), lr, betas[0], betas[1], eps, step, st), "hashgrid_bwd_tiles_adam")

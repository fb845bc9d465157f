"""Flat, channel-last storage of a K-Planes plane set (all scales, all 6 planes in ONE fp32 buffer).

Reference layout: nn.ParameterList of [1, C, reso[b], reso[a]] tensors per scale
(init_kplanes_field, NS/fields/kplanes_field.py:47-74).  Here: one contiguous buffer, plane p of scale s
at float offset off[s][p], stored [H][W][C] so a texel's C features are one contiguous (<=128 B) line.
The flat buffer is also what Adam and the RCCL gradient all-reduce sweep.
"""
import itertools
from typing import List, Sequence

import torch
from torch import nn

from . import _lib


def coo_combs(n_coords: int):
    return list(itertools.combinations(range(n_coords), 2))


class PlaneSet(nn.Module):
    def __init__(self, C: int, resolutions: Sequence[Sequence[int]], concat: bool, a: float = 0.1, b: float = 0.5,
                 device=None, generator: torch.Generator = None):
        """resolutions: one [x,y,z(,t)] list per scale (time is not multiplied by the caller's scale)."""
        super().__init__()
        assert 1 <= len(resolutions) <= _lib.MAX_SCALES
        self.C = C
        self.concat = bool(concat)
        self.resolutions = [list(r) for r in resolutions]
        self.n_coords = len(self.resolutions[0])
        self.combs = coo_combs(self.n_coords)
        self.offsets: List[List[int]] = []
        self.shapes: List[List[tuple]] = []  # (H, W) per plane
        off = 0
        for reso in self.resolutions:
            o, sh = [], []
            for (ca, cb) in self.combs:
                H, W = reso[cb], reso[ca]
                o.append(off)
                sh.append((H, W))
                off += H * W * C
            self.offsets.append(o)
            self.shapes.append(sh)
        self.numel = off
        flat = torch.empty(off, dtype=torch.float32, device=device)
        # init: uniform(a,b) for space planes, ones for planes that contain the time axis (kplanes_field.py:68-71)
        for s in range(len(self.resolutions)):
            for p, (ca, cb) in enumerate(self.combs):
                H, W = self.shapes[s][p]
                view = flat[self.offsets[s][p]: self.offsets[s][p] + H * W * C]
                if self.n_coords == 4 and 3 in (ca, cb):
                    view.fill_(1.0)
                else:
                    view.copy_(torch.rand(H * W * C, generator=generator, device="cpu").to(flat.device) * (b - a) + a)
        self.planes = nn.Parameter(flat)

    # ---- descriptor for the C ABI ----
    def desc(self) -> _lib.KPlanesDesc:
        d = _lib.KPlanesDesc()
        d.n_scales = len(self.resolutions)
        d.C = self.C
        d.concat = int(self.concat)
        d.n_coords = self.n_coords
        for s, reso in enumerate(self.resolutions):
            for k, r in enumerate(reso):
                d.res[s][k] = r
            for p, o in enumerate(self.offsets[s]):
                d.off[s][p] = o
        return d

    def space_desc(self) -> _lib.KPlanesDesc:
        """The same buffer seen as a static scene: the planes XY, XZ, YZ of every scale (freeze_time_planes, kplanes_field.py:95-99)."""
        d = self.desc()
        if self.n_coords == 4:
            d.n_coords = 3
            for s in range(len(self.resolutions)):
                for k, p in enumerate((0, 1, 3)):
                    d.off[s][k] = self.offsets[s][p]
        return d

    @property
    def out_dim(self) -> int:
        return self.C * len(self.resolutions) if self.concat else self.C

    def plane_view(self, s: int, p: int, buf: torch.Tensor = None) -> torch.Tensor:
        """[H,W,C] view of plane p at scale s (of `buf`, default the parameter buffer)."""
        buf = self.planes if buf is None else buf
        H, W = self.shapes[s][p]
        o = self.offsets[s][p]
        return buf[o: o + H * W * self.C].view(H, W, self.C)

    # ---- reference-layout import/export (checkpoint I/O, parity tests) ----
    def to_reference(self, buf: torch.Tensor = None) -> List[List[torch.Tensor]]:
        """List over scales of 6 tensors [1,C,H,W] (detached copies)."""
        return [[self.plane_view(s, p, buf).detach().permute(2, 0, 1)[None].contiguous() for p in range(len(self.combs))]
                for s in range(len(self.resolutions))]

    @torch.no_grad()
    def load_reference(self, grids: Sequence[Sequence[torch.Tensor]]):
        for s, g in enumerate(grids):
            for p, t in enumerate(g):
                self.plane_view(s, p).copy_(t[0].permute(1, 2, 0))

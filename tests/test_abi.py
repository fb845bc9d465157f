"""CPU: the C-ABI library builds/loads and exports every symbol include/snerf.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "snerf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(snerf_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_header_symbols():
    from soccernerfs_amd import _lib, build

    build.build(verbose=False)
    l = _lib.lib()
    syms = header_symbols()
    assert syms, "no symbols parsed from include/snerf.h"
    for s in syms:
        assert hasattr(l, s), f"libsnerf.so does not export {s}"
    assert sorted(_lib.EXPORTS) == syms, (sorted(set(syms) - set(_lib.EXPORTS)), sorted(set(_lib.EXPORTS) - set(syms)))
    assert l.snerf_abi_version() == _lib.ABI_VERSION
    assert l.snerf_target_arch() == b"gfx950"


def test_struct_sizes_match_header():
    from soccernerfs_amd import _lib

    assert ctypes.sizeof(_lib.KPlanesDesc) == 16 + 8 * 4 * 4 + 8 * 6 * 8
    assert ctypes.sizeof(_lib.Coords) == 16 + 5 * 8 + 24


def test_ops_refuse_cpu_tensors():
    import torch
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    ps = PlaneSet(8, [[4, 4, 4, 2]], concat=False)
    with pytest.raises(RuntimeError, match="HIP device tensor"):
        ops.interpolate_kplanes(torch.zeros(4, 4), ps)


def test_plane_set_roundtrip_reference_layout():
    import torch
    from soccernerfs_amd.plane_set import PlaneSet

    ps = PlaneSet(8, [[5, 4, 3, 2], [10, 8, 6, 2]], concat=True)
    ref = ps.to_reference()
    assert ref[0][2].shape == (1, 8, 2, 5)  # XT plane: H = time, W = x
    assert torch.all(ref[0][2] == 1.0) and float(ref[0][0].min()) >= 0.1
    g = [[torch.rand_like(t) for t in sc] for sc in ref]
    ps.load_reference(g)
    back = ps.to_reference()
    for a, b in zip(sum(g, []), sum(back, [])):
        assert torch.equal(a, b)
    assert ps.numel == sum(t.numel() for t in sum(g, []))

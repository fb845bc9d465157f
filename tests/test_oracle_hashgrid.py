"""CPU: the static hash-grid oracle (oracle/hashgrid_oracle.py) -- level geometry against the library's host-side layout function,
structural properties of the published algorithm, and the full NeRFPlayer field restatement against the reference's own model (G13).
The tcnn HashGrid itself is third-party and absent from the reference tree: PARITY UNPINNED for its arithmetic (see the oracle header);
what G13 pins is the reference's model wiring around it."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hashgrid_oracle as HG
from oracle import kplanes_oracle as KO
from oracle import tgrid_oracle as TO
from tests.conftest import load_golden

NERFPLAYER = (16, 2, 16, 1.4472692012786865, 19)


def test_level_geometry_nerfplayer_grid():
    """16 levels from 16 to 4096 vertices per axis (nerfplayer_field.py:242-252: per_level_scale 1.447 = (4096/16)^(1/15))."""
    scales, res, off = HG.level_geometry(NERFPLAYER[0], NERFPLAYER[2], NERFPLAYER[3], NERFPLAYER[4])
    assert res[0] == 16 and res[-1] == 4096 and float(scales[0]) == 15.0
    rows = np.diff(off)
    assert rows[0] == 4096 and rows[1] == 13824 and all(r % 8 == 0 for r in rows)  # dense levels: res^3 rounded up to 8
    assert all(r == 1 << 19 for r in rows[5:]) and off[-1] == 6299960


@pytest.mark.parametrize("cfg", [(3, 16, 2, 16, 1.4472692012786865, 19), (3, 8, 4, 16, 2.0, 15), (2, 12, 2, 4, 1.5, 12), (3, 5, 2, 16, 1.3, 17),
                                 (3, 16, 2, 16, 1.38, 18), (1, 6, 8, 8, 1.7, 14)])
def test_library_layout_matches_oracle(cfg):
    """snerf_hashgrid_layout is host-only arithmetic: scale / resolution / offsets bit-identical to the oracle's."""
    from soccernerfs_amd import _lib

    D, L, F, base, pls, log2 = cfg
    d = _lib.HashgridDesc()
    d.D, d.F, d.L = D, F, L
    rows = _lib.lib().snerf_hashgrid_layout(C.byref(d), base, pls, log2)
    scales, res, off = HG.level_geometry(L, base, pls, log2, D)
    assert rows == off[-1] and list(d.offsets)[:L + 1] == off and list(d.resolution)[:L] == res
    assert np.array_equal(np.asarray(list(d.scale)[:L], np.float32), scales)
    d.L = 40
    assert _lib.lib().snerf_hashgrid_layout(C.byref(d), base, pls, log2) < 0


def test_encode_structure():
    """Vertices reproduce table rows; dense levels are a plain trilinear lattice; values are continuous across cell borders."""
    gen = torch.Generator().manual_seed(0)
    L, F, base, pls, log2 = 2, 2, 4, 2.0, 12
    scales, res, off = HG.level_geometry(L, base, pls, log2)
    table = torch.rand(off[-1], F, generator=gen)
    # level 0: scale 3, res 4, dense; pos = 3x + 0.5 -> x = (i - 0.5) / 3 sits exactly on vertex i
    i = torch.tensor([[1, 2, 3], [2, 1, 1]])
    x = (i.float() - 0.5) / 3.0
    out = HG.encode(x, table, L, F, base, pls, log2)
    want = table[i[:, 0] + 4 * i[:, 1] + 16 * i[:, 2]]
    torch.testing.assert_close(out[:, :F], want, rtol=1e-5, atol=1e-6)
    a = HG.encode(torch.tensor([[0.5 - 1e-6, 0.3, 0.7]]), table, L, F, base, pls, log2)
    b = HG.encode(torch.tensor([[0.5 + 1e-6, 0.3, 0.7]]), table, L, F, base, pls, log2)
    torch.testing.assert_close(a, b, rtol=0, atol=1e-4)
    # coordinate gradient = finite difference inside a cell
    x0 = torch.tensor([[0.31, 0.42, 0.77]], requires_grad=True)
    y = HG.encode(x0, table, L, F, base, pls, log2)
    y[0, 3].backward()
    eps = 1e-3
    for d in range(3):
        dx = torch.zeros(1, 3)
        dx[0, d] = eps
        fd = (HG.encode(x0.detach() + dx, table, L, F, base, pls, log2)[0, 3] - HG.encode(x0.detach() - dx, table, L, F, base, pls, log2)[0, 3]) / (2 * eps)
        assert abs(float(fd) - float(x0.grad[0, d])) < 2e-2 * max(1.0, abs(float(fd)))


def test_oracle_field_reproduces_reference_full_model_golden():
    """G13 (the reference's NerfplayerModel run on the CPU): from the stored final sample bins, the oracle's restatement of NerfplayerField
    gives the reference's weights, composited rgb and rendered decomposition probabilities."""
    g = load_golden("g13_nerfplayer_full")
    o, d, times = g["origins"], g["directions"], g["times"]
    aabb = torch.tensor([[-1.0] * 3, [1.0] * 3])
    eb = g["ebins_2"]
    pos = o[:, None, :] + d[:, None, :] * ((eb[:, :-1] + eb[:, 1:]) / 2)[..., None]
    scale = TO.resolve_scale(4, 16, 2.0, 2048)  # desired_resolution = 1024 * (aabb.max - aabb.min) (nerfplayer_field.py:277)
    offsets = TO.level_offsets(4, 16, scale, 12)
    assert offsets[-1] == g["param_field.newness_field.embeddings"].shape[0]
    enc = {"offsets": offsets, "log2_scale": float(np.log2(scale)), "base_res": 16, "gridtype": 0, "level_dim": 2, "table": TO.channel_table(8, 2)}
    params = {str(n): g["param_" + str(n)] for n in g["param_names"]}
    dens, rgb, probs = HG.nerfplayer_field_forward(pos, times, aabb, (4, 2, 16, 1.4472692012786865, 12), enc, params)
    w = KO.get_weights(eb[:, 1:] - eb[:, :-1], dens)
    torch.testing.assert_close(w, g["weights_2"], rtol=1e-4, atol=1e-6)
    comp = (w[..., None] * rgb).sum(1) + g["bg"] * (1 - w.sum(1, keepdim=True))
    torch.testing.assert_close(comp, g["rgb"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close((w[..., None] * probs).sum(1), g["probs"], rtol=1e-4, atol=1e-6)
    pm = g["probs"].mean(0)
    torch.testing.assert_close((0.01 * pm[1] + pm[2]) * 0.1, torch.as_tensor(g["loss_prob_loss"]), rtol=1e-5, atol=1e-8)  # nerfplayer.py:336-341

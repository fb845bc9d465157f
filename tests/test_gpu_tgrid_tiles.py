"""GPU: the owner-computes (tiled) backward of the temporal hash grid (csrc/tgrid_tiles.hip, ABI 14) against the per-sample atomic kernel that restates
the reference's kernel_grid_backward (NS/field_components/cuda/csrc/temporal_gridencoder.cu:283-370; itself pinned against the oracle and the reference's
known-answer test in tests/test_gpu_tgrid.py), and its fused Adam form against the unfused sequence scatter -> temporal TV -> snerf_adam_step_tv (which
tests/test_gpu_nerfplayer_trainer.py pins against torch.optim.Adam)."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(gen, R, S, mode):
    from soccernerfs_amd import ops

    o = ((torch.rand(R, 3, generator=gen) * 2 - 1) * 0.9).to(DEV)
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1).to(DEV)
    edges = torch.sort(torch.rand(R, S + 1, generator=gen) ** 3 * 1.6, dim=-1).values.to(DEV).contiguous()  # clustered; some samples leave the box
    times = torch.rand(R, generator=gen).to(DEV)
    times[0], times[1] = 0.0, 1.0
    if mode == "rays":
        return ops.coords_from_rays(o, d, times, edges, [[-1.0] * 3, [1.0] * 3], False), times, (o, d, edges)
    mid = (edges[:, :-1] + edges[:, 1:]) / 2
    pts = (((o[:, None, :] + d[:, None, :] * mid[..., None]) + 1.0) / 2.0).reshape(R * S, 3).contiguous()
    return ops.coords_from_points(pts), times, (pts,)


def _per_sample_bwd(enc, co, times, spr, B, gout):
    from soccernerfs_amd import _lib, ops

    ref = torch.zeros_like(enc.embeddings)
    os.environ["SNERF_TGRID_RUNS"] = "0"  # tgrid_kernel<true>: one atomic per sample, level, corner and live column, as the reference's kernel
    try:
        _lib.check(_lib.lib().snerf_tgrid_encode_bwd(C.byref(enc.desc), C.byref(co), None, ops._ptr(times), spr, C.c_int64(B), ops._ptr(gout), ops._ptr(ref),
                                                     ops._stream()), "tgrid_encode_bwd")
        torch.cuda.synchronize()
    finally:
        os.environ.pop("SNERF_TGRID_RUNS", None)
    return ref


CASES = [
    # the preset's main grid at a smaller table (dense levels, hashed levels, 66-float rows), default split: levels below 2^16 rows atomic
    (dict(input_dim=3, temporal_dim=64, num_levels=16, level_dim=2, log2_hashmap_size=17, desired_resolution=2048), 0, -1),
    # every level tiled, small tiles (tile boundaries inside dense levels whose row count is not a multiple of the tile)
    (dict(input_dim=3, temporal_dim=64, num_levels=16, level_dim=2, log2_hashmap_size=15, desired_resolution=2048), 5, 0),
    # a proposal grid of the preset (34-float rows), C = 4 and C = 8 tables, a tiled (non-hashed) grid type
    (dict(input_dim=3, temporal_dim=32, num_levels=5, level_dim=2, log2_hashmap_size=17, base_resolution=16, per_level_scale=1.4142135), 0, -1),
    (dict(input_dim=3, temporal_dim=12, num_levels=4, level_dim=4, log2_hashmap_size=12, base_resolution=5, per_level_scale=1.7), 4, 0),
    (dict(input_dim=3, temporal_dim=24, num_levels=3, level_dim=8, log2_hashmap_size=11, base_resolution=4, per_level_scale=2.0), 3, 1),
    (dict(input_dim=3, temporal_dim=7, num_levels=3, level_dim=1, log2_hashmap_size=10, base_resolution=6, per_level_scale=1.5, gridtype="tiled"), 6, 0),
]


@pytest.mark.parametrize("mode", ["rays", "points"])
@pytest.mark.parametrize("kw,sh,lc", CASES)
def test_tiled_scatter_equals_the_per_sample_atomic_kernel(kw, sh, lc, mode):
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder, TiledTableBackward

    gen = torch.Generator().manual_seed(5)
    enc = TemporalGridEncoder(**kw).to(DEV)
    for R, S in ((300, 48), (129, 37)):
        B = R * S
        co, times, keep = _batch(gen, R, S, mode)
        gout = torch.randn(B, enc.output_dim, generator=gen).to(DEV)
        gout[5:9] = 0.0                        # samples without any gradient produce no record
        gout[11, : enc.level_dim] = 0.0        # ... nor does a (sample, level) pair whose channels are all zero
        ref = _per_sample_bwd(enc, co, times, S, B, gout)
        tb = TiledTableBackward(enc, B, tile_rows_log2=sh, first_tiled_level=lc)
        assert tb.plan.n_tiles == tb.plan.tile_start[enc.num_levels] and (sh == 0 or tb.plan.tile_rows_log2 == sh)
        got = torch.zeros_like(ref)
        tb.bin(co, times, S, gout)
        tb.coarse_levels(co, times, S, gout, got)
        tb.scatter(gout, got)
        torch.cuda.synchronize()
        n_rec = int(tb.tile_base[tb.plan.n_tiles])
        assert n_rec <= tb.plan.record_capacity and (n_rec > 0) == (tb.plan.first_tiled_level < enc.num_levels)
        scale = float(ref.abs().max())
        assert scale > 0
        # the same products (w * (g * wt), factors in the same order) summed in another order: a few ulps of the largest partial sum
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6 * scale)
        assert float((got - ref).norm() / ref.norm()) < 1e-6
        assert bool(((got != 0) == (ref != 0)).all())  # exactly the same entries are touched
        # it ACCUMULATES: a second pass over the same records doubles the tiled levels' share, whatever the buffer held
        again = got.clone()
        tb.coarse_levels(co, times, S, gout, again)
        tb.scatter(gout, again)
        torch.testing.assert_close(again, 2 * ref, rtol=1e-5, atol=4e-6 * scale)


@pytest.mark.parametrize("ws", ["1", "0"])
@pytest.mark.parametrize("tv", [False, True])
@pytest.mark.parametrize("kw,sh,lc", CASES[:4])
def test_fused_adam_equals_scatter_then_tv_then_adam(kw, sh, lc, tv, ws, monkeypatch):
    """ws = 0 (default): one workgroup per tile; ws = 1: the persistent, wave-specialised tile kernel (builder waves sum tile k + 1 while streamer waves run Adam over tile k; tiles handed out by a
    ticket); ws = 0: one workgroup per tile (SNERF_TGRID_TILES_WS=0).  Three optimiser steps, so the ticket words are reused across launches."""
    from soccernerfs_amd import _lib, ops

    monkeypatch.setenv("SNERF_TGRID_TILES_WS", ws)
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder, TiledTableBackward

    gen = torch.Generator().manual_seed(9)
    enc = TemporalGridEncoder(**kw).to(DEV)
    with torch.no_grad():
        enc.embeddings.copy_(((torch.rand(enc.embeddings.shape, generator=gen) - 0.5) * 0.2).to(DEV))
    rows, gc = enc.embeddings.shape
    R, S = 257, 48
    B = R * S
    L = _lib.lib()
    p_ref, p_new = enc.embeddings.detach().clone(), enc.embeddings.detach().clone()
    z = lambda: torch.zeros(rows, gc, device=DEV)
    m_ref, v_ref, m_new, v_new, g_ref, g_new = z(), z(), z(), z(), z(), z()
    srow, part = torch.zeros(rows, device=DEV), torch.zeros(64, 16, device=DEV)
    tb = TiledTableBackward(enc, B, tile_rows_log2=sh, first_tiled_level=lc)
    lr, eps = 1e-2, 1e-12
    for step in range(1, 4):
        co, times, keep = _batch(gen, R, S, "rays")
        gout = (torch.randn(B, enc.output_dim, generator=gen) * 1e-3).to(DEV)
        ca, cb = enc._index_list_host[step * 7 % len(enc._index_list_host)]
        # ---- unfused: run-length atomic scatter -> TV sign from the OLD table -> snerf_adam_step_tv (clears the gradient) ----
        _lib.check(L.snerf_tgrid_encode_bwd(C.byref(enc.desc), C.byref(co), None, ops._ptr(times), S, C.c_int64(B), ops._ptr(gout), ops._ptr(g_ref), ops._stream()))
        if tv:
            _lib.check(L.snerf_tgrid_tv_sign(ops._ptr(p_ref), C.c_int64(rows), gc, ca, cb, 0.1, ops._ptr(part), 64, ops._ptr(srow), ops._stream()))
            _lib.check(L.snerf_adam_step_tv(ops._ptr(p_ref), ops._ptr(g_ref), ops._ptr(m_ref), ops._ptr(v_ref), C.c_int64(rows), gc, ca, cb, ops._ptr(srow), lr, 0.9, 0.999,
                                            eps, step, 1.0, 1, None, ops._stream()))
        else:
            ops.adam_step(p_ref.view(-1), g_ref.view(-1), m_ref.view(-1), v_ref.view(-1), step, lr, eps=eps, zero_grad=True)
        # ---- fused ----
        if tv:  # the TV step of the fused path comes from ITS old table
            _lib.check(L.snerf_tgrid_tv_sign(ops._ptr(p_new), C.c_int64(rows), gc, ca, cb, 0.1, ops._ptr(part), 64, ops._ptr(srow), ops._stream()))
        tb.bin(co, times, S, gout)
        tb.coarse_levels(co, times, S, gout, g_new)
        tb.scatter_adam(gout, g_new, p_new, m_new, v_new, lr, step, eps, tv_cols=(ca, cb) if tv else None, srow=srow if tv else None)
        torch.cuda.synchronize()
        assert float(g_new.abs().max()) == 0.0 and float(g_ref.abs().max()) == 0.0  # both leave a cleared gradient buffer
        # Adam's first steps move a parameter by ~lr whatever the gradient's size, so an element whose tiny gradient differs in its last bits (another
        # summation order) may move differently: compare moments tightly, parameters on all but a vanishing fraction of elements
        torch.testing.assert_close(m_new, m_ref, rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(v_new, v_ref, rtol=2e-4, atol=1e-15)
        bad = (p_new - p_ref).abs() > 1e-5
        assert float(bad.float().mean()) < 1e-4, float(bad.float().mean())
    assert float((p_new - enc.embeddings).abs().max()) > 1e-3  # the table did move

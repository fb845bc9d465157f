"""GPU parity: K-Planes gather fwd/bwd (HIP, through the C ABI) vs the CPU oracle and the golden vectors.
Tolerances: features rtol 1e-5 / atol 1e-6; plane gradients (atomic order) rtol 1e-4."""
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU test needs a HIP device"
    return torch.device("cuda:0")


def _plane_set_from_reference(grids, C, concat):
    from soccernerfs_amd.plane_set import PlaneSet

    reso = []
    for g in grids:
        # plane 0 = XY: [1,C,Y,X]; plane 1 = XZ: [1,C,Z,X]; plane 2 = XT: [1,C,T,X]
        reso.append([g[0].shape[3], g[0].shape[2], g[1].shape[2], g[2].shape[2]])
    ps = PlaneSet(C, reso, concat=concat)
    ps.load_reference(grids)
    return ps.to(_dev())


@pytest.mark.parametrize("tag", ["single", "multi", "prop"])
def test_gather_matches_golden(tag):
    from soccernerfs_amd import ops

    g = load_golden("g5_interp")
    C, n_scales, concat = [int(v) for v in g[f"{tag}_meta"]]
    grids = [[g[f"{tag}_plane_{s}_{p}"] for p in range(6)] for s in range(n_scales)]
    ps = _plane_set_from_reference(grids, C, bool(concat))
    pts = g[f"{tag}_pts"].to(_dev())
    feats = ops.interpolate_kplanes(pts, ps)
    torch.testing.assert_close(feats.cpu(), g[f"{tag}_feats"], rtol=1e-5, atol=1e-6)
    feats.backward(g[f"{tag}_gout"].to(_dev()))
    got = ps.to_reference(ps.planes.grad.cpu())
    for s in range(n_scales):
        for p in range(6):
            torch.testing.assert_close(got[s][p], g[f"{tag}_grad_{s}_{p}"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("C,ms,concat,lo,hi", [(32, (1, 2, 4), True, -1.2, 1.2), (8, (1,), False, -0.1, 1.1), (16, (1, 3), False, -1.0, 1.0)])
def test_gather_matches_oracle_random(C, ms, concat, lo, hi):
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(7)
    base = (11, 9, 7, 5)
    grids = []
    for m in ms:
        reso = [r * m for r in base[:3]] + [base[3]]
        grids.append([torch.rand(1, C, reso[b], reso[a], generator=gen) + 0.2 for (a, b) in KO.COO_COMBS])
    N = 3001  # ragged: not a multiple of the block size
    pts = torch.rand(N, 4, generator=gen) * (hi - lo) + lo
    leaves = [[t.clone().requires_grad_(True) for t in sc] for sc in grids]
    ref = KO.interpolate_kplanes(pts, leaves, concat)
    gout = torch.rand(ref.shape, generator=gen) - 0.5
    ref.backward(gout)
    ps = _plane_set_from_reference(grids, C, concat)
    out = ops.interpolate_kplanes(pts.to(_dev()), ps)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    out.backward(gout.to(_dev()))
    got = ps.to_reference(ps.planes.grad.cpu())
    for s in range(len(ms)):
        for p in range(6):
            torch.testing.assert_close(got[s][p], leaves[s][p].grad, rtol=1e-4, atol=2e-6)


def test_gather_from_rays_matches_oracle():
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(11)
    R, S, C = 37, 48, 8
    grids = [torch.rand(1, C, r2, r1, generator=gen) + 0.1 for (r1, r2) in [(12, 10), (12, 8), (12, 4), (10, 8), (10, 4), (8, 4)]]
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    o = torch.rand(R, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    times = torch.rand(R, 1, generator=gen)
    eb = torch.cumsum(torch.rand(R, S + 1, generator=gen) * 0.06, -1)
    pos = KO.sample_positions(o, d, eb[:, :-1], eb[:, 1:])
    for rescale in (True, False):
        p = KO.normalize_positions(pos, aabb)
        p = p * 2 - 1 if rescale else p
        pts = torch.cat([p, (times * 2 - 1)[:, None, :].expand(R, S, 1)], -1).reshape(-1, 4)
        ref = KO.interpolate_kplanes(pts, [grids], False)
        ps = _plane_set_from_reference([grids], C, False)
        dev = _dev()
        out = ops.interpolate_kplanes_rays(ps, o.to(dev), d.to(dev), times.to(dev), eb.to(dev), aabb, rescale)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)


def test_gather_full_size_properties():
    """BASELINE config-2 sizes (5 scales x 6 planes, C=32, 64*4096 samples): size-independent properties.
    Constant planes k => every feature equals k^6 and every plane's gradient sums to k^5 * sum(gout)."""
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    dev = _dev()
    reso = [[64 * m, 64 * m, 64 * m, 100] for m in (1, 2, 4, 8, 16)]
    ps = PlaneSet(32, reso, concat=True, device=dev)
    k = 1.25
    with torch.no_grad():
        ps.planes.fill_(k)
    N = 64 * 4096
    pts = torch.rand(N, 4, device=dev) * 2.2 - 1.1
    out = ops.interpolate_kplanes(pts, ps)
    assert out.shape == (N, 160)
    torch.testing.assert_close(out, torch.full_like(out, k**6), rtol=2e-6, atol=0)
    gout = torch.rand_like(out) - 0.25
    out.backward(gout)
    g = ps.planes.grad
    for s in range(5):
        expect = (gout[:, s * 32:(s + 1) * 32].double().sum(0) * k**5).cpu()
        for p in range(6):
            got = ps.plane_view(s, p, g).double().sum((0, 1)).cpu()
            torch.testing.assert_close(got, expect, rtol=2e-4, atol=1e-3)
    # linearity of the backward in grad_out
    ps.planes.grad = None
    out2 = ops.interpolate_kplanes(pts, ps)
    out2.backward(2.0 * gout)
    torch.testing.assert_close(ps.planes.grad, 2.0 * g, rtol=1e-3, atol=1e-3)


def test_gather_empty_and_errors():
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    dev = _dev()
    ps = PlaneSet(8, [[4, 4, 4, 2]], concat=False, device=dev)
    out = ops.interpolate_kplanes(torch.zeros(0, 4, device=dev), ps)
    assert out.shape == (0, 8)
    bad = PlaneSet(8, [[4, 4, 4, 2]], concat=False, device=dev)
    bad.C = 12
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.interpolate_kplanes(torch.zeros(4, 4, device=dev), bad)


@pytest.mark.parametrize("C,ms,concat,N", [(32, (1, 2, 4), True, 3001), (8, (1,), False, 5000), (32, (1, 2, 4, 8, 16), True, 64 * 512)])
def test_sorted_scatter_equals_direct_scatter_and_oracle(C, ms, concat, N):
    """The sorted variant (counting sort by texel key + run-length combining) gives the same plane gradients as the
    sample-major scatter (and, for the small cases, as the oracle's autograd)."""
    import ctypes as Ct
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.plane_set import PlaneSet

    dev = _dev()
    gen = torch.Generator().manual_seed(21)
    base = (11, 9, 7, 5) if N < 20000 else (64, 64, 64, 100)
    ps = PlaneSet(C, [[r * m for r in base[:3]] + [base[3]] for m in ms], concat=concat, generator=gen)
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, generator=gen) + 0.2)
    ref_grids = [[t.clone().requires_grad_(True) for t in sc] for sc in ps.to_reference()]
    ps = ps.to(dev)
    pts = torch.rand(N, 4, generator=gen) * 2.2 - 1.1
    pts[: N // 3, 3] = 0.25  # many samples share a time row, as rays of one image do
    pts[: N // 8] = pts[0]   # heavy duplicates: long equal-key runs
    gout = torch.rand(N, ps.out_dim, generator=gen) - 0.5
    ptsd, goutd = pts.to(dev), gout.to(dev)
    co = ops.coords_from_points(ptsd)
    direct = torch.zeros_like(ps.planes)
    desc = ps.desc()
    _lib.check(_lib.lib().snerf_kplanes_gather_bwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(goutd), ops._ptr(direct),
                                                   ops._stream()))
    ss = ops.SortedScatter(ps, N, dev)
    ss.sort(co)
    # the order of every plane is a permutation of 0..N-1
    sn = ss.sorted_rec[:, 0].contiguous().view(torch.int32).view(-1, N)
    assert torch.equal(torch.sort(sn, dim=1).values, torch.arange(N, device=dev, dtype=torch.int32).expand_as(sn))
    got = torch.zeros_like(ps.planes)
    ss.scatter(ps.planes, co, goutd, got)
    torch.testing.assert_close(got, direct, rtol=1e-4, atol=1e-5)
    # bf16 gradient vectors between pass A and pass B (the production path with 16-bit MLP operands): every contribution is rounded to 8
    # significant bits before the fp32 accumulation -> relative L2 error ~2^-9 / sqrt(#contributions), no bias
    ss16 = ops.SortedScatter(ps, N, dev, gvec_dtype=torch.bfloat16)
    ss16.sort(co)
    got16 = torch.zeros_like(ps.planes)
    ss16.scatter(ps.planes, co, goutd, got16)
    assert float((got16 - direct).norm() / direct.norm()) < 3e-3
    assert abs(float((got16 - direct).sum() / direct.abs().sum())) < 1e-4
    if N < 20000:
        ref = KO.interpolate_kplanes(pts, ref_grids, concat)
        ref.backward(gout)
        g = ps.to_reference(got.cpu())
        for s in range(len(ms)):
            for p in range(6):
                torch.testing.assert_close(g[s][p], ref_grids[s][p].grad, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("ms,N,zeros", [((1, 2, 4), 3000, False), ((1, 2), 1500, True), ((1, 2, 4, 8, 16), 100000, False), ((1,), 257, True)])
def test_quotient_scatter_equals_direct_scatter(ms, N, zeros):
    """Quotient form of the sorted scatter -- g_q = (gfeat .* feat) ./ v_q with v_q re-interpolated in pass B -- against the sample-major
    scatter and the product-form sorted scatter.  `zeros`: exact zeros planted in the planes, so that features vanish and the listed rows go
    through the exact fix-up (where the quotient alone would lose a plane's gradient)."""
    import ctypes as Ct
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.plane_set import PlaneSet

    dev = _dev()
    gen = torch.Generator().manual_seed(33)
    base = (11, 9, 7, 5) if N < 20000 else (64, 64, 64, 100)
    ps = PlaneSet(32, [[r * m for r in base[:3]] + [base[3]] for m in ms], concat=True, generator=gen)
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, generator=gen) * 1.2 - 0.3)  # both signs, values near zero included
        if zeros:
            # whole texels of scale 0 / plane 0 and a sprinkle everywhere: interpolated values that are exactly 0.0f
            ps.planes[: 32 * 40] = 0.0
            ps.planes[torch.rand(ps.numel, generator=gen) < 0.02] = 0.0
        if zeros:
            ps.plane_view(0, 2)[-1, -1, :] = 3e-39  # a SUBNORMAL texel: counts as zero in the quotient form (v_rcp_f32 may flush it)
    ps = ps.to(dev)
    pts = torch.rand(N, 4, generator=gen) * 2.2 - 1.1
    pts[: N // 3, 3] = 0.25
    pts[: N // 8] = pts[0]
    if zeros:
        pts[N // 2: N // 2 + 64] = -1.0  # the all-zero corner texel of plane 0 exactly (weights 1, 0, 0, 0)
        pts[N // 2 + 64: N // 2 + 96] = 1.0  # the subnormal last texel of plane 2 exactly
    gout = torch.rand(N, ps.out_dim, generator=gen) - 0.5
    ptsd, goutd = pts.to(dev), gout.to(dev)
    co = ops.coords_from_points(ptsd)
    desc = ps.desc()
    L = _lib.lib()
    direct = torch.zeros_like(ps.planes)
    _lib.check(L.snerf_kplanes_gather_bwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(goutd), ops._ptr(direct), ops._stream()))
    feat = torch.empty(N, ps.out_dim, device=dev)
    _lib.check(L.snerf_kplanes_gather_fwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(feat), ops._stream()))
    # planted zeros make thousands of features vanish at once.  (r06, ADVICE) The DEFAULT list holds the worst case (every feature of every sample),
    # so no input -- imported all-zero planes included -- can overflow it; a smaller list is the caller's explicit choice
    ss = ops.SortedScatter(ps, N, dev, quotient=True)
    assert ss.gvec is None and ss.fix_capacity == N * ps.out_dim
    assert ops.SortedScatter(ps, N, dev, quotient=True, fix_capacity=7).fix_capacity == 7
    ss.sort(co)
    got = torch.zeros_like(ps.planes)
    ss.scatter_quotient(ps.planes, co, goutd, feat, got)
    n_fix = int(ss.fix_count.item())
    assert (n_fix > 0) == zeros
    ss.check_fix_overflow()  # nothing was dropped
    if zeros:
        # the same scatter with a list that is too short must not lose gradient terms SILENTLY: the fix-up records the demanded count
        # and check_fix_overflow raises
        small = ops.SortedScatter(ps, N, dev, quotient=True, fix_capacity=max(n_fix // 2, 1))
        small.sort(co)
        small.scatter_quotient(ps.planes, co, goutd, feat, torch.zeros_like(ps.planes))
        assert int(small.fix_peak.item()) == n_fix
        with pytest.raises(RuntimeError, match="fix list holds"):
            small.check_fix_overflow()
    scale = float(direct.abs().max())
    torch.testing.assert_close(got, direct, rtol=1e-4, atol=2e-6 * scale)
    assert float((got - direct).norm() / direct.norm()) < 2e-6
    # scale ranges (the multi-GPU exchange scatters the finest scale first) add up to the whole
    if len(ms) > 1:
        parts = torch.zeros_like(ps.planes)
        ss.quotient_prepare(goutd, feat)
        ss.quotient_scatter_scales(ps.planes, co, goutd, parts, len(ms) - 1, len(ms))
        ss.quotient_scatter_scales(ps.planes, co, goutd, parts, 0, len(ms) - 1)
        torch.testing.assert_close(parts, got, rtol=1e-4, atol=2e-6 * scale)


@pytest.mark.gpu
def test_quotient_scatter_when_the_product_of_six_normal_values_underflows():
    """A feature can vanish although no plane value does: six values of 2e-7 multiply to 6.4e-41, below the smallest normal float.  The
    quotient G / v_q has lost (almost) all bits there; quotient_prepare zeroes G for the channel, lists the row, and the fix-up adds the exact
    product-form term to EVERY plane (these gradients used to be dropped or mangled)."""
    import ctypes as Ct
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.plane_set import PlaneSet

    dev = _dev()
    gen = torch.Generator().manual_seed(4)
    ps = PlaneSet(32, [[11, 9, 7, 5], [22, 18, 14, 5]], concat=True, generator=gen)
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, generator=gen) * 0.8 + 0.2)
        for q in range(6):
            ps.plane_view(0, q)[0:2, 0:2, :] = 2e-7  # the first cell of every plane of scale 0
    ps = ps.to(dev)
    N = 600
    pts = torch.rand(N, 4, generator=gen) * 2 - 1
    pts[:64] = -1.0 + 0.04 + 0.1 * torch.rand(64, 4, generator=gen)  # inside the first cell of every axis (cell widths 0.2 .. 0.5)
    gout = torch.rand(N, ps.out_dim, generator=gen) + 0.5
    ptsd, goutd = pts.to(dev), gout.to(dev)
    co = ops.coords_from_points(ptsd)
    desc = ps.desc()
    L = _lib.lib()
    direct = torch.zeros_like(ps.planes)
    _lib.check(L.snerf_kplanes_gather_bwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(goutd), ops._ptr(direct), ops._stream()))
    feat = torch.empty(N, ps.out_dim, device=dev)
    _lib.check(L.snerf_kplanes_gather_fwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(feat), ops._stream()))
    assert float(feat[:64, :32].abs().max()) < 1.2e-38  # the features of scale 0 underflowed ...
    ss = ops.SortedScatter(ps, N, dev, quotient=True, fix_capacity=N * ps.out_dim)  # 64 samples x 32 channels vanish at once
    ss.sort(co)
    got = torch.zeros_like(ps.planes)
    ss.scatter_quotient(ps.planes, co, goutd, feat, got)
    assert int(ss.fix_count.item()) >= 64
    torch.testing.assert_close(got, direct, rtol=1e-4, atol=2e-6 * float(direct.abs().max()))
    for q in range(6):  # ... but every plane's gradient there (g * (2e-7)^5 ~ 3e-34 per sample) arrives, to fp32 accuracy
        a, b = ps.plane_view(0, q, got)[0:2, 0:2, :], ps.plane_view(0, q, direct)[0:2, 0:2, :]
        assert float(b.abs().min()) > 1e-36
        torch.testing.assert_close(a, b, rtol=2e-4, atol=0)


@pytest.mark.gpu
def test_quotient_scatter_with_one_subnormal_plane_value_beside_a_normal_feature():
    """ONE plane value is subnormal (not zero) while the other five multiply to more than 1, so the feature is a normal number and
    G = gfeat * feat is usable -- but v_rcp_f32 may flush the subnormal divisor.  Pass B divides by it in IEEE arithmetic there; that plane's
    gradient (gfeat * the others' product) used to be dropped silently."""
    import ctypes as Ct
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.plane_set import PlaneSet

    dev = _dev()
    gen = torch.Generator().manual_seed(9)
    ps = PlaneSet(32, [[11, 9, 7, 5], [22, 18, 14, 5]], concat=True, generator=gen)
    with torch.no_grad():
        ps.planes.copy_(torch.rand(ps.numel, generator=gen) * 2.0 + 2.0)  # every value in [2, 4): five of them multiply to >= 32
        ps.plane_view(1, 4)[0:2, 0:2, :] = 3e-39                           # the first cell of plane YT of scale 1: subnormal
    ps = ps.to(dev)
    N = 500
    pts = torch.rand(N, 4, generator=gen) * 2 - 1
    pts[:80, 1] = -1.0 + 0.05 * torch.rand(80, generator=gen)  # y and t inside the first cell of scale 1 (cell widths 2 / 17 and 0.5)
    pts[:80, 3] = -1.0 + 0.3 * torch.rand(80, generator=gen)
    gout = torch.rand(N, ps.out_dim, generator=gen) + 0.5
    ptsd, goutd = pts.to(dev), gout.to(dev)
    co = ops.coords_from_points(ptsd)
    desc = ps.desc()
    L = _lib.lib()
    direct = torch.zeros_like(ps.planes)
    _lib.check(L.snerf_kplanes_gather_bwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(goutd), ops._ptr(direct), ops._stream()))
    feat = torch.empty(N, ps.out_dim, device=dev)
    _lib.check(L.snerf_kplanes_gather_fwd(Ct.byref(desc), ops._ptr(ps.planes), Ct.byref(co), Ct.c_int64(N), ops._ptr(feat), ops._stream()))
    f1 = feat[:80, 32:64].abs()
    assert float(f1.min()) >= 1.17549435e-38 and float(f1.max()) < 1e-35  # normal numbers, carried by a subnormal factor
    ss = ops.SortedScatter(ps, N, dev, quotient=True)
    ss.sort(co)
    got = torch.zeros_like(ps.planes)
    ss.scatter_quotient(ps.planes, co, goutd, feat, got)
    assert int(ss.fix_count.item()) == 0  # nothing vanished: no fix-up involved
    a, b = ps.plane_view(1, 4, got)[0:2, 0:2, :], ps.plane_view(1, 4, direct)[0:2, 0:2, :]
    assert float(b.abs().min()) > 1.0  # gfeat * (>= 32) summed over the cell's samples
    torch.testing.assert_close(a, b, rtol=2e-4, atol=0)
    torch.testing.assert_close(got, direct, rtol=1e-4, atol=2e-6 * float(direct.abs().max()))

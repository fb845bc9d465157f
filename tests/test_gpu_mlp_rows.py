"""The wave-owns-rows backward of the 64-wide nets (csrc/mlp_rows.hip: no barrier inside the loop, activations handed from layer to layer
in registers through a permuted contraction index) against

* the workgroup-tile kernel it replaces (snerf_mlp_bwd_tile = round 4's kernel): same operand roundings, so gX agrees to the last bits of
  the fp32 accumulations (bit for bit where the matrix core's sum does not depend on the slot order) and the weight gradients to the
  order of the fp32 sums;
* an emulation in float64 of exactly what the kernel rounds (operands to bf16 / fp16, everything else exact).

Shapes: K-Planes color_net 15 -> 64 -> 64 -> 3 (Sigmoid), proposal nets 8 -> 64 -> 1 (trunc_exp head through gaux), NeRFPlayer mlp_base
32 -> 64 -> 16 and a 20-wide one-hidden-layer net; ragged, tiny and preset-sized N; strided inputs / outputs as the trainer passes them
(NS/fields/kplanes_field.py:249-273,397-407)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _desc(d_in, d_out, n_hidden, out_act, operands):
    from soccernerfs_amd import _lib

    d = _lib.MlpDesc()
    d.d_in, d.d_out, d.hidden, d.n_hidden, d.hidden_act, d.out_act = d_in, d_out, 64, n_hidden, 1, out_act
    d.operands = {"bf16": 1, "fp16": 2}[operands]
    return d


def _bwd(fn, d, W, X, ldx, N, gY, ldgy, aux_col, gaux, ldgx):
    from soccernerfs_amd import _lib, ops

    gX = torch.full((N, ldgx), 7.0, device=DEV)  # sentinel: columns >= d_in must stay untouched
    gW = torch.zeros_like(W)
    _lib.check(fn(C.byref(d), ops._ptr(W), ops._ptr(X), ldx, C.c_int64(N), ops._ptr(gY) if gY is not None else None, ldgy, aux_col,
                  ops._ptr(gaux) if gaux is not None else None, ops._ptr(gX), ldgx, ops._ptr(gW), ops._stream()), "mlp_bwd")
    torch.cuda.synchronize()
    return gX, gW


CASES = [  # d_in, d_out, n_hidden, out_act, ldx, ldgy, ldgx, aux, N
    (15, 3, 2, 1, 16, 3, 16, False, 1111),     # color_net as the trainer calls it: X = h[:, :15] (stride 16), gX -> gh[:, :15]
    (15, 3, 2, 0, 15, 3, 15, False, 31),       # less than one 32-sample pair
    (8, 1, 1, 0, 8, 1, 8, True, 4096 * 3 + 5),  # proposal net: the density gradient arrives through gaux only
    (8, 1, 1, 0, 8, 1, 8, False, 64),
    (32, 16, 1, 0, 32, 16, 32, True, 777),     # NeRFPlayer mlp_base 32 -> 64 -> 16 with the density head in column 0
    (20, 5, 1, 1, 20, 5, 20, False, 2049),     # a 32-wide layer 0 (natural-k 16x16x32 products), Sigmoid outputs
    (15, 3, 2, 1, 15, 3, 15, False, 500),      # rows that are not float4-granular: served by the workgroup-tile kernel (same results)
    (15, 3, 2, 1, 16, 3, 16, False, 262144),   # the preset's 4096 rays x 64 samples
]


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES)
def test_rows_backward_matches_tile_kernel_and_emulation(case, operands):
    from soccernerfs_amd import _lib

    d_in, d_out, nh, out_act, ldx, ldgy, ldgx, aux, N = case
    L = _lib.lib()
    d = _desc(d_in, d_out, nh, out_act, operands)
    assert L.snerf_mlp_supported(C.byref(d))
    gen = torch.Generator().manual_seed(1000 + d_in + N % 97)
    dims = [d_in] + [64] * nh + [d_out]
    Ws = [((torch.rand(dims[i], dims[i + 1], generator=gen) * 2 - 1) * (6.0 / (dims[i] + dims[i + 1])) ** 0.5) for i in range(len(dims) - 1)]
    W = torch.cat([w.reshape(-1) for w in Ws]).to(DEV)
    X = (torch.rand(N, ldx, generator=gen) - 0.3).to(DEV)
    gY = (torch.rand(N, ldgy, generator=gen) - 0.5).to(DEV) if not (aux and d_out == 1) else None
    gaux = (torch.rand(N, generator=gen) - 0.5).to(DEV) if aux else None
    aux_col = 0 if aux else -1
    gX_r, gW_r = _bwd(L.snerf_mlp_bwd, d, W, X, ldx, N, gY, ldgy, aux_col, gaux, ldgx)
    gX_t, gW_t = _bwd(L.snerf_mlp_bwd_tile, d, W, X, ldx, N, gY, ldgy, aux_col, gaux, ldgx)
    assert float((gX_r[:, d_in:] - 7.0).abs().max() if ldgx > d_in else 0.0) == 0.0  # padding columns untouched
    assert bool(torch.isfinite(gX_r).all()) and bool(torch.isfinite(gW_r).all())
    # ---- against the kernel it replaces ----
    a, b = gX_r[:, :d_in], gX_t[:, :d_in]
    scale = float(b.abs().max())
    n_diff = int((a != b).sum())
    print(f"rows vs tile ({operands}, {d_in}->{'x'.join(['64'] * nh)}->{d_out}, N={N}): gX differs in {n_diff} of {a.numel()} elements, "
          f"max |diff| / max |gX| = {float((a - b).abs().max()) / scale:.2e}")
    # same roundings, another association order inside the fp32 accumulations; a pre-activation within that distance of zero may take the other
    # ReLU branch for a whole sample row (rare): bound the bulk tightly and the outlier fraction
    bad = (a - b).abs() > 1e-5 * scale + 1e-4 * b.abs()
    assert float(bad.float().mean()) < 2e-3, float(bad.float().mean())
    torch.testing.assert_close(gW_r, gW_t, rtol=2e-3, atol=2e-4 * float(gW_t.abs().max()))
    # ---- against a float64 emulation of what the kernel rounds ----
    dt, GS = (torch.bfloat16, 1.0) if operands == "bf16" else (torch.float16, 8192.0)
    rd = lambda t: t.float().to(dt).double()
    rg = lambda t: ((t * GS).float().clamp(-65504.0, 65504.0) if operands == "fp16" else t.float()).to(dt).double() / GS
    x = rd(X[:, :d_in].double())
    Wd = [rd(w.to(DEV).double()) for w in Ws]
    acts = [x]
    for l in range(nh):
        acts.append(rd(torch.relu(acts[-1] @ Wd[l])))
    z = acts[-1] @ Wd[nh]
    g = torch.zeros(N, d_out, dtype=torch.float64, device=DEV)
    if gY is not None:
        g = gY[:, :d_out].double() * (torch.sigmoid(z) * (1 - torch.sigmoid(z)) if out_act == 1 else 1.0)
    if aux:
        g[:, 0] += gaux.double() * torch.exp(z[:, 0].clamp(-15, 15))
    g = rg(g)
    gWs = [None] * (nh + 1)
    gWs[nh] = acts[nh].t() @ g
    for l in range(nh, 0, -1):
        g = rg((g @ Wd[l].t()) * (acts[l] > 0))
        gWs[l - 1] = acts[l - 1].t() @ g
    gx = g @ Wd[0].t()
    gw = torch.cat([w.reshape(-1) for w in gWs])
    bad = (gX_r[:, :d_in].double() - gx).abs() > 2e-3 * float(gx.abs().max()) + 1e-2 * gx.abs()
    assert float(bad.float().mean()) < 5e-3, float(bad.float().mean())
    assert float((gX_r[:, :d_in].double() - gx).abs().mean()) < 2e-3 * float(gx.abs().mean() + 1e-20)
    assert float((gW_r.double() - gw).norm() / gw.norm()) < 5e-3


def test_rows_backward_fixed_point_weight_gradient_is_reproducible():
    """snerf_mlp_bwd_fx through the rows kernel: the waves of a workgroup are summed in wave order and the workgroups meet in 2^50-scaled
    integers, so two launches give the same bits (deterministic mode of the trainers)."""
    from soccernerfs_amd import _lib, ops

    L = _lib.lib()
    d = _desc(15, 3, 2, 1, "bf16")
    gen = torch.Generator().manual_seed(3)
    N = 50000
    W = ((torch.rand(L.snerf_mlp_param_count(C.byref(d)), generator=gen) - 0.5) * 0.5).to(DEV)
    X = (torch.rand(N, 16, generator=gen) - 0.3).to(DEV)
    gY = (torch.rand(N, 3, generator=gen) - 0.5).to(DEV)
    outs = []
    for _ in range(2):
        gX = torch.zeros(N, 16, device=DEV)
        fx = torch.zeros(W.numel(), dtype=torch.int64, device=DEV)
        _lib.check(L.snerf_mlp_bwd_fx(C.byref(d), ops._ptr(W), ops._ptr(X), 16, C.c_int64(N), ops._ptr(gY), 3, -1, None, ops._ptr(gX), 16, ops._ptr(fx),
                                      ops._stream()), "mlp_bwd_fx")
        torch.cuda.synchronize()
        outs.append((gX, fx))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert int(outs[0][1].abs().max()) > 0


@pytest.mark.parametrize("shape", [(15, 3, 2, 16), (8, 1, 1, 8)])
def test_weight_gradient_through_the_replica_workspace(shape):
    """snerf_mlp_bwd_ws + snerf_mlp_gw_reduce = snerf_mlp_bwd: same gX bit for bit, same weight gradient up to the order of the fp32 sums;
    the workspace is zero again after the reduce, and the reduce ACCUMULATES into gW."""
    from soccernerfs_amd import _lib, ops

    d_in, d_out, nh, ldx = shape
    L = _lib.lib()
    d = _desc(d_in, d_out, nh, 1, "bf16")
    gen = torch.Generator().manual_seed(9)
    N = 70001
    W = ((torch.rand(L.snerf_mlp_param_count(C.byref(d)), generator=gen) - 0.5) * 0.5).to(DEV)
    X = (torch.rand(N, ldx, generator=gen) - 0.3).to(DEV)
    gY = (torch.rand(N, d_out, generator=gen) - 0.5).to(DEV)
    gX_a, gW_a = _bwd(L.snerf_mlp_bwd, d, W, X, ldx, N, gY, d_out, -1, None, ldx)
    ws = torch.zeros(int(L.snerf_mlp_gw_workspace_floats(C.byref(d))), device=DEV)
    gX_b = torch.full((N, ldx), 7.0, device=DEV)
    _lib.check(L.snerf_mlp_bwd_ws(C.byref(d), ops._ptr(W), ops._ptr(X), ldx, C.c_int64(N), ops._ptr(gY), d_out, -1, None, ops._ptr(gX_b), ldx, ops._ptr(ws),
                                  ops._stream()), "mlp_bwd_ws")
    torch.cuda.synchronize()
    assert torch.equal(gX_a, gX_b)
    n = W.numel()
    stride = ws.numel() // 16
    assert int((ws.view(16, stride)[:, :n].abs().sum(1) > 0).sum()) == 16  # every replica received a share
    gW_b = torch.ones_like(W)  # ACCUMULATED: starts at 1
    _lib.check(L.snerf_mlp_gw_reduce(C.byref(d), ops._ptr(ws), ops._ptr(gW_b), ops._stream()), "mlp_gw_reduce")
    torch.cuda.synchronize()
    assert float(ws.abs().max()) == 0.0
    torch.testing.assert_close(gW_b - 1.0, gW_a, rtol=1e-3, atol=2e-5 * float(gW_a.abs().max()) + 1e-6)


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
@pytest.mark.parametrize("d_in,N,quotient", [(160, 1000, False), (160, 4096 * 64, True), (192, 333, True), (32, 77, False), (96, 5000, True)])
def test_sigma_net_rows_backward_matches_tile_kernel(d_in, N, quotient, operands):
    """csrc/mlp_rows128.hip (sigma_net d_in -> 128 -> 16 from the 16-bit feature tile; chain phases per wave, cooperative weight-gradient phases)
    against the workgroup-tile kernel it replaces (SNERF_MLP_SIGMA_ROWS=0): gX / G to the last bits of the fp32 accumulations, weight gradients to
    the order of the sums, the fix list of the quotient epilogue identical as a set."""
    import os

    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.tcnn_compat import Network

    dt = torch.bfloat16 if operands == "bf16" else torch.float16
    net = Network(d_in, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 1},
                  operands=operands).to(DEV)
    gen = torch.Generator().manual_seed(d_in + N % 13)
    x = torch.rand(N, d_in, generator=gen) - 0.3
    x[torch.rand(N, d_in, generator=gen) < 0.002] = 0.0
    x16 = x.to(DEV).to(dt)
    gy = (torch.rand(N, 16, generator=gen) - 0.5).to(DEV)
    gy[:, 15] = 0.0  # the trainer's layout: column 15 receives its gradient through gaux (density head)
    gaux = (torch.rand(N, generator=gen) - 0.5).to(DEV)
    L = _lib.lib()
    res = {}
    for rows in ("1", "0"):
        os.environ["SNERF_MLP_SIGMA_ROWS"] = rows
        try:
            out, gw = torch.full((N, d_in), 7.0, device=DEV), torch.zeros_like(net.params)
            if quotient:
                cap = max(N * d_in // 50, 64)
                fl = torch.full((2 * cap,), -1, dtype=torch.int32, device=DEV)
                cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
                _lib.check(L.snerf_mlp_bwd_x16_quotient(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x16), d_in, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux),
                                                        ops._ptr(out), d_in, ops._ptr(fl), cap, ops._ptr(cnt[0:1]), ops._ptr(cnt[1:2]), ops._ptr(gw), ops._stream()))
                torch.cuda.synchronize()
                n = int(cnt[0])
                assert n <= cap
                res[rows] = (out, gw, set(fl[:2 * n].view(n, 2)[:, 0].cpu().tolist()))
            else:
                _lib.check(L.snerf_mlp_bwd_x16(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x16), d_in, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux),
                                               ops._ptr(out), d_in, ops._ptr(gw), ops._stream()))
                torch.cuda.synchronize()
                res[rows] = (out, gw, None)
        finally:
            os.environ.pop("SNERF_MLP_SIGMA_ROWS", None)
    a, b = res["1"][0], res["0"][0]
    scale = float(b.abs().max())
    assert scale > 0 and bool(torch.isfinite(a).all())
    print(f"sigma rows vs tile ({operands}, {d_in}->128->16, N={N}, quotient={quotient}): differs in {int((a != b).sum())} of {a.numel()} elements, "
          f"max |diff| / max = {float((a - b).abs().max()) / scale:.2e}")
    bad = (a - b).abs() > 1e-5 * scale + 1e-4 * b.abs()
    assert float(bad.float().mean()) < 2e-3, float(bad.float().mean())
    torch.testing.assert_close(res["1"][1], res["0"][1], rtol=2e-3, atol=2e-4 * float(res["0"][1].abs().max()))
    if quotient:
        # the listed elements (vanished feature, non-zero gradient) agree except where a gradient is exactly zero in one kernel and a last-bit
        # residue in the other (a ReLU-dead row): compare through the symmetric difference
        sa, sb = res["1"][2], res["0"][2]
        assert len(sa ^ sb) <= max(2, len(sb) // 200), (len(sa), len(sb), len(sa ^ sb))


@pytest.mark.parametrize("d_in,N,quotient", [(160, 777, False), (160, 777, True), (192, 333, True), (32, 130, False)])
def test_sigma_net_rows_backward_with_a_padded_row_stride_is_the_same_bits(d_in, N, quotient):
    """csrc/mlp_rows128.hip on a ROW-STRIDED 16-bit input (ldx = d_in + 8, the pad columns holding large garbage; N not a multiple of its 128-row
    tile) must give the bits it gives on the contiguous copy of the same rows -- gX / G, the weight gradients' value (order of the sums aside), the fix list --
    so a layout or tail bug of the default sigma_net backward cannot hide inside the rows-vs-tile tolerance above."""
    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.tcnn_compat import Network

    net = Network(d_in, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 1}, operands="bf16").to(DEV)
    gen = torch.Generator().manual_seed(31 + d_in)
    x = torch.rand(N, d_in, generator=gen) - 0.3
    x[torch.rand(N, d_in, generator=gen) < 0.01] = 0.0
    xc = x.to(DEV).to(torch.bfloat16).contiguous()
    ld = d_in + 8
    xp = torch.full((N, ld), 1.0e4, device=DEV, dtype=torch.bfloat16)
    xp[:, :d_in] = xc
    gy = (torch.rand(N, 16, generator=gen) - 0.5).to(DEV)
    gy[:, 15] = 0.0
    gaux = (torch.rand(N, generator=gen) - 0.5).to(DEV)
    L = _lib.lib()
    res = []
    for X, ldx in ((xc, d_in), (xp, ld)):
        out, gw = torch.full((N, d_in), 7.0, device=DEV), torch.zeros_like(net.params)
        if quotient:
            cap = N * d_in
            fl = torch.full((2 * cap,), -1, dtype=torch.int32, device=DEV)
            cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
            _lib.check(L.snerf_mlp_bwd_x16_quotient(C.byref(net.desc), ops._ptr(net.params), ops._ptr(X), ldx, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux),
                                                    ops._ptr(out), d_in, ops._ptr(fl), cap, ops._ptr(cnt[0:1]), ops._ptr(cnt[1:2]), ops._ptr(gw), ops._stream()))
            torch.cuda.synchronize()
            n = int(cnt[0])
            ent = fl[:2 * n].view(n, 2).cpu()
            res.append((out, gw, sorted(zip(ent[:, 0].tolist(), ent[:, 1].tolist()))))
        else:
            _lib.check(L.snerf_mlp_bwd_x16(C.byref(net.desc), ops._ptr(net.params), ops._ptr(X), ldx, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux), ops._ptr(out),
                                           d_in, ops._ptr(gw), ops._stream()))
            torch.cuda.synchronize()
            res.append((out, gw, None))
    assert float(res[0][0].abs().max()) > 0
    assert torch.equal(res[0][0], res[1][0])  # per-sample quantities: the same bits whatever the row stride
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-5 * float(res[0][1].abs().max()))  # summed across workgroups: order of the atomics
    if quotient:
        assert res[0][2] == res[1][2] and len(res[0][2]) > 0  # the same (element, gradient bits) entries

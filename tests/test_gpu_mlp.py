"""GPU parity: fp32-MFMA tiny MLP fwd/bwd vs the CPU oracle (bias-free Linear stacks = the tcnn stand-in).
fp32 tolerance: outputs rtol 1e-5/atol 1e-6; gradients rtol 1e-4."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # d_in, hidden, n_hidden, d_out, out_act
    (8, 64, 1, 1, "None"),       # K-Planes proposal sigma_net
    (15, 64, 2, 3, "Sigmoid"),   # K-Planes color_net
    (32, 128, 1, 16, "None"),    # sigma_net, single scale
    (160, 128, 1, 16, "None"),   # sigma_net, k-planes preset
    (64, 128, 1, 16, "None"),
    (10, 16, 1, 1, "None"),      # nerfplayer-nerfacto proposal
    (32, 64, 1, 16, "None"),     # nerfplayer mlp_base
    (63, 64, 2, 3, "Sigmoid"),   # nerfplayer mlp_head
]


@pytest.mark.parametrize("d_in,hidden,n_hidden,d_out,out_act", SHAPES)
@pytest.mark.parametrize("N", [1000, 64])
def test_mlp_matches_oracle(d_in, hidden, n_hidden, d_out, out_act, N):
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd.tcnn_compat import Network

    gen = torch.Generator().manual_seed(d_in * 7 + N)
    net = Network(d_in, d_out, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": out_act,
                                "n_neurons": hidden, "n_hidden_layers": n_hidden})
    ws = [w.clone().requires_grad_(True) for w in net.linear_weights()]
    x = (torch.rand(N, d_in, generator=gen) * 2 - 1).requires_grad_(True)
    ref = KO.mlp(x, ws, out_act=out_act)
    gy = torch.rand(ref.shape, generator=gen) - 0.5
    ref.backward(gy)
    net = net.to(DEV)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = net(xg)
    torch.testing.assert_close(y.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    y.backward(gy.to(DEV))
    torch.testing.assert_close(xg.grad.cpu(), x.grad, rtol=1e-4, atol=1e-6)
    got = net.linear_weights(net.params.grad.cpu())
    for a, b in zip(got, ws):  # sums over N samples in a different association order (and atomics across workgroups)
        torch.testing.assert_close(a, b.grad, rtol=1e-4, atol=2e-6 * max(1.0, float(b.grad.abs().max())))


def test_mlp_exp_head_and_strided_input():
    """sigma_net: 16 outputs = 15 geo + density_before_activation; density = trunc_exp(col 15); colour net reads
    the first 15 columns of that [N,16] buffer in place (row stride 16)."""
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd.tcnn_compat import Network

    gen = torch.Generator().manual_seed(9)
    N = 777
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 1}
    sig = Network(64, 16, cfg)
    col = Network(15, 3, {**cfg, "output_activation": "Sigmoid", "n_neurons": 64, "n_hidden_layers": 2}, seed=3)
    ws = [w.clone().requires_grad_(True) for w in sig.linear_weights()]
    wc = [w.clone().requires_grad_(True) for w in col.linear_weights()]
    x = (torch.rand(N, 64, generator=gen) - 0.3) * 4
    x[0] = 30.0  # drives the density pre-activation outside trunc_exp's [-15,15] clamp
    x = x.requires_grad_(True)
    h = KO.mlp(x, ws)
    dens = KO.trunc_exp(h[:, 15:])
    rgb = KO.mlp(h[:, :15], wc, out_act="Sigmoid")
    gd, gr = torch.rand(N, 1, generator=gen), torch.rand(N, 3, generator=gen) - 0.5
    (dens * gd).sum().backward(retain_graph=True)
    (rgb * gr).sum().backward()
    sig, col = sig.to(DEV), col.to(DEV)
    xg = x.detach().to(DEV).requires_grad_(True)
    hg, dg = sig.forward_with_exp_head(xg, 15)
    rg = col(hg[:, :15])  # strided view, no copy
    torch.testing.assert_close(dg.cpu(), dens.detach()[:, 0], rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(rg.cpu(), rgb.detach(), rtol=1e-5, atol=1e-6)
    ((dg * gd[:, 0].to(DEV)).sum() + (rg * gr.to(DEV)).sum()).backward()
    torch.testing.assert_close(xg.grad.cpu(), x.grad, rtol=2e-4, atol=1e-5)
    for a, b in zip(sig.linear_weights(sig.params.grad.cpu()), ws):
        torch.testing.assert_close(a, b.grad, rtol=2e-4, atol=1e-4 * float(b.grad.abs().max()))
    for a, b in zip(col.linear_weights(col.params.grad.cpu()), wc):
        torch.testing.assert_close(a, b.grad, rtol=2e-4, atol=2e-6)


def test_mlp_full_size_linearity():
    """Config-2 size (64*4096 samples): output is linear in the last layer's weights; gradient wrt X of sum(y)
    is the same for every sample when hidden activations are all positive."""
    from soccernerfs_amd.tcnn_compat import Network

    N = 64 * 4096
    net = Network(160, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128,
                            "n_hidden_layers": 1}).to(DEV)
    x = torch.rand(N, 160, device=DEV)
    y1 = net(x)
    with torch.no_grad():
        net.params[160 * 128:] *= 2.0
    y2 = net(x)
    torch.testing.assert_close(y2, 2 * y1, rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        net.params.abs_()
    xg = x.clone().requires_grad_(True)
    net(xg).sum().backward()
    g = xg.grad
    torch.testing.assert_close(g, g[:1].expand_as(g), rtol=1e-5, atol=1e-6)


def test_mlp_unsupported_shape_raises():
    from soccernerfs_amd.tcnn_compat import Network

    # a width the fused table does not hold runs through the dense-layer kernels ...
    net = Network(8, 1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 48,
                         "n_hidden_layers": 1}).to(DEV)
    assert not net.fused and net(torch.zeros(4, 8, device=DEV)).shape == (4, 1)
    with pytest.raises(RuntimeError, match="needs a shape the fused kernels"):
        net.forward_with_exp_head(torch.zeros(4, 8, device=DEV), 0)
    # ... and so do widths beyond one 128 x 128 weight block (tiled by the entry points: test_dense_layers_wider_than_one_block)
    wide = Network(8, 1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 256,
                          "n_hidden_layers": 2}).to(DEV)
    assert not wide.fused and wide(torch.zeros(4, 8, device=DEV)).shape == (4, 1)


@pytest.mark.parametrize("dims,act,out_act", [
    ((160, 1), "None", "None"),            # the linear decoder's density layer at the preset's feature width (kplanes_field.py:236-246)
    ((3, 128, 480), "ReLU", "None"),       # its basis net, 3 F = 480 outputs
    ((3, 128, 128, 96), "ReLU", "None"),
    ((300, 200, 5), "None", "Sigmoid"),    # K and M blocks at once, ragged last blocks
    ((8, 256, 256, 1), "ReLU", "None"),    # a ReLU layer with 256 inputs: the activation is applied by the last K block only
])
def test_dense_layers_wider_than_one_block(dims, act, out_act):
    """snerf_dense_fwd / _bwd tile layers wider than 128 over the 128 x 128 kernels: column blocks are separate launches, K blocks accumulate
    and the last one applies the activation.  Against the oracle's Linear stack, forward, input gradient and weight gradients."""
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(sum(dims))
    N = 777
    ws = [((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * (3.0 / dims[i]) ** 0.5).requires_grad_(True) for i in range(len(dims) - 1)]
    x = (torch.rand(N, dims[0], generator=gen) * 2 - 1).requires_grad_(True)
    ref = KO.mlp(x, ws, out_act=out_act, hidden_act=act)
    gy = torch.rand(ref.shape, generator=gen) - 0.5
    ref.backward(gy)
    params = torch.cat([w.detach().t().reshape(-1) for w in ws]).to(DEV).requires_grad_(True)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = ops.dense_net_forward(xg, params, list(dims), act, out_act)
    torch.testing.assert_close(y.cpu(), ref.detach(), rtol=2e-5, atol=2e-6)
    y.backward(gy.to(DEV))
    # sums of up to 480 products with cancellation, in another association order: the absolute bound scales with the gradient's size
    torch.testing.assert_close(xg.grad.cpu(), x.grad, rtol=1e-4, atol=2e-6 * max(1.0, float(x.grad.abs().max())))
    off = 0
    for i, w in enumerate(ws):
        n = dims[i] * dims[i + 1]
        got = params.grad[off: off + n].view(dims[i], dims[i + 1]).t().cpu()
        torch.testing.assert_close(got, w.grad, rtol=1e-4, atol=2e-6 * max(1.0, float(w.grad.abs().max())))
        off += n


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(160, 16, 128, "None"), (8, 1, 64, "None"), (96, 16, 128, "Sigmoid"), (50, 5, 128, "None"),
                                   (192, 16, 128, "None")])  # 6 scales x 32: BASELINE config 3 (weights of layer 0 register-resident in the backward)
def test_16bit_operand_mlp(shape, operands):
    """desc.operands = 1 / 2: bf16 / fp16 MFMA operands, fp32 accumulation.  (a) equals an emulation that rounds exactly the tensors the kernel rounds
    (inputs, weights, hidden activations, upstream gradients) and accumulates in fp32; (b) stays within SURVEY §8d's tolerance for
    the bf16 MLP path against the exact fp32 kernels (density rtol 2e-2, rgb atol 4e-3)."""
    from soccernerfs_amd.tcnn_compat import Network

    d_in, d_out, hidden, out_act = shape
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": out_act, "n_neurons": hidden, "n_hidden_layers": 1}
    dt, GS = (torch.bfloat16, 1.0) if operands == "bf16" else (torch.float16, 8192.0)
    _bf = lambda t: t.to(dt).to(torch.float32)                # operand rounding
    _bg = lambda t: (t * GS).to(dt).to(torch.float32) / GS    # gradient tiles: rounded after the power-of-two scale (fp16)
    net = Network(d_in, d_out, cfg, operands=operands).to(DEV)
    ref = Network(d_in, d_out, cfg).to(DEV)
    with torch.no_grad():
        ref.params.copy_(net.params)
    gen = torch.Generator().manual_seed(11)
    N = 1000  # ragged against the 32-sample tile
    x = (torch.rand(N, d_in, generator=gen) - 0.3).to(DEV)
    go = (torch.rand(N, d_out, generator=gen) - 0.5).to(DEV)
    gaux = (torch.rand(N, generator=gen) - 0.5).to(DEV)
    W0, WO = [w.t().contiguous() for w in net.linear_weights()]  # [in, out]
    xg = x.clone().requires_grad_(True)
    y, aux = net.forward_with_exp_head(xg, 0)
    (y * go).sum().backward(retain_graph=True)
    gx_y, gw_y = xg.grad.clone(), net.params.grad.clone()
    # ---- (a) emulation ----
    a1 = torch.relu(_bf(x) @ _bf(W0))
    z = _bf(a1) @ _bf(WO)
    y_em = torch.sigmoid(z) if out_act == "Sigmoid" else z
    # elementwise: a hidden activation that sits on a bf16 rounding boundary may round the other way (fp32 accumulation order), which moves
    # an output by one bf16 ulp of that activation times a weight; the mean error shows that nothing systematic is off
    torch.testing.assert_close(y, y_em, rtol=5e-3, atol=2e-3)
    assert float((y - y_em).detach().abs().mean()) < 2e-5 * max(1.0, float(y_em.abs().mean()))
    torch.testing.assert_close(aux, torch.exp(z[:, 0]), rtol=5e-3, atol=2e-3)
    gpre = go * (y_em * (1 - y_em) if out_act == "Sigmoid" else 1.0)
    gwo_em = _bf(a1).t() @ _bg(gpre)
    gz = (_bg(gpre) @ _bf(WO).t()) * (a1 > 0)
    gw0_em = _bf(x).t() @ _bg(gz)
    gx_em = _bg(gz) @ _bf(W0).t()
    got0, goto = [w.t() for w in net.linear_weights(gw_y)]
    for got, want in ((gx_y, gx_em), (got0, gw0_em), (goto, gwo_em)):
        torch.testing.assert_close(got, want, rtol=5e-3, atol=5e-3 * float(want.abs().max()))
        assert float((got - want).abs().mean()) < 2e-4 * float(want.abs().mean() + 1e-12)
    # ---- (b) against the exact fp32 path ----
    xr = x.clone().requires_grad_(True)
    yr, auxr = ref.forward_with_exp_head(xr, 0)
    torch.testing.assert_close(aux, auxr, rtol=2e-2, atol=1e-3)
    torch.testing.assert_close(y, yr, rtol=2e-2, atol=4e-3)
    (yr * go).sum().backward()
    rel = lambda u, v: float((u - v).norm() / (v.norm() + 1e-20))
    # gradients: relative L2 distance to the exact path (operand rounding + ReLU masks of near-zero pre-activations; measured 3-4 % for bf16)
    lim = 8e-2 if operands == "bf16" else 2e-2  # fp16 measured 1.1 %
    assert rel(gx_y, xr.grad) < lim and rel(gw_y, ref.params.grad) < lim
    # trunc_exp head gradient and a forward-only call without the head
    net.params.grad = None
    xg2 = x.clone().requires_grad_(True)
    y2, aux2 = net.forward_with_exp_head(xg2, 0)
    (aux2 * gaux).sum().backward()
    gpre2 = torch.zeros_like(go)
    gpre2[:, 0] = gaux * torch.exp(z[:, 0].clamp(-15, 15))
    gx2_em = _bg((_bg(gpre2) @ _bf(WO).t()) * (a1 > 0)) @ _bf(W0).t()
    torch.testing.assert_close(xg2.grad, gx2_em, rtol=5e-3, atol=5e-3 * float(gx2_em.abs().max()))
    torch.testing.assert_close(net(x), y.detach(), rtol=0, atol=0)
    with pytest.raises(ValueError):
        Network(15, 3, {**cfg, "n_hidden_layers": 2, "n_neurons": 128}, operands=operands)  # not in the 16-bit shape table


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(15, 3, "Sigmoid"), (63, 3, "Sigmoid"), (15, 3, "None")])
def test_16bit_operand_two_hidden_layers(shape, operands):
    """The colour net (15 -> 64 -> 64 -> 3, NS/fields/kplanes_field.py:263-273) with 16-bit MFMA operands: equals an emulation that rounds what the
    kernel rounds, and stays within SURVEY 8d's bf16 tolerance (rgb atol 4e-3) of the exact fp32 kernels."""
    from soccernerfs_amd.tcnn_compat import Network

    d_in, d_out, out_act = shape
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": out_act, "n_neurons": 64, "n_hidden_layers": 2}
    dt, GS = (torch.bfloat16, 1.0) if operands == "bf16" else (torch.float16, 8192.0)
    _bf = lambda t: t.to(dt).to(torch.float32)
    _bg = lambda t: (t * GS).to(dt).to(torch.float32) / GS
    net = Network(d_in, d_out, cfg, operands=operands).to(DEV)
    ref = Network(d_in, d_out, cfg).to(DEV)
    with torch.no_grad():
        ref.params.copy_(net.params)
    gen = torch.Generator().manual_seed(5)
    N = 1111
    x = (torch.rand(N, d_in, generator=gen) - 0.3).to(DEV)
    go = (torch.rand(N, d_out, generator=gen) - 0.5).to(DEV)
    W0, W1, WO = [w.t().contiguous() for w in net.linear_weights()]
    xg = x.clone().requires_grad_(True)
    y = net(xg)
    (y * go).sum().backward()
    a1 = torch.relu(_bf(x) @ _bf(W0))
    a2 = torch.relu(_bf(a1) @ _bf(W1))
    z = _bf(a2) @ _bf(WO)
    y_em = torch.sigmoid(z) if out_act == "Sigmoid" else z
    torch.testing.assert_close(y, y_em, rtol=5e-3, atol=2e-3)
    assert float((y - y_em).detach().abs().mean()) < 5e-5 * max(1.0, float(y_em.abs().mean()))
    gpre = go * (y_em * (1 - y_em) if out_act == "Sigmoid" else 1.0)
    gwo_em = _bf(a2).t() @ _bg(gpre)
    gz2 = (_bg(gpre) @ _bf(WO).t()) * (a2 > 0)
    gw1_em = _bf(a1).t() @ _bg(gz2)
    gz1 = (_bg(gz2) @ _bf(W1).t()) * (a1 > 0)
    gw0_em = _bf(x).t() @ _bg(gz1)
    gx_em = _bg(gz1) @ _bf(W0).t()
    g0, g1, g_o = [w.t() for w in net.linear_weights(net.params.grad)]
    for got, want in ((xg.grad, gx_em), (g0, gw0_em), (g1, gw1_em), (g_o, gwo_em)):
        # a pre-activation within rounding of zero can take the other ReLU branch (fp32 accumulation order of the MFMA vs torch's matmul): that
        # changes the whole gradient row of that sample -> a handful of elements (0.1 % seen) may sit outside the elementwise tolerance
        bad = (got - want).abs() > 1e-2 * float(want.abs().max()) + 1e-2 * want.abs()
        assert float(bad.float().mean()) < 2e-2, float(bad.float().mean())
        assert float((got - want).abs().mean()) < 1e-3 * float(want.abs().mean() + 1e-12)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    torch.testing.assert_close(y, yr, rtol=2e-2, atol=4e-3)
    (yr * go).sum().backward()
    rel = lambda u, v: float((u - v).norm() / (v.norm() + 1e-20))
    lim = 1e-1 if operands == "bf16" else 3e-2
    assert rel(xg.grad, xr.grad) < lim and rel(net.params.grad, ref.params.grad) < lim


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
def test_backward_from_16bit_input_equals_backward_from_its_fp32_image(operands):
    """snerf_mlp_bwd_x16: X handed over in the operand type (the feature tile snerf_kplanes_field_fwd writes) gives the same gX as
    snerf_mlp_bwd on the fp32 image of those values, and the same gW up to the order of the float atomics."""
    import ctypes as C

    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.tcnn_compat import Network

    dt = torch.bfloat16 if operands == "bf16" else torch.float16
    net = Network(160, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 1},
                  operands=operands).to(DEV)
    N = 1000
    gen = torch.Generator().manual_seed(5)
    x16 = (torch.rand(N, 160, generator=gen) - 0.3).to(DEV).to(dt)
    x32 = x16.float()
    gy = (torch.rand(N, 16, generator=gen) - 0.5).to(DEV)
    gaux = (torch.rand(N, generator=gen) - 0.5).to(DEV)
    import os

    L = _lib.lib()
    # since round 5 a 16-bit input takes the wave-owns-rows kernel (csrc/mlp_rows128.hip), an fp32 input the workgroup-tile kernel: the bit-for-bit
    # statement is about the SAME kernel fed both forms (SNERF_MLP_SIGMA_ROWS=0 selects the tile kernel for the 16-bit input too); the two kernels
    # agree to the last bits of the fp32 accumulations (tests/test_gpu_mlp_rows.py)
    os.environ["SNERF_MLP_SIGMA_ROWS"] = "0"
    try:
        out = []
        for fn, x in ((L.snerf_mlp_bwd, x32), (L.snerf_mlp_bwd_x16, x16)):
            gx, gw = torch.full((N, 160), 7.0, device=DEV), torch.zeros_like(net.params)
            _lib.check(fn(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x), 160, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux), ops._ptr(gx), 160,
                          ops._ptr(gw), ops._stream()))
            out.append((gx, gw))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("SNERF_MLP_SIGMA_ROWS", None)
    assert torch.equal(out[0][0], out[1][0])
    torch.testing.assert_close(out[0][1], out[1][1], rtol=1e-4, atol=1e-5 * float(out[0][1].abs().max()))
    # fp32-operand nets refuse a 16-bit input
    net32 = Network(160, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 1}).to(DEV)
    rc = L.snerf_mlp_bwd_x16(C.byref(net32.desc), ops._ptr(net32.params), ops._ptr(x16), 160, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux), None, 160,
                             ops._ptr(torch.zeros_like(net32.params)), ops._stream())
    assert rc != 0 and b"16-bit" in L.snerf_last_error()
    # the 16-B loads need ldx % 8 == 0 and an aligned base; ragged N and a padded row stride work
    rc = L.snerf_mlp_bwd_x16(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x16), 164, C.c_int64(N // 2), ops._ptr(gy), 16, 15, ops._ptr(gaux), None, 160,
                             ops._ptr(torch.zeros_like(net.params)), ops._stream())
    assert rc != 0 and b"multiple of 8" in L.snerf_last_error()
    Nr = 777
    xp = torch.zeros(Nr, 168, device=DEV, dtype=dt)
    xp[:, :160] = x16[:Nr]
    res = []
    for fn, x, ld in ((L.snerf_mlp_bwd, x32[:Nr].contiguous(), 160), (L.snerf_mlp_bwd_x16, xp, 168)):
        gx, gw = torch.empty(Nr, 160, device=DEV), torch.zeros_like(net.params)
        _lib.check(fn(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x), ld, C.c_int64(Nr), ops._ptr(gy), 16, 15, ops._ptr(gaux), ops._ptr(gx), 160,
                      ops._ptr(gw), ops._stream()))
        res.append((gx, gw))
    # (default dispatch: tile kernel on the fp32 image, rows kernel on the padded 16-bit tile -- equal up to the fp32 accumulation order)
    scale = float(res[0][0].abs().max())
    assert float(((res[0][0] - res[1][0]).abs() > 1e-5 * scale + 1e-4 * res[0][0].abs()).float().mean()) < 2e-3
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-3, atol=2e-4 * float(res[0][1].abs().max()))


@pytest.mark.parametrize("d_in,N", [(160, 1000), (192, 333), (32, 64)])
def test_quotient_epilogue_of_the_sigma_net_backward(d_in, N):
    """snerf_mlp_bwd_x16_quotient: G = gX .* X16 formed from the LDS-resident feature tile -- bit for bit the product of snerf_mlp_bwd_x16's gX
    with the fp32 image of X16 -- zero where X16 vanished (zero / subnormal), those elements listed with their gX; the weight gradients are the
    plain kernel's; the OTHER list counter is reset for the next step; ragged N."""
    import ctypes as C

    from soccernerfs_amd import _lib, ops
    from soccernerfs_amd.tcnn_compat import Network

    net = Network(d_in, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 1},
                  operands="bf16").to(DEV)
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(N, d_in, generator=gen) - 0.3
    x[torch.rand(N, d_in, generator=gen) < 0.01] = 0.0   # vanished features
    x[3, 5], x[N - 1, d_in - 1] = 3e-39, -2e-39          # subnormal: vanished too
    x16 = x.to(DEV).to(torch.bfloat16)
    gy = (torch.rand(N, 16, generator=gen) - 0.5).to(DEV)
    gaux = (torch.rand(N, generator=gen) - 0.5).to(DEV)
    L = _lib.lib()
    gx, gw = torch.empty(N, d_in, device=DEV), torch.zeros_like(net.params)
    _lib.check(L.snerf_mlp_bwd_x16(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x16), d_in, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux), ops._ptr(gx),
                                   d_in, ops._ptr(gw), ops._stream()))
    G, gw2 = torch.full((N, d_in), 7.0, device=DEV), torch.zeros_like(net.params)
    cap = N * d_in
    fix_list = torch.full((2 * cap,), -1, dtype=torch.int32, device=DEV)
    counts = torch.tensor([0, 12345], dtype=torch.int32, device=DEV)
    _lib.check(L.snerf_mlp_bwd_x16_quotient(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x16), d_in, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux),
                                            ops._ptr(G), d_in, ops._ptr(fix_list), cap, ops._ptr(counts[0:1]), ops._ptr(counts[1:2]), ops._ptr(gw2), ops._stream()))
    torch.cuda.synchronize()
    xf = x16.float()
    vanished = xf.abs() < 1.17549435e-38
    assert int(vanished.sum()) > 2
    want = torch.where(vanished, torch.zeros_like(gx), gx * xf)
    assert torch.equal(G, want)
    torch.testing.assert_close(gw2, gw, rtol=1e-4, atol=1e-5 * float(gw.abs().max()))
    n = int(counts[0])
    assert int(counts[1]) == 0  # the next step's counter was reset by this launch
    listed = vanished & (gx != 0)
    assert n == int(listed.sum())
    ent = fix_list[:2 * n].view(n, 2).cpu()
    order = torch.argsort(ent[:, 0])
    idx = ent[order, 0].long()
    assert torch.equal(idx, torch.nonzero(listed.reshape(-1).cpu()).reshape(-1))
    assert torch.equal(ent[order, 1].contiguous().view(torch.float32), gx.reshape(-1).cpu()[idx])
    # a single counter (fix_count_next = NULL) is reset by the call itself; shapes outside the sigma_net family are refused
    counts[0] = 999
    _lib.check(L.snerf_mlp_bwd_x16_quotient(C.byref(net.desc), ops._ptr(net.params), ops._ptr(x16), d_in, C.c_int64(N), ops._ptr(gy), 16, 15, ops._ptr(gaux),
                                            ops._ptr(G), d_in, ops._ptr(fix_list), cap, ops._ptr(counts[0:1]), None, ops._ptr(gw2), ops._stream()))
    assert int(counts[0]) == n
    net64 = Network(32, 1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1},
                    operands="bf16").to(DEV)
    rc = L.snerf_mlp_bwd_x16_quotient(C.byref(net64.desc), ops._ptr(net64.params), ops._ptr(x16), d_in, C.c_int64(N), None, 1, 0, ops._ptr(gaux),
                                      ops._ptr(G), d_in, ops._ptr(fix_list), cap, ops._ptr(counts[0:1]), None, ops._ptr(torch.zeros_like(net64.params)), ops._stream())
    assert rc != 0 and b"sigma_net shapes" in L.snerf_last_error()


def test_linear_decoder_pieces_against_torch_and_edge_cases():
    """csrc/linear_decoder.hip: trunc_exp (forward exp, backward with the exponent clamped to [-15, 15]: NS/field_components/activations.py:25-41)
    and basis_rgb (sigmoid of the feature / basis contraction, kplanes_field.py:349-354) against the oracle's formulas under autograd; empty
    inputs, the clamp's two sides, a row-strided feature tensor, F = 4 (one lane of sixteen active) and F = 160."""
    from oracle import kplanes_oracle as KO
    from soccernerfs_amd import ops

    gen = torch.Generator().manual_seed(2)
    x = torch.cat([torch.randn(1000, generator=gen) * 3, torch.tensor([-40.0, -15.0, 15.0, 20.0, 0.0])]).requires_grad_(True)
    g = torch.randn(x.shape, generator=gen)
    ref = KO.trunc_exp(x)
    ref.backward(g)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = ops.trunc_exp(xg)
    y.backward(g.to(DEV))
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=2e-6, atol=0)
    torch.testing.assert_close(xg.grad.cpu(), x.grad, rtol=2e-6, atol=0)
    assert float(xg.grad[-2].cpu() / g[-2]) == pytest.approx(float(torch.exp(torch.tensor(15.0))), rel=1e-6)  # x = 20: the backward stops at e^15
    assert ops.trunc_exp(torch.zeros(0, 1, device=DEV)).shape == (0, 1)
    for F, N in ((4, 33), (64, 1000), (160, 257)):
        feat = (torch.rand(N, F, generator=gen) * 2 - 1).requires_grad_(True)
        basis = (torch.rand(N, 3 * F, generator=gen) * 2 - 1).requires_grad_(True)
        w = torch.rand(N, 3, generator=gen) - 0.5
        ref = torch.sigmoid(torch.sum(feat[:, None, :] * basis.view(N, 3, F), dim=-1))
        (ref * w).sum().backward()
        fg, bg = feat.detach().to(DEV).requires_grad_(True), basis.detach().to(DEV).requires_grad_(True)
        out = ops.basis_rgb(fg, bg)
        (out * w.to(DEV)).sum().backward()
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(fg.grad.cpu(), feat.grad, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(bg.grad.cpu(), basis.grad, rtol=1e-4, atol=1e-6)
    assert ops.basis_rgb(torch.zeros(0, 32, device=DEV), torch.zeros(0, 96, device=DEV)).shape == (0, 3)
    with pytest.raises(ValueError, match="multiple of 4"):
        ops.basis_rgb(torch.zeros(5, 6, device=DEV), torch.zeros(5, 18, device=DEV))
    with pytest.raises(RuntimeError, match="HIP device tensor"):
        ops.trunc_exp(torch.zeros(4))
    # wide dense layers: nothing to do for N = 0
    assert ops.dense_net_forward(torch.zeros(0, 160, device=DEV), torch.zeros(160, device=DEV), [160, 1], "None", "None").shape == (0, 1)

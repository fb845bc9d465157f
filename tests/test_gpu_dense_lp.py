"""GPU: single dense layers with bf16 MFMA operands (csrc/dense_lp.hip: snerf_dense_fwd_lp / snerf_dense_bwd_lp) against a float64 emulation of exactly what
they round (X, W and dZ to bf16; products and sums exact), at the layer shapes of the full NeRFPlayer's nets; the fixed-point weight-gradient path; and
NerfplayerFullTrainer(mlp_operands="bf16") against the exact-fp32 trainer."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _p(t):
    return C.c_void_p(t.data_ptr())


def _bf(x):
    return x.to(torch.bfloat16).double()


# (K, M, act): deformation net 3 -> 128 -> 128 -> 128 -> 3, static MLP 33 -> 64 -> 32, colour head 15 -> 64 -> 64 -> 64 -> 3 (sigmoid), odd sizes
SHAPES = [(3, 128, 1), (128, 128, 1), (128, 3, 0), (33, 64, 1), (64, 32, 0), (15, 64, 1), (64, 64, 1), (64, 3, 2), (100, 20, 1), (1, 1, 0), (32, 128, 2)]


@pytest.mark.parametrize("K,M,act", SHAPES)
@pytest.mark.parametrize("N", [50000, 77])
def test_dense_lp_matches_bf16_emulation(K, M, act, N):
    from soccernerfs_amd import _lib, ops

    L = _lib.lib()
    assert L.snerf_dense_lp_supported(K, M, 1) == 1 and L.snerf_dense_lp_supported(K, M, 0) == 0 and L.snerf_dense_lp_supported(129, M, 1) == 0
    gen = torch.Generator().manual_seed(K * 1000 + M + N)
    ldx, ldy = K + 3, M + 1  # strided rows
    W = ((torch.rand(K, M, generator=gen) - 0.5) * 0.6).to(DEV)
    Xw = (torch.rand(N, ldx, generator=gen) - 0.4).to(DEV)
    X = Xw[:, :K]
    Yw = torch.full((N, ldy), 7.0, device=DEV)
    st = ops._stream()
    _lib.check(L.snerf_dense_fwd_lp(_p(W), K, M, act, _p(Xw), ldx, C.c_int64(N), _p(Yw), ldy, 1, st), "fwd_lp")
    Y = Yw[:, :M]
    assert bool((Yw[:, M] == 7.0).all())  # nothing written past the layer's columns
    z = _bf(X) @ _bf(W)
    ref = torch.relu(z) if act == 1 else (torch.sigmoid(z) if act == 2 else z)
    torch.testing.assert_close(Y.double(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max())))
    # backward from the kernel's own stored output
    gYw = (torch.rand(N, ldy, generator=gen) - 0.5).to(DEV)
    gY = gYw[:, :M]
    gXw = torch.full((N, ldx), 5.0, device=DEV)
    gW = torch.zeros(K, M, device=DEV)
    _lib.check(L.snerf_dense_bwd_lp(_p(W), K, M, act, _p(Xw), ldx, C.c_int64(N), _p(Yw), ldy, _p(gYw), ldy, _p(gXw), ldx, _p(gW), None, 1, st), "bwd_lp")
    # dZ in the kernel's own fp32 arithmetic (so that its bf16 rounding lands on the same side), then rounded as the kernel rounds it
    dz = torch.where(Y > 0, gY, torch.zeros_like(gY)) if act == 1 else (gY * Y * (1 - Y) if act == 2 else gY.clone())
    dzb = _bf(dz)
    gx_ref = dzb @ _bf(W).t()
    gw_ref = _bf(X).t() @ dzb
    assert bool((gXw[:, K:] == 5.0).all())
    torch.testing.assert_close(gXw[:, :K].double(), gx_ref, rtol=1e-5, atol=1e-5 * max(1.0, float(gx_ref.abs().max())))
    # the weight gradient sums N products in fp32 (MFMA accumulators per workgroup, float atomics across workgroups)
    torch.testing.assert_close(gW.double(), gw_ref, rtol=1e-4, atol=3e-6 * float(gw_ref.abs().max()) + 1e-7)
    # fixed-point cells: same values, bit-identical between launches; gX unchanged; weight gradient only (gX NULL)
    cells = []
    for k in range(2):
        fx = torch.zeros(K * M, dtype=torch.int64, device=DEV)
        gX2 = torch.empty(N, ldx, device=DEV)
        _lib.check(L.snerf_dense_bwd_lp(_p(W), K, M, act, _p(Xw), ldx, C.c_int64(N), _p(Yw), ldy, _p(gYw), ldy, _p(gX2) if k == 0 else None, ldx, None, _p(fx), 1, st), "fx")
        if k == 0:
            assert torch.equal(gX2[:, :K], gXw[:, :K])
        cells.append(fx)
    assert torch.equal(cells[0], cells[1])
    out = torch.empty(K * M, device=DEV)
    ops.fx_to_float(cells[0], out)
    torch.testing.assert_close(out.view(K, M).double(), gw_ref, rtol=1e-4, atol=3e-6 * float(gw_ref.abs().max()) + 1e-7)
    # argument checks: both accumulators, unsupported operand type, widths beyond 128
    assert L.snerf_dense_bwd_lp(_p(W), K, M, act, _p(Xw), ldx, C.c_int64(N), _p(Yw), ldy, _p(gYw), ldy, None, ldx, _p(gW), _p(cells[0]), 1, st) != 0
    assert L.snerf_dense_fwd_lp(_p(W), K, M, act, _p(Xw), ldx, C.c_int64(N), _p(Yw), ldy, 2, st) != 0
    assert L.snerf_dense_fwd_lp(_p(W), K, M, act, _p(Xw), ldx, C.c_int64(0), _p(Yw), ldy, 1, st) == 0


def _trainers(det=False):
    from tests.conftest import load_golden
    from tests.test_gpu_hashgrid import _full_model
    from tests.test_gpu_nerfplayer_full_trainer import _pairs
    from soccernerfs_amd.nerfplayer_full_trainer import NerfplayerFullTrainer

    g, gb = load_golden("g13_nerfplayer_full"), load_golden("g13b_nerfplayer_dynamics")
    model, _ = _full_model(g)
    R = int(g["R"])
    pairs = _pairs(model)
    t = lambda k: g[k].to(DEV).contiguous()
    rays = {"origins": t("origins"), "directions": t("directions"), "times": t("times")}

    def make(operands):
        tr = NerfplayerFullTrainer(model.config, R, aabb_scale=1.0, device=DEV, lr=float(gb["lr0"]), adam_eps=float(gb["eps"]), warm_up_end=int(gb["warm_up_end"]),
                                   max_steps=int(gb["max_steps"]), seed=0, mlp_operands=operands, deterministic=det)
        with torch.no_grad():
            for name, p in pairs.items():
                tr.views[name].copy_(p.detach().reshape(tr.views[name].shape))
        return tr

    return make, rays, t("target"), gb


def test_full_trainer_bf16_operands_close_to_fp32_on_one_step():
    """Same parameters, batch and draws through the exact-fp32 trainer and the bf16-operand one: rendered colours, loss terms and every gradient tensor agree
    to operand-rounding accuracy (2^-9 per operand, a few layers deep)."""
    make, rays, target, gb = _trainers()
    outs = {}
    for op in ("fp32", "bf16"):
        tr = make(op)
        assert tr._dense_operands == (1 if op == "bf16" else 0) and tr.decode.desc.operands == tr._dense_operands  # fused nets whose shape has a 16-bit kernel follow
        rng = {"t_rand": gb["t_rand"][0].to(DEV), "u": [gb["u0"][0].to(DEV), gb["u1"][0].to(DEV)], "bg": gb["bg"][0].to(DEV)}
        tr.tv_rows = [int(x) for x in gb["tv_rows"][0]]
        rgb = tr.forward(rays, rng, 1.0).clone()
        tr.backward(target, rng, proposal_grads=True)
        outs[op] = (rgb, {k: float(v) for k, v in tr.loss_dict().items()}, {k: v.clone() for k, v in tr.gviews.items()})
    (ra, la, ga), (rb, lb, gb_) = outs["fp32"], outs["bf16"]
    assert float((ra - rb).abs().max()) <= 1e-2, float((ra - rb).abs().max())
    for k in la:
        assert abs(la[k] - lb[k]) <= 3e-2 * max(abs(la[k]), 1e-6), (k, la[k], lb[k])
    # relative L2 deviation of every gradient tensor.  The deformation net's gradient comes through the hash grid's COORDINATE gradient, which is piecewise
    # constant in the deformed position: a bf16-level change of the deformation moves samples across cells of the fine levels (4096 per unit), so that one
    # tensor differs by tens of per cent between ANY two arithmetics of this model (the reference runs the net in fp16)
    rel = {name: float((ga[name] - gb_[name]).norm() / ga[name].norm().clamp_min(1e-20)) for name in ga}
    print("bf16 vs fp32 operands, one step: max |rgb| deviation", float((ra - rb).abs().max()), "relative L2 gradient deviation by tensor:",
          {k: round(v, 4) for k, v in rel.items()})
    # measured: deform 0.30, hash 0.13, newness 0.095 (the three tables read AT the deformed positions), static MLP 0.045, every other tensor <= 0.025
    for name, v in rel.items():
        bound = 0.6 if name == "field.deform" else (0.25 if name in ("field.hash", "field.newness", "field.decomp") else 0.1)
        assert v <= bound, (name, v)


def test_full_trainer_bf16_operands_follow_g13b():
    """G13b's 50 steps with bf16 operands (deterministic mode, so ONE reproducible run): value by value for the first steps at operand-rounding accuracy, mean
    rendered probabilities within 0.1 of the reference's run at every step (the reference restarted one fp32 rounding away moves by 0.053), the run ends static."""
    make, rays, target, gb = _trainers(det=True)
    tr = make("bf16")
    dev = []
    for step in range(int(gb["steps"])):
        rng = {"t_rand": gb["t_rand"][step].to(DEV), "u": [gb["u0"][step].to(DEV), gb["u1"][step].to(DEV)], "bg": gb["bg"][step].to(DEV)}
        tr.tv_rows = [int(x) for x in gb["tv_rows"][step]]
        tr.train_step(rays, target, rng)
        ld = tr.loss_dict()
        if step < 3:
            ref, got = float(gb["loss_rgb_loss"][step]), float(ld["rgb_loss"])
            assert abs(got - ref) <= 5e-2 * abs(ref), (step, got, ref)
        dev.append(float((tr.rendered_probs().mean(0).cpu() - gb["probs_mean"][step]).abs().max()))
    end = tr.rendered_probs().mean(0).cpu()
    print("bf16 operands, G13b: |mean probabilities - reference's| by step:", [round(x, 3) for x in dev[::5]], "end", [round(float(x), 3) for x in end])
    assert max(dev) <= 0.1, dev
    assert abs(float(end[0]) - float(gb["probs_mean"][-1][0])) <= 0.1 and float(ld["rgb_loss"]) < 0.8 * float(gb["loss_rgb_loss"][0])


@pytest.mark.parametrize("shape", [(3, 3, 128, 3, "None"), (33, 32, 64, 1, "None"), (15, 3, 64, 3, "Sigmoid")])
def test_layer_chained_network_with_bf16_operands(shape):
    """tcnn_compat.Network for a shape outside the fused kernels' table, operands="bf16": chained from csrc/dense_lp.hip's layers through autograd; values and
    gradients within operand-rounding distance of the exact-fp32 network with the same parameters."""
    from soccernerfs_amd.tcnn_compat import Network

    d_in, d_out, hidden, nh, out_act = shape
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": out_act, "n_neurons": hidden, "n_hidden_layers": nh}
    a, b = Network(d_in, d_out, cfg, operands="fp32").to(DEV), Network(d_in, d_out, cfg, operands="bf16", chained_16bit=True).to(DEV)
    assert not a.fused and not b.fused and b.dense_operands == 1 and a.dense_operands == 0
    with torch.no_grad():
        b.params.copy_(a.params)
    gen = torch.Generator().manual_seed(9)
    x = (torch.rand(1000, d_in, generator=gen) - 0.5).to(DEV)
    go = (torch.rand(1000, d_out, generator=gen) - 0.5).to(DEV)
    outs = []
    for net in (a, b):
        xi = x.clone().requires_grad_(True)
        y = net(xi)
        y.backward(go)
        outs.append((y.detach(), xi.grad, net.params.grad))
    for u, v in zip(outs[0], outs[1]):
        assert float((u - v).norm() / u.norm()) <= 0.1, float((u - v).norm() / u.norm())  # measured up to 0.064 (three ReLU layers: units near zero flip)
    with pytest.raises(ValueError):
        Network(3, 3, {**cfg, "n_neurons": 256}, operands="bf16", chained_16bit=True)  # wider than the 16-bit single layers
    with pytest.raises(ValueError):
        Network(3, 3, {**cfg, "n_hidden_layers": 3, "n_neurons": 128}, operands="fp16", chained_16bit=True)  # the single layers are bf16 only
    with pytest.raises(ValueError):
        Network(3, 3, {**cfg, "n_hidden_layers": 3, "n_neurons": 128}, operands="bf16")  # not asked for: the fused-table contract stands

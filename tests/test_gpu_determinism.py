"""GPU: reproducibility machinery of the fused trainer.

* deterministic mode (fixed-point gradient accumulation): two runs from one seed give bit-identical parameters, and agree with the
  float-atomic mode within the atomic-order tolerance;
* skip-step semantics (the reference's GradScaler, NS/engine/trainer.py:394-408): a non-finite gradient skips the WHOLE optimiser step of
  its parameter group -- parameters and moments untouched, gradient cleared, Adam's own step counter not advanced;
* snerf_weights_bwd raises the group's flag exactly where autograd would have produced a non-finite gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SMALL = dict(aabb_scale=1.5, spacetime_resolution=(16, 16, 16, 4), multiscale_res=(1, 2), feature_dim=32,
             proposal_resolutions=((24, 24, 24, 4), (32, 32, 32, 4)), proposal_feature_dim=8, num_proposal_samples_per_ray=(64, 32),
             num_nerf_samples_per_ray=16, warm_up_end=2)


def _inputs(R, steps, seed=3):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(steps):
        o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 1.2
        d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
        rays = {"origins": o, "directions": d, "times": torch.rand(R, 1, generator=gen)}
        rng = {"t_rand": torch.rand(R, 65, generator=gen), "u": [torch.rand(R, 33, generator=gen), torch.rand(R, 17, generator=gen)],
               "bg": torch.rand(R, 3, generator=gen)}
        out.append((rays, torch.rand(R, 3, generator=gen), rng))
    return out


def _run(cfg_kw, R=512, steps=4):
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    tr = KPlanesTrainer(KPlanesTrainConfig(**SMALL, **cfg_kw), R, DEV)
    g = lambda z: z.to(DEV).contiguous()
    for rays, target, rng in _inputs(R, steps):
        tr.train_step({k: g(v) for k, v in rays.items()}, g(target), {"t_rand": g(rng["t_rand"]), "u": [g(u) for u in rng["u"]], "bg": g(rng["bg"])})
    tr.synchronize()
    return tr


@pytest.mark.parametrize("operands", ["fp32", "bf16"])
def test_deterministic_mode_is_bit_reproducible(operands):
    a = _run(dict(deterministic=True, mlp_operands=operands))
    b = _run(dict(deterministic=True, mlp_operands=operands))
    assert torch.equal(a.params, b.params) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    assert float(a.grads.abs().max()) == 0.0 and int(a.grads_fx.abs().max()) == 0
    # ... and it is the same optimisation as the float-atomic mode, up to the accumulation order of the atomics.  Like with like: the product form of
    # the scatter on both sides (deterministic mode has no quotient form; with 16-bit operands the default path's G = gX .* X16 also carries the
    # operand rounding of the features)
    c = _run(dict(deterministic=False, mlp_operands=operands, quotient_scatter=False))
    assert a.step == c.step == 4
    torch.testing.assert_close(a.params, c.params, rtol=0, atol=2e-3)
    assert float((a.params - c.params).abs().mean()) < 2e-5
    # the DEFAULT float-atomic path (quotient scatter; bf16: G formed in the sigma_net backward's epilogue from the rounded features): Adam turns a
    # 2^-9 relative change of a near-zero gradient into a full-size step now and then, so a handful of parameters may sit ~lr away after four steps
    d = _run(dict(deterministic=False, mlp_operands=operands))
    assert d.quotient_scatter and d.quotient_epilogue == (operands == "bf16")
    diff = (a.params - d.params).abs()
    assert float(diff.mean()) < 5e-5 and float((diff > 2e-3).float().mean()) < 2e-3 and float(diff.max()) < 4.5e-2
    assert a.skipped_steps()["fields"] == {"adam_steps": 4, "skipped": 0, "dropped_elements": 0}


def test_adam_skip_step_semantics():
    """A flagged step leaves p, m, v alone, clears g and does not advance Adam's counter; the next step uses t, not the scheduler step."""
    from soccernerfs_amd import ops

    n = 4096 + 3
    torch.manual_seed(0)
    p0 = torch.randn(n, device=DEV)
    mk = lambda: (p0.clone(), torch.randn(n, device=DEV, generator=torch.Generator(DEV).manual_seed(1)), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV))
    # reference: two plain steps (host-side step numbers 1, 2)
    p, g, m, v = mk()
    g1 = g.clone()
    ops.adam_step(p, g, m, v, 1, 1e-2, zero_grad=True)
    g.copy_(g1 * 0.5)
    ops.adam_step(p, g, m, v, 2, 2e-2, zero_grad=True)
    # with the device-side state: step, SKIPPED step (flag set by a producer), step
    q, h, mq, vq = mk()
    dyn = ops.new_adam_dyn(DEV)
    ops.adam_prepare(dyn, 1e-2)
    ops.adam_step(q, h, mq, vq, 0, 1e-2, zero_grad=True, dyn=dyn)
    snap = (q.clone(), mq.clone(), vq.clone())
    h.copy_(torch.full_like(h, float("nan")))
    dyn[0] = 1  # what snerf_weights_bwd does
    ops.adam_prepare(dyn, 1.5e-2)
    ops.adam_step(q, h, mq, vq, 0, 1.5e-2, zero_grad=True, dyn=dyn)
    assert torch.equal(q, snap[0]) and torch.equal(mq, snap[1]) and torch.equal(vq, snap[2]) and float(h.abs().max()) == 0.0
    assert dyn.cpu().tolist()[:4] == [0, 1, 1, 0]
    h.copy_(g1 * 0.5)
    ops.adam_prepare(dyn, 2e-2)
    ops.adam_step(q, h, mq, vq, 0, 2e-2, zero_grad=True, dyn=dyn)
    assert dyn.cpu().tolist()[:4] == [0, 2, 1, 0]
    torch.testing.assert_close(q, p, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(mq, m, rtol=1e-6, atol=0)
    # "drop_elements": the step is taken, non-finite elements contribute nothing and are counted
    h.copy_(g1)
    h[5] = float("inf")
    h[n - 1] = float("nan")  # in the scalar tail
    dyn[0] = 1
    ops.adam_prepare(dyn, 1e-2, policy="drop_elements")
    before = q.clone()
    ops.adam_step(q, h, mq, vq, 0, 1e-2, zero_grad=True, dyn=dyn)
    assert dyn.cpu().tolist()[:4] == [0, 3, 1, 2]
    assert not torch.equal(q, before) and bool(torch.isfinite(q).all())


def test_planes_sweep_skips_with_its_group():
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    ps = PlaneSet(8, [[6, 5, 4, 3]], concat=False).to(DEV)
    n = ps.numel
    p_in = ps.planes.detach().clone()
    p_out, g, m, v = torch.zeros(n, device=DEV), torch.randn(n, device=DEV), torch.rand(n, device=DEV), torch.rand(n, device=DEV)
    m0, v0 = m.clone(), v.clone()
    dyn = ops.new_adam_dyn(DEV)
    dyn[0] = 1
    ops.adam_prepare(dyn, 1e-2)
    losses = torch.zeros(ops.REG_SLOTS, 16, device=DEV)
    ops.adam_planes_step(ps, p_in, p_out, g, m, v, (1e-3, 1e-3, 1e-3), losses, 0, 1e-2, dyn=dyn)
    assert torch.equal(p_out, p_in) and torch.equal(m, m0) and torch.equal(v, v0) and float(g.abs().max()) == 0.0
    assert float(losses.sum()) > 0  # the regulariser VALUES are still reported


def test_weights_bwd_raises_flag_on_inf_density_with_zero_width_bin():
    import ctypes as C

    from soccernerfs_amd import _lib, ops

    R, S = 8, 16
    torch.manual_seed(0)
    dens = torch.rand(R, S, device=DEV) * 5
    eb = torch.cumsum(torch.rand(R, S + 1, device=DEV) * 0.1 + 0.01, dim=1)
    gw = torch.randn(R, S, device=DEV)
    out = torch.empty(R, S, device=DEV)
    flag = torch.zeros(8, dtype=torch.int32, device=DEV)
    call = lambda: _lib.check(_lib.lib().snerf_weights_bwd(ops._ptr(dens), ops._ptr(eb), ops._ptr(gw), R, S, ops._ptr(out), 0, ops._ptr(flag), ops._stream()))
    call()
    assert int(flag[0]) == 0 and bool(torch.isfinite(out).all())
    dens[3, 7] = float("inf")
    eb[3, 8] = eb[3, 7]  # zero-width bin: delta * sigma = 0 * inf
    call()
    assert int(flag[0]) == 1 and bool(torch.isfinite(out).all())

"""The stock-PyTorch stand-in trainer (oracle/torch_standin.StandinTrainer; baseline code for PSNR runs) on the CPU: its step equals
the oracle's forward / loss dict / adam_step applied by hand, proposal updates follow the sampler's schedule, eval forward runs."""
import torch

from oracle import kplanes_oracle as KO
from oracle import torch_standin as TS

TINY = dict(base_res=(8, 8, 8, 4), multiscale=(1, 2), prop_res=((8, 8, 8, 4), (16, 16, 16, 4)))


def _rays(R, gen):
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.9
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    return {"origins": o, "directions": d, "times": torch.rand(R, 1, generator=gen)}


def test_standin_step_is_the_oracle_step_with_two_adams():
    R, S = 32, (16, 8, 4)
    tr = TS.StandinTrainer("cpu", R, seed=3, model=TINY, samples=S)
    P = KO.make_kplanes_params(seed=3, **TINY)
    leaves = KO.all_param_tensors(P)
    for x in leaves:
        x.requires_grad_(True)
    m = [torch.zeros_like(x) for x in leaves]
    v = [torch.zeros_like(x) for x in leaves]
    gen = torch.Generator().manual_seed(3)  # the trainer's own draw stream, replayed
    g2 = torch.Generator().manual_seed(11)
    for step in range(3):
        rays, target = _rays(R, g2), torch.rand(R, 3, generator=g2)
        rgb = tr.train_step(rays, target)
        rnd = lambda *s: torch.rand(*s, generator=gen)
        rng = {"t_rand": rnd(R, S[0] + 1), "u": [rnd(R, S[1] + 1), rnd(R, S[2] + 1)], "bg": rnd(R, 3)}
        out = KO.kplanes_forward(P, rays, rng, S[:2], S[2], anneal=KO.anneal_value(step))
        torch.testing.assert_close(rgb, out["rgb"].detach())
        loss = sum(KO.kplanes_loss_dict(P, out, target).values())
        grads = torch.autograd.grad(loss, leaves)
        with torch.no_grad():
            for x, g, mm, vv in zip(leaves, grads, m, v):
                KO.adam_step(x, g, mm, vv, step + 1, 1e-2 * KO.cosine_lr_factor(step))
    mine = torch.cat([x.detach().reshape(-1) for x in [t for sc in P["prop_grids"] for t in sc] + [w for lv in P["prop_sigma"] for w in lv]
                      + [t for sc in P["field_grids"] for t in sc] + list(P["field_sigma"]) + list(P["field_color"])])
    torch.testing.assert_close(tr.params, mine, rtol=1e-4, atol=1e-6)
    assert tr.step == 3 and tr.skipped_steps() == {"proposal_networks": 0, "fields": 0}
    img = tr.forward(_rays(R, g2), None, 1.0, training=False)
    assert img.shape == (R, 3) and bool(torch.isfinite(img).all()) and float(img.min()) >= 0 and float(img.max()) <= 1


def test_standin_skips_an_optimiser_whose_gradient_is_not_finite():
    tr = TS.StandinTrainer("cpu", 8, seed=0, model=TINY, samples=(8, 4, 4))
    g = torch.Generator().manual_seed(0)
    before = tr.params.clone()
    tr.train_step(_rays(8, g), torch.full((8, 3), float("nan")))
    assert tr.skipped_steps()["fields"] == 1
    n_prop = sum(x.numel() for x in tr.groups["proposal_networks"])
    torch.testing.assert_close(tr.params[n_prop:], before[n_prop:], rtol=0, atol=0)

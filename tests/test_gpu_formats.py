"""GPU: a checkpoint in the reference's file format (G15: the reference's own KPlanesModel state after three Adam steps) loaded into this
package's model renders what the reference model renders; training resumes from the imported Adam moments."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_checkpoint_renders_the_same(tmp_path):
    import shutil

    from soccernerfs_amd import checkpoint as CK
    from soccernerfs_amd.rays import RayBundle
    from tests.test_formats_cpu import _small_kplanes

    shutil.copy(os.path.join(GOLD, "g15_step-000000002.ckpt"), tmp_path / "step-000000002.ckpt")
    model = _small_kplanes()
    start, moments = CK.load_checkpoint(str(tmp_path), model)
    assert start == 3 and len(moments) == 7  # 3 plane sets + 4 nets
    model = model.to(DEV).eval()
    model.scene_box.aabb = model.scene_box.aabb.to(DEV)
    g = np.load(os.path.join(GOLD, "g15_checkpoint.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(DEV).contiguous()
    R = g["origins"].shape[0]
    rb = RayBundle(origins=t("origins"), directions=t("directions"), pixel_area=torch.ones(R, 1, device=DEV),
                   camera_indices=torch.zeros(R, 1, dtype=torch.long, device=DEV), times=t("times"))
    with torch.no_grad():
        out = model(rb)
    torch.testing.assert_close(out["rgb"].cpu(), torch.from_numpy(g["rgb"]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out["accumulation"].cpu(), torch.from_numpy(g["accumulation"]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out["depth"].cpu(), torch.from_numpy(g["depth"]), rtol=1e-4, atol=1e-4)
    # resume: FusedAdam seeded with the imported moments takes the step torch.optim.Adam would take from the reference state
    from soccernerfs_amd.optimizers import FusedAdam

    params = {n: p for n, p in model.named_parameters() if n in moments}
    opt = FusedAdam(list(params.values()), lr=1e-2, eps=1e-12)
    ref_params = {n: p.detach().clone().requires_grad_(True) for n, p in params.items()}
    ref_opt = torch.optim.Adam(list(ref_params.values()), lr=1e-2, eps=1e-12)
    gen = torch.Generator().manual_seed(0)
    for n, p in params.items():
        m, v, step = moments[n]
        opt.state[p].update(step=step, exp_avg=m.to(DEV).reshape(-1).clone(), exp_avg_sq=v.to(DEV).reshape(-1).clone())
        ref_opt.state[ref_params[n]].update(step=torch.tensor(float(step)), exp_avg=m.to(DEV).reshape(p.shape).clone(),
                                            exp_avg_sq=v.to(DEV).reshape(p.shape).clone())
        grad = (torch.rand(p.shape, generator=gen) - 0.5).to(DEV)
        p.grad = grad.clone()
        ref_params[n].grad = grad.clone()
    opt.step()
    ref_opt.step()
    for n, p in params.items():
        torch.testing.assert_close(p.detach(), ref_params[n].detach(), rtol=1e-5, atol=1e-7, msg=n)


def test_fused_trainer_checkpoint_roundtrip_and_reference_names(tmp_path):
    """The fused trainer writes a file with exactly the reference checkpoint's keys, and a second trainer that loads it continues
    bit-identically (same parameters, Adam moments, step count and sampler state -> same next step)."""
    from soccernerfs_amd import checkpoint as CK
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    def make(seed):
        cfg = KPlanesTrainConfig(seed=seed, mlp_operands="fp32", multiscale_res=(1, 2), spacetime_resolution=(8, 8, 8, 4), feature_dim=8, proposal_feature_dim=8,
                            proposal_resolutions=((8, 8, 8, 4), (16, 16, 16, 4)), num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8,
                            sigma_net_hidden_dim=64, rgb_net_hidden_dim=64)
        return KPlanesTrainer(cfg, 256, device=torch.device(DEV))

    gen = torch.Generator(device=DEV).manual_seed(1)
    R = 256
    rays = {"origins": (torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1) * 0.5,
            "directions": torch.nn.functional.normalize(torch.rand(R, 3, device=DEV, generator=gen) * 2 - 1, dim=-1),
            "times": torch.rand(R, 1, device=DEV, generator=gen)}
    target = torch.rand(R, 3, device=DEV, generator=gen)
    S0, S1 = 32, 16

    def draws(seed):
        g2 = torch.Generator(device=DEV).manual_seed(seed)
        r = lambda *s: torch.rand(*s, device=DEV, generator=g2)
        return {"t_rand": r(R, S0 + 1), "u": [r(R, S1 + 1), r(R, 9)], "bg": r(R, 3)}

    a = make(0)
    for i in range(4):
        a.train_step(rays, target, draws(10 + i))
    path = a.save_checkpoint(str(tmp_path))
    assert os.path.basename(path) == "step-000000003.ckpt"
    saved = torch.load(path, map_location="cpu", weights_only=False)
    ref = torch.load(os.path.join(GOLD, "g15_step-000000002.ckpt"), map_location="cpu", weights_only=False)
    assert set(saved["pipeline"]) == set(ref["pipeline"])  # same model family and sizes as G15
    for k, v in ref["pipeline"].items():
        assert tuple(saved["pipeline"][k].shape) == tuple(v.shape), k
    for g in ("fields", "proposal_networks"):
        assert set(saved["optimizers"][g]["state"]) == set(ref["optimizers"][g]["state"])
    b = make(99)
    assert b.load_checkpoint(str(tmp_path)) == 4 and b.step == 4
    a.synchronize()
    assert torch.equal(a.params, b.params) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    b._steps_since_update = a._steps_since_update
    a.train_step(rays, target, draws(8))
    b.train_step(rays, target, draws(8))
    a.synchronize(), b.synchronize()
    torch.testing.assert_close(a.params, b.params, rtol=1e-5, atol=1e-7)  # atomic accumulation order differs run to run
    # a reference-written file loads too (params + Adam moments), and training continues finite
    import shutil
    d2 = tmp_path / "ref"
    d2.mkdir()
    shutil.copy(os.path.join(GOLD, "g15_step-000000002.ckpt"), d2 / "step-000000002.ckpt")
    c = make(5)
    assert c.load_checkpoint(str(d2)) == 3
    want = CK.import_optimizer_states(c._named_module(), ref["optimizers"])["field.grids.planes"][0]
    torch.testing.assert_close(c.mviews["field.planes"].cpu(), want.reshape(-1), rtol=0, atol=0)
    c.train_step(rays, target, draws(9))
    c.synchronize()
    assert bool(torch.isfinite(c.params).all())

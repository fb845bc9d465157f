"""The synthetic Broadcast-style scene (data plumbing of bench.py / tools/train_psnr.py) on the CPU: the analytic shader is deterministic, the
round-4 "textured" variant keeps the default scene where it adds nothing (sky, pitch lines) and adds structure elsewhere; frame times follow the
reference's parser formula (NS/data/dataparsers/broadcaststyle_dataparser.py:408-412)."""
import torch

from soccernerfs_amd import synthetic as S


def _rays(cam, H=54, W=96):
    cams = S.make_cameras(20, W, H)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    x = (xs + 0.5 - cams["cx"][cam]) / cams["fx"][cam]
    y = -(ys + 0.5 - cams["cy"][cam]) / cams["fy"][cam]
    d = torch.stack([x, y, -torch.ones_like(x)], -1).reshape(-1, 3) @ cams["c2w"][cam, :3, :3].T
    d = torch.nn.functional.normalize(d, dim=-1)
    return cams["c2w"][cam, :3, 3].expand_as(d).contiguous(), d


def test_cell_hash_is_a_fixed_function():
    ix, iy = torch.tensor([0.0, 1.0, -3.0, 117.0]), torch.tensor([0.0, 2.0, 5.0, -40.0])
    a, b = S._cell_hash(ix, iy, 3), S._cell_hash(ix, iy, 3)
    assert torch.equal(a, b) and float(a.min()) >= 0.0 and float(a.max()) < 1.0
    assert not torch.equal(a, S._cell_hash(ix, iy, 4))
    # known answers: a change of the mixing constants would silently change every committed textured-scene PSNR figure
    torch.testing.assert_close(a, torch.tensor([int(v) / 65536.0 for v in (a * 65536).round().tolist()]), rtol=0, atol=0)


def test_textured_variant_adds_structure_and_keeps_the_rest():
    o, d = _rays(5)
    t = torch.full((o.shape[0],), 0.3)
    base, tex = S.shade(o, d, t, "default"), S.shade(o, d, t, "textured")
    assert torch.equal(tex, S.shade(o, d, t, "textured"))  # deterministic
    assert float(base.min()) >= 0 and float(tex.max()) <= 1
    changed = (base - tex).abs().sum(-1) > 1e-6
    assert 0.3 < float(changed.float().mean()) < 1.0  # grass grain / board / crowd / extra players, but not the sky and not the lines
    assert bool((~changed).any())  # sky and pitch lines are the default scene's
    # more high-frequency content: the mean absolute horizontal difference grows by more than half (2.0x at this size)
    img = lambda c: c.view(54, 96, 3)
    assert (img(tex)[:, 1:] - img(tex)[:, :-1]).abs().mean() > 1.5 * (img(base)[:, 1:] - img(base)[:, :-1]).abs().mean()
    # the dynamic content moves: two times differ somewhere, in both variants
    for v in ("default", "textured"):
        assert not torch.equal(S.shade(o, d, t, v), S.shade(o, d, torch.full_like(t, 0.7), v))


def test_frame_times_follow_the_parser():
    t3, t4 = S.frame_times(100, 3), S.frame_times(100, 4)
    assert len(t3) == 33 and len(t4) == 25 and float(t3[0]) == 0.0 and float(t3[-1]) == 1.0 and float(t4[-1]) == 1.0
    assert bool((t3[1:] > t3[:-1]).all())

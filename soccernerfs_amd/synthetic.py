"""Synthetic "Broadcast-style" dynamic scene (the real dataset is not available offline).

Geometry constants follow the reference's Broadcast-style parser defaults:
20 pinhole cameras (19 train + 1 eval) on one side of the pitch, 1920x1080 downscaled by 2 -> 960x540,
100 frames at 25 fps subsampled by fps_downsample (3 -> 33 frames per camera, 627 train images),
camera translations scaled so max|t| = 1, scene_scale 1.5 => aabb [-1.5,1.5]^3, times = frame/max_frame in [0,1]
(NS/data/dataparsers/broadcaststyle_dataparser.py:166-232,408-480).  Content is an analytic scene ray-cast on
the GPU with torch ops (variants "default" / "textured"; "stadium" = the stadium-players scene of config 4, below): a static textured ground plane plus three "players" (two stacked spheres each, a few pixels wide,
moving a few body widths during the clip) and a ball on a parabolic arc.  Together they cover ~0.5 % of a frame and their
motion sweeps ~2 % of the pixels over the clip, so temporal-difference (IST) maps are sparse as in the real footage
(REF/data/README.md:17; the round-1 scene swept 25 %).  Data plumbing only -- not part of the measured hot path.
"""
import math
from typing import Dict

import torch


def make_cameras(n_cams: int = 20, width: int = 960, height: int = 540, device="cpu") -> Dict[str, torch.Tensor]:
    c2w = []
    for i in range(n_cams):
        ang = math.radians(-60 + 120 * i / max(n_cams - 1, 1))  # an arc along one touchline
        pos = torch.tensor([math.sin(ang) * 1.0, -math.cos(ang) * 1.0, 0.35 + 0.15 * (i % 3)])
        target = torch.tensor([0.15 * math.sin(3 * i), 0.1 * math.cos(2 * i), 0.0])
        fwd = torch.nn.functional.normalize(target - pos, dim=0)
        right = torch.nn.functional.normalize(torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])), dim=0)
        up = torch.linalg.cross(right, fwd)
        c2w.append(torch.cat([torch.stack([right, up, -fwd], dim=1), pos[:, None]], dim=1))
    c2w = torch.stack(c2w).float()
    c2w[:, :, 3] /= c2w[:, :, 3].abs().max()  # auto-scale translations to max|t| = 1 (parser :433-438)
    focal = 0.9 * width
    return {"c2w": c2w.to(device), "fx": torch.full((n_cams,), focal, device=device), "fy": torch.full((n_cams,), focal, device=device),
            "cx": torch.full((n_cams,), width / 2.0, device=device), "cy": torch.full((n_cams,), height / 2.0, device=device),
            "width": width, "height": height}


def make_novel_cameras(n: int = 3, width: int = 960, height: int = 540, n_train_cams: int = 20, device="cpu") -> Dict[str, torch.Tensor]:
    """Evaluation-only cameras BETWEEN the training cameras of make_cameras (half-way angles, mid height): interpolated novel views,
    rendered from the analytic scene.  They share make_cameras' translation scale so that both sets live in one world frame."""
    ref = make_cameras(n_train_cams, width, height)
    raw_max = max(max(abs(math.sin(math.radians(-60 + 120 * i / (n_train_cams - 1)))), abs(math.cos(math.radians(-60 + 120 * i / (n_train_cams - 1)))),
                      0.35 + 0.15 * (i % 3)) for i in range(n_train_cams))
    c2w = []
    for j in range(n):
        i = (j + 0.5) * (n_train_cams - 1) / n  # between two training cameras
        i = math.floor(i) + 0.5
        ang = math.radians(-60 + 120 * i / (n_train_cams - 1))
        pos = torch.tensor([math.sin(ang) * 1.0, -math.cos(ang) * 1.0, 0.42]) / raw_max
        target = torch.tensor([0.1 * math.sin(2 * j + 1), 0.08 * math.cos(3 * j), 0.0])
        fwd = torch.nn.functional.normalize(target - pos, dim=0)
        right = torch.nn.functional.normalize(torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])), dim=0)
        up = torch.linalg.cross(right, fwd)
        c2w.append(torch.cat([torch.stack([right, up, -fwd], dim=1), pos[:, None]], dim=1))
    c2w = torch.stack(c2w).float()
    out = {k: (v[:1].expand(n).clone() if torch.is_tensor(v) and v.dim() == 1 else v) for k, v in ref.items()}
    out["c2w"] = c2w.to(device)
    for k in ("fx", "fy", "cx", "cy"):
        out[k] = out[k].to(device)
    return out


def frame_times(n_frames: int = 100, fps_downsample: int = 3) -> torch.Tensor:
    """linspace(0, n-1, int(n/f)) frame ids -> time = frame / max_frame (parser :408-412,:476-479)."""
    ids = torch.linspace(0, n_frames - 1, int(n_frames / fps_downsample)).long()
    return ids.float() / float(n_frames - 1)


def _sphere_hit(o, d, c, r):
    oc = o - c
    b = (oc * d).sum(-1)
    disc = b * b - ((oc * oc).sum(-1) - r * r)
    t = -b - torch.sqrt(disc.clamp_min(0))
    return torch.where((disc > 0) & (t > 0), t, torch.full_like(t, float("inf")))


def _cell_hash(ix: torch.Tensor, iy: torch.Tensor, salt: int) -> torch.Tensor:
    """Integer lattice -> pseudo-random value in [0,1) (a fixed integer mix: the same on every device)."""
    h = (ix.long() * 374761393 + iy.long() * 668265263 + salt * 2246822519) & 0xFFFFFFFF
    h = ((h ^ (h >> 13)) * 1274126177) & 0xFFFFFFFF
    return ((h ^ (h >> 16)) & 0xFFFF).float() / 65536.0


def shade(o: torch.Tensor, d: torch.Tensor, time: torch.Tensor, variant: str = "default") -> torch.Tensor:
    """Analytic colour in [0,1] for rays (o,d [N,3], unit d) at times [N].  variant "textured" (round 4; PSNR studies on content that does
    not saturate): the same pitch with fine grass / wear texture on it, an advertising board and a crowd stand behind the far touchline,
    and ten players instead of three."""
    if variant == "stadium":
        return shade_stadium(o, d, time)
    textured = variant == "textured"
    assert variant in ("default", "textured")
    N = o.shape[0]
    sky = torch.stack([0.55 + 0.2 * d[:, 2], 0.7 + 0.15 * d[:, 2], 0.95 * torch.ones_like(d[:, 2])], -1).clamp(0, 1)
    # ground plane z = -0.1: mown-stripe pitch + lines
    tz = (-0.1 - o[:, 2]) / torch.where(d[:, 2].abs() < 1e-6, torch.full_like(d[:, 2], -1e-6), d[:, 2])
    tg = torch.where(tz > 0, tz, torch.full_like(tz, float("inf")))
    pg = o + d * tg.clamp(max=1e4)[:, None]
    stripe = ((pg[:, 0] * 6).floor() % 2)
    green = torch.stack([0.12 + 0.05 * stripe, 0.45 + 0.12 * stripe, 0.15 + 0.03 * stripe], -1)
    line = ((pg[:, 0].abs() - 1.0).abs() < 0.01) | ((pg[:, 1].abs() - 0.65).abs() < 0.01) | (pg[:, 0].abs() < 0.008)
    if textured:  # grass grain (three octaves of a lattice hash, bilinear in the coarsest), worn patches in front of the goals
        gx, gy = pg[:, 0].clamp(-2, 2), pg[:, 1].clamp(-2, 2)
        grain = sum(a * (_cell_hash((gx * f).floor(), (gy * f).floor(), k) - 0.5) for k, (f, a) in enumerate(((40.0, 0.10), (110.0, 0.08), (300.0, 0.05))))
        wear = torch.exp(-(((gx.abs() - 0.85) / 0.12) ** 2 + (gy / 0.2) ** 2))
        green = (green * (1.0 + grain[:, None]) * (1 - 0.45 * wear[:, None]) + wear[:, None] * torch.tensor([0.20, 0.15, 0.08], device=o.device)).clamp(0, 1)
    green = torch.where(line[:, None], torch.ones_like(green) * 0.92, green)
    inside = (pg[:, 0].abs() < 1.45) & (pg[:, 1].abs() < 1.45)
    col = torch.where((torch.isfinite(tg) & inside)[:, None], green, sky)
    depth = torch.where(torch.isfinite(tg) & inside, tg, torch.full_like(tg, float("inf")))
    if textured:  # far side (the cameras sit at y < 0): advertising board y = 0.8, z in [-0.1, -0.04]; crowd stand y = 1.2, z in [-0.1, 0.5]
        for y_w, z_hi, kind in ((1.2, 0.5, "crowd"), (0.8, -0.04, "board")):
            tw = (y_w - o[:, 1]) / torch.where(d[:, 1].abs() < 1e-6, torch.full_like(d[:, 1], 1e-6), d[:, 1])
            pw = o + d * tw.clamp(min=0, max=1e4)[:, None]
            hit = (tw > 0) & (tw < depth) & (pw[:, 0].abs() < 1.45) & (pw[:, 2] > -0.1) & (pw[:, 2] < z_hi)
            if kind == "crowd":  # seats: 1.5 cm cells of three hashed colour channels, darker towards the top
                ix, iz = (pw[:, 0].clamp(-2, 2) * 66).floor(), (pw[:, 2].clamp(-1, 1) * 66).floor()
                c = torch.stack([_cell_hash(ix, iz, 11 + k) for k in range(3)], -1) * 0.7 + 0.15
                c = c * (1.0 - 0.5 * ((pw[:, 2] + 0.1) / 0.6).clamp(0, 1))[:, None]
            else:  # board: 0.29-wide panels of a hashed base colour with vertical bars (lettering) in the complementary one
                panel = (pw[:, 0].clamp(-2, 2) * 3.5).floor()
                base = torch.stack([_cell_hash(panel, torch.zeros_like(panel), 31 + k) for k in range(3)], -1)
                bars = (_cell_hash((pw[:, 0].clamp(-2, 2) * 90).floor(), ((pw[:, 2] + 0.1) * 50).floor(), 7) > 0.55).float()
                c = base * (1 - bars[:, None]) + (1 - base) * bars[:, None]
            col = torch.where(hit[:, None], c, col)
            depth = torch.where(hit, tw, depth)
    # dynamic content: three players (torso + head) that move a few body widths during the clip, and a ball on a parabolic arc
    tt = time
    z0 = torch.full_like(tt, -0.08)
    head_up = torch.tensor([0.0, 0.0, 0.028], device=o.device)
    players = (
        (torch.stack([0.25 + 0.07 * tt, -0.20 + 0.04 * torch.sin(3.14 * tt), z0], -1), (0.85, 0.1, 0.1)),
        (torch.stack([-0.35 + 0.05 * torch.cos(3.14 * tt), 0.15 + 0.06 * tt, z0], -1), (0.1, 0.15, 0.8)),
        (torch.stack([-0.05 - 0.06 * tt, 0.35 - 0.05 * tt * tt, z0], -1), (0.95, 0.85, 0.1)),
    )
    if textured:  # seven more players on deterministic tracks (straight runs and arcs of a few body widths)
        kits = ((0.85, 0.1, 0.1), (0.1, 0.15, 0.8), (0.95, 0.85, 0.1), (0.9, 0.9, 0.9), (0.1, 0.1, 0.1), (0.9, 0.4, 0.05), (0.5, 0.1, 0.6))
        for k in range(7):
            x0, y0 = -0.9 + 0.27 * k, 0.45 * math.sin(1.7 * k + 0.4)
            players = players + ((torch.stack([x0 + 0.08 * math.cos(k) * tt + 0.02 * torch.sin(6.28 * tt + k), y0 + 0.08 * math.sin(k) * tt, z0], -1), kits[k]),)
    ball_c = torch.stack([-0.25 + 0.4 * tt, 0.08 * torch.sin(6.28 * tt), -0.09 + 0.5 * tt * (1 - tt)], -1)
    spheres = [(ball_c, 0.008, (0.95, 0.95, 0.9))]
    for body_c, shirt in players:
        spheres.append((body_c, 0.02, shirt))
        spheres.append((body_c + head_up, 0.01, (0.9, 0.75, 0.6)))
    for c, r, rgb in spheres:
        t = _sphere_hit(o, d, c, r)
        hit = t < depth
        n = torch.nn.functional.normalize(o + d * t.clamp(max=1e4)[:, None] - c, dim=-1)
        lam = (n * torch.tensor([0.3, -0.4, 0.85], device=o.device)).sum(-1).clamp(0.15, 1.0)
        sc = torch.tensor(rgb, device=o.device)[None, :] * lam[:, None]
        col = torch.where(hit[:, None], sc, col)
        depth = torch.where(hit, t, depth)
    return col.clamp(0, 1)


# ---- "stadium-players" (BASELINE.json configs[3]; REF/data/README.md:21-25, NS/data/dataparsers/stadiumwide_dataparser.py:94-112) ----
# "several players and balls interacting all over the field, captured by 30 wide-angle cameras placed high up in the bleachers ... much more
# distant from the field.  Six additional cameras, used exclusively for evaluation, are placed near the players."  scene_scale 1.0 => aabb
# [-1,1]^3, fps_downsample 1 => every one of the 100 frames, camera translations auto-scaled to max|t| = 1.
STADIUM_GROUND_Z = -0.06
_ST_PX, _ST_PY = 0.72, 0.46      # pitch half-extents
_ST_BX, _ST_BY = 0.78, 0.54      # where the bleachers start to rise
_ST_SLOPE = 1.6


def make_stadium_cameras(n_train: int = 30, n_eval: int = 6, width: int = 960, height: int = 540, device="cpu") -> Dict[str, torch.Tensor]:
    """Cameras 0 .. n_train-1: wide-angle (hfov ~78 deg), on an ellipse high in the bleachers all around the pitch; cameras n_train ..: the
    evaluation-only ones near the players (hfov ~56 deg).  One world frame: translations divided by the training cameras' max|t|."""
    c2w, focal = [], []
    look = lambda pos, target: (lambda fwd: (lambda right: torch.cat([torch.stack([right, torch.linalg.cross(right, fwd), -fwd], dim=1), pos[:, None]], dim=1))(
        torch.nn.functional.normalize(torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])), dim=0)))(torch.nn.functional.normalize(target - pos, dim=0))
    for i in range(n_train):
        a = 2 * math.pi * i / n_train
        pos = torch.tensor([1.0 * math.cos(a), 0.80 * math.sin(a), 0.44 + 0.05 * (i % 3)])
        c2w.append(look(pos, torch.tensor([0.12 * math.cos(3 * i), 0.10 * math.sin(2 * i), STADIUM_GROUND_Z])))
        focal.append(0.62 * width)
    for j in range(n_eval):
        a = 2 * math.pi * (j + 0.5) / n_eval
        pos = torch.tensor([0.40 * math.cos(a), 0.30 * math.sin(a), 0.05 + 0.03 * (j % 2)])
        c2w.append(look(pos, torch.tensor([0.10 * math.cos(a + 2.0), 0.08 * math.sin(a + 2.0), STADIUM_GROUND_Z + 0.02])))
        focal.append(0.95 * width)
    c2w = torch.stack(c2w).float()
    c2w[:, :, 3] /= c2w[:n_train, :, 3].abs().max()  # auto_scale_poses (stadiumwide_dataparser.py:102): max|t| of the TRAINING poses = 1
    n = n_train + n_eval
    f = torch.tensor(focal, dtype=torch.float32, device=device)
    return {"c2w": c2w.to(device), "fx": f, "fy": f.clone(), "cx": torch.full((n,), width / 2.0, device=device), "cy": torch.full((n,), height / 2.0, device=device),
            "width": width, "height": height, "n_train": n_train, "n_eval": n_eval}


def stadium_tracks(tt: torch.Tensor):
    """Centres of the 14 players' bodies and of the 2 balls at times tt [N] -> ([14][N,3], [2][N,3]).  Deterministic runs, arcs and passes
    spread over the whole pitch; a ball travels between two players on a parabola."""
    z0 = torch.full_like(tt, STADIUM_GROUND_Z + 0.010)
    players = []
    for k in range(14):
        x0 = -0.60 + 0.092 * k + 0.05 * math.sin(2.3 * k)
        y0 = 0.34 * math.sin(1.7 * k + 0.4)
        vx, vy = 0.10 * math.cos(1.3 * k), 0.08 * math.sin(0.9 * k + 1.0)
        players.append(torch.stack([x0 + vx * tt + 0.015 * torch.sin(6.28 * tt + k), y0 + vy * tt + 0.012 * torch.cos(5.0 * tt + 2 * k), z0], -1))
    balls = []
    for b, (pa, pb, h) in enumerate(((2, 9, 0.10), (11, 5, 0.06))):
        s = (tt * (1.0 + 0.5 * b)) % 1.0
        c = players[pa] * (1 - s)[:, None] + players[pb] * s[:, None]
        balls.append(torch.stack([c[:, 0], c[:, 1], STADIUM_GROUND_Z + 0.004 + 4 * h * s * (1 - s)], -1))
    return players, balls


def shade_stadium(o: torch.Tensor, d: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
    dev = o.device
    sky = torch.stack([0.50 + 0.2 * d[:, 2], 0.66 + 0.15 * d[:, 2], 0.93 * torch.ones_like(d[:, 2])], -1).clamp(0, 1)
    inf = torch.full_like(d[:, 0], float("inf"))
    # ground: pitch (mown stripes, lines) inside the apron
    tz = (STADIUM_GROUND_Z - o[:, 2]) / torch.where(d[:, 2].abs() < 1e-6, torch.full_like(d[:, 2], -1e-6), d[:, 2])
    tg = torch.where(tz > 0, tz, inf)
    pg = o + d * tg.clamp(max=1e4)[:, None]
    on_ground = torch.isfinite(tg) & (pg[:, 0].abs() <= _ST_BX) & (pg[:, 1].abs() <= _ST_BY)
    stripe = ((pg[:, 0] * 9).floor() % 2)
    gx, gy = pg[:, 0].clamp(-1, 1), pg[:, 1].clamp(-1, 1)
    grain = 0.08 * (_cell_hash((gx * 90).floor(), (gy * 90).floor(), 3) - 0.5) + 0.05 * (_cell_hash((gx * 260).floor(), (gy * 260).floor(), 4) - 0.5)
    green = torch.stack([0.11 + 0.05 * stripe, 0.43 + 0.12 * stripe, 0.14 + 0.03 * stripe], -1) * (1.0 + grain[:, None])
    line = (((pg[:, 0].abs() - _ST_PX).abs() < 0.004) & (pg[:, 1].abs() < _ST_PY)) | (((pg[:, 1].abs() - _ST_PY).abs() < 0.004) & (pg[:, 0].abs() < _ST_PX)) \
        | ((pg[:, 0].abs() < 0.003) & (pg[:, 1].abs() < _ST_PY)) | (((pg[:, 0] ** 2 + pg[:, 1] ** 2).sqrt() - 0.09).abs() < 0.003)
    apron = (pg[:, 0].abs() > _ST_PX + 0.02) | (pg[:, 1].abs() > _ST_PY + 0.02)
    ground = torch.where(line[:, None], torch.full_like(green, 0.93), torch.where(apron[:, None], torch.tensor([0.45, 0.33, 0.25], device=dev).expand_as(green), green))
    col = torch.where(on_ground[:, None], ground.clamp(0, 1), sky)
    depth = torch.where(on_ground, tg, inf)
    # bleachers: four planes rising outward from the apron; seats = hashed cells, darker with height
    for axis, sign in ((0, 1.0), (0, -1.0), (1, 1.0), (1, -1.0)):
        b0 = _ST_BX if axis == 0 else _ST_BY
        nrm = torch.zeros(3, device=dev)
        nrm[axis], nrm[2] = -_ST_SLOPE * sign, 1.0
        dd = STADIUM_GROUND_Z - _ST_SLOPE * b0
        den = (d * nrm).sum(-1)
        ts = (dd - (o * nrm).sum(-1)) / torch.where(den.abs() < 1e-6, torch.full_like(den, 1e-6), den)
        ps = o + d * ts.clamp(min=0, max=1e4)[:, None]
        along, across = ps[:, 1 - axis], ps[:, axis] * sign
        # the corner belongs to the plane that is higher there
        other_h = STADIUM_GROUND_Z + _ST_SLOPE * (along.abs() - (_ST_BY if axis == 0 else _ST_BX))
        hit = (ts > 1e-4) & (ts < depth) & (across >= b0) & (across <= 1.08) & (along.abs() <= 1.08) & (ps[:, 2] >= other_h - 1e-4)
        ia, ih = (along.clamp(-2, 2) * 120).floor(), ((ps[:, 2] - STADIUM_GROUND_Z).clamp(0, 1) * 120).floor()
        seat = torch.stack([_cell_hash(ia, ih, 11 + k + 5 * axis + (1 if sign > 0 else 0)) for k in range(3)], -1) * 0.65 + 0.15
        seat = seat * (1.0 - 0.45 * ((ps[:, 2] - STADIUM_GROUND_Z) / 0.45).clamp(0, 1))[:, None]
        col = torch.where(hit[:, None], seat, col)
        depth = torch.where(hit, ts, depth)
    players, balls = stadium_tracks(time)
    kits = ((0.85, 0.1, 0.1), (0.1, 0.15, 0.8), (0.95, 0.85, 0.1), (0.92, 0.92, 0.92), (0.08, 0.08, 0.08), (0.9, 0.4, 0.05), (0.5, 0.1, 0.6))
    head_up = torch.tensor([0.0, 0.0, 0.013], device=dev)
    spheres = [(c, 0.0035, (0.96, 0.96, 0.92)) for c in balls]
    for k, c in enumerate(players):
        spheres.append((c, 0.009, kits[k % 7]))
        spheres.append((c + head_up, 0.0045, (0.9, 0.75, 0.6)))
    light = torch.tensor([0.3, -0.4, 0.85], device=dev)
    for c, r, rgb in spheres:
        t = _sphere_hit(o, d, c, r)
        hit = t < depth
        n = torch.nn.functional.normalize(o + d * t.clamp(max=1e4)[:, None] - c, dim=-1)
        lam = (n * light).sum(-1).clamp(0.15, 1.0)
        col = torch.where(hit[:, None], torch.tensor(rgb, device=dev)[None, :] * lam[:, None], col)
        depth = torch.where(hit, t, depth)
    return col.clamp(0, 1)


def render_dataset(cams: Dict[str, torch.Tensor], times: torch.Tensor, cam_ids, device, chunk_rows: int = 135, variant: str = "default") -> Dict[str, torch.Tensor]:
    """uint8 images [M,H,W,3] on `device` for every (camera in cam_ids) x (time), plus per-image camera tables
    (one 'camera' per image, as nerfstudio's Cameras object holds them: c2w/intrinsics repeated per frame + times)."""
    from . import ops

    H, W = cams["height"], cams["width"]
    M = len(cam_ids) * len(times)
    imgs = torch.empty(M, H, W, 3, dtype=torch.uint8, device=device)
    tab = {k: [] for k in ("c2w", "fx", "fy", "cx", "cy", "times", "cam_id")}
    m = 0
    for c in cam_ids:
        for t in times.tolist():
            for k in ("c2w", "fx", "fy", "cx", "cy"):
                tab[k].append(cams[k][c])
            tab["times"].append(t)
            tab["cam_id"].append(c)
            m += 1
    table = {k: torch.stack(v).to(device).contiguous() for k, v in tab.items() if k not in ("times", "cam_id")}
    table["times"] = torch.tensor(tab["times"], dtype=torch.float32, device=device)
    table["cam_id"] = torch.tensor(tab["cam_id"], dtype=torch.int64, device=device)
    xs = torch.arange(W, device=device)
    for m in range(M):
        for r0 in range(0, H, chunk_rows):
            rows = torch.arange(r0, min(r0 + chunk_rows, H), device=device)
            yy, xx = torch.meshgrid(rows, xs, indexing="ij")
            idx = torch.stack([torch.full_like(yy, m), yy, xx], -1).reshape(-1, 3)
            rays = ops.generate_rays(idx, table["fx"], table["fy"], table["cx"], table["cy"], table["c2w"], table["times"])
            col = shade(rays["origins"], rays["directions"], rays["times"][:, 0], variant)
            imgs[m, r0:r0 + rows.numel()] = (col.view(rows.numel(), W, 3) * 255.0 + 0.5).to(torch.uint8)
    return {"images": imgs, **table, "width": W, "height": H}

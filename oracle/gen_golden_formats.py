"""G14 / G15: on-disk formats, captured from the reference's own code.

G14  Broadcaststyle._generate_dataparser_outputs (NS/data/dataparsers/broadcaststyle_dataparser.py:261-527) run on two small synthetic
     `transforms.json` trees (global intrinsics / per-frame intrinsics, missing files, a camera outside the split, masks, fps
     down-sampling 1 and 3): file selection, poses after auto-scaling, intrinsics after the down-scale, times, ids, scene box.
     -> tests/golden/g14_dataparser.json (inputs: the json text and the list of files that exist; outputs as lists).
G14b the same for Stadiumwide (NS/data/dataparsers/stadiumwide_dataparser.py) -> tests/golden/g14b_stadiumwide.json.
G14c the same for Stadium (NS/data/dataparsers/stadium_dataparser.py) -> tests/golden/g14c_stadium.json.
G15  a checkpoint holding what the reference's Trainer.save_checkpoint (NS/engine/trainer.py:353-380) saves -- the dict
     {"step", "pipeline": pipeline.state_dict(), "optimizers": {group: Adam.state_dict()}, "scalers": GradScaler.state_dict()} --
     for the reference's own small KPlanesModel and torch.optim.Adam objects after three steps, plus that model's eval-mode outputs
     on fixed rays and the Adam moments by parameter name.  (The Trainer module itself does not import here -- it pulls the viewer
     and its tornado / websocket dependencies -- so the ten-line writer is restated; every tensor, name and ordering in the file comes
     from the reference's objects.)
     -> tests/golden/g15_step-000000002.ckpt (the file, as written) and tests/golden/g15_checkpoint.npz.

    python oracle/gen_golden_formats.py        # build container only

TEST INFRASTRUCTURE ONLY (header as oracle/_refimport.py)."""
import json
import os
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle._refimport import import_reference  # noqa: E402
from oracle.gen_golden import npy  # noqa: E402

import_reference()


def _pose(gen):
    m = torch.eye(4)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=gen))
    m[:3, :3] = q
    m[:3, 3] = torch.randn(3, generator=gen) * 40.0
    return [[float(v) for v in row] for row in m]


def make_tree(root: Path, per_frame: bool, gen):
    cams = ["Camera_1", "Camera_2", "Camera_3", "Camera_20", "global_1"]
    T = 7
    meta = {"frames": []}
    intr = {"fl_x": 1000.0, "fl_y": 1010.0, "cx": 480.0, "cy": 270.5, "w": 960, "h": 541}
    if not per_frame:
        meta.update(intr)
        meta["k1"] = 0.01
    existing = []
    for c in cams:
        pose = _pose(gen)
        for t in range(T):
            fr = {"file_path": f"images/{c}_{t:04d}.png", "transform_matrix": pose}
            if per_frame:
                fr.update({k: (v + (1.0 if k.startswith("fl") else 0) * cams.index(c)) for k, v in intr.items()})
                fr["k1"] = 0.001 * t
                fr["mask_path"] = f"masks/{c}_{t:04d}.png"
                fr["depth_file_path"] = f"depth-maps/{c}_{t:04d}.png"
            meta["frames"].append(fr)
            if not (c == "Camera_2" and t == 3):  # one missing file
                existing.append(f"images/2x/{c}_{t:04d}.png")
    (root / "images" / "2x").mkdir(parents=True)
    for f in existing:
        (root / f).touch()
    text = json.dumps(meta)
    (root / "transforms.json").write_text(text)
    return text, existing


def g14():
    from nerfstudio.data.dataparsers.broadcaststyle_dataparser import BroadcaststyleDataParserConfig

    gen = torch.Generator().manual_seed(3)
    cases = []
    for per_frame, fps, extra in ((False, 3.0, {}), (True, 1.0, {}), (True, 2.0, {}),
                                  (True, 1.0, {"depth_maps": "depth-maps", "depth_mask": "mask"}),
                                  (True, 1.0, {"depth_maps": "depth-maps_field", "depth_mask": "none", "static": True, "static_timestep": 2,
                                               "cap_box_floor": True, "scene_scale": 2.0, "scale_factor": 0.5, "auto_scale_poses": False}),
                                  (False, 1.0, {"orientation_method": "up", "center_method": "poses"})):
        tmp = Path(tempfile.mkdtemp())
        try:
            text, existing = make_tree(tmp, per_frame, gen)
            case = {"transforms": text, "existing": existing, "fps_downsample": fps, "options": extra, "splits": {}}
            for split in ("train", "val"):
                cfg = BroadcaststyleDataParserConfig(data=tmp, fps_downsample=fps, **extra)
                out = cfg.setup().get_dataparser_outputs(split)
                cam = out.cameras
                case["splits"][split] = {
                    "image_filenames": [str(Path(f).relative_to(tmp)) for f in out.image_filenames],
                    "mask_filenames": None if out.mask_filenames is None else [str(Path(f).relative_to(tmp)) for f in out.mask_filenames],
                    "c2w": cam.camera_to_worlds.tolist(), "fx": cam.fx.flatten().tolist(), "fy": cam.fy.flatten().tolist(),
                    "cx": cam.cx.flatten().tolist(), "cy": cam.cy.flatten().tolist(), "height": cam.height.flatten().tolist(),
                    "width": cam.width.flatten().tolist(), "times": cam.times.flatten().tolist(), "ids": cam.ids.flatten().tolist(),
                    "depth_filenames": None if out.metadata["depth_filenames"] is None else [str(Path(f).relative_to(tmp)) for f in out.metadata["depth_filenames"]],
                    "static": bool(out.metadata["static"]),
                    "distortion": cam.distortion_params.tolist(), "aabb": out.scene_box.aabb.tolist(), "scale": out.dataparser_scale,
                    "transform": out.dataparser_transform.tolist()}
            cases.append(case)
        finally:
            shutil.rmtree(tmp)
    path = os.path.join(ROOT, "tests", "golden", "g14_dataparser.json")
    json.dump(cases, open(path, "w"))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", [(len(c["splits"]["train"]["image_filenames"]), len(c["splits"]["val"]["image_filenames"])) for c in cases])


def g14b():
    """Stadiumwide._generate_dataparser_outputs (NS/data/dataparsers/stadiumwide_dataparser.py) on a synthetic tree: ring cameras of three
    groups + two close-up cameras, 4 time steps, per-frame intrinsics."""
    from nerfstudio.data.dataparsers.stadiumwide_dataparser import StadiumwideDataParserConfig

    gen = torch.Generator().manual_seed(9)
    names = [f"Ext Left-Left-{i}" for i in (0, 3, 9)] + [f"Middle-Right-{i}" for i in (1, 5)] + ["Ext Op Left-High Behind Left-9", "Center", "Shooter"]
    cases = []
    for extra in ({"nb_train_cameras": 110}, {"nb_train_cameras": 12, "closeup_training": True, "fps_downsample": 2.0}):
        tmp = Path(tempfile.mkdtemp())
        try:
            meta = {"frames": []}
            existing = []
            for ci, c in enumerate(names):
                pose = _pose(gen)
                for t in range(4):
                    meta["frames"].append({"file_path": f"images/{c}_{t:04d}.png", "transform_matrix": pose, "fl_x": 900.0 + ci, "fl_y": 905.0, "cx": 480.0,
                                           "cy": 270.0, "w": 960, "h": 540})
                    existing.append(f"images/2x/{c}_{t:04d}.png")
            (tmp / "images" / "2x").mkdir(parents=True)
            for f in existing:
                (tmp / f).touch()
            text = json.dumps(meta)
            (tmp / "transforms.json").write_text(text)
            case = {"transforms": text, "existing": existing, "options": extra, "splits": {}}
            for split in ("train", "val"):
                out = StadiumwideDataParserConfig(data=tmp, **extra).setup().get_dataparser_outputs(split)
                cam = out.cameras
                case["splits"][split] = {"image_filenames": [str(Path(f).relative_to(tmp)) for f in out.image_filenames],
                                         "c2w": cam.camera_to_worlds.tolist(), "fx": cam.fx.flatten().tolist(), "times": cam.times.flatten().tolist(),
                                         "ids": cam.ids.flatten().tolist(), "aabb": out.scene_box.aabb.tolist(), "scale": out.dataparser_scale,
                                         "height": cam.height.flatten().tolist(), "width": cam.width.flatten().tolist()}
            cases.append(case)
        finally:
            shutil.rmtree(tmp)
    path = os.path.join(ROOT, "tests", "golden", "g14b_stadiumwide.json")
    json.dump(cases, open(path, "w"))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", [(len(c["splits"]["train"]["image_filenames"]), len(c["splits"]["val"]["image_filenames"])) for c in cases])


def g14c():
    """Stadium._generate_dataparser_outputs (NS/data/dataparsers/stadium_dataparser.py: the parser the nerfplayer presets name) on synthetic
    trees: `images_2/<group>-<camera>_<time>.png`, pose orientation "up" + centring "poses" (its defaults), 95 % camera split."""
    from nerfstudio.data.dataparsers.stadium_dataparser import StadiumDataParserConfig

    gen = torch.Generator().manual_seed(13)
    cases = []
    for names, extra, splits in (([f"Ext Left-Left-{i}" for i in range(10)] + [f"Left-Middle-{i}" for i in range(10)], {}, ("train", "val")),
                                 (["Ext Left-Left-0", "Ext Left-Left-3", "Middle-Right-5", "Ext Op Left-High Behind Left-9"],
                                  {"train_split_percentage": 0.5, "orientation_method": "none", "center_method": "none"}, ("train",))):
        tmp = Path(tempfile.mkdtemp())
        try:
            meta = {"frames": [], "fl_x": 800.0, "fl_y": 805.0, "cx": 480.0, "cy": 270.0, "w": 960, "h": 540}
            existing = []
            for c in names:
                pose = _pose(gen)
                for t in range(3):
                    meta["frames"].append({"file_path": f"images/{c}_{t:04d}.png", "transform_matrix": pose, "depth_file_path": f"depths/{c}_{t:04d}.png"})
                    existing.append(f"images_2/{c}_{t:04d}.png")
            (tmp / "images_2").mkdir(parents=True)
            for f in existing:
                (tmp / f).touch()
            text = json.dumps(meta)
            (tmp / "transforms.json").write_text(text)
            case = {"transforms": text, "existing": existing, "options": extra, "splits": {}}
            for split in splits:
                out = StadiumDataParserConfig(data=tmp, **extra).setup().get_dataparser_outputs(split)
                cam = out.cameras
                case["splits"][split] = {"image_filenames": [str(Path(f).relative_to(tmp)) for f in out.image_filenames],
                                         "depth_filenames": [str(Path(f).relative_to(tmp)) for f in out.metadata["depth_filenames"]],
                                         "c2w": cam.camera_to_worlds.tolist(), "fx": cam.fx.flatten().tolist(), "times": cam.times.flatten().tolist(),
                                         "ids": cam.ids.flatten().tolist(), "aabb": out.scene_box.aabb.tolist(), "scale": out.dataparser_scale,
                                         "transform": out.dataparser_transform.tolist()}
            cases.append(case)
        finally:
            shutil.rmtree(tmp)
    path = os.path.join(ROOT, "tests", "golden", "g14c_stadium.json")
    json.dump(cases, open(path, "w"))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", [{k: len(v["image_filenames"]) for k, v in c["splits"].items()} for c in cases])


def g15():
    import nerfstudio.models.kplanes as km
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.data.scene_box import SceneBox

    km.DynMetric = lambda *a, **k: None
    torch.manual_seed(5)
    gen = torch.Generator().manual_seed(5)
    mc = km.KPlanesModelConfig(multiscale_res=(1, 2), spacetime_resolution=(8, 8, 8, 4), feature_dim=8,
                               proposal_net_args_list=[{"feature_dim": 8, "resolution": [8, 8, 8, 4]}, {"feature_dim": 8, "resolution": [16, 16, 16, 4]}],
                               num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8, disable_viewing_dependent=True)
    model = km.KPlanesModel(mc, scene_box=SceneBox(aabb=torch.tensor([[-1.5] * 3, [1.5] * 3])), num_train_data=4)
    groups = model.get_param_groups()
    opts = {k: torch.optim.Adam(v, lr=1e-2, eps=1e-12) for k, v in groups.items()}
    R = 24
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.5
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    times = torch.rand(R, 1, generator=gen)
    rb = lambda: RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1), camera_indices=torch.zeros(R, 1, dtype=torch.long), times=times)
    target = torch.rand(R, 3, generator=gen)
    model.train()
    for step in range(3):
        for op in opts.values():
            op.zero_grad()
        out = model(rb())
        md = model.get_metrics_dict(out, {"image": target})
        ld = model.get_loss_dict(out, {"image": target}, md)
        sum(ld.values()).backward()
        for op in opts.values():
            op.step()
        model.proposal_sampler.step_cb(step)
    # trainer.py:364-374; pipeline.state_dict() prefixes the model with `_model.` (base_pipeline.py:109-113: VanillaPipeline._model)
    dst = os.path.join(ROOT, "tests", "golden", "g15_step-000000002.ckpt")
    torch.save({"step": 2, "pipeline": {"_model." + k: v for k, v in model.state_dict().items()},
                "optimizers": {k: v.state_dict() for k, v in opts.items()}, "scalers": torch.cuda.amp.GradScaler(enabled=False).state_dict()}, dst)
    model.eval()
    with torch.no_grad():
        out = model(rb())
    g = {"origins": o, "directions": d, "times": times, "rgb": out["rgb"], "accumulation": out["accumulation"], "depth": out["depth"]}
    names = {}
    for grp, params in groups.items():
        pnames = {id(p): n for n, p in model.named_parameters()}
        for i, p in enumerate(params):
            st = opts[grp].state.get(p, {})
            if "exp_avg" in st:
                key = f"{grp}/{i}/{pnames[id(p)]}"
                g["m_" + key] = st["exp_avg"]
                g["v_" + key] = st["exp_avg_sq"]
                names[key] = 1
    g["moment_keys"] = np.array(sorted(names))
    path = os.path.join(ROOT, "tests", "golden", "g15_checkpoint.npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in g.items()})
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB and", path, os.path.getsize(path) // 1024, "KiB;", len(names), "moment tensors")


if __name__ == "__main__":
    g14()
    g14b()
    g14c()
    g15()

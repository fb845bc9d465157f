"""CPU: on-disk formats against fixtures captured from the reference's own code (oracle/gen_golden_formats.py):
G14 Broadcast-style `transforms.json` parser, G15 nerfstudio checkpoint file (names, shapes, ordering, Adam moments)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case_idx", [0, 1, 2, 3, 4, 5])
def test_broadcaststyle_parser_matches_reference(tmp_path, case_idx):
    from soccernerfs_amd.dataparsers import BroadcaststyleDataParserConfig

    case = json.load(open(os.path.join(GOLD, "g14_dataparser.json")))[case_idx]
    (tmp_path / "transforms.json").write_text(case["transforms"])
    for f in case["existing"]:
        p = tmp_path / f
        p.parent.mkdir(parents=True, exist_ok=True)
        p.touch()
    for split, want in case["splits"].items():
        out = BroadcaststyleDataParserConfig(data=tmp_path, fps_downsample=case["fps_downsample"], **case.get("options", {})).setup().get_dataparser_outputs(split)
        assert [str(Path(f).relative_to(tmp_path)) for f in out.image_filenames] == want["image_filenames"]
        got_masks = None if out.mask_filenames is None else [str(Path(f).relative_to(tmp_path)) for f in out.mask_filenames]
        assert got_masks == want["mask_filenames"]
        got_depth = out.metadata["depth_filenames"]
        assert (None if got_depth is None else [str(Path(f).relative_to(tmp_path)) for f in got_depth]) == want["depth_filenames"]
        assert out.metadata["static"] == want["static"]
        cam = out.cameras
        M = len(want["image_filenames"])
        assert len(cam) == M
        torch.testing.assert_close(cam.camera_to_worlds, torch.tensor(want["c2w"]), rtol=1e-6, atol=1e-7)
        for k in ("fx", "fy", "cx", "cy"):
            torch.testing.assert_close(getattr(cam, k), torch.tensor(want[k]), rtol=1e-6, atol=0)
        assert [cam.height] * M == want["height"] and [cam.width] * M == want["width"]
        torch.testing.assert_close(cam.times, torch.tensor(want["times"]), rtol=0, atol=0)
        assert cam.ids.dtype == torch.uint8 and cam.ids.tolist() == want["ids"]
        torch.testing.assert_close(cam.distortion_params.expand(M, 6) if cam.distortion_params.dim() == 1 else cam.distortion_params,
                                   torch.tensor(want["distortion"]).expand(M, 6), rtol=1e-6, atol=0)
        torch.testing.assert_close(out.scene_box.aabb, torch.tensor(want["aabb"]), rtol=0, atol=0)
        assert abs(out.dataparser_scale - want["scale"]) <= 1e-7 * abs(want["scale"])
        torch.testing.assert_close(out.dataparser_transform, torch.tensor(want["transform"]), rtol=1e-6, atol=1e-7)


def test_parser_errors_and_options(tmp_path):
    from soccernerfs_amd.dataparsers import BroadcaststyleDataParserConfig, auto_orient_and_center_poses

    case = json.load(open(os.path.join(GOLD, "g14_dataparser.json")))[0]
    (tmp_path / "transforms.json").write_text(case["transforms"])
    with pytest.raises(AssertionError):  # no image exists
        BroadcaststyleDataParserConfig(data=tmp_path).setup().get_dataparser_outputs("train")
    with pytest.raises(KeyError):  # the reference's other split tables name cameras its CAM_IDS does not hold
        BroadcaststyleDataParserConfig(data=tmp_path, cam_split_setup="real").setup().get_dataparser_outputs("train")
    poses = torch.eye(4)[None].repeat(3, 1, 1)
    poses[:, :3, 3] = torch.tensor([[1.0, 2, 3], [3, 2, 1], [2, 2, 2]])
    centred, tf = auto_orient_and_center_poses(poses, "none", "poses")
    torch.testing.assert_close(centred[:, :3, 3].mean(0), torch.zeros(3))
    torch.testing.assert_close(tf[:, 3], -torch.tensor([2.0, 2, 2]))
    with pytest.raises(NotImplementedError):
        auto_orient_and_center_poses(poses, "pca", "none")


def test_image_cache_loader(tmp_path):
    from PIL import Image
    from soccernerfs_amd.dataparsers import load_image_cache

    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (5, 7, 3), dtype=np.uint8) for _ in range(3)]
    files = []
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(tmp_path / f"Camera_1_{i:04d}.png")
        files.append(tmp_path / f"Camera_1_{i:04d}.png")
    cache = load_image_cache(files)
    assert cache.dtype == torch.uint8 and cache.shape == (3, 5, 7, 3)
    assert np.array_equal(cache.numpy(), np.stack(imgs))


def _small_kplanes():
    from soccernerfs_amd.kplanes import KPlanesModel, KPlanesModelConfig
    from soccernerfs_amd.scene_colliders import SceneBox

    cfg = KPlanesModelConfig(multiscale_res=(1, 2), spacetime_resolution=(8, 8, 8, 4), feature_dim=8,
                             proposal_net_args_list=[{"feature_dim": 8, "resolution": [8, 8, 8, 4]}, {"feature_dim": 8, "resolution": [16, 16, 16, 4]}],
                             num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8, disable_viewing_dependent=True)
    return KPlanesModel(cfg, SceneBox(aabb=torch.tensor([[-1.5] * 3, [1.5] * 3])), num_train_data=4)


def test_checkpoint_names_shapes_and_moments_match_reference():
    """Export of this package's model carries exactly the reference checkpoint's keys / shapes; importing the reference file and
    exporting again reproduces every tensor bit for bit; the Adam moments land on the right parameters."""
    from soccernerfs_amd import checkpoint as CK

    ref = torch.load(os.path.join(GOLD, "g15_step-000000002.ckpt"), map_location="cpu", weights_only=False)
    assert set(ref) == {"step", "pipeline", "optimizers", "scalers"} and ref["step"] == 2
    model = _small_kplanes()
    mine = CK.reference_state_dict(model)
    assert set(mine) == set(ref["pipeline"])
    for k, v in ref["pipeline"].items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
    CK.load_reference_state_dict(model, ref["pipeline"])
    again = CK.reference_state_dict(model)
    for k, v in ref["pipeline"].items():
        assert torch.equal(again[k], v.float()), k
    # optimiser state: reference order -> this package's parameters -> reference order
    moments = CK.import_optimizer_states(model, ref["optimizers"])
    assert set(moments) == {n for n, p in model.named_parameters() if p.requires_grad and p.numel() > 0 and "aabb" not in n}
    back = CK.export_optimizer_states(model, moments, {g: {k: v for k, v in sd["param_groups"][0].items() if k != "params"} for g, sd in ref["optimizers"].items()})
    for g, sd in ref["optimizers"].items():
        assert back[g]["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
        assert set(back[g]["state"]) == set(sd["state"])
        for i, st in sd["state"].items():
            assert torch.equal(back[g]["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(back[g]["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
            assert float(back[g]["state"][i]["step"]) == float(st["step"]) == 3.0
    # and against the moments the generator stored by parameter NAME (guards the index -> parameter mapping)
    g = np.load(os.path.join(GOLD, "g15_checkpoint.npz"))
    for key in [str(k) for k in g["moment_keys"]]:
        grp, idx, name = key.split("/")
        torch.testing.assert_close(back[grp]["state"][int(idx)]["exp_avg"], torch.from_numpy(g["m_" + key]), rtol=0, atol=0)
        assert ("_model." + name) in ref["pipeline"]


def test_checkpoint_save_load_roundtrip(tmp_path):
    from soccernerfs_amd import checkpoint as CK

    a, b = _small_kplanes(), _small_kplanes()
    with torch.no_grad():
        for p in a.parameters():
            if p.requires_grad and p.numel():
                p.uniform_(-1, 1)
    CK.save_checkpoint(str(tmp_path), 10, a)
    path = CK.save_checkpoint(str(tmp_path), 12, a)
    assert os.path.basename(path) == "step-000000012.ckpt" and sorted(os.listdir(tmp_path)) == ["step-000000012.ckpt"]  # only the latest kept
    start, moments = CK.load_checkpoint(str(tmp_path), b)
    assert start == 13 and moments == {}
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n
    with pytest.raises(FileNotFoundError):
        CK.load_checkpoint(str(tmp_path), b, load_step=11)
    sd = CK.reference_state_dict(a)
    sd.pop("_model.field.grids.0.3")
    with pytest.raises(KeyError):
        CK.load_reference_state_dict(b, sd)


def test_checkpoint_names_for_the_nerfplayer_models():
    """Temporal-grid / hash-grid models: the export carries the reference's names incl. the encoder's registered index buffers (G12 / G13
    hold the reference's parameter names)."""
    from tests.conftest import load_golden
    from soccernerfs_amd import checkpoint as CK
    from soccernerfs_amd.nerfplayer import NerfplayerModel, NerfplayerModelConfig
    from soccernerfs_amd.scene_colliders import SceneBox

    g = load_golden("g13_nerfplayer_full")
    cfg = NerfplayerModelConfig(
        num_levels=4, features_per_level=2, log2_hashmap_size=12, temporal_dim=8,
        proposal_net_args_list=[{"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 32},
                                {"hidden_dim": 16, "temporal_dim": 4, "log2_hashmap_size": 9, "num_levels": 3, "max_res": 64}],
        num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8)
    model = NerfplayerModel(cfg, SceneBox(aabb=torch.tensor([[-1.0] * 3, [1.0] * 3])), num_train_data=int(g["num_images"]))
    sd = CK.reference_state_dict(model, prefix="")
    for name in [str(n) for n in g["param_names"]]:
        assert name in sd and tuple(sd[name].shape) == tuple(g["param_" + name].shape), name
    assert "field.newness_field.index_list" in sd and "proposal_networks.0.encoding.offsets" in sd
    ref_state = {str(n): g["param_" + str(n)] for n in g["param_names"]}
    CK.load_reference_state_dict(model, ref_state, prefix="", strict=False)
    again = CK.reference_state_dict(model, prefix="")
    for name, v in ref_state.items():
        assert torch.equal(again[name], v), name


@pytest.mark.parametrize("case_idx", [0, 1])
def test_stadiumwide_parser_matches_reference(tmp_path, case_idx):
    """G14b: the stadium-wide scene's parser (config 4's data): camera ids from the group names, eval = the close-up cameras, train = ring
    cameras spread evenly over the 110, every parsed camera in the pose scaling."""
    from soccernerfs_amd.dataparsers import StadiumwideDataParserConfig, get_cam_id

    assert get_cam_id("Ext Left-Left-0") == 0 and get_cam_id("Middle-Right-5") == 25 and get_cam_id("Ext Op Left-High Behind Left-9") == 109
    assert get_cam_id("Shooter") == 115
    case = json.load(open(os.path.join(GOLD, "g14b_stadiumwide.json")))[case_idx]
    (tmp_path / "transforms.json").write_text(case["transforms"])
    for f in case["existing"]:
        p = tmp_path / f
        p.parent.mkdir(parents=True, exist_ok=True)
        p.touch()
    for split, want in case["splits"].items():
        out = StadiumwideDataParserConfig(data=tmp_path, **case["options"]).setup().get_dataparser_outputs(split)
        assert [str(Path(f).relative_to(tmp_path)) for f in out.image_filenames] == want["image_filenames"]
        cam = out.cameras
        M = len(want["image_filenames"])
        torch.testing.assert_close(cam.camera_to_worlds, torch.tensor(want["c2w"]), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(cam.fx, torch.tensor(want["fx"]), rtol=1e-6, atol=0)
        torch.testing.assert_close(cam.times, torch.tensor(want["times"]), rtol=0, atol=0)
        assert cam.ids.tolist() == want["ids"] and [cam.height] * M == want["height"] and [cam.width] * M == want["width"]
        torch.testing.assert_close(out.scene_box.aabb, torch.tensor(want["aabb"]), rtol=0, atol=0)
        assert abs(out.dataparser_scale - want["scale"]) <= 1e-7 * abs(want["scale"])
        assert "depth_filenames" not in out.metadata


@pytest.mark.parametrize("case_idx", [0, 1])
def test_stadium_parser_matches_reference(tmp_path, case_idx):
    """G14c: the parser the nerfplayer presets name (stadium_dataparser.py): `images_<k>/` layout, ids from `<group>-<camera>`, pose
    orientation "up" + centring "poses", percentage camera split (incl. its index-vs-id comparison on a sparse ring), depth lists."""
    from soccernerfs_amd.dataparsers import StadiumDataParserConfig

    case = json.load(open(os.path.join(GOLD, "g14c_stadium.json")))[case_idx]
    (tmp_path / "transforms.json").write_text(case["transforms"])
    for f in case["existing"]:
        p = tmp_path / f
        p.parent.mkdir(parents=True, exist_ok=True)
        p.touch()
    for split, want in case["splits"].items():
        out = StadiumDataParserConfig(data=tmp_path, **case["options"]).setup().get_dataparser_outputs(split)
        assert [str(Path(f).relative_to(tmp_path)) for f in out.image_filenames] == want["image_filenames"]
        assert [str(Path(f).relative_to(tmp_path)) for f in out.metadata["depth_filenames"]] == want["depth_filenames"]
        cam = out.cameras
        torch.testing.assert_close(cam.camera_to_worlds, torch.tensor(want["c2w"]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(cam.fx, torch.tensor(want["fx"]), rtol=1e-6, atol=0)
        torch.testing.assert_close(cam.times, torch.tensor(want["times"]), rtol=0, atol=0)
        assert cam.ids.tolist() == want["ids"]
        torch.testing.assert_close(out.scene_box.aabb, torch.tensor(want["aabb"]), rtol=0, atol=0)
        assert abs(out.dataparser_scale - want["scale"]) <= 1e-6 * abs(want["scale"])
        torch.testing.assert_close(out.dataparser_transform, torch.tensor(want["transform"]), rtol=1e-5, atol=1e-6)
        assert set(out.metadata) == {"depth_filenames", "depth_unit_scale_factor"}
    with pytest.raises(ValueError):
        StadiumDataParserConfig(data=tmp_path, **case["options"]).setup().get_dataparser_outputs("bogus")

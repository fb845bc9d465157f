"""On-disk format of the Broadcast-style release: `transforms.json` + `<k>x/` image directories, parsed as
NS/data/dataparsers/broadcaststyle_dataparser.py:261-547 does (`Broadcaststyle._generate_dataparser_outputs`), plus the image loader
that fills the resident uint8 image cache (NS/data/datasets/base_dataset.py:60-95).

Host-side logic only (json + file names + a few 4x4 matrices); no kernels.  What is built: the "all" camera split (the only entry of
the reference's SETUPS table whose camera names exist in its CAM_IDS table, :44-73,75-192), per-frame or global intrinsics, time steps
from the file names, fps down-sampling, pose auto-scaling, `orientation_method`/`center_method` "none" (the parser's defaults) and
"poses" centring, masks / depth file lists.  Distortion parameters are carried but the ray generator here is pinhole-only."""
import json
from dataclasses import dataclass, field
from pathlib import Path, PurePath
from typing import Dict, List, Optional

import numpy as np
import torch

from .cameras import Cameras
from .scene_colliders import SceneBox

# camera name -> unique id (:44-73)
CAM_IDS = {**{f"Camera_{i + 1}": i for i in range(20)}, **{f"global_{i + 1}": 20 + i for i in range(8)}}
# train / eval camera split (:75-192); the other entries of the reference's table name cameras that CAM_IDS does not hold
SETUPS = {"all": {"train": [f"Camera_{i + 1}" for i in range(19)], "eval": ["Camera_20"]}}


@dataclass
class DataparserOutputs:
    """NS/data/dataparsers/base_dataparser.py:44-77."""

    image_filenames: List[Path]
    cameras: Cameras
    scene_box: SceneBox
    mask_filenames: Optional[List[Path]] = None
    dataparser_scale: float = 1.0
    dataparser_transform: torch.Tensor = field(default_factory=lambda: torch.eye(4)[:3])
    metadata: Dict = field(default_factory=dict)


@dataclass
class BroadcaststyleDataParserConfig:
    """:195-236 (defaults of the parser class; the k-planes preset keeps them)."""

    data: Path = Path("data/broadcaststyle/")
    scale_factor: float = 1.0
    downscale_factor: Optional[int] = 2
    scene_scale: float = 1.5
    orientation_method: str = "none"
    center_method: str = "none"
    auto_scale_poses: bool = True
    depth_unit_scale_factor: float = 0.01
    depth_maps: str = "none"
    depth_mask: str = "mask"
    cam_split_setup: str = "all"
    cap_box_floor: bool = False
    static: bool = False
    static_allimgs: bool = False
    static_timestep: int = -1
    fps_downsample: float = 3.0

    def setup(self) -> "Broadcaststyle":
        return Broadcaststyle(self)


def _distortion(src: Dict) -> torch.Tensor:
    """camera_utils.get_distortion_params: [k1, k2, k3, k4, p1, p2]."""
    return torch.tensor([float(src.get(k, 0.0)) for k in ("k1", "k2", "k3", "k4", "p1", "p2")])


def rotation_matrix(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Rotation taking direction a to direction b (Rodrigues; NS/cameras/camera_utils.py:404-429).  Exactly opposite vectors are not
    handled (the reference perturbs one of them randomly)."""
    a = a / torch.linalg.norm(a)
    b = b / torch.linalg.norm(b)
    v = torch.linalg.cross(a, b)
    c = torch.dot(a, b)
    if c < -1 + 1e-8:
        raise ValueError("rotation_matrix: opposite vectors")
    s = torch.linalg.norm(v)
    k = torch.tensor([[0.0, -float(v[2]), float(v[1])], [float(v[2]), 0.0, -float(v[0])], [-float(v[1]), float(v[0]), 0.0]])
    return torch.eye(3) + k + k @ k * ((1 - c) / (s ** 2 + 1e-8))


def auto_orient_and_center_poses(poses: torch.Tensor, method: str = "none", center_method: str = "none"):
    """NS/cameras/camera_utils.py:470-574 for the methods "none" (the Broadcast-style default) and "up" (mean camera up-axis to +z);
    centring "none" / "poses"."""
    origins = poses[..., :3, 3]
    if center_method == "poses":
        translation = torch.mean(origins, dim=0)
    elif center_method == "none":
        translation = torch.zeros(3)
    else:
        raise NotImplementedError(f"center_method {center_method!r} (built: 'none', 'poses')")
    if method == "up":
        up = torch.mean(poses[:, :3, 1], dim=0)
        up = up / torch.linalg.norm(up)
        rotation = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rotation, rotation @ -translation[..., None]], dim=-1)
        return transform @ poses, transform
    if method != "none":
        raise NotImplementedError(f"orientation method {method!r} (built: 'none', 'up')")
    transform = torch.eye(4)
    transform[:3, 3] = -translation
    transform = transform[:3, :]
    return transform @ poses, transform


class Broadcaststyle:
    empty_scene_dir = "broadcaststyle_empty/"  # :263-264
    has_depth = True

    def __init__(self, config: BroadcaststyleDataParserConfig):
        self.config = config
        self.downscale_factor = None

    def _cam_id(self, name: str) -> int:
        return int(CAM_IDS[name])

    def _split_cameras(self, split: str):
        """-> (cameras of this split, cameras of the other split or None = keep every parsed camera for the pose scaling)."""
        setup_split = "train" if split == "train" else "eval"
        other_split = "eval" if setup_split == "train" else "train"
        return ([CAM_IDS[c] for c in SETUPS[self.config.cam_split_setup][setup_split]],
                [CAM_IDS[c] for c in SETUPS[self.config.cam_split_setup][other_split]])

    def _get_fname(self, filepath: PurePath, data_dir: Path, downsample_folder_prefix="images_") -> Path:
        """:529-547: <dir>/<k>x/<name>."""
        self.downscale_factor = self.config.downscale_factor
        old = data_dir / filepath
        return old.parent / f"{self.config.downscale_factor}x" / old.name

    guard_zero_time = True

    def _keep_time_step(self, time_step: int) -> bool:
        """Static-scene options (:329-335)."""
        cfg = self.config
        if cfg.static and not cfg.static_allimgs:
            return time_step == (0 if cfg.static_timestep == -1 else cfg.static_timestep)
        return True

    def _depth_path(self, frame: Dict) -> Optional[str]:
        """:379-389."""
        cfg = self.config
        if not (self.has_depth and "depth_file_path" in frame and cfg.depth_maps != "none"):
            return None
        dp = frame["depth_file_path"]
        if cfg.depth_mask != "none":
            dp = dp.replace("depth-maps", "depth-maps-" + cfg.depth_mask)
        if cfg.depth_maps != "depth-maps":
            dp = dp.replace("depth-maps", cfg.depth_maps)
        return dp

    def _select(self, cam_uids: List[int], times: List[int], split: str, split_cams) -> List[int]:
        """fps down-sampling (:405-412: keep the time steps linspace(0, T-1, int(T / fps_downsample))) and the split's cameras."""
        times_filter = np.arange(max(times) + 1)
        if self.config.fps_downsample > 1:
            base = max(times) + 1
            times_filter = np.linspace(0, base - 1, int(base / self.config.fps_downsample)).astype(np.int32)
        return [i for i in range(len(cam_uids)) if cam_uids[i] in split_cams and times[i] in times_filter]

    def _metadata(self, depth_filenames) -> Dict:
        cfg = self.config
        if not self.has_depth:
            return {"static": cfg.static}
        return {"depth_filenames": depth_filenames, "depth_unit_scale_factor": cfg.depth_unit_scale_factor, "static": cfg.static}

    def _frame_metadata(self, fname: Path):
        """:242-259: `<camera name>_<time step>.<ext>`."""
        head, tail = fname.name.rsplit("_", 1)
        return self._cam_id(head), int(tail.split(".")[0])

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        cfg = self.config
        data = Path(cfg.data)
        if cfg.static and cfg.static_timestep == -1:
            data = data.parent / self.empty_scene_dir
        if data.suffix == ".json":
            meta, data_dir = json.load(open(data)), data.parent
        else:
            meta, data_dir = json.load(open(data / "transforms.json")), data
        fixed = {k: k in meta for k in ("fl_x", "fl_y", "cx", "cy", "h", "w")}
        distort_fixed = any(k in meta for k in ("k1", "k2", "k3", "p1", "p2"))
        per = {k: [] for k in fixed}
        image_filenames, mask_filenames, depth_filenames, poses, distort, times, cam_uids = [], [], [], [], [], [], []
        split_cams, other_cams = self._split_cameras(split)
        for frame in meta["frames"]:
            fname = self._get_fname(PurePath(frame["file_path"]), data_dir)
            if not fname.exists():
                continue
            cam_id, time_step = self._frame_metadata(fname)
            if other_cams is not None and cam_id not in split_cams and cam_id not in other_cams:
                continue
            if not self._keep_time_step(time_step):
                continue
            cam_uids.append(cam_id)
            times.append(time_step)
            for k in fixed:
                if not fixed[k]:
                    assert k in frame, f"{k} not specified in frame"
                    per[k].append(int(frame[k]) if k in ("h", "w") else float(frame[k]))
            if not distort_fixed:
                distort.append(_distortion(frame))
            image_filenames.append(fname)
            poses.append(np.array(frame["transform_matrix"]))
            if "mask_path" in frame:
                mask_filenames.append(self._get_fname(PurePath(frame["mask_path"]), data_dir, downsample_folder_prefix="masks_"))
            dp = self._depth_path(frame)
            if dp is not None:
                depth_filenames.append(self._get_fname(PurePath(dp), data_dir, downsample_folder_prefix="depths_"))
        assert len(image_filenames) != 0, "No image files found. Check the file_paths in transforms.json."
        assert len(mask_filenames) in (0, len(image_filenames)), "Different number of image and mask filenames."
        assert len(depth_filenames) in (0, len(image_filenames)), "Different number of image and depth filenames."

        indices = self._select(cam_uids, times, split, split_cams)

        orientation = meta.get("orientation_override", cfg.orientation_method)
        poses = torch.from_numpy(np.array(poses).astype(np.float32))
        poses, transform = auto_orient_and_center_poses(poses, method=orientation, center_method=cfg.center_method)
        scale = 1.0
        if cfg.auto_scale_poses:
            scale /= float(torch.max(torch.abs(poses[:, :3, 3])))  # over ALL parsed cameras (both splits), :433-438
        scale *= cfg.scale_factor
        poses[:, :3, 3] *= scale

        sel = lambda lst: [lst[i] for i in indices]
        a = cfg.scene_scale
        lo = [-a, -a, -0.1] if cfg.cap_box_floor else [-a, -a, -a]
        scene_box = SceneBox(aabb=torch.tensor([lo, [a, a, a]], dtype=torch.float32))
        if meta.get("camera_model", "OPENCV") not in ("OPENCV", "PERSPECTIVE", "PINHOLE", "SIMPLE_PINHOLE"):
            raise NotImplementedError(f"camera model {meta['camera_model']!r}: only perspective cameras are built")
        idx = torch.tensor(indices, dtype=torch.long)
        val = lambda k, dt: (dt(meta[k]) if fixed[k] else torch.tensor(per[k], dtype=torch.float32 if dt is float else torch.int32)[idx])
        fx, fy, cx, cy = val("fl_x", float), val("fl_y", float), val("cx", float), val("cy", float)
        height, width = val("h", int), val("w", int)
        tmax = max(times)
        t = torch.tensor(times, dtype=torch.float32)[idx]
        t = t / tmax if (tmax != 0 or not self.guard_zero_time) else t
        ids = torch.tensor(cam_uids, dtype=torch.float32)[idx].to(torch.uint8)  # Cameras._init_get_ids (cameras.py:263-272) stores uint8
        dist = _distortion(meta) if distort_fixed else torch.stack(distort, dim=0)[idx]
        if isinstance(height, torch.Tensor):
            if len(set(height.tolist())) != 1 or len(set(width.tolist())) != 1:
                raise NotImplementedError("cameras of different image sizes")
            height, width = int(height[0]), int(width[0])
        cameras = Cameras(camera_to_worlds=poses[idx][:, :3, :4], fx=fx, fy=fy, cx=cx, cy=cy, width=width, height=height, times=t, ids=ids,
                          distortion_params=dist)
        assert self.downscale_factor is not None
        cameras.rescale_output_resolution(1.0 / self.downscale_factor)
        return DataparserOutputs(image_filenames=sel(image_filenames), cameras=cameras, scene_box=scene_box,
                                 mask_filenames=sel(mask_filenames) if mask_filenames else None, dataparser_scale=scale,
                                 dataparser_transform=transform,
                                 metadata=self._metadata(sel(depth_filenames) if depth_filenames else None))


# ---- stadium-wide scene (NS/data/dataparsers/stadiumwide_dataparser.py): 110 ring cameras in 11 named groups of 10 + 6 close-up cameras ----
CAMERA_LOCATIONS = ["Ext Left-Left", "Left-Middle", "Middle-Right", "Right-Ext Right", "Ext Right-High Behind Right",
                    "High Behind Right-Ext Op Right", "Ext Op Right-Op Right", "Op Right-Op Middle", "Op Middle-Op Left", "Op Left-Ext Op Left",
                    "Ext Op Left-High Behind Left"]  # :49-61, ids 10 * group + local id
CLOSE_CAMERAS = {"Center": 110, "GoalLeft": 111, "GoalRight": 112, "PlayerLeft": 113, "PlayerRight": 114, "Shooter": 115}  # :63-70


def get_cam_id(cam_name: str) -> int:
    """stadiumwide_dataparser.py:73-79."""
    if "-" in cam_name:
        group, local = cam_name.rsplit("-", 1)
        return CAMERA_LOCATIONS.index(group) * 10 + int(local)
    return CLOSE_CAMERAS[cam_name]


@dataclass
class StadiumwideDataParserConfig(BroadcaststyleDataParserConfig):
    """:83-118: as the Broadcast-style config with these defaults and two extra fields (no depth maps)."""

    data: Path = Path("data/stadiumwide/")
    scene_scale: float = 1.0
    cam_split_setup: str = "low"
    fps_downsample: float = 1.0
    nb_train_cameras: int = 110
    closeup_training: bool = False

    def setup(self) -> "Stadiumwide":
        return Stadiumwide(self)


class Stadiumwide(Broadcaststyle):
    """Same file layout; camera ids from the group names; eval = the six close-up cameras, train = nb_train_cameras ring cameras spread
    evenly over the 110 (:271-281); every parsed camera takes part in the pose scaling."""

    empty_scene_dir = "stadium_players_empty/"
    has_depth = False

    def _cam_id(self, name: str) -> int:
        return get_cam_id(name)

    def _split_cameras(self, split: str):
        cams = list(range(110, 116))
        if split == "train":
            cams = np.linspace(0, 109, self.config.nb_train_cameras).astype(np.int32).tolist()
            if self.config.closeup_training:
                cams = cams + list(range(110, 116))
        return cams, None


@dataclass
class StadiumDataParserConfig:
    """NS/data/dataparsers/stadium_dataparser.py:72-110 (the parser the nerfplayer presets name, method_configs.py:573,627)."""

    data: Path = Path("data/stadium/")
    scale_factor: float = 1.0
    downscale_factor: Optional[int] = 2
    scene_scale: float = 1.5
    orientation_method: str = "up"
    center_method: str = "poses"
    auto_scale_poses: bool = True
    train_split_percentage: float = 0.95
    depth_unit_scale_factor: float = 1e-3
    camera_location: str = "Op Right-Op Middle"  # unused by the reference's code path as it stands (its filter is commented out)
    # fields the shared parsing code reads; the stadium parser has no such options
    cap_box_floor: bool = False
    static: bool = False
    static_allimgs: bool = False
    static_timestep: int = -1
    fps_downsample: float = 1.0

    def setup(self) -> "Stadium":
        return Stadium(self)


class Stadium(Broadcaststyle):
    """Files `<group name>-<camera in group>_<time step>.<ext>` under `images_<k>/`; unique camera id = 10 * group + camera; the
    training split = ceil(95 %) of the cameras, evenly spaced, the rest evaluates (:289-306); depth maps listed whenever present."""

    has_depth = True
    guard_zero_time = False  # :367 divides by max(times) unconditionally

    def _get_fname(self, filepath: PurePath, data_dir: Path, downsample_folder_prefix="images_") -> Path:
        """:404-433 with an explicit downscale factor (the automatic choice opens image files and is not built)."""
        if self.config.downscale_factor is None:
            raise NotImplementedError("automatic downscale factor")
        self.downscale_factor = self.config.downscale_factor
        if self.downscale_factor > 1:
            return data_dir / f"{downsample_folder_prefix}{self.downscale_factor}" / filepath.name
        return data_dir / filepath

    def _frame_metadata(self, fname: Path):
        """:120-144."""
        loc, rest = fname.name.rsplit("-", 1)
        cam, tail = rest.split("_")[:2]
        return CAMERA_LOCATIONS.index(loc) * 10 + int(cam), int(tail.split(".")[0])

    def _split_cameras(self, split: str):
        return None, None  # chosen after parsing, from the cameras actually present

    def _keep_time_step(self, time_step: int) -> bool:
        return True

    def _depth_path(self, frame: Dict) -> Optional[str]:
        return frame.get("depth_file_path")

    def _select(self, cam_uids, times, split, split_cams):
        import math

        num_cams = len(np.unique(cam_uids))
        num_train = math.ceil(num_cams * self.config.train_split_percentage)
        i_train = np.linspace(0, num_cams - 1, num_train, dtype=int)
        i_eval = np.setdiff1d(np.arange(num_cams), i_train)
        if split == "train":
            chosen = i_train
        elif split in ("val", "test"):
            chosen = i_eval
        else:
            raise ValueError(f"Unknown dataparser split {split}")
        # as the reference: camera INDICES are compared with camera IDS (the same thing when every camera of the ring is present)
        return [i for i in range(len(cam_uids)) if cam_uids[i] in chosen]

    def _metadata(self, depth_filenames) -> Dict:
        return {"depth_filenames": depth_filenames, "depth_unit_scale_factor": self.config.depth_unit_scale_factor}


def load_image_cache(image_filenames: List[Path]) -> torch.Tensor:
    """InputDataset.get_numpy_image (base_dataset.py:60-80) for every file: uint8 [M,H,W,3] (RGBA files are alpha-composited over white
    only by get_image's float path, :82-95; the resident cache keeps 8-bit RGB and rejects other layouts)."""
    from PIL import Image

    out = []
    for f in image_filenames:
        im = np.array(Image.open(f), dtype="uint8")
        if im.ndim == 2:
            im = im[:, :, None].repeat(3, axis=2)
        if im.shape[2] != 3:
            raise NotImplementedError(f"{f}: {im.shape[2]} channels; the image cache holds 8-bit RGB")
        out.append(torch.from_numpy(im))
    return torch.stack(out)

"""Pixel samplers with the interface of NS/data/pixel_samplers.py (PixelSampler :24-128, DynamicBasedPixelSampler :329-426)
and the IST weight maps of NS/data/datasets/dynamic_dataset.py:328-470 -- all on the device, no host loop, no sync."""
import ctypes as C
from math import floor
from typing import Dict, Optional

import torch

from . import _lib, ops


def temporal_neighbours(cam_ids: torch.Tensor, cam_times: torch.Tensor, ist_range: float):
    """CSR neighbour lists: same camera id and 0.01 < |dt| <= ist_range (dynamic_dataset.py:419-429).  Small (M x M) torch ops."""
    ids, t = cam_ids.reshape(-1), cam_times.reshape(-1).float()
    dt = (t[:, None] - t[None, :]).abs()
    adj = (ids[:, None] == ids[None, :]) & (dt <= ist_range) & (dt > 0.01)
    counts = adj.sum(1)
    off = torch.zeros(ids.numel() + 1, dtype=torch.int32, device=ids.device)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    idx = adj.nonzero()[:, 1].to(torch.int32).contiguous()
    if idx.numel() == 0:
        idx = torch.zeros(1, dtype=torch.int32, device=ids.device)
    return off, idx


def compute_ist(images: torch.Tensor, cam_ids: torch.Tensor, cam_times: torch.Tensor, ist_range: float, alpha: float = 0.15) -> torch.Tensor:
    """DynamicDataset.compute_ist: images [M,H,W,3] (uint8 or float32 in [0,1], on the HIP device) -> fp16 maps [M,H,W]."""
    if not images.is_cuda or images.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError("compute_ist: images must be a uint8 or float32 HIP device tensor")
    images = images.contiguous()
    M, H, W = images.shape[:3]
    off, idx = temporal_neighbours(cam_ids.to(images.device), cam_times.to(images.device), ist_range)
    out = torch.empty(M, H, W, dtype=torch.float16, device=images.device)
    _lib.check(_lib.lib().snerf_ist_maps(ops._ptr(images), 0 if images.dtype == torch.uint8 else 1, M, H, W, ops._ptr(off), ops._ptr(idx), alpha,
                                         ops._ptr(out), ops._stream()), "ist_maps")
    return out


def compute_isg(images: torch.Tensor, cam_ids: torch.Tensor, isg_gamma: float = 5e-2) -> torch.Tensor:
    """DynamicDataset.compute_isg (dynamic_dataset.py:215-326): images [M,H,W,3] (uint8 or float32 in [0,1], HIP device), cam_ids [M]
    -> fp16 maps [M,H,W] = mean_c r^2 / (r^2 + gamma^2) with r = image - median image of its camera."""
    if not images.is_cuda or images.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError("compute_isg: images must be a uint8 or float32 HIP device tensor")
    images = images.contiguous()
    M, H, W = images.shape[:3]
    dev = images.device
    ids = cam_ids.reshape(-1).to(dev)
    uniq, inv = torch.unique(ids, return_inverse=True)  # camera slots in ascending id order; small [M] ops
    order = torch.argsort(inv, stable=True)
    counts = torch.bincount(inv, minlength=uniq.numel())
    off = torch.zeros(uniq.numel() + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    max_frames = int(counts.max())
    med = torch.empty(uniq.numel(), H, W, 3, dtype=images.dtype, device=dev)
    out = torch.empty(M, H, W, dtype=torch.float16, device=dev)
    cam_img, img_cam = order.to(torch.int32).contiguous(), inv.to(torch.int32).contiguous()  # named: they must outlive the launch
    _lib.check(_lib.lib().snerf_isg_maps(ops._ptr(images), 0 if images.dtype == torch.uint8 else 1, M, H, W, uniq.numel(), ops._ptr(off),
                                         ops._ptr(cam_img), ops._ptr(img_cam), max_frames, isg_gamma, ops._ptr(med), ops._ptr(out), ops._stream()),
               "isg_maps")
    return out


def weights_cache_name(kind: str, param: float, num_images: int, height: int, eval_split: bool = False) -> str:
    """File name of the reference's offline weight cache next to the images (a torch.save'd fp16 [M,H,W] tensor):
    `ist-weights-<range with . -> _>-<split>-<N>-<H>p.pt` (dynamic_dataset.py:362-363) / `isg-weights-<gamma>-<split>-<N>-<H>p.pt` (:237)."""
    split = "eval" if eval_split else "train"
    if kind == "ist":
        return f"ist-weights-{str(param).replace('.', '_')}-{split}-{num_images}-{height}p.pt"
    if kind == "isg":
        return f"isg-weights-{param}-{split}-{num_images}-{height}p.pt"
    raise ValueError("kind must be 'ist' or 'isg'")


def load_or_compute_weights(path: str, num_images: int, compute):
    """The reference's offline cache protocol (dynamic_dataset.py:365-378, 464-468): load the .pt file if it exists and has one map per
    image, otherwise call compute() and save the result."""
    import os

    if os.path.exists(path):
        w = torch.load(path)
        if w.shape[0] == num_images:
            return w
    w = compute()
    torch.save(w, path)
    return w


def pick_cached_images(cam_times, cam_ids, num_to_sample: int, pick_mode: str = "randsteps", rng=None):
    """Which images go into the on-device image cache when it is smaller than the dataset: CacheDataloader._get_batch_list
    (NS/data/utils/dataloaders.py:105-175).  cam_times / cam_ids: one entry per dataset image.  "normal": a random subset;
    "randsteps": all cameras at the same random time steps (first and last always included); "lowfps": every k-th time step;
    short-falls are topped up with random images.  rng: a `random.Random` (the reference uses the global `random` module)."""
    import random as _random
    from math import ceil

    rng = rng if rng is not None else _random
    times_all = [float(t) for t in (cam_times.reshape(-1).tolist() if hasattr(cam_times, "reshape") else cam_times)]
    ids_all = [int(i) for i in (cam_ids.reshape(-1).tolist() if hasattr(cam_ids, "reshape") else cam_ids)]
    total = len(times_all)
    if total == num_to_sample:
        pick_mode = "normal"
    if pick_mode == "normal":
        indices = rng.sample(range(total), k=num_to_sample)
    elif pick_mode in ("randsteps", "lowfps"):
        times = list(set(times_all))  # the reference does not sort this list (:129)
        if pick_mode == "randsteps":
            steps = int(num_to_sample / len(set(ids_all)))
            picked = [times[0], times[-1]] + rng.sample(times[1:-1], k=steps - 2)
        else:
            k = ceil(total / num_to_sample)
            picked = times[::k]
            if len(times) % k != 0:
                picked = picked[:-1]
        indices = [i for i in range(total) if times_all[i] in picked]
        left = num_to_sample - len(indices)
        if left > 0:
            indices += rng.sample([x for x in range(total) if x not in indices], k=left)
    else:
        raise ValueError("Unknown pick_mode: " + pick_mode)
    if len(indices) != num_to_sample:
        raise RuntimeError("Not enough images to sample from.")
    return indices


class PixelSampler:
    """Uniform pixel sampler (pixel_samplers.py:24-128)."""

    def __init__(self, num_rays_per_batch: int, keep_full_image: bool = False, **kwargs) -> None:
        self.num_rays_per_batch, self.keep_full_image = num_rays_per_batch, keep_full_image

    def set_num_rays_per_batch(self, num_rays_per_batch: int):
        self.num_rays_per_batch = num_rays_per_batch

    def sample_method(self, batch_size: int, num_images: int, image_height: int, image_width: int, mask=None, batch=None, device="cuda"):
        if mask is not None:
            raise NotImplementedError("masked sampling is not used by the soccer datasets")
        # :74-77 floor(rand(R,3) * [M,H,W]).long(), one kernel
        return ops.sample_pixels_uniform(torch.rand((batch_size, 3), device=device), num_images, image_height, image_width)[0]

    def collate_image_dataset_batch(self, batch: Dict, num_rays_per_batch: int, keep_full_image: bool = False):
        """:81-128: batch["image"] [M,H,W,3] (uint8 or float), batch["image_idx"] [M]."""
        device = batch["image"].device
        M, H, W, _ = batch["image"].shape
        indices = self.sample_method(num_rays_per_batch, M, H, W, batch=batch, device=device)
        if batch.get("time_key") is not None:
            # optional, not in the reference: the batch in order of the images' frame time (batch["time_key"], batch["n_time_keys"] =
            # ops.image_time_keys(times)).  A batch is a set, so this is free, and it makes every time-plane / temporal-grid gather coherent
            indices = ops.sort_rays_by_time(indices, batch["time_key"], int(batch["n_time_keys"]))
        c, y, x = indices[:, 0], indices[:, 1], indices[:, 2]
        out = {}
        for key, value in batch.items():
            if key in ("image_idx", "iter_steps", "ist_cdf", "ist_nonempty", "ist_nnz", "time_key", "n_time_keys") or value is None or not isinstance(value, torch.Tensor):
                continue
            v = value[c, y, x]
            out[key] = v.float() / 255.0 if v.dtype == torch.uint8 else v
        indices = indices.clone()
        indices[:, 0] = batch["image_idx"][c]
        out["indices"] = indices
        if keep_full_image:
            out["full_image"] = batch["image"]
        return out

    def sample(self, image_batch: Dict):
        return self.collate_image_dataset_batch(image_batch, self.num_rays_per_batch, keep_full_image=self.keep_full_image)


class DynamicBasedPixelSampler(PixelSampler):
    """IST/ISG importance sampling (pixel_samplers.py:329-426): floor(is_pixel_ratio * R) rays are drawn from the weight maps,
    10 * ceil(num_ist / M) per image from randomly chosen non-empty images -- without replacement inside an image whenever it has enough
    non-zero pixels, as torch.multinomial is called at :400-402 --, the rest uniformly; active after iters_to_start_ist."""

    def __init__(self, num_rays_per_batch: int, keep_full_image: bool = False, is_pixel_ratio: float = 0.15, iters_to_start_ist: int = 2000,
                 **kwargs) -> None:
        super().__init__(num_rays_per_batch, keep_full_image)
        self.is_pixel_ratio, self.iters_to_start_ist = is_pixel_ratio, iters_to_start_ist

    @staticmethod
    def prepare(batch: Dict) -> Dict:
        """Once per image-cache refresh: per-image inclusive prefix sums of the maps + the list of non-empty maps."""
        w = batch["ist_weights"]
        M = w.shape[0]
        batch["ist_cdf"] = torch.cumsum(w.reshape(M, -1).float(), dim=1).contiguous()
        batch["ist_nonempty"] = (batch["ist_cdf"][:, -1] > 0).nonzero()[:, 0].contiguous()
        batch["ist_nnz"] = (w.reshape(M, -1) > 0).sum(1).to(torch.int32).contiguous()  # len(torch.nonzero(weight_map)) of :400-402
        return batch

    def sample_method(self, batch_size: int, num_images: int, image_height: int, image_width: int, mask=None, batch: Optional[Dict] = None,
                      device="cuda"):
        assert batch is not None, "Batch information must be provided for DynamicBasedPixelSampler"
        if batch.get("ist_weights") is None or not (batch.get("iter_steps", 0) > self.iters_to_start_ist):
            return super().sample_method(batch_size, num_images, image_height, image_width, device=device)
        if "ist_cdf" not in batch:
            self.prepare(batch)
        num_ist = floor(self.is_pixel_ratio * batch_size)
        per_image = 10 * (-(-num_ist // num_images))  # :369
        nonempty = batch["ist_nonempty"]
        k = min(-(-num_ist // per_image), int(nonempty.numel()))
        n = min(num_ist, k * per_image)  # "rare case where pixels_per_image times num_images is less than num_ist" (:413-416)
        parts = []
        if n > 0:
            chosen = nonempty[torch.randperm(nonempty.numel(), device=device)[:k]].contiguous()  # random.shuffle + skip empty maps (:376-399)
            u = torch.rand(n, device=device)
            idx = torch.empty(n, 3, dtype=torch.int64, device=device)
            _lib.check(_lib.lib().snerf_ist_sample(ops._ptr(batch["ist_cdf"]), image_height, image_width, ops._ptr(chosen), ops._ptr(batch["ist_nnz"]),
                                                   per_image, ops._ptr(u), n, ops._ptr(idx), ops._stream()), "ist_sample")
            parts.append(idx)
        parts.append(super().sample_method(batch_size - n, num_images, image_height, image_width, device=device))
        return torch.cat(parts, dim=0)

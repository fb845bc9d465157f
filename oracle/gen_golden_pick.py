"""G10c: image-cache pick modes -- the reference's own CacheDataloader._get_batch_list (NS/data/utils/dataloaders.py:105-175) with a seeded
`random`, on a toy dataset (5 cameras x 12 time steps); the picked dataset indices per mode.  Build container only:

    python oracle/gen_golden_pick.py     # writes tests/golden/g10c_pick.npz

TEST INFRASTRUCTURE ONLY (header as oracle/_refimport.py)."""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle._refimport import import_reference  # noqa: E402

import_reference()
import cv2  # noqa: E402,F401
if "torch._six" not in sys.modules:  # removed from torch 2.x; nerfstudio_collate.py:26 only needs string_classes
    six = types.ModuleType("torch._six")
    six.string_classes = (str, bytes)
    sys.modules["torch._six"] = six
import nerfstudio.data.utils.dataloaders as DL  # noqa: E402
from nerfstudio.data.datasets.dynamic_dataset import DynamicDataset  # noqa: E402

n_cam, n_t = 5, 12
times = torch.linspace(0, 1, n_t).repeat(n_cam)[:, None]
ids = torch.arange(n_cam).repeat_interleave(n_t)[:, None]


class FakeDataset(DynamicDataset):  # isinstance check in _get_batch_list (:113)
    def __init__(self, pick_mode):
        self.pick_mode = pick_mode
        self.cameras = types.SimpleNamespace(times=times, ids=ids)

    def __len__(self):
        return n_cam * n_t

    def __getitem__(self, idx):
        return {"image_idx": idx}


res = {"times": times[:, 0].numpy(), "ids": ids[:, 0].numpy()}
for mode, k in (("normal", 20), ("randsteps", 20), ("randsteps", 33), ("lowfps", 30), ("lowfps", 20)):
    for seed in (0, 1):
        random.seed(seed)
        fake = types.SimpleNamespace(dataset=FakeDataset(mode), num_images_to_sample_from=k, num_workers=1)
        batch_list = DL.CacheDataloader._get_batch_list(fake)
        res[f"{mode}_{k}_{seed}"] = np.array([b["image_idx"] for b in batch_list])
out = os.path.join(ROOT, "tests", "golden", "g10c_pick.npz")
np.savez_compressed(out, **res)
print("wrote", out, {k: len(v) for k, v in res.items()})

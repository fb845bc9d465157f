"""G10b: the reference's own DynamicDataset.compute_isg (NS/data/datasets/dynamic_dataset.py:215-326) on the synthetic clip of G10
(tests/golden/g10_ist.npz: 4 cameras x 8 frames, one camera id with a single image).  Build container only:

    python oracle/gen_golden_isg.py      # writes tests/golden/g10b_isg.npz

TEST INFRASTRUCTURE ONLY (header as oracle/_refimport.py)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle._refimport import import_reference  # noqa: E402

import_reference()
import cv2  # noqa: E402,F401  (shim)
import nerfstudio.data.datasets.dynamic_dataset as DD  # noqa: E402

g10 = np.load(os.path.join(ROOT, "tests", "golden", "g10_ist.npz"))
u8 = torch.from_numpy(g10["images_u8"])
ids = torch.from_numpy(g10["cam_ids"])
tms = torch.from_numpy(g10["cam_times"])
imgs = u8.float() / 255.0  # what base_dataset.py:82 hands over
res = {}
for gamma in (5e-2, 2e-1):
    fake = types.SimpleNamespace(isg_gamma=gamma, eval_dataset=False, cameras=types.SimpleNamespace(times=tms[:, None], ids=ids[:, None]),
                                 _dataparser_outputs=None)
    w = DD.DynamicDataset.compute_isg(fake, {"image": imgs, "image_idx": torch.arange(imgs.shape[0])}, "cpu", offline=False)
    res["isg_" + str(gamma).replace(".", "_")] = w.float().numpy()
out = os.path.join(ROOT, "tests", "golden", "g10b_isg.npz")
np.savez_compressed(out, **res)
print("wrote", out, os.path.getsize(out) // 1024, "KiB", {k: v.shape for k, v in res.items()})

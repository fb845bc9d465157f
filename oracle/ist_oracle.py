"""CPU oracle for the IST weight maps -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates DynamicDataset.compute_ist (NS/data/datasets/dynamic_dataset.py:398-464) in plain PyTorch.  Pinned by golden vectors
captured by calling the reference method itself on a small synthetic clip (tests/golden/g10_ist.npz, oracle/gen_golden.py).
The pixel draw of DynamicBasedPixelSampler is RNG-specific (random.shuffle + torch.multinomial): `sample` restates its loop
(NS/data/pixel_samplers.py:369-411) with the random choices as explicit inputs -- per-image counts, the replacement flag of :400-402
and the sequential-removal distribution of torch.multinomial(replacement=False) follow the reference; which uniform lands on which
pixel is this build's own convention (inverse CDF), pinned kernel == oracle bit for bit (tests/test_gpu_ist.py)."""
import numpy as np
import torch


def sample(cdf, chosen, nnz, per_image: int, u):
    """cdf [M, H*W] fp32 inclusive prefix sums of the weight maps; chosen: image of each slot (the shuffled, non-empty images the
    reference's loop visits, :376-399); nnz [M] = len(torch.nonzero(weight_map)); u [n] uniform draws.  Slot j takes draws
    [j*per_image, min((j+1)*per_image, n)) (:393-397) -- without replacement iff nnz[image] >= the slot's draw count (:400-402).
    Returns flat pixel indices [n] and image ids [n]."""
    cdf = np.asarray(cdf, dtype=np.float32)
    u = np.asarray(u, dtype=np.float32)
    n, HW = len(u), cdf.shape[1]
    pix, img_of = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    for slot, img in enumerate(chosen):
        d0 = slot * per_image
        if d0 >= n:
            break
        cnt = min(per_image, n - d0)
        c = cdf[int(img)].astype(np.float64)
        without = int(nnz[int(img)]) >= cnt
        rem_i, rem_w, removed = [], [], 0.0
        total = c[HW - 1]
        for k in range(cnt):
            target = np.float64(u[d0 + k]) * (total - removed)
            lo, hi = 0, HW - 1
            while lo < hi:
                mid = (lo + hi) >> 1
                adj = sum(w for i, w in zip(rem_i, rem_w) if i <= mid) if without else 0.0
                if c[mid] - adj > target:
                    hi = mid
                else:
                    lo = mid + 1
            if without:
                w = c[lo] - (c[lo - 1] if lo > 0 else 0.0)
                rem_i.append(lo)
                rem_w.append(w)
                removed += w
            pix[d0 + k], img_of[d0 + k] = lo, int(img)
    return pix, img_of


def compute_ist(images, cam_ids, cam_times, ist_range: float, alpha: float = 0.15):
    """images [M,H,W,3] float32 in [0,1]; cam_ids, cam_times [M] -> fp16 [M,H,W]."""
    M, H, W = images.shape[:3]
    out = torch.zeros(M, H, W)
    ids, t = cam_ids.reshape(-1), cam_times.reshape(-1)
    for i in range(M):
        same = torch.where(ids == ids[i])[0]
        dt = (t[same] - t[i]).abs()
        close = same[(dt <= ist_range) & (dt > 0.01)]
        if len(close) == 0:
            out[i] = 1.0
            continue
        md = torch.zeros_like(images[i])
        for j in close:
            md = torch.maximum(md, (images[i] - images[j]).abs())
        md = md.mean(dim=2)
        out[i] = torch.where(md > alpha, md, torch.zeros_like(md))
    return out.to(torch.float16)


def compute_isg(images, cam_ids, isg_gamma: float = 5e-2):
    """DynamicDataset.compute_isg (NS/data/datasets/dynamic_dataset.py:284-317): images [M,H,W,3] float32 in [0,1]; cam_ids [M]
    -> fp16 [M,H,W].  Pinned by tests/golden/g10b_isg.npz (oracle/gen_golden_isg.py calls the reference method itself)."""
    ids = cam_ids.reshape(-1)
    out = torch.zeros(images.shape[:3])
    med = {int(c): torch.median(images[ids == c], dim=0).values for c in torch.unique(ids)}
    for i in range(images.shape[0]):
        sq = torch.square(images[i] - med[int(ids[i])])
        out[i] = (1.0 / 3) * torch.sum(sq / (sq + isg_gamma**2), dim=-1)
    return out.to(torch.float16)

"""CPU oracle for the IST weight maps -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates DynamicDataset.compute_ist (NS/data/datasets/dynamic_dataset.py:398-464) in plain PyTorch.  Pinned by golden vectors
captured by calling the reference method itself on a small synthetic clip (tests/golden/g10_ist.npz, oracle/gen_golden.py).
The pixel draw of DynamicBasedPixelSampler is RNG-specific (random.shuffle + torch.multinomial), so only its counts and
distribution are checked (tests/test_gpu_ist.py)."""
import torch


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

"""Full-image evaluation metrics the models report (NS/models/kplanes.py:291-293,454-498): PSNR with data_range 1 and SSIM as
`torchmetrics.functional.structural_similarity_index_measure` computes it with its defaults (Wang et al. 2004: 11x11 Gaussian window,
sigma 1.5, K1 0.01, K2 0.03, reflect padding cropped from the result, data_range = the larger of the two images' value spans when not
given).  torchmetrics is a third-party dependency absent from the reference tree and from this image: PARITY UNPINNED for SSIM (restated
from the paper and the library's documented defaults; CPU restatement in oracle/metrics_oracle.py).  LPIPS needs pretrained weights
(network) and is out of scope.

Evaluated once per rendered IMAGE on device tensors through torch's convolution (a library call, not a hot path)."""
from typing import Optional

import torch
import torch.nn.functional as F


def psnr(preds: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """PeakSignalNoiseRatio(data_range=1.0): 10 log10(data_range^2 / MSE) over all elements."""
    mse = torch.mean((preds.float() - target.float()) ** 2)
    return 10.0 * torch.log10(torch.as_tensor(data_range ** 2, device=mse.device) / mse)


def _gaussian_window(size: int, sigma: float, device, dtype) -> torch.Tensor:
    d = torch.arange((1 - size) / 2, (1 + size) / 2, step=1, dtype=dtype, device=device)
    g = torch.exp(-((d / sigma) ** 2) / 2)
    g = g / g.sum()
    return g[:, None] @ g[None, :]


def structural_similarity_index_measure(preds: torch.Tensor, target: torch.Tensor, kernel_size: int = 11, sigma: float = 1.5,
                                        data_range: Optional[float] = None, k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """preds, target [B,C,H,W] -> mean SSIM over the batch."""
    if preds.shape != target.shape or preds.dim() != 4:
        raise ValueError(f"expected two [B,C,H,W] tensors of the same shape, got {tuple(preds.shape)} and {tuple(target.shape)}")
    preds, target = preds.float(), target.float()
    if data_range is None:
        data_range = float(torch.maximum(preds.max() - preds.min(), target.max() - target.min()))
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    C = preds.shape[1]
    pad = (kernel_size - 1) // 2
    win = _gaussian_window(kernel_size, sigma, preds.device, preds.dtype)[None, None].expand(C, 1, kernel_size, kernel_size)
    p = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t], dim=0)
    out = F.conv2d(stack, win, groups=C)
    B = preds.shape[0]
    mu_p, mu_t, pp, tt, pt = (out[i * B:(i + 1) * B] for i in range(5))
    s_p, s_t, s_pt = pp - mu_p ** 2, tt - mu_t ** 2, pt - mu_p * mu_t
    ssim_full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_p + s_t + c2))
    ssim = ssim_full[..., pad:-pad, pad:-pad]  # drop the border the padding influenced
    return ssim.reshape(B, -1).mean(-1).mean()

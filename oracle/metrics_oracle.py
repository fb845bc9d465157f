"""CPU restatement (numpy, float64, direct window sums) of the SSIM the reference reports through
torchmetrics.functional.structural_similarity_index_measure (NS/models/kplanes.py:292,473): Wang et al. 2004 with the library's
defaults -- 11x11 Gaussian window (sigma 1.5), K1 = 0.01, K2 = 0.03, reflect padding by 5 whose border is cropped from the map again,
data_range = max value span of the two images when not given.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: torchmetrics is third-party, absent from /root/reference and from this image; the reference's
tests hold no SSIM values."""
import numpy as np


def ssim(preds: np.ndarray, target: np.ndarray, kernel_size: int = 11, sigma: float = 1.5, data_range=None, k1=0.01, k2=0.03) -> float:
    """[B,C,H,W] arrays -> mean SSIM."""
    p, t = preds.astype(np.float64), target.astype(np.float64)
    if data_range is None:
        data_range = max(p.max() - p.min(), t.max() - t.min())
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    d = np.arange(kernel_size) - (kernel_size - 1) / 2
    g = np.exp(-((d / sigma) ** 2) / 2)
    g /= g.sum()
    win = np.outer(g, g)
    pad = (kernel_size - 1) // 2
    pp = np.pad(p, ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode="reflect")
    tp = np.pad(t, ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode="reflect")
    H, W = p.shape[2:]

    def filt(a):  # valid correlation with the window: [B,C,H,W]
        out = np.zeros(p.shape)
        for i in range(kernel_size):
            for j in range(kernel_size):
                out += win[i, j] * a[:, :, i:i + H, j:j + W]
        return out

    mu_p, mu_t = filt(pp), filt(tp)
    s_p, s_t, s_pt = filt(pp * pp) - mu_p ** 2, filt(tp * tp) - mu_t ** 2, filt(pp * tp) - mu_p * mu_t
    m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_p + s_t + c2))
    return float(m[:, :, pad:-pad, pad:-pad].reshape(p.shape[0], -1).mean(-1).mean())

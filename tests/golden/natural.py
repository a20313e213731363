"""A deterministic "natural-ish" test frame: smooth gradients + hard edges + flat areas + 2 % noise.

Why: U[0,1) noise (the reference's rand_*_Input tensors) calibrates every ReLU-fed domain AND the image domain to zero point -128
and saturates the 16-bit bias constant on most channels; a real image has min > 0, so the reference's own calibration
(test.py:185-217) yields zero_0 < -128 together with its consistent scale -- a (scale, zero) pair no noise fixture has.

Only IEEE add / multiply / compare on float64 grids plus a seeded torch CPU generator (the same one bench.py's pool frames rely on):
no transcendental function, so the frame is the same bits wherever it is generated; every fixture records its SHA-256 and the tests
check it before they use the frame.  Used by tests/golden/make_golden.py (build container) and by the tests / bench (GPU box).
"""
import numpy as np


def natural_frame(channels: int, H: int, W: int, seed: int = 2024) -> np.ndarray:
    """(1, channels, H, W) float32 in [0.03, 0.97]."""
    import torch
    g = torch.Generator().manual_seed(seed)
    yy = (np.arange(H, dtype=np.float64) + 0.5)[:, None] / H
    xx = (np.arange(W, dtype=np.float64) + 0.5)[None, :] / W
    out = np.zeros((channels, H, W), np.float64)
    # geometry shared by the channels (objects), tint per channel
    nrect = 10
    r = torch.rand((nrect, 6), generator=g, dtype=torch.float64).numpy()
    tint = torch.rand((channels, nrect + 4), generator=g, dtype=torch.float64).numpy()
    for c in range(channels):
        t = tint[c]
        # smooth part: a linear ramp, a quadratic bowl and a bilinear term -- slopes differ per channel
        img = 0.20 + 0.45 * (t[0] * xx + (1.0 - t[0]) * yy) + 0.25 * (xx - t[1]) * (xx - t[1]) + 0.15 * (xx - 0.5) * (yy - t[2])
        # periodic texture without sin(): a triangle wave of period W/24 columns, amplitude 0.04
        ph = xx * 24.0 + yy * 3.0 * t[3]
        tri = np.abs(ph - np.floor(ph) - 0.5) * 2.0
        img = img + 0.04 * (tri - 0.5)
        for k in range(nrect):
            x0, y0, w, h, amp, kind = r[k]
            x1, y1 = x0 + 0.04 + 0.20 * w, y0 + 0.10 + 0.45 * h
            inside = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1)
            if kind < 0.3:      # flat area: one constant value per channel
                img = np.where(inside, 0.15 + 0.7 * t[4 + k], img)
            else:               # an object with hard edges on top of the gradient
                img = np.where(inside, img + (amp - 0.5) * 0.5 * (0.5 + t[4 + k]), img)
        out[c] = img
    noise = torch.randn((channels, H, W), generator=g, dtype=torch.float32).numpy().astype(np.float64)
    out = out + 0.02 * noise
    out = np.minimum(np.maximum(out, 0.03), 0.97)
    return out.astype(np.float32)[None]


def interesting_crop(x: np.ndarray, h: int, w: int, step: int = 4):
    """Top-left corner (y, x) of the h x w window of frame x (1, C, H, W) with the most hard-edge pixels (|horizontal or vertical step|
    > 0.06 in channel 0) -- an object boundary, a flat area and gradient inside one small crop.  Integer arithmetic on exact compares:
    deterministic."""
    a = x[0, 0].astype(np.float64)
    e = np.zeros(a.shape, np.int64)
    e[:, 1:] += (np.abs(a[:, 1:] - a[:, :-1]) > 0.06)
    e[1:, :] += (np.abs(a[1:, :] - a[:-1, :]) > 0.06)
    c = np.zeros((a.shape[0] + 1, a.shape[1] + 1), np.int64)
    c[1:, 1:] = e.cumsum(0).cumsum(1)
    best, arg = -1, (0, 0)
    for y in range(0, a.shape[0] - h + 1, step):
        for xx in range(0, a.shape[1] - w + 1, step):
            s = c[y + h, xx + w] - c[y, xx + w] - c[y + h, xx] + c[y, xx]
            if s > best:
                best, arg = s, (y, xx)
    return arg

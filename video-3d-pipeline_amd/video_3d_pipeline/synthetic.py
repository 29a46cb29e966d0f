"""Deterministic synthetic stereo input (SURVEY.md section 8d recipe).

There is no network for datasets, so tests, smoke() and bench.py all draw frames from here:
a blurred random texture viewed through a piecewise-planar disparity field (three
fronto-parallel rectangles at d = 12, 28, 44 over a 4 -> 20 ramp), squeezed into a
side-by-side BGR frame the way a 3D blu-ray stores it, plus a matching 2x guide frame.
"""
import numpy as np


def _blur(img, sigma):
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(img.astype(np.float32), sigma=(sigma, sigma, 0) if img.ndim == 3 else sigma)


def gt_disparity(W, H):
    """ground-truth disparity d*(x, y) in [1, 62], float32 HxW"""
    x = np.arange(W, dtype=np.float32)[None, :]
    d = np.broadcast_to(4.0 + 16.0 * x / max(W - 1, 1), (H, W)).copy()
    for (fx0, fx1, fy0, fy1, dv) in ((0.10, 0.35, 0.15, 0.55, 12.0), (0.40, 0.70, 0.30, 0.80, 28.0),
                                     (0.72, 0.92, 0.10, 0.45, 44.0)):
        d[int(fy0 * H):int(fy1 * H), int(fx0 * W):int(fx1 * W)] = dv
    return d


_pair_cache = {}


def stereo_pair(W, H, frame_idx=0, margin=64):
    """full-width BGR left/right views (HxWx3 u8) with L(x) ~ R(x - d*) (read-only: the last few pairs are memoised,
    sbs_frame and guide_frame of one index share the texture synthesis)"""
    key = (W, H, frame_idx, margin)
    hit = _pair_cache.get(key)
    if hit is None:
        hit = _stereo_pair(W, H, frame_idx, margin)
        for a in hit:
            a.setflags(write=False)
        if len(_pair_cache) >= 4:
            _pair_cache.pop(next(iter(_pair_cache)), None)
        _pair_cache[key] = hit
    return hit


def _stereo_pair(W, H, frame_idx, margin):
    rng = np.random.default_rng(1234 + frame_idx)
    T = _blur(rng.integers(0, 256, (H, W + 2 * margin, 3)), 1.5)
    T = np.clip((T - 127.5) * 2.5 + 127.5, 0, 255)           # restore contrast lost to the blur
    left = T[:, margin:margin + W]
    d = gt_disparity(W, H)
    xs = np.arange(W, dtype=np.float32)[None, :] + d + margin  # R(x) = T(x + d)
    x0 = np.floor(xs).astype(np.int64)
    w = (xs - x0)[..., None]
    x0 = np.clip(x0, 0, T.shape[1] - 2)
    rows = np.arange(H)[:, None]
    right = T[rows, x0] * (1 - w) + T[rows, x0 + 1] * w
    return np.rint(left).astype(np.uint8), np.rint(right).astype(np.uint8)


def sbs_frame(W, H, frame_idx=0):
    """side-by-side BGR frame HxWx3 u8: each eye squeezed to W/2 by 2:1 area averaging"""
    assert W % 2 == 0
    left, right = stereo_pair(W, H, frame_idx)

    def squeeze(a):
        a = a.astype(np.uint16)
        return ((a[:, 0::2] + a[:, 1::2] + 1) >> 1).astype(np.uint8)

    return np.ascontiguousarray(np.hstack([squeeze(left), squeeze(right)]))


def guide_frame(W, H, frame_idx=0, scale=2):
    """2x 'original 4K' luma guide (scale*H x scale*W u8) of the left view, + N(0,2) noise"""
    left, _ = stereo_pair(W, H, frame_idx)
    luma = (left[..., 2].astype(np.float32) * 0.299 + left[..., 1] * 0.587 + left[..., 0] * 0.114)
    from scipy.ndimage import zoom
    g = zoom(luma, scale, order=3, mode="nearest", grid_mode=True)
    rng = np.random.default_rng(99991 + frame_idx)
    g = g + rng.normal(0.0, 2.0, g.shape)
    return np.clip(np.rint(g), 0, 255).astype(np.uint8)


def gray_pair(W, H, frame_idx=0):
    """full-width gray left/right u8 (no SBS squeeze) -- the direct StereoSGBM input"""
    left, right = stereo_pair(W, H, frame_idx)

    def gray(a):
        return ((a[..., 2].astype(np.int32) * 9798 + a[..., 1].astype(np.int32) * 19235
                 + a[..., 0].astype(np.int32) * 3735 + (1 << 14)) >> 15).astype(np.uint8)

    return gray(left), gray(right)

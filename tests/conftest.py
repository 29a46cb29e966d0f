"""pytest configuration: `gpu` marker, import paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-3d-pipeline_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure): builds oracle/liboracle.so on first use"""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def native():
    """the product's C-ABI binding; GPU tests call the hot path only through this"""
    import torch
    from video_3d_pipeline import _native as N
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    N.lib()
    return N


def textured_pair(W, H, seed, max_disp=40):
    """random blurred texture with a smooth random disparity field; returns gray left/right u8"""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    T = gaussian_filter(rng.integers(0, 256, (H, W + 160)).astype(np.float32), 1.3)
    T = np.clip((T - 127.5) * 3.0 + 127.5, 0, 255)
    d = gaussian_filter(rng.uniform(2, max_disp, (H, W)).astype(np.float32), 12.0)
    left = T[:, 80:80 + W]
    xs = np.arange(W, dtype=np.float32)[None, :] + d + 80
    x0 = np.floor(xs).astype(np.int64)
    w = xs - x0
    rows = np.arange(H)[:, None]
    right = T[rows, x0] * (1 - w) + T[rows, x0 + 1] * w
    return np.rint(left).astype(np.uint8), np.rint(right).astype(np.uint8)


def mismatch_report(a, b, name):
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        return f"{name}: shape {a.shape} vs {b.shape}"
    bad = np.argwhere(a != b)
    if bad.size == 0:
        return ""
    first = tuple(bad[0])
    return (f"{name}: {len(bad)} of {a.size} elements differ ({100.0 * len(bad) / a.size:.4f}%), "
            f"first at {first}: got {a[first]} want {b[first]}; max abs diff "
            f"{np.abs(a.astype(np.int64) - b.astype(np.int64)).max()}")

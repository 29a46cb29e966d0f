"""Seeded fuzz of the guided upscale (sizes down to one pixel, every radius, odd guide sizes, non-integer scales) and of the
SBS split / Lanczos unsqueeze against the oracle (test aid; oracle/ is the checker).  usage: python tools/fuzz_guided.py [count]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np
from video_3d_pipeline import _native as N
from oracle import oracle as O
count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0; worst = 0.0
for seed in range(count):
    rng = np.random.default_rng(5000 + seed)
    Wlo, Hlo = int(rng.integers(1, 90)), int(rng.integers(1, 70))
    Wg, Hg = int(rng.integers(1, 260)), int(rng.integers(1, 200))
    if seed % 3 == 0: Wg, Hg = 2 * Wlo, 2 * Hlo
    r = int(rng.choice([4, 8, 8, 8, 1, 2, 3, 5, 7, 12, 16])); eps = float(10.0 ** rng.uniform(-4, -1))
    depth = (rng.uniform(0, 63, (Hlo, Wlo)) * (rng.random((Hlo, Wlo)) > 0.2)).astype(np.float32)
    guide = rng.integers(0, 256, (Hg, Wg), dtype=np.uint8)
    if seed % 4 == 1: guide[:] = int(rng.integers(0, 256))
    want = O.guided_upscale(depth, guide, r, eps)
    got = N.guided_upscale(N.to_device(depth), N.to_device(guide), r, eps).cpu().numpy().astype(np.float64)
    rng_ = float(want.max() - want.min()) or 1.0
    err = (np.abs(got - want) / np.maximum(np.abs(want), 1e-6 * rng_)).max()
    worst = max(worst, err)
    if not np.isfinite(got).all() or err > 1e-3:
        bad += 1; print("GUIDED MISMATCH seed", seed, (Wlo, Hlo), (Wg, Hg), r, eps, err)
print(f"guided fuzz {count} cases: mismatches = {bad}, worst rel err {worst:.2e}")
bad = 0
for seed in range(count):
    rng = np.random.default_rng(9000 + seed)
    W, H = 2 * int(rng.integers(1, 700)), int(rng.integers(1, 40))
    sbs = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if seed % 3 == 0: sbs = np.where(rng.random((H, W, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, axis=2)
    for unsq in (True, False):
        wl, wr = O.sbs_to_gray(sbs, unsq); gl, gr = N.sbs_to_gray(N.to_device(sbs), unsq)
        bl, br = O.split_sbs(sbs, unsq); cl, cr = N.split_sbs(N.to_device(sbs), unsq)
        if not (np.array_equal(gl.cpu().numpy(), wl) and np.array_equal(gr.cpu().numpy(), wr)
                and np.array_equal(cl.cpu().numpy(), bl) and np.array_equal(cr.cpu().numpy(), br)):
            bad += 1; print("SBS MISMATCH seed", seed, W, H, unsq)
print(f"sbs split fuzz {count} cases x 2 modes: mismatches = {bad}")

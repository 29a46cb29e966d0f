"""fused guided filter, int16 disparity + exact 2x: integer stage 1 (gf_int1 = 1) against the f64 stage 1 (0): bits and time"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
rng = np.random.default_rng(0)
for (Wlo, Hlo, r, band) in [(320, 180, 8, 432), (117, 75, 8, 16), (150, 48, 4, 40), (9, 10, 8, 432), (250, 150, 4, 64), (1920, 1080, 8, 432), (500, 350, 8, 100)]:
    d = rng.integers(-16, 1024, (2, Hlo, Wlo)).astype(np.int16); d[rng.random(d.shape) < 0.1] = -16
    g = rng.integers(0, 256, (2, 2 * Hlo, 2 * Wlo), dtype=np.uint8)
    dd, gg = N.to_device(d), N.to_device(g)
    N.set_option("gf_band", band)
    N.set_option("gf_int1", 0); a = N.guided_upscale_batch(dd, gg, r, 1e-3)
    N.set_option("gf_int1", 1); b = N.guided_upscale_batch(dd, gg, r, 1e-3)
    torch.cuda.synchronize()
    print(f"{2 * Wlo}x{2 * Hlo} r={r} band={band}:", "identical" if torch.equal(a, b) else f"MISMATCH {int((a != b).sum())} px, max {float((a - b).abs().max()):.3e}")
N.set_option("gf_band", 432)
W, H, B = 1920, 1080, int(os.environ.get("QB_BATCH", "34"))
disp = N.to_device(np.stack([(syn.gt_disparity(W, H) * 16).astype(np.int16)] * B)); guide = N.to_device(np.stack([syn.guide_frame(W, H, 0)] * B))
out = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda")
for rnd in range(3):
    for i1 in (0, 1):
        N.set_option("gf_int1", i1)
        for _ in range(2): N.guided_upscale_batch(disp, guide, 8, 1e-3, out)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(5): N.guided_upscale_batch(disp, guide, 8, 1e-3, out)
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
        print(f"gf_int1={i1}: {ms:.3f} ms / {B} frames = {ms / B * 1000:.1f} us/frame")

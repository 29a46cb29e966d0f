import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
W, H, B = 1920, 1080, int(os.environ.get("QB_BATCH", "30"))
depth = N.to_device(np.stack([syn.gt_disparity(W, H).astype(np.float32)] * B)); guide = N.to_device(np.stack([syn.guide_frame(W, H, 0)] * B))
out = torch.empty((B, 2*H, 2*W), dtype=torch.float32, device="cuda")
for b1, b2 in ((40, 270), (60, 270), (90, 270), (120, 270), (180, 270), (60, 180), (60, 540), (90, 135), (90, 360)):
    N.set_option("gf_band1", b1); N.set_option("gf_band2", b2)
    t = timeit(lambda: N.guided_upscale_batch(depth, guide, 8, 1e-3, out)); print(f"bands {b1}/{b2}: {t:.3f} ms / {B} frames = {t / B:.4f} ms/frame")

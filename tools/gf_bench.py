import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
W, H = 1920, 1080
depth = N.to_device(syn.gt_disparity(W, H).astype(np.float32)); guide = N.to_device(syn.guide_frame(W, H, 0))
out = torch.empty((2*H, 2*W), dtype=torch.float32, device="cuda")
for band in os.environ.get("BANDS", "24,32,48,64,96,128").split(","):
    os.environ["V3D_GF_BAND"] = band
    t = timeit(lambda: N.guided_upscale(depth, guide, 8, 1e-3, out)); print(f"band {band}: {t:.3f} ms")
os.environ["V3D_GF_TILED"] = "1"
t = timeit(lambda: N.guided_upscale(depth, guide, 8, 1e-3, out)); print(f"tiled: {t:.3f} ms")

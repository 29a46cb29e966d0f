"""SBS split + SGBM prefilter stage times under the library V3D_HIP_LIB selects (34 frames)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
import envopts; envopts.select_variant_lib(N)
W, H, B = 1920, 1080, 34
sbs = N.to_device(np.stack([syn.sbs_frame(W, H, i % 4) for i in range(B)]))
lg = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); rg = torch.empty_like(lg)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
best = min(t(lambda: N.sbs_to_gray_batch(sbs, True, (lg, rg))) for _ in range(3))
m = N.StereoSGBM(W, H, B); out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
pf = 1e9
for _ in range(3):
    m.profile(True)
    for _ in range(5): m.compute(lg, rg, out)
    torch.cuda.synchronize(); calls, st = m.read_profile(); pf = min(pf, st["prefilter"] / calls)
print(f"split+gray {best:.3f} ms   prefilter {pf:.3f} ms   (34 frames)")

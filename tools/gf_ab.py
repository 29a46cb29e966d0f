"""guided upscale (int16 disparity in, 4K float out) of one batch under the library V3D_HIP_LIB selects: ms per batch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
import envopts; envopts.select_variant_lib(N); envopts.apply_lib_options(N)
if os.environ.get('GF_BAND'): N.set_option('gf_band', int(os.environ['GF_BAND']))
W, H, B = 1920, 1080, int(os.environ.get("QB_BATCH", "34"))
disp = N.to_device(np.stack([(syn.gt_disparity(W, H) * 16).astype(np.int16)] * B)); guide = N.to_device(np.stack([syn.guide_frame(W, H, 0)] * B))
out = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda")
for _ in range(2): N.guided_upscale_batch(disp, guide, 8, 1e-3, out)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): N.guided_upscale_batch(disp, guide, 8, 1e-3, out)
    e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 5)
print(f"guided batch {B}: {best:.3f} ms = {best / B * 1000:.1f} us/frame  checksum {float(out[0].double().sum().item()):.6e}")

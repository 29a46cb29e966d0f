"""Guided upscale of one 30-frame batch, a few times (target program for rocprofv3 passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
import envopts; envopts.apply_lib_options(N)
W, H, B = 1920, 1080, int(os.environ.get("QB_BATCH", "30"))
depth = N.to_device(np.stack([syn.gt_disparity(W, H).astype(np.float32)] * B)); guide = N.to_device(np.stack([syn.guide_frame(W, H, 0)] * B))
out = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda")
for _ in range(2): N.guided_upscale_batch(depth, guide, 8, 1e-3, out)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(5): N.guided_upscale_batch(depth, guide, 8, 1e-3, out)
e1.record(); torch.cuda.synchronize(); print("guided batch", B, ":", e0.elapsed_time(e1) / 5, "ms", "checksum", float(out.double().sum().item()))

"""PCIe-inclusive rate of the NumPy surface: host SBS frames in -> host float32 depth / 4K depth out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch, io, contextlib
from video_3d_pipeline import synthetic as syn
from video_3d_pipeline.depth import HybridStereoDepthExtractor
from video_3d_pipeline.upscale import SimpleDepthUpscaler
W, H, B = 1920, 1080, 8
frames = [syn.sbs_frame(W, H, i % 2) for i in range(B)]
guide = syn.guide_frame(W, H, 0)
with contextlib.redirect_stdout(io.StringIO()):
    ex = HybridStereoDepthExtractor(work_dir="/tmp/hr", cache_dir="/tmp/hr", stereo_only=True)
    up = SimpleDepthUpscaler()
be = ex.backend
def fused():
    d = be.sbs_to_disparity(frames, True)          # H2D 8 x 6.2 MB, kernels, result stays on device
    return be.depth_to_host(d)                     # D2H 8 x 8.3 MB through pinned memory
for _ in range(2): fused()
t0 = time.perf_counter(); n = 5
for _ in range(n): out = fused()
t1 = time.perf_counter()
print(f"sbs_to_disparity (host in, host f32 out): {(t1 - t0) / n * 1e3:.1f} ms / {B} frames = {B * n / (t1 - t0):.1f} fps")
with contextlib.redirect_stdout(io.StringIO()):
    pairs = [ex.split_sbs_frame(f, True) for f in frames]
    for _ in range(2): ex.process_frame_batch(pairs)
    t0 = time.perf_counter()
    for _ in range(n): ex.process_frame_batch(pairs)
    t1 = time.perf_counter()
print(f"process_frame_batch (reference NumPy surface, BGR pairs in, f32 out): {(t1 - t0) / n * 1e3:.1f} ms / {B} pairs = {B * n / (t1 - t0):.1f} fps")
d0 = out[0]
for _ in range(2): up.upscale_frame(d0, guide)
t0 = time.perf_counter()
for _ in range(n): q = up.upscale_frame(d0, guide)
t1 = time.perf_counter()
print(f"upscale_frame (host depth + 4K guide in, host 4K f32 out): {(t1 - t0) / n * 1e3:.1f} ms/frame")

"""End-to-end rate of the file-to-file CLI paths on synthetic clips (decode from .npy, GPU path, 16-bit PNG out):
   depth.py: SBS clip -> depth_%06d.png;  upscale.py: depth PNGs + 4K clip -> depth4k_%06d.png."""
import os, sys, time, shutil, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np
from video_3d_pipeline import synthetic as syn
from video_3d_pipeline.depth import HybridStereoDepthExtractor
from video_3d_pipeline.upscale import SimpleDepthUpscaler
W, H = 1920, 1080
N = int(os.environ.get("CLI_FRAMES", "120")); NU = int(os.environ.get("CLI_UP_FRAMES", "24"))
work = "/tmp/cli_rate"; shutil.rmtree(work, ignore_errors=True); os.makedirs(work)
base = [syn.sbs_frame(W, H, i) for i in range(4)]
clip = os.path.join(work, "sbs.npy"); np.save(clip, np.stack([base[i % 4] for i in range(N)]))
g = [syn.guide_frame(W, H, i) for i in range(2)]
clip4k = os.path.join(work, "v4k.npy"); np.save(clip4k, np.stack([np.repeat(g[i % 2][..., None], 3, axis=2) for i in range(NU)]))
for bs in (8, 30):
    with contextlib.redirect_stdout(io.StringIO()):
        ex = HybridStereoDepthExtractor(work_dir=work, cache_dir=work, stereo_only=True, batch_size=bs)
        ex.process_video_sbs(clip, max_frames=bs, force_reprocess=True)        # warm-up (library load, workspace)
        t0 = time.perf_counter()
        out = ex.process_video_sbs(clip, force_reprocess=True)
        t1 = time.perf_counter()
    print(f"depth CLI path, batch {bs}: {N} frames in {t1 - t0:.2f} s = {N / (t1 - t0):.1f} fps  ({len(os.listdir(out))} PNGs)")
with contextlib.redirect_stdout(io.StringIO()):
    up = SimpleDepthUpscaler()
    ddir = os.path.join(work, "d24"); os.makedirs(ddir)
    for i, f in enumerate(sorted(os.listdir(out))[:NU]): shutil.copy(os.path.join(out, f), os.path.join(ddir, f"depth_{i:06d}.png"))
    up.process_depth_upscaling(ddir, clip4k, output_path=os.path.join(work, "w.mp4"), force_reprocess=True)
    t0 = time.perf_counter()
    up.process_depth_upscaling(ddir, clip4k, output_path=os.path.join(work, "o.mp4"), force_reprocess=True)
    t1 = time.perf_counter()
print(f"upscale CLI path: {NU} frames in {t1 - t0:.2f} s = {NU / (t1 - t0):.1f} fps")

"""where the upscale CLI path spends its wall time (cProfile of process_depth_upscaling on a synthetic clip)"""
import os, sys, time, shutil, io, contextlib, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np
from video_3d_pipeline import synthetic as syn
from video_3d_pipeline.upscale import SimpleDepthUpscaler
from video_3d_pipeline.utils import write_png16
W, H, NU = 1920, 1080, int(os.environ.get("CLI_UP_FRAMES", "32"))
work = "/tmp/up_prof"; shutil.rmtree(work, ignore_errors=True); os.makedirs(work + "/d")
rng = np.random.default_rng(0)
base = (syn.gt_disparity(W, H) * 1000).astype(np.uint16)
for i in range(NU): write_png16(f"{work}/d/depth_{i:06d}.png", base + i)
g = [syn.guide_frame(W, H, i) for i in range(2)]
np.save(work + "/v4k.npy", np.stack([np.repeat(g[i % 2][..., None], 3, axis=2) for i in range(NU)]))
with contextlib.redirect_stdout(io.StringIO()):
    up = SimpleDepthUpscaler()
    up.process_depth_upscaling(work + "/d", work + "/v4k.npy", output_path=work + "/w.json", force_reprocess=True)
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    up.process_depth_upscaling(work + "/d", work + "/v4k.npy", output_path=work + "/o.json", force_reprocess=True)
    pr.disable(); t1 = time.perf_counter()
print(f"{NU} frames in {t1 - t0:.3f} s = {NU / (t1 - t0):.1f} fps")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[-3500:])

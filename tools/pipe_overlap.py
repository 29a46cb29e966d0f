"""can the guided upscale of batch i-1 (f64-VALU bound) hide under the SGBM of batch i (HBM bound)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
W, H, B = 1920, 1080, 30
sbs = N.to_device(np.stack([syn.sbs_frame(W, H, i % 2) for i in range(B)]))
guide = N.to_device(np.stack([syn.guide_frame(W, H, i % 2) for i in range(B)]))
lg = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); rg = torch.empty_like(lg)
disp = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
depth = [torch.empty((B, H, W), dtype=torch.float32, device="cuda") for _ in range(2)]
out4k = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda")
m = N.StereoSGBM(W, H, B)
def front(i):
    N.sbs_to_gray_batch(sbs, True, (lg, rg)); m.compute(lg, rg, disp); N.disp_to_depth(disp, depth[i & 1])
def back(i):
    N.guided_upscale_batch(depth[i & 1], guide, 8, 1e-3, out4k)
def serial(n):
    for i in range(n): front(i); back(i)
side = torch.cuda.Stream()
def piped(n):
    main = torch.cuda.current_stream()
    ev_f = [None] * n; ev_b = [None] * n
    for i in range(n):
        if i >= 2: main.wait_event(ev_b[i - 2])          # depth[i&1] free again
        front(i)
        ev_f[i] = torch.cuda.Event(); ev_f[i].record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_f[i]); back(i)
            ev_b[i] = torch.cuda.Event(); ev_b[i].record(side)
    main.wait_stream(side)
def timeit(fn, n=10):
    fn(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    fn(n); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for r in range(2):
    print(f"serial {timeit(serial):.3f} ms/step   piped {timeit(piped):.3f} ms/step   lockstep errors {m.sync_errors()}")

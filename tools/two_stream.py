"""does running two half-batches on two streams overlap the memory-bound and the instruction-bound kernels?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
W, H, B = 1920, 1080, 30
L, R = syn.gray_pair(W, H, 0)
Ld = N.to_device(np.stack([L] * B)); Rd = N.to_device(np.stack([R] * B))
out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
def timeit(fn, n=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
m = N.StereoSGBM(W, H, B, options={"vdd_dpl": 8})          # 2 x 225 lock-step workgroups stay co-resident
t1 = timeit(lambda: m.compute(Ld, Rd, out)); print(f"one stream, batch 30: {t1:.3f} ms  ({t1 / B:.4f} ms/frame)")
ref = out.clone(); m.close()
h = B // 2
ms = [N.StereoSGBM(W, H, h, options={"vdd_dpl": 8}) for _ in range(2)]
ss = [torch.cuda.Stream() for _ in range(2)]
def two():
    cur = torch.cuda.current_stream()
    for i in range(2):
        ss[i].wait_stream(cur)
        with torch.cuda.stream(ss[i]):
            ms[i].compute(Ld[i * h:(i + 1) * h], Rd[i * h:(i + 1) * h], out[i * h:(i + 1) * h])
    for i in range(2): cur.wait_stream(ss[i])
t2 = timeit(two); print(f"two streams, 2 x 15: {t2:.3f} ms  ({t2 / B:.4f} ms/frame)  same result: {bool((out == ref).all())}  errors: {[x.sync_errors() for x in ms]}")

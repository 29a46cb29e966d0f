"""scratch timing of the individual stages (development aid; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

W, H = 1920, 1080
B = int(os.environ.get("QB_BATCH", "8"))
print("device", torch.cuda.get_device_name(0))
L, R = syn.gray_pair(W, H, 0)
Ld = N.to_device(np.stack([L] * B)); Rd = N.to_device(np.stack([R] * B))
m = N.StereoSGBM(W, H, B)
print("workspace GB", m.workspace_bytes / 1e9)
out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
d = m.compute(Ld[0].contiguous(), Rd[0].contiguous())
dn = d.cpu().numpy()
print("valid frac", (dn >= 0).mean(), "mean disp", dn[dn >= 0].mean() / 16, "checksum", int(dn.astype(np.int64).sum()))
for b in sorted({1, 2, B}):
    l, r = Ld[:b].contiguous(), Rd[:b].contiguous()
    t = timeit(lambda: m.compute(l, r, out[:b]))
    print(f"sgbm batch {b}: {t:.3f} ms/batch  {t / b:.3f} ms/frame  {1e3 * b / t:.1f} fps  alg {1.2911616 * b / t:.2f} TB/s")
sbs = N.to_device(syn.sbs_frame(W, H, 0))
t = timeit(lambda: N.sbs_to_gray(sbs, True)); print(f"sbs_to_gray: {t:.3f} ms")
depth = N.disp_to_depth(d)
t = timeit(lambda: N.disp_to_depth(d)); print(f"disp_to_depth: {t:.3f} ms")
guide = N.to_device(syn.guide_frame(W, H, 0))
t = timeit(lambda: N.guided_upscale(depth, guide, 8, 1e-3)); print(f"guided_upscale 4K: {t:.3f} ms  alg {0.1907712 / t:.2f} TB/s")
t = timeit(lambda: N.depth_to_u16(depth)); print(f"depth_to_u16: {t:.3f} ms")
h, w, C = 270, 480, 256
fl = torch.randn((h, w, C), device="cuda").to(torch.bfloat16); fr = torch.randn((h, w, C), device="cuda").to(torch.bfloat16)
flow = torch.rand((2, h, w), device="cuda") * 4 - 2
t = timeit(lambda: N.corr_lookup(fl, fr, flow, 4, 0)); print(f"corr 1x9: {t:.3f} ms")

"""Fused guided filter (k_gff: a/b handed stage-1 -> stage-2 waves through LDS) vs the two-sweep form (a/b through HBM):
bit-identity of the outputs and time per batch, over band heights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
W, H, B = 1920, 1080, int(os.environ.get("QB_BATCH", "30"))
rng = np.random.default_rng(0)
d = syn.gt_disparity(W, H).astype(np.float32); d[rng.random(d.shape) < 0.1] = 0.0
depth = N.to_device(np.stack([d] * B)); guide = N.to_device(np.stack([syn.guide_frame(W, H, i % 2) for i in range(B)]))
ref = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda"); out = torch.empty_like(ref)
N.set_option("gf_fused", 0)
t0 = timeit(lambda: N.guided_upscale_batch(depth, guide, 8, 1e-3, ref)); print(f"two sweeps      : {t0:.3f} ms / {B} frames")
N.set_option("gf_fused", 1)
for cols, band in [(c, int(x)) for c in (256, 512) for x in os.environ.get("BANDS", "135,180,270,360,540,1080,2160").split(",")]:
    N.set_option("gf_cols", cols); N.set_option("gf_band", band); out.zero_()
    t = timeit(lambda: N.guided_upscale_batch(depth, guide, 8, 1e-3, out))
    rel = ((out.double() - ref.double()).abs() / ref.double().abs().clamp_min(1e-6 * float(ref.max()))).max().item()
    print(f"fused, {cols} cols, band {band:4d}: {t:.3f} ms / {B} frames   identical to two-sweep: {bool(torch.equal(out, ref))}  max rel diff {rel:.2e}")
for r in (4,):
    N.set_option("gf_fused", 0); N.guided_upscale_batch(depth[:2], guide[:2], r, 1e-3, ref[:2])
    N.set_option("gf_fused", 1); N.set_option("gf_cols", 256); N.set_option("gf_band", 432); N.guided_upscale_batch(depth[:2], guide[:2], r, 1e-3, out[:2])
    print(f"r = {r}: identical {bool(torch.equal(out[:2], ref[:2]))}")

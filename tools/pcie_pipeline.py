"""PCIe-inclusive rate of the full path with the copies overlapped: SBS frames + 4K guides start in PINNED HOST memory,
the float32 4K depth ends in pinned host memory.  Three streams (H2D, compute, D2H), double-buffered device tensors:
step s+1's inputs upload and step s-1's result downloads while step s computes.  Never reported as bench.py's `value`
(that is HBM-resident by contract); DESIGN.md quotes this figure as the host-link-inclusive rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
W, H, S = 1920, 1080, 2
B = int(os.environ.get("QB_BATCH", "30")); STEPS = int(os.environ.get("STEPS", "12"))
dev = torch.device("cuda", 0)
base_sbs = [syn.sbs_frame(W, H, i) for i in range(2)]; base_g = [syn.guide_frame(W, H, i, S) for i in range(2)]
h_sbs = torch.from_numpy(np.stack([base_sbs[i % 2] for i in range(B)])).pin_memory()
h_gui = torch.from_numpy(np.stack([base_g[i % 2] for i in range(B)])).pin_memory()
h_out = [torch.empty((B, H * S, W * S), dtype=torch.float32).pin_memory() for _ in range(2)]
d_sbs = [torch.empty_like(h_sbs, device=dev) for _ in range(2)]
d_gui = [torch.empty_like(h_gui, device=dev) for _ in range(2)]
d_out = [torch.empty((B, H * S, W * S), dtype=torch.float32, device=dev) for _ in range(2)]
lg = torch.empty((B, H, W), dtype=torch.uint8, device=dev); rg = torch.empty_like(lg)
disp = torch.empty((B, H, W), dtype=torch.int16, device=dev); depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
m = N.StereoSGBM(W, H, B)
s_in, s_out, s_cmp = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)

def run(nsteps):
    in_ready = [None, None]; cmp_done = [None, None]; out_done = [None, None]
    def upload(s):
        k = s & 1
        with torch.cuda.stream(s_in):
            if cmp_done[k] is not None: s_in.wait_event(cmp_done[k])          # step s-2 has consumed this input slot
            d_sbs[k].copy_(h_sbs, non_blocking=True); d_gui[k].copy_(h_gui, non_blocking=True)
            e = torch.cuda.Event(); e.record(s_in); in_ready[k] = e
    upload(0)
    for s in range(nsteps):
        k = s & 1
        if s + 1 < nsteps: upload(s + 1)
        s_cmp.wait_event(in_ready[k])
        if out_done[k] is not None: s_cmp.wait_event(out_done[k])              # step s-2's result has left this slot
        N.sbs_to_gray_batch(d_sbs[k], True, (lg, rg)); m.compute(lg, rg, disp); N.disp_to_depth(disp, depth)
        N.guided_upscale_batch(depth, d_gui[k], 8, 1e-3, d_out[k])
        e = torch.cuda.Event(); e.record(s_cmp); cmp_done[k] = e
        with torch.cuda.stream(s_out):
            s_out.wait_event(e)
            h_out[k].copy_(d_out[k], non_blocking=True)
            e2 = torch.cuda.Event(); e2.record(s_out); out_done[k] = e2
    torch.cuda.synchronize()

run(3)
t0 = time.perf_counter(); run(STEPS); t1 = time.perf_counter()
fps = B * STEPS / (t1 - t0)
mb = (h_sbs.numel() + h_gui.numel() + h_out[0].numel() * 4) / B / 1e6
print(f"host-pinned in -> host-pinned out, copies overlapped: {STEPS} steps x {B} frames in {t1 - t0:.3f} s = {fps:.0f} frames/s "
      f"({mb:.1f} MB over PCIe per frame = {fps * mb / 1e3:.1f} GB/s both directions, lock-step timeouts {m.sync_errors()})")
# sequential reference: same work, one stream, blocking copies
def seq(n):
    for _ in range(n):
        d_sbs[0].copy_(h_sbs); d_gui[0].copy_(h_gui)
        N.sbs_to_gray_batch(d_sbs[0], True, (lg, rg)); m.compute(lg, rg, disp); N.disp_to_depth(disp, depth)
        N.guided_upscale_batch(depth, d_gui[0], 8, 1e-3, d_out[0]); h_out[0].copy_(d_out[0])
    torch.cuda.synchronize()
seq(2); t0 = time.perf_counter(); seq(6); t1 = time.perf_counter()
print(f"same, not overlapped: {B * 6 / (t1 - t0):.0f} frames/s")

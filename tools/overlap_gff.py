"""Does the guided filter of step s hide behind the horizontal SGM pass of step s+1?  k_gff is latency / VALU bound and moves
little HBM traffic, k_hfused is HBM bound with VALU slack: two streams, the filter ordered behind the lock-step pass (k_vdd needs
every CU slot) of the next step.  Prints ms per step serial and pipelined."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
import envopts; envopts.select_variant_lib(N)
W, H, B = 1920, 1080, int(os.environ.get("QB_BATCH", "34")); STEPS = int(os.environ.get("STEPS", "12"))
sbs = N.to_device(np.stack([syn.sbs_frame(W, H, i % 4) for i in range(B)])); guide = N.to_device(np.stack([syn.guide_frame(W, H, i % 4) for i in range(B)]))
lg = torch.empty((B, H, W), dtype=torch.uint8, device="cuda"); rg = torch.empty_like(lg)
disp = [torch.empty((B, H, W), dtype=torch.int16, device="cuda") for _ in range(2)]
out = [torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device="cuda") for _ in range(2)]
m = N.StereoSGBM(W, H, B, options={'hf_persist': int(os.environ.get('HF_PERSIST', '1'))})
main, side = torch.cuda.current_stream(), torch.cuda.Stream(priority=int(os.environ.get('SIDE_PRIO', '0')))

def sgbm(k):
    N.sbs_to_gray_batch(sbs, True, (lg, rg)); m.compute(lg, rg, disp[k])

def serial(n):
    for s in range(n):
        sgbm(s & 1); N.guided_upscale_batch(disp[s & 1], guide, 8, 1e-3, out[s & 1])

def pipelined(n):
    gdone = [None, None]; ready = None
    for s in range(n + 1):
        if s < n:
            if gdone[s & 1] is not None: main.wait_event(gdone[s & 1])      # filter of step s-2 has released disp[s & 1]; (and of s-1: below)
            if s >= 1 and gdone[(s - 1) & 1] is not None and False: pass
            sgbm(s & 1)
            ev = torch.cuda.Event(); ev.record(main)
        if s >= 1:                                                            # filter of step s-1 behind the lock-step pass of step s
            k = (s - 1) & 1
            side.wait_event(ready)
            if s < n: m.stream_wait_lockstep(side)
            with torch.cuda.stream(side):
                N.guided_upscale_batch(disp[k], guide, 8, 1e-3, out[k])
                g = torch.cuda.Event(); g.record(side); gdone[k] = g
            if s < n: main.wait_event(g) if False else None
        if s < n: ready = ev
    main.wait_event(gdone[(n - 1) & 1])

def timeit(fn, n):
    fn(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); fn(n); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rnd in range(3):
    a = timeit(serial, STEPS); b = timeit(pipelined, STEPS)
    print(f"batch {B}: serial {a:.3f} ms/step ({B / a * 1000:.0f} fps)   pipelined {b:.3f} ms/step ({B / b * 1000:.0f} fps)   timeouts {m.sync_errors()}")
ref = [o.clone() for o in out]; serial(2); torch.cuda.synchronize()
print("outputs equal:", all(torch.equal(a, b) for a, b in zip(ref, out)))

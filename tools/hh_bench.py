"""MODE_HH (8 paths) timing: lock-step bottom-up pass vs three k_chain launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
W, H, B = 1920, 1080, int(os.environ.get("QB_BATCH", "30"))
L, R = syn.gray_pair(W, H, 0)
Ld = N.to_device(np.stack([L] * B)); Rd = N.to_device(np.stack([R] * B))
out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
ref = None
for vdd in (1, 0):
    m = N.StereoSGBM(W, H, B, mode=1, options={"lockstep": vdd})
    for _ in range(2): m.compute(Ld, Rd, out)
    torch.cuda.synchronize(); m.profile(True)
    for _ in range(5): m.compute(Ld, Rd, out)
    torch.cuda.synchronize()
    calls, st = m.read_profile()
    chk = int(out.to(torch.int64).sum().item()); ref = ref or chk
    tot = sum(st.values()) / calls
    print(f"HH VDD={vdd}: {tot:.3f} ms/batch = {tot / B:.4f} ms/frame  " + " ".join(f"{k.replace('chain_', '')}={x / calls:.3f}" for k, x in st.items() if x / calls > 0.01), "OK" if chk == ref else "MISMATCH", "errors", m.sync_errors())
    m.close()

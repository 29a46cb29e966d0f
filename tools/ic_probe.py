"""does a cost-volume working set that fits the 256 MB Infinity Cache run the chain kernels faster?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
W = 1920
for (H, B) in ((1080, 8), (64, 4), (32, 8), (32, 4), (128, 2), (256, 1)):
    L, R = syn.gray_pair(W, 1080, 0)
    L, R = L[:H], R[:H]
    Ld = N.to_device(np.stack([L] * B)); Rd = N.to_device(np.stack([R] * B))
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    m = N.StereoSGBM(W, H, B)
    for _ in range(3): m.compute(Ld, Rd, out)
    torch.cuda.synchronize(); m.profile(True)
    for _ in range(20): m.compute(Ld, Rd, out)
    torch.cuda.synchronize()
    calls, st = m.read_profile()
    V = (W - 64) * H * 64 * 2 * B
    def bw(name, nv): return V * nv / (st[name] / calls * 1e-3) / 1e12
    print(f"H={H:5d} B={B} C+S={2 * V / 1e6:7.1f} MB | cost {st['cost'] / calls * 1e3:7.1f} us {bw('cost', 1):5.2f} TB/s | v2 {st['chain_v2'] / calls * 1e3:7.1f} us {bw('chain_v2', 2):5.2f} | d1 {st['chain_d1'] / calls * 1e3:7.1f} us {bw('chain_d1', 3):5.2f} | d3 {st['chain_d3'] / calls * 1e3:7.1f} us {bw('chain_d3', 3):5.2f} | hf {st['chain_h4_wta'] / calls * 1e3:7.1f} us {bw('chain_h4_wta', 3):5.2f}")
    m.close()

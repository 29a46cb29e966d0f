"""A/B of SGBM variants in ONE process: VARIANTS="HFUSED=1,DPL=8;..." -> v3d_sgbm_set_option (tools/envopts.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
import envopts
envopts.select_variant_lib(N)
W, H = int(os.environ.get("QB_W", "1920")), int(os.environ.get("QB_H", "1080"))
B = int(os.environ.get("QB_BATCH", "8"))
L, R = syn.gray_pair(W, H, 0)
Ld = N.to_device(np.stack([L] * B)); Rd = N.to_device(np.stack([R] * B))
out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
variants = [v for v in os.environ.get("VARIANTS", "HFUSED=0,DPL=8;HFUSED=1,DPL=8;HFUSED=1,DPL=4;HFUSED=0,DPL=4").split(";")]
ref = None
ms = {}
stage_min = {}
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for v in variants:
        env = dict(os.environ)
        for kv in v.split(","):
            k, val = kv.split("="); env["V3D_" + ("CHAIN_DPL" if k == "DPL" else k)] = val
        m = N.StereoSGBM(W, H, B, options=envopts.sgbm_options(env))
        for _ in range(2): m.compute(Ld, Rd, out)
        torch.cuda.synchronize()
        m.profile(True)
        for _ in range(5): m.compute(Ld, Rd, out)
        torch.cuda.synchronize()
        calls, st = m.read_profile()
        chk = int(out.to(torch.int64).sum().item())
        if ref is None: ref = chk
        tot = sum(st.values()) / calls
        ms.setdefault(v, []).append(tot)
        for k, x in st.items(): stage_min.setdefault(v, {})[k] = min(stage_min.get(v, {}).get(k, 1e9), x / calls)
        if rnd == 0:
            print(f"{v:22s} total {tot:7.3f} ms/batch  " + " ".join(f"{k.replace('chain_', '')}={x / calls:.3f}" for k, x in st.items() if x / calls > 0.01) + ("  OK" if chk == ref else "  MISMATCH") + f" chk={chk}")
        m.close()
for v in variants: print(f"{v:22s} min {min(ms[v]):.3f} ms/batch -> {min(ms[v]) / B:.3f} ms/frame   stage minima: " + " ".join(f"{k.replace('chain_', '')}={x:.3f}" for k, x in stage_min[v].items() if x > 0.01))

timeout -k 10 300 python -m pytest tests/test_guided_gpu.py tests/test_golden_gpu.py tests/test_host_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
python - <<'PY'
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/video-3d-pipeline_amd')
import numpy as np, torch
from video_3d_pipeline import _native as N, synthetic as syn
W, H, B = 1920, 1080, 30
depth = N.to_device(np.stack([syn.gt_disparity(W, H).astype(np.float32)] * B)); guide = N.to_device(np.stack([syn.guide_frame(W, H, 0)] * B))
out = torch.empty((B, 2*H, 2*W), dtype=torch.float32, device="cuda")
for _ in range(2): N.guided_upscale_batch(depth, guide, 8, 1e-3, out)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(10): N.guided_upscale_batch(depth, guide, 8, 1e-3, out)
e1.record(); torch.cuda.synchronize(); print("guided batch 30:", e0.elapsed_time(e1) / 10, "ms")
PY

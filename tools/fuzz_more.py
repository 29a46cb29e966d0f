"""Extra seeds of tests/test_sgbm_gpu.py's fuzz generator against the oracle (bug hunting beyond the committed 28 cases;
uses oracle/ as the checker, so it is a test aid, not product code).  usage: python tools/fuzz_more.py [first] [count]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
import numpy as np
import test_sgbm_gpu as T
from video_3d_pipeline import _native as N
from oracle import oracle as O
bad = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 28
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for seed in range(first, first + count):
    W, H, L, R, kw = T._fuzz_case(seed)
    want = O.sgbm_compute(L, R, O.default_params(**kw))
    m = N.StereoSGBM(max_width=W, max_height=H, **kw)
    got = m.compute(N.to_device(L), N.to_device(R)).cpu().numpy()
    e = m.sync_errors(); m.close()
    if e or not np.array_equal(got, want):
        bad += 1; print("MISMATCH seed", seed, W, H, kw, "errs", e, "ndiff", int((got != want).sum()))
print(f"fuzz seeds {first}..{first + count - 1}: mismatches = {bad}")

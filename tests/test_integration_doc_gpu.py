"""INTEGRATION.md section B is the text a reference maintainer would paste into the reference (a ctypes stub over
libv3d_hip.so at the OpenCV call sites of depth.py:315-341, 344-374).  This test extracts those Python blocks VERBATIM,
executes them against the built library and checks compute() / blend() against the oracle: the document cannot rot."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _section_b_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):text.index("## Entry points")]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def test_integration_md_binding_runs_and_matches_the_oracle(native, oracle):
    import torch
    blocks = _section_b_blocks()
    assert len(blocks) == 2, "INTEGRATION.md section B: expected the SGBM stub and the blend stub"
    ns = {}
    for b in blocks:
        assert '"libv3d_hip.so"' in b or "_lib." in b
        exec(compile(b.replace('"libv3d_hip.so"', repr(native.lib_path())), "INTEGRATION.md", "exec"), ns)
    from video_3d_pipeline import synthetic as syn
    W, H = 320, 180
    left, right = oracle.sbs_to_gray(syn.sbs_frame(W, H, 3), True)
    want = oracle.sgbm_compute(left, right)
    h = ns["StereoSGBM_create"](W, H, 2, uniquenessRatio=10)            # keyword arguments as at depth.py:315-325
    lg = torch.from_numpy(np.stack([left, left])).cuda()
    rg = torch.from_numpy(np.stack([right, right])).cuda()
    got = ns["compute"](h, lg, rg)
    torch.cuda.synchronize()
    assert got.dtype == torch.int16 and tuple(got.shape) == (2, H, W)
    assert np.array_equal(got[0].cpu().numpy(), want) and np.array_equal(got[1].cpu().numpy(), want)
    mono = np.random.default_rng(1).random((48, 64)).astype(np.float32) * 9 + 1          # stands for DPT's predicted_depth[0]
    blend = ns["blend"](got[0].contiguous(), torch.from_numpy(mono).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(blend.cpu().numpy(), oracle.mono_blend(want, mono))
    ns["_lib"].v3d_sgbm_destroy(h)

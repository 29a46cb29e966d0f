"""Generates tests/golden/*.npz -- small input/expected-output vectors for the hot path.

PARITY UNPINNED: the reference holds no tests or fixtures and its arithmetic
lives in OpenCV, which cannot be imported in the build container, so these vectors are produced by the
repo's own CPU oracle (oracle/liboracle.so) on seeded inputs.  They pin the oracle against accidental
change and give the GPU tests a frozen target that does not depend on rebuilding the oracle.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import textured_pair  # noqa: E402
from oracle import oracle as O  # noqa: E402
from video_3d_pipeline import synthetic as syn  # noqa: E402


def main():
    for (W, H, seed) in ((160, 96, 1), (320, 180, 2)):
        L, R = textured_pair(W, H, seed)
        raw = O.sgbm_raw(L, R)
        disp = O.sgbm_compute(L, R)
        disp_hh = O.sgbm_compute(L, R, O.default_params(mode=1))
        np.savez_compressed(os.path.join(HERE, f"sgbm_{W}x{H}.npz"), left=L, right=R, raw=raw, disp=disp, disp_hh=disp_hh)
    # SBS pre-stage + depth post-stage
    sbs = syn.sbs_frame(192, 64, 3)
    gl, gr = O.sbs_to_gray(sbs, True)
    sl, sr = O.sbs_to_gray(sbs, False)
    d = O.sgbm_compute(gl, gr)
    dep = O.disp_to_depth(d)
    np.savez_compressed(os.path.join(HERE, "prepost_192x64.npz"), sbs=sbs, left_gray=gl, right_gray=gr,
                        left_gray_squeezed=sl, right_gray_squeezed=sr, disp=d, depth=dep, u16=O.depth_to_u16(dep),
                        taps_025=O.lanczos4_taps(0.25), taps_075=O.lanczos4_taps(0.75))
    # guided upscale
    rng = np.random.default_rng(7)
    depth = syn.gt_disparity(96, 54).astype(np.float32)
    depth[rng.random(depth.shape) < 0.1] = 0.0
    guide = syn.guide_frame(96, 54, 1)
    np.savez_compressed(os.path.join(HERE, "guided_96x54.npz"), depth=depth, guide=guide,
                        q=O.guided_upscale(depth, guide, 8, 1e-3).astype(np.float64))
    # correlation lookup (inputs already bf16-representable)
    import torch
    fl = torch.from_numpy(rng.normal(0, 1, (128, 6, 20)).astype(np.float32)).to(torch.bfloat16).float().numpy()
    fr = torch.from_numpy(rng.normal(0, 1, (128, 6, 20)).astype(np.float32)).to(torch.bfloat16).float().numpy()
    flow = rng.uniform(-2, 2, (2, 6, 20)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "corr_128x6x20.npz"), fl=fl, fr=fr, flow=flow,
                        out_1x9=O.corr_lookup(fl, fr, flow, 2, 0), out_3x3=O.corr_lookup(fl, fr, flow, 2, 1))
    blend()
    print("golden vectors written to", HERE)


def blend():
    """depth.py:344-374: disparity (with invalid pixels) blended with a 48x48 'network output' (upscaled) and with a
    map larger than the frame (downscaled); checked here against the NumPy transcription of depth.py:359-374 on the oracle's resize"""
    rng = np.random.default_rng(11)
    L, R = textured_pair(200, 60, 4)
    d16 = O.sgbm_compute(L, R)
    mono_small = (rng.random((48, 48)).astype(np.float32) * 20 + 3)
    mono_big = (rng.random((97, 333)).astype(np.float32) * 5 - 1)
    out = {}
    for tag, m in (("small", mono_small), ("big", mono_big)):
        r = O.resize_linear_f32(m, 200, 60)
        nb = 0.7 * (d16.astype(np.float32) / 16.0) + 0.3 * ((r - r.min()) / (r.max() - r.min()) * 64)
        nb[nb <= 0] = 0
        out[f"mono_{tag}"] = m
        out[f"resized_{tag}"] = r
        out[f"blend_{tag}"] = O.mono_blend(d16, m)
        assert np.array_equal(out[f"blend_{tag}"], nb.astype(np.float32))       # the oracle == depth.py:359-374 written in NumPy float32
    np.savez_compressed(os.path.join(HERE, "blend_200x60.npz"), disp16=d16, **out)


if __name__ == "__main__":
    blend() if sys.argv[1:] == ["blend"] else main()

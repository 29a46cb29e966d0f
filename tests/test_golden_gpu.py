"""GPU path vs the committed golden vectors (tests/golden/, made by make_golden.py from the oracle)."""
import os

import numpy as np
import pytest
import torch

from conftest import mismatch_report

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["sgbm_160x96.npz", "sgbm_320x180.npz"])
def test_sgbm_golden(native, name):
    z = np.load(os.path.join(GOLD, name))
    L, R = native.to_device(z["left"]), native.to_device(z["right"])
    H, W = z["left"].shape
    m = native.StereoSGBM(W, H)
    assert not mismatch_report(m.debug_raw(L, R).cpu().numpy(), z["raw"], "raw")
    assert not mismatch_report(m.compute(L, R).cpu().numpy(), z["disp"], "disp")
    m.close()
    m = native.StereoSGBM(W, H, mode=1)
    assert not mismatch_report(m.compute(L, R).cpu().numpy(), z["disp_hh"], "disp_hh")
    m.close()


def test_prepost_golden(native):
    z = np.load(os.path.join(GOLD, "prepost_192x64.npz"))
    sbs = native.to_device(z["sbs"])
    gl, gr = native.sbs_to_gray(sbs, True)
    assert not mismatch_report(gl.cpu().numpy(), z["left_gray"], "left") and not mismatch_report(gr.cpu().numpy(), z["right_gray"], "right")
    sl, sr = native.sbs_to_gray(sbs, False)
    assert not mismatch_report(sl.cpu().numpy(), z["left_gray_squeezed"], "left sq") and not mismatch_report(sr.cpu().numpy(), z["right_gray_squeezed"], "right sq")
    m = native.StereoSGBM(192, 64)
    d = m.compute(gl, gr)
    assert not mismatch_report(d.cpu().numpy(), z["disp"], "disp")
    dep = native.disp_to_depth(d)
    assert np.array_equal(dep.cpu().numpy(), z["depth"])
    assert not mismatch_report(native.depth_to_u16(dep).cpu().numpy().view(np.uint16), z["u16"], "u16")
    m.close()


def test_guided_golden(native):
    z = np.load(os.path.join(GOLD, "guided_96x54.npz"))
    got = native.guided_upscale(native.to_device(z["depth"]), native.to_device(z["guide"]), 8, 1e-3).cpu().numpy().astype(np.float64)
    want = z["q"]
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-6 * np.abs(want).max())
    assert rel.max() <= 1e-3, rel.max()


def test_corr_golden(native):
    z = np.load(os.path.join(GOLD, "corr_128x6x20.npz"))
    fl = torch.from_numpy(z["fl"]).permute(1, 2, 0).contiguous().to("cuda", torch.bfloat16)
    fr = torch.from_numpy(z["fr"]).permute(1, 2, 0).contiguous().to("cuda", torch.bfloat16)
    flow = torch.from_numpy(z["flow"]).cuda()
    for pat, key in ((0, "out_1x9"), (1, "out_3x3")):
        got = native.corr_lookup(fl, fr, flow, 2, pat).cpu().numpy()
        assert np.abs(got - z[key]).max() <= 2e-2 * max(1.0, np.abs(z[key]).max())

"""ctypes front-end of the CPU oracle (liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (video_3d_pipeline) never does.  PARITY UNPINNED -- see v3d_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SgbmParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
        "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "v3d_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def default_params(**kw):
    p = SgbmParams()
    lib().orc_sgbm_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _gray_pair(left, right):
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    assert left.ndim == 2 and left.shape == right.shape
    return left, right


def sgbm_compute(left, right, params=None):
    """cv2.StereoSGBM.compute restatement: int16 HxW, x16, -16 invalid (depth.py:341)."""
    left, right = _gray_pair(left, right)
    p = params or default_params()
    H, W = left.shape
    out = np.empty((H, W), np.int16)
    rc = lib().orc_sgbm_compute(C.byref(p), _p(left, C.c_uint8), _p(right, C.c_uint8), W, H, _p(out, C.c_int16))
    if rc:
        raise RuntimeError(f"orc_sgbm_compute rc={rc}")
    return out


def sgbm_raw(left, right, params=None, want_S=False):
    left, right = _gray_pair(left, right)
    p = params or default_params()
    H, W = left.shape
    D = p.numDisparities
    out = np.empty((H, W), np.int16)
    S = np.empty((H, W - D, D), np.int16) if want_S else None
    rc = lib().orc_sgbm_raw(C.byref(p), _p(left, C.c_uint8), _p(right, C.c_uint8), W, H, _p(out, C.c_int16),
                            _p(S, C.c_int16) if want_S else None)
    if rc:
        raise RuntimeError(f"orc_sgbm_raw rc={rc}")
    return (out, S) if want_S else out


def cost_volume(left, right, params=None):
    left, right = _gray_pair(left, right)
    p = params or default_params()
    H, W = left.shape
    D = p.numDisparities
    out = np.empty((H, W - D, D), np.int16)
    rc = lib().orc_sgbm_cost_volume(C.byref(p), _p(left, C.c_uint8), _p(right, C.c_uint8), W, H, _p(out, C.c_int16))
    if rc:
        raise RuntimeError(f"orc_sgbm_cost_volume rc={rc}")
    return out


def median3x3(img):
    img = np.ascontiguousarray(img, np.int16)
    out = np.empty_like(img)
    lib().orc_median3x3_i16(_p(img, C.c_int16), img.shape[1], img.shape[0], _p(out, C.c_int16))
    return out


def filter_speckles(img, new_val=-16, max_size=100, max_diff=512):
    out = np.array(img, np.int16, order="C", copy=True)
    lib().orc_filter_speckles(_p(out, C.c_int16), out.shape[1], out.shape[0], new_val, max_size, max_diff)
    return out


def sbs_to_gray(sbs, unsqueeze=True):
    sbs = np.ascontiguousarray(sbs, np.uint8)
    H, W, _ = sbs.shape
    ow = W if unsqueeze else W // 2
    L = np.empty((H, ow), np.uint8)
    R = np.empty((H, ow), np.uint8)
    rc = lib().orc_sbs_to_gray(_p(sbs, C.c_uint8), W, H, int(unsqueeze), _p(L, C.c_uint8), _p(R, C.c_uint8))
    if rc == -1:
        raise ValueError("SBS frame width must be even")
    if rc:
        raise RuntimeError(f"orc_sbs_to_gray rc={rc}")
    return L, R


def split_sbs(sbs, unsqueeze=True):
    sbs = np.ascontiguousarray(sbs, np.uint8)
    H, W, _ = sbs.shape
    ow = W if unsqueeze else W // 2
    L = np.empty((H, ow, 3), np.uint8)
    R = np.empty((H, ow, 3), np.uint8)
    rc = lib().orc_split_sbs(_p(sbs, C.c_uint8), W, H, int(unsqueeze), _p(L, C.c_uint8), _p(R, C.c_uint8))
    if rc == -1:
        raise ValueError("SBS frame width must be even")
    return L, R


def bgr_to_gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty(bgr.shape[:2], np.uint8)
    lib().orc_bgr_to_gray(_p(bgr, C.c_uint8), out.size, _p(out, C.c_uint8))
    return out


def lanczos4_taps(fx):
    t = np.empty(8, np.int16)
    lib().orc_lanczos4_taps(C.c_float(fx), _p(t, C.c_int16))
    return t


def disp_to_depth(disp16):
    d = np.ascontiguousarray(disp16, np.int16)
    out = np.empty(d.shape, np.float32)
    lib().orc_disp_to_depth(_p(d, C.c_int16), d.size, _p(out, C.c_float))
    return out


def depth_to_u16(depth):
    d = np.ascontiguousarray(depth, np.float32)
    out = np.empty(d.shape, np.uint16)
    lib().orc_depth_to_u16(_p(d, C.c_float), d.size, _p(out, C.c_uint16))
    return out


def guided_upscale(depth_lo, guide, r=8, eps=1e-3):
    d = np.ascontiguousarray(depth_lo, np.float32)
    g = np.ascontiguousarray(guide, np.uint8)
    assert g.ndim == 2
    out = np.empty(g.shape, np.float64)
    f = lib().orc_guided_upscale
    f.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int,
                  C.c_double, C.POINTER(C.c_double)]
    rc = f(_p(d, C.c_float), d.shape[1], d.shape[0], _p(g, C.c_uint8), g.shape[1], g.shape[0], r, eps,
           _p(out, C.c_double))
    if rc:
        raise RuntimeError(f"orc_guided_upscale rc={rc}")
    return out


def bilinear_resize(src, Wd, Hd):
    s = np.ascontiguousarray(src, np.float32)
    out = np.empty((Hd, Wd), np.float64)
    lib().orc_bilinear_resize(_p(s, C.c_float), s.shape[1], s.shape[0], Wd, Hd, _p(out, C.c_double))
    return out


def resize_linear_f32(src, Wd, Hd):
    """cv2.resize(src, (Wd, Hd)) with the default INTER_LINEAR on a float32 image (depth.py:353-354)"""
    s = np.ascontiguousarray(src, np.float32)
    out = np.empty((Hd, Wd), np.float32)
    lib().orc_resize_linear_f32(_p(s, C.c_float), s.shape[1], s.shape[0], Wd, Hd, _p(out, C.c_float))
    return out


def mono_blend(disp16, mono, w_stereo=0.7, w_mono=0.3):
    """depth.py:344-374: float32 HxW = clamp(w_stereo * disp16/16 + w_mono * minmax64(resize(mono)))"""
    d = np.ascontiguousarray(disp16, np.int16)
    m = np.ascontiguousarray(mono, np.float32)
    out = np.empty(d.shape, np.float32)
    f = lib().orc_mono_blend
    f.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_float,
                  C.POINTER(C.c_float)]
    f.restype = None
    f(_p(d, C.c_int16), d.shape[1], d.shape[0], _p(m, C.c_float), m.shape[1], m.shape[0], w_stereo, w_mono, _p(out, C.c_float))
    return out


def corr_lookup(fl, fr, flow, groups=4, pattern=0):
    fl = np.ascontiguousarray(fl, np.float32)
    fr = np.ascontiguousarray(fr, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    Cc, h, w = fl.shape
    out = np.empty((groups * 9, h, w), np.float32)
    rc = lib().orc_corr_lookup(_p(fl, C.c_float), _p(fr, C.c_float), _p(flow, C.c_float), Cc, h, w, groups,
                               pattern, _p(out, C.c_float))
    if rc:
        raise RuntimeError(f"orc_corr_lookup rc={rc}")
    return out

"""ctypes binding of libv3d_hip.so (include/v3d_hip.h) over PyTorch-ROCm device buffers.

PyTorch is plumbing only: it owns device memory and streams; every kernel on the hot path is
in libv3d_hip.so.  There is NO CPU fallback: if the library is missing or a call fails this
module raises.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libv3d_hip.so")
_lib = None

# every symbol include/v3d_hip.h declares
EXPORTS = (
    "v3d_sgbm_default_params", "v3d_sgbm_create", "v3d_sgbm_destroy", "v3d_sgbm_workspace_bytes",
    "v3d_sgbm_compute", "v3d_sgbm_compute_batch", "v3d_sgbm_debug_cost_volume", "v3d_sgbm_debug_raw",
    "v3d_median3x3_i16", "v3d_filter_speckles", "v3d_sbs_to_gray", "v3d_split_sbs", "v3d_disp_to_depth",
    "v3d_depth_to_u16", "v3d_guided_upscale_ws_bytes", "v3d_guided_upscale", "v3d_bgr_to_gray",
    "v3d_corr_ws_bytes", "v3d_corr_lookup", "v3d_last_error", "v3d_version",
    "v3d_sbs_to_gray_batch", "v3d_guided_upscale_batch", "v3d_guided_upscale_disp16_batch",
    "v3d_sgbm_sync_errors", "v3d_sgbm_set_lockstep", "v3d_sgbm_profile", "v3d_sgbm_profile_stage_count", "v3d_sgbm_profile_stage_name", "v3d_sgbm_profile_read",
    "v3d_mono_blend_ws_bytes", "v3d_mono_blend", "v3d_mono_blend_batch",
    "v3d_sgbm_poll_errors", "v3d_sgbm_stream_wait_lockstep", "v3d_sgbm_set_option", "v3d_sgbm_get_option", "v3d_set_option", "v3d_get_option", "v3d_round_to_u16",
)

ERR_LOCKSTEP = -4      # V3D_ERR_LOCKSTEP


class NativeError(RuntimeError):
    pass


class LockstepTimeout(NativeError):
    """a lock-step SGM pass timed out on an over-subscribed GPU; the call's output was invalidated on the device.
    React with StereoSGBM.set_lockstep(False) and recompute (include/v3d_hip.h)."""


class SgbmParams(C.Structure):
    """mirror of v3d_sgbm_params == keyword arguments of cv2.StereoSGBM_create (depth.py:315-325)"""
    _fields_ = [(n, C.c_int) for n in (
        "minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
        "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")]


def lib_path():
    return _LIB_PATH


def use_library(path):
    """load another build of libv3d_hip.so instead of the in-tree one (development tools only -- tools/envopts.py; the
    product never calls this and reads no environment variable).  Must come before the first lib()."""
    global _LIB_PATH
    if _lib is not None:
        raise NativeError("use_library() after the library was loaded")
    _LIB_PATH = os.path.abspath(path)


def lib():
    """load libv3d_hip.so (built by `make -C video-3d-pipeline_amd/csrc` or __graft_entry__.build())"""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NativeError(
                f"{_LIB_PATH} not found: build it with `make -C video-3d-pipeline_amd/csrc` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(_LIB_PATH)
        vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
        L.v3d_last_error.restype = C.c_char_p
        L.v3d_version.restype = C.c_char_p
        L.v3d_sgbm_default_params.argtypes = [C.POINTER(SgbmParams)]
        L.v3d_sgbm_default_params.restype = None
        L.v3d_sgbm_create.argtypes = [C.POINTER(SgbmParams), ci, ci, ci, ci, C.POINTER(vp)]
        L.v3d_sgbm_destroy.argtypes = [vp]
        L.v3d_sgbm_destroy.restype = None
        L.v3d_sgbm_workspace_bytes.argtypes = [vp]
        L.v3d_sgbm_workspace_bytes.restype = sz
        L.v3d_sgbm_compute.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp]
        L.v3d_sgbm_profile.argtypes = [vp, ci]
        L.v3d_sgbm_sync_errors.argtypes = [vp]
        L.v3d_sgbm_set_lockstep.argtypes = [vp, C.c_int]
        L.v3d_sgbm_poll_errors.argtypes = [vp]
        L.v3d_sgbm_stream_wait_lockstep.argtypes = [vp, vp]
        L.v3d_sgbm_set_option.argtypes = [vp, C.c_char_p, ci]
        L.v3d_sgbm_get_option.argtypes = [vp, C.c_char_p, C.POINTER(ci)]
        L.v3d_set_option.argtypes = [C.c_char_p, ci]
        L.v3d_get_option.argtypes = [C.c_char_p, C.POINTER(ci)]
        L.v3d_mono_blend_ws_bytes.argtypes = [ci]
        L.v3d_mono_blend_ws_bytes.restype = sz
        L.v3d_mono_blend.argtypes = [vp, ci, ci, vp, ci, ci, C.c_float, C.c_float, vp, vp, vp]
        L.v3d_mono_blend_batch.argtypes = [vp, ci, ci, ci, vp, ci, ci, sz, C.c_float, C.c_float, vp, vp, vp]
        L.v3d_sgbm_profile_stage_name.argtypes = [ci]
        L.v3d_sgbm_profile_stage_name.restype = C.c_char_p
        L.v3d_sgbm_profile_read.argtypes = [vp, C.POINTER(C.c_double), ci]
        L.v3d_sgbm_compute_batch.argtypes = [vp, vp, vp, ci, ci, ci, ci, sz, vp, vp]
        L.v3d_sgbm_debug_cost_volume.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp]
        L.v3d_sgbm_debug_raw.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp]
        L.v3d_median3x3_i16.argtypes = [vp, ci, ci, vp, vp]
        L.v3d_filter_speckles.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp]
        L.v3d_sbs_to_gray.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
        L.v3d_split_sbs.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
        L.v3d_sbs_to_gray_batch.argtypes = [vp, ci, ci, ci, ci, sz, ci, vp, vp, vp]
        L.v3d_guided_upscale_batch.argtypes = [vp, ci, ci, sz, vp, ci, ci, sz, ci, ci, C.c_float, vp, vp, vp]
        L.v3d_guided_upscale_disp16_batch.argtypes = [vp, ci, ci, sz, vp, ci, ci, sz, ci, ci, C.c_float, vp, vp, vp]
        L.v3d_disp_to_depth.argtypes = [vp, sz, vp, vp]
        L.v3d_depth_to_u16.argtypes = [vp, sz, vp, vp, vp]
        L.v3d_round_to_u16.argtypes = [vp, sz, vp, vp]
        L.v3d_guided_upscale_ws_bytes.argtypes = [ci, ci]
        L.v3d_guided_upscale_ws_bytes.restype = sz
        L.v3d_guided_upscale.argtypes = [vp, ci, ci, vp, ci, ci, ci, C.c_float, vp, vp, vp]
        L.v3d_bgr_to_gray.argtypes = [vp, sz, vp, vp]
        L.v3d_corr_ws_bytes.argtypes = [ci, ci, ci]
        L.v3d_corr_ws_bytes.restype = sz
        L.v3d_corr_lookup.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        cls = LockstepTimeout if rc == ERR_LOCKSTEP else NativeError
        raise cls(f"{what} failed (rc={rc}): {lib().v3d_last_error().decode()}")


def set_option(key, value):
    """library-wide tuning switch (v3d_set_option): gf_band1, gf_band2, gf_tiled, gf_fused, corr_gather"""
    _check(lib().v3d_set_option(key.encode(), int(value)), f"v3d_set_option({key})")


def get_option(key):
    """current value of a library-wide switch (v3d_get_option)"""
    v = C.c_int()
    _check(lib().v3d_get_option(key.encode(), C.byref(v)), f"v3d_get_option({key})")
    return v.value


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NativeError(f"{what}: expected a device tensor")
    if t.dtype != dtype or not t.is_contiguous():
        raise NativeError(f"{what}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return C.c_void_p(t.data_ptr())


def resolve_device(device=None):
    """one place that turns None / 'cuda' / 'cuda:i' / i / torch.device into a device WITH an index: a bare 'cuda'
    means the CURRENT device (torch.cuda.set_device), never silently GPU 0"""
    if isinstance(device, int):
        return torch.device("cuda", device)
    d = torch.device("cuda") if device is None else torch.device(device)
    if d.type != "cuda":
        raise NativeError(f"device {device!r}: this build only has the MI355X (HIP) path")
    return d if d.index is not None else torch.device("cuda", torch.cuda.current_device())


def default_params(**kw):
    p = SgbmParams()
    lib().v3d_sgbm_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown StereoSGBM parameter {k!r}")
        setattr(p, k, int(v))
    return p


class StereoSGBM:
    """GPU stand-in for the object cv2.StereoSGBM_create returns (depth.py:315-325); `.compute`
    mirrors depth.py:341 on device tensors."""

    def __init__(self, max_width, max_height, max_batch=1, device=None, options=None, **params):
        """device: None = the current device (what torch.cuda.set_device / LOCAL_RANK selected), an index, or a
        torch.device; options: {key: int} for v3d_sgbm_set_option (tuning switches, results never change)"""
        self.params = default_params(**params)
        self.max_width, self.max_height, self.max_batch = int(max_width), int(max_height), int(max_batch)
        self.device = resolve_device(device)
        h = C.c_void_p()
        _check(lib().v3d_sgbm_create(C.byref(self.params), self.device.index, self.max_width, self.max_height,
                                     self.max_batch, C.byref(h)), "v3d_sgbm_create")
        self._h = h
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, key, value):
        _check(lib().v3d_sgbm_set_option(self._h, key.encode(), int(value)), f"v3d_sgbm_set_option({key})")

    def get_option(self, key):
        v = C.c_int()
        _check(lib().v3d_sgbm_get_option(self._h, key.encode(), C.byref(v)), f"v3d_sgbm_get_option({key})")
        return v.value

    def poll_errors(self):
        """non-blocking: lock-step time-outs reported so far by finished calls (0 = healthy)"""
        return int(lib().v3d_sgbm_poll_errors(self._h))

    def stream_wait_lockstep(self, stream):
        """make the torch stream `stream` wait for the lock-step pass of the latest compute() (order collectives behind it)"""
        _check(lib().v3d_sgbm_stream_wait_lockstep(self._h, C.c_void_p(stream.cuda_stream)), "v3d_sgbm_stream_wait_lockstep")

    def close(self):
        if getattr(self, "_h", None):
            lib().v3d_sgbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass

    @property
    def workspace_bytes(self):
        return int(lib().v3d_sgbm_workspace_bytes(self._h))

    def compute(self, left, right, out=None):
        """left/right: uint8 [H,W] or [N,H,W] device tensors -> int16 disparity x16 (-16 invalid)"""
        batched = left.dim() == 3
        l3 = left if batched else left[None]
        r3 = right if batched else right[None]
        n, H, W = l3.shape
        if r3.shape != l3.shape:
            raise NativeError("left/right shape mismatch")
        if out is None:
            out = torch.empty((n, H, W), dtype=torch.int16, device=l3.device)
        o3 = out if out.dim() == 3 else out[None]
        with torch.cuda.device(self.device):
            _check(lib().v3d_sgbm_compute_batch(self._h, _dev(l3, torch.uint8, "left"), _dev(r3, torch.uint8, "right"),
                                                n, W, H, W, H * W, _dev(o3, torch.int16, "out"), _stream()),
                   "v3d_sgbm_compute_batch")
        return out if batched else o3[0]

    def sync_errors(self):
        """device-synchronise; number of lock-step (k_vdd) workgroups that timed out on a neighbour (0 = healthy)"""
        return int(lib().v3d_sgbm_sync_errors(self._h))

    def set_lockstep(self, enable: bool):
        """switch the co-resident lock-step pass on/off for later compute() calls (off = one launch per direction);
        synchronises and clears the time-out counter"""
        _check(lib().v3d_sgbm_set_lockstep(self._h, int(bool(enable))), "v3d_sgbm_set_lockstep")

    def profile(self, enable=True):
        """per-stage HIP-event timing on the current stream: enable, run compute(), synchronize, read_profile()"""
        _check(lib().v3d_sgbm_profile(self._h, int(bool(enable))), "v3d_sgbm_profile")

    def read_profile(self):
        """-> (calls, {stage: total_ms}) ; the stream must be synchronised first"""
        n = lib().v3d_sgbm_profile_stage_count()
        buf = (C.c_double * n)()
        calls = lib().v3d_sgbm_profile_read(self._h, buf, n)
        if calls < 0:
            _check(calls, "v3d_sgbm_profile_read")
        return calls, {lib().v3d_sgbm_profile_stage_name(i).decode(): buf[i] for i in range(n)}

    def debug_cost_volume(self, left, right):
        H, W = left.shape
        out = torch.empty((H, W - 64, 64), dtype=torch.int16, device=left.device)
        _check(lib().v3d_sgbm_debug_cost_volume(self._h, _dev(left, torch.uint8, "left"), _dev(right, torch.uint8, "right"),
                                                W, H, W, _dev(out, torch.int16, "C"), _stream()), "v3d_sgbm_debug_cost_volume")
        return out

    def debug_raw(self, left, right, want_S=False):
        H, W = left.shape
        out = torch.empty((H, W), dtype=torch.int16, device=left.device)
        S = torch.empty((H, W - 64, 64), dtype=torch.int16, device=left.device) if want_S else None
        _check(lib().v3d_sgbm_debug_raw(self._h, _dev(left, torch.uint8, "left"), _dev(right, torch.uint8, "right"),
                                        W, H, W, _dev(out, torch.int16, "disp"),
                                        _dev(S, torch.int16, "S") if want_S else None, _stream()), "v3d_sgbm_debug_raw")
        return (out, S) if want_S else out


def median3x3(img):
    H, W = img.shape
    out = torch.empty_like(img)
    _check(lib().v3d_median3x3_i16(_dev(img, torch.int16, "img"), W, H, _dev(out, torch.int16, "out"), _stream()),
           "v3d_median3x3_i16")
    return out


def filter_speckles(img, new_val=-16, max_size=100, max_diff=512):
    H, W = img.shape
    out = img.clone()
    ws = torch.empty(3 * H * W, dtype=torch.int32, device=img.device)
    _check(lib().v3d_filter_speckles(_dev(out, torch.int16, "img"), W, H, new_val, max_size, max_diff,
                                     _dev(ws, torch.int32, "ws"), _stream()), "v3d_filter_speckles")
    return out


def sbs_to_gray(sbs, unsqueeze=True):
    """sbs: uint8 [H,W,3] BGR device tensor -> (left_gray, right_gray) uint8 [H, W or W/2]"""
    H, W, ch = sbs.shape
    if ch != 3:
        raise NativeError("expected HxWx3")
    if W % 2:
        raise ValueError("SBS frame width must be even")
    ow = W if unsqueeze else W // 2
    L = torch.empty((H, ow), dtype=torch.uint8, device=sbs.device)
    R = torch.empty_like(L)
    _check(lib().v3d_sbs_to_gray(_dev(sbs, torch.uint8, "sbs"), W, H, W * 3, int(bool(unsqueeze)),
                                 _dev(L, torch.uint8, "L"), _dev(R, torch.uint8, "R"), _stream()), "v3d_sbs_to_gray")
    return L, R


def sbs_to_gray_batch(sbs, unsqueeze=True, out=None):
    """sbs: uint8 [N,H,W,3] -> (left, right) uint8 [N,H,outW], one launch"""
    n, H, W, ch = sbs.shape
    if W % 2:
        raise ValueError("SBS frame width must be even")
    ow = W if unsqueeze else W // 2
    if out is None:
        out = (torch.empty((n, H, ow), dtype=torch.uint8, device=sbs.device), torch.empty((n, H, ow), dtype=torch.uint8, device=sbs.device))
    _check(lib().v3d_sbs_to_gray_batch(_dev(sbs, torch.uint8, "sbs"), n, W, H, W * 3, H * W * 3, int(bool(unsqueeze)),
                                       _dev(out[0], torch.uint8, "L"), _dev(out[1], torch.uint8, "R"), _stream()), "v3d_sbs_to_gray_batch")
    return out


def split_sbs(sbs, unsqueeze=True):
    H, W, ch = sbs.shape
    if W % 2:
        raise ValueError("SBS frame width must be even")
    ow = W if unsqueeze else W // 2
    L = torch.empty((H, ow, 3), dtype=torch.uint8, device=sbs.device)
    R = torch.empty_like(L)
    _check(lib().v3d_split_sbs(_dev(sbs, torch.uint8, "sbs"), W, H, W * 3, int(bool(unsqueeze)),
                               _dev(L, torch.uint8, "L"), _dev(R, torch.uint8, "R"), _stream()), "v3d_split_sbs")
    return L, R


def bgr_to_gray(bgr):
    out = torch.empty(bgr.shape[:-1], dtype=torch.uint8, device=bgr.device)
    _check(lib().v3d_bgr_to_gray(_dev(bgr, torch.uint8, "bgr"), out.numel(), _dev(out, torch.uint8, "gray"), _stream()),
           "v3d_bgr_to_gray")
    return out


def disp_to_depth(disp16, out=None):
    if out is None:
        out = torch.empty(disp16.shape, dtype=torch.float32, device=disp16.device)
    _check(lib().v3d_disp_to_depth(_dev(disp16, torch.int16, "disp16"), disp16.numel(), _dev(out, torch.float32, "out"),
                                   _stream()), "v3d_disp_to_depth")
    return out


def mono_blend(disp16, mono, w_stereo=0.7, w_mono=0.3, out=None):
    """depth.py:344-374 on the device.  disp16: int16 [H,W] or [N,H,W]; mono: float32 [mh,mw] or [N,mh,mw] (any size)
    -> float32 depth like disp16's shape: clamp0(w_stereo * disp16/16 + w_mono * minmax64(resize(mono)))"""
    batched = disp16.dim() == 3
    d3 = disp16 if batched else disp16[None]
    m3 = mono if mono.dim() == 3 else mono[None]
    n, H, W = d3.shape
    if m3.shape[0] != n:
        raise NativeError(f"mono batch {m3.shape[0]} != disparity batch {n}")
    mh, mw = m3.shape[1:]
    if out is None:
        out = torch.empty(d3.shape, dtype=torch.float32, device=d3.device)
    o3 = out if out.dim() == 3 else out[None]
    ws = torch.empty(max(int(lib().v3d_mono_blend_ws_bytes(n)), 16), dtype=torch.uint8, device=d3.device)
    _check(lib().v3d_mono_blend_batch(_dev(d3, torch.int16, "disp16"), n, W, H, _dev(m3, torch.float32, "mono"), mw, mh, mh * mw,
                                      float(w_stereo), float(w_mono), _dev(o3, torch.float32, "out"), _dev(ws, torch.uint8, "ws"),
                                      _stream()), "v3d_mono_blend_batch")
    return out if batched else o3[0]


def depth_to_u16(depth):
    out = torch.empty(depth.shape, dtype=torch.int16, device=depth.device)   # torch has no uint16 math; raw bits
    ws = torch.empty(2, dtype=torch.float32, device=depth.device)
    _check(lib().v3d_depth_to_u16(_dev(depth, torch.float32, "depth"), depth.numel(), _dev(out, torch.int16, "out"),
                                  _dev(ws, torch.float32, "ws"), _stream()), "v3d_depth_to_u16")
    return out


def round_to_u16(depth):
    """float32 device tensor -> clamp(rint(x), 0, 65535) as uint16 bit patterns in an int16 tensor (torch has no uint16 math)"""
    out = torch.empty(depth.shape, dtype=torch.int16, device=depth.device)
    _check(lib().v3d_round_to_u16(_dev(depth, torch.float32, "depth"), depth.numel(), _dev(out, torch.int16, "out"), _stream()),
           "v3d_round_to_u16")
    return out


_gf_ws = {}


def guided_upscale(depth_lo, guide, r=8, eps=1e-3, out=None):
    """depth_lo f32 [Hlo,Wlo], guide u8 [Hhi,Whi] -> f32 [Hhi,Whi] (upscale.py's scale step, re-specified)"""
    Hlo, Wlo = depth_lo.shape
    Hhi, Whi = guide.shape
    if out is None:
        out = torch.empty((Hhi, Whi), dtype=torch.float32, device=guide.device)
    key = (Whi, Hhi, guide.device.index)
    ws = _gf_ws.get(key)
    if ws is None:
        ws = torch.empty(int(lib().v3d_guided_upscale_ws_bytes(Whi, Hhi)), dtype=torch.uint8, device=guide.device)
        _gf_ws[key] = ws
    _check(lib().v3d_guided_upscale(_dev(depth_lo, torch.float32, "depth_lo"), Wlo, Hlo, _dev(guide, torch.uint8, "guide"),
                                    Whi, Hhi, int(r), float(eps), _dev(out, torch.float32, "out"),
                                    _dev(ws, torch.uint8, "ws"), _stream()), "v3d_guided_upscale")
    return out


def guided_upscale_batch(depth_lo, guide, r=8, eps=1e-3, out=None):
    """depth_lo [N,Hlo,Wlo] (contiguous): float32 depth, or the matcher's int16 disparity x16 (then `/16` and `<= 0 -> 0`
    of depth.py:341, 374 happen inside the filter's loads: same bits, no float plane); guide u8 [N,Hhi,Whi] (frames may be
    strided) -> f32 [N,Hhi,Whi], one launch"""
    n, Hlo, Wlo = depth_lo.shape
    _, Hhi, Whi = guide.shape
    if guide.stride(2) != 1 or guide.stride(1) != Whi:
        raise NativeError("guide frames must be dense HxW images")
    if out is None:
        out = torch.empty((n, Hhi, Whi), dtype=torch.float32, device=guide.device)
    key = (Whi, Hhi, n, guide.device.index)
    ws = _gf_ws.get(key)
    if ws is None:
        ws = torch.empty(int(lib().v3d_guided_upscale_ws_bytes(Whi, Hhi)) * n, dtype=torch.uint8, device=guide.device)
        _gf_ws[key] = ws
    if guide.dtype != torch.uint8 or not guide.is_cuda:
        raise NativeError("guide: expected a uint8 device tensor")
    if depth_lo.dtype == torch.int16:
        fn, name, src = lib().v3d_guided_upscale_disp16_batch, "v3d_guided_upscale_disp16_batch", _dev(depth_lo, torch.int16, "disp16")
    else:
        fn, name, src = lib().v3d_guided_upscale_batch, "v3d_guided_upscale_batch", _dev(depth_lo, torch.float32, "depth_lo")
    _check(fn(src, Wlo, Hlo, Hlo * Wlo, C.c_void_p(guide.data_ptr()), Whi, Hhi, guide.stride(0), n, int(r), float(eps),
              _dev(out, torch.float32, "out"), _dev(ws, torch.uint8, "ws"), _stream()), name)
    return out


def corr_lookup(fl, fr, flow, groups=4, pattern=0):
    """fl, fr: bf16 [h,w,C]; flow f32 [2,h,w] -> f32 [groups*9,h,w]"""
    h, w, Cc = fl.shape
    out = torch.empty((groups * 9, h, w), dtype=torch.float32, device=fl.device)
    ws = torch.empty(max(int(lib().v3d_corr_ws_bytes(Cc, h, w)), 16), dtype=torch.uint8, device=fl.device)
    _check(lib().v3d_corr_lookup(_dev(fl, torch.bfloat16, "fl"), _dev(fr, torch.bfloat16, "fr"),
                                 _dev(flow, torch.float32, "flow"), Cc, h, w, groups, pattern,
                                 _dev(out, torch.float32, "out"), _dev(ws, torch.uint8, "ws"), _stream()), "v3d_corr_lookup")
    return out


def to_device(a, device="cuda"):
    a = np.ascontiguousarray(a)
    if not a.flags.writeable:          # e.g. a memory-mapped clip: torch wants a writable buffer
        a = a.copy()
    return torch.from_numpy(a).to(device)

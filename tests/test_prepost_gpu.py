"""GPU parity of the steps either side of the matcher: depth.py:250-268 (split + Lanczos4 unsqueeze),
:274-275/:337-338 (gray), :341/:374 (/16 + clamp), :397-406 (min-max to uint16).  Integer work: bit-exact."""
import numpy as np
import pytest

from conftest import mismatch_report

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H", [(64, 8), (322, 45), (1030, 9), (2050, 3), (960, 540), (1920, 1080)])
@pytest.mark.parametrize("unsqueeze", [True, False])
def test_sbs_to_gray_bit_exact(native, oracle, W, H, unsqueeze):
    rng = np.random.default_rng(W * 3 + H)
    sbs = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    wl, wr = oracle.sbs_to_gray(sbs, unsqueeze)
    gl, gr = native.sbs_to_gray(native.to_device(sbs), unsqueeze)
    assert not mismatch_report(gl.cpu().numpy(), wl, "left gray"), mismatch_report(gl.cpu().numpy(), wl, "left gray")
    assert not mismatch_report(gr.cpu().numpy(), wr, "right gray"), mismatch_report(gr.cpu().numpy(), wr, "right gray")


def test_split_sbs_bgr_bit_exact(native, oracle):
    rng = np.random.default_rng(5)
    sbs = rng.integers(0, 256, (37, 130, 3), dtype=np.uint8)
    # saturating content: hard black/white edges overshoot under Lanczos
    sbs[:, 20:40] = 255
    sbs[:, 40:60] = 0
    for unsq in (True, False):
        wl, wr = oracle.split_sbs(sbs, unsq)
        gl, gr = native.split_sbs(native.to_device(sbs), unsq)
        assert not mismatch_report(gl.cpu().numpy(), wl, "left bgr")
        assert not mismatch_report(gr.cpu().numpy(), wr, "right bgr")


def test_odd_width_raises_value_error(native):
    sbs = native.to_device(np.zeros((4, 7, 3), np.uint8))
    with pytest.raises(ValueError):
        native.sbs_to_gray(sbs)


def test_bgr_to_gray(native, oracle):
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (33, 71, 3), dtype=np.uint8)
    got = native.bgr_to_gray(native.to_device(img)).cpu().numpy()
    assert not mismatch_report(got, oracle.bgr_to_gray(img), "gray")


def test_disp_to_depth_and_u16(native, oracle):
    rng = np.random.default_rng(7)
    d = (rng.integers(-1, 64 * 16, (200, 333))).astype(np.int16)
    d[rng.random(d.shape) < 0.2] = -16
    dep = native.disp_to_depth(native.to_device(d))
    want = oracle.disp_to_depth(d)
    assert np.array_equal(dep.cpu().numpy(), want)
    u = native.depth_to_u16(dep).cpu().numpy().view(np.uint16)
    assert not mismatch_report(u, oracle.depth_to_u16(want), "u16")
    flat = native.to_device(np.full((5, 9), 3.25, np.float32))
    assert (native.depth_to_u16(flat).cpu().numpy() == 0).all()


def test_sbs_to_gray_batch_equals_single(native, oracle):
    rng = np.random.default_rng(8)
    sbs = rng.integers(0, 256, (3, 21, 66, 3), dtype=np.uint8)
    L, R = native.sbs_to_gray_batch(native.to_device(sbs), True)
    for i in range(3):
        wl, wr = oracle.sbs_to_gray(sbs[i], True)
        assert not mismatch_report(L[i].cpu().numpy(), wl, f"left[{i}]") and not mismatch_report(R[i].cpu().numpy(), wr, f"right[{i}]")


@pytest.mark.parametrize("off", [1, 2, 3])
@pytest.mark.parametrize("unsqueeze", [True, False])
def test_sbs_from_an_unaligned_base_pointer(native, oracle, off, unsqueeze):
    """the C API takes any pointer.  A frame that starts `off` bytes into an allocation (first byte of the
    buffer = first byte the kernel may touch) must neither read before the buffer nor change a single output sample."""
    import torch
    from video_3d_pipeline import synthetic as syn
    sbs = syn.sbs_frame(538, 37, 5)                          # odd row count, W*3 = 1614 bytes per row: rows land on every alignment
    H, W, _ = sbs.shape
    buf = torch.zeros(off + sbs.size, dtype=torch.uint8, device="cuda")
    buf[off:] = native.to_device(sbs).reshape(-1)
    view = buf[off:].view(H, W, 3)
    assert view.data_ptr() % 4 == off and view.is_contiguous()
    gl, gr = native.sbs_to_gray(view, unsqueeze)
    wl, wr = oracle.sbs_to_gray(sbs, unsqueeze)
    assert np.array_equal(gl.cpu().numpy(), wl) and np.array_equal(gr.cpu().numpy(), wr)
    bl, br = native.split_sbs(view, unsqueeze)
    ol, orr = oracle.split_sbs(sbs, unsqueeze)
    assert np.array_equal(bl.cpu().numpy(), ol) and np.array_equal(br.cpu().numpy(), orr)


def test_round_to_u16_is_rint_and_clamp(native):
    """v3d_round_to_u16 (the 16-bit sample of the 4K PNG sink): numpy.rint + clip, halves to even, out-of-range and NaN"""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-50, 70000, 50000), np.arange(0, 300) + 0.5, [-0.5, 65534.5, 65535.5, 1e9, -1e9, np.nan]]).astype(np.float32)
    want = np.clip(np.rint(np.nan_to_num(x, nan=0.0)), 0, 65535).astype(np.uint16)
    got = native.round_to_u16(native.to_device(x)).cpu().numpy().view(np.uint16)
    assert np.array_equal(got, want)

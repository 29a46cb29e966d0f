"""CPU tests of the oracle (test infrastructure): known-answer cases of SURVEY.md 8c, an independent
NumPy restatement on tiny images, scipy cross-checks, and the committed golden vectors.
The oracle's parity to OpenCV is UNPINNED (no cv2, no reference fixtures) -- these tests pin its
internal consistency, not its agreement with cv2."""
import os

import numpy as np
import pytest

import np_sgm
from conftest import textured_pair

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _shift_pair(H=40, W=180, k=7, seed=0):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    T = gaussian_filter(rng.integers(0, 256, (H, W + 64)).astype(np.float32), 1.2)
    T = np.clip((T - 127) * 3 + 127, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(T[:, 32:32 + W]), np.ascontiguousarray(T[:, 32 + k:32 + k + W])


def test_default_params_match_depth_py(oracle):
    p = oracle.default_params()
    assert (p.minDisparity, p.numDisparities, p.blockSize) == (0, 64, 5)
    assert (p.P1, p.P2) == (8 * 3 * 5 ** 2, 32 * 3 * 5 ** 2)          # depth.py:319-320
    assert (p.disp12MaxDiff, p.uniquenessRatio, p.speckleWindowSize, p.speckleRange) == (1, 10, 100, 32)
    assert p.preFilterCap == 0 and p.mode == 0


def test_shifted_texture_gives_k(oracle):
    L, R = _shift_pair(k=7)
    d = oracle.sgbm_compute(L, R)
    inter = d[8:-8, 64 + 16:-16]
    assert (inter >= 0).all()
    assert (((inter + 8) >> 4) == 7).all() and (np.abs(inter - 16 * 7) <= 8).all()


def test_identical_constant_and_invalid_columns(oracle):
    L, _ = _shift_pair()
    d0 = oracle.sgbm_compute(L, L)
    assert (d0[:, 64:] == 0).all() and (d0[:, :64] == -16).all()
    c = np.full((30, 150), 93, np.uint8)
    dc = oracle.sgbm_compute(c, c)
    assert (dc[:, 64:] == 0).all() and (dc[:, :64] == -16).all()


@pytest.mark.parametrize("W,H,seed", [(80, 9, 1), (72, 6, 2), (90, 5, 3)])
def test_cost_volume_vs_numpy(oracle, W, H, seed):
    L, R = textured_pair(W, H, seed, max_disp=20)
    assert np.array_equal(oracle.cost_volume(L, R).astype(np.int32), np_sgm.cost_volume(L, R))


@pytest.mark.parametrize("mode,dirs", [(0, np_sgm.DIRS5), (1, np_sgm.DIRS8)])
def test_aggregation_and_wta_vs_numpy(oracle, mode, dirs):
    W, H = 82, 7
    L, R = textured_pair(W, H, 5, max_disp=15)
    raw, S = oracle.sgbm_raw(L, R, oracle.default_params(mode=mode), want_S=True)
    C = np_sgm.cost_volume(L, R)
    Sn = np_sgm.aggregate(C, dirs=dirs)
    assert np.array_equal(S.astype(np.int32), Sn)
    assert np.array_equal(raw.astype(np.int32), np_sgm.wta(Sn, W))


def test_single_row_path_normalisation(oracle):
    """1xN strip: the left->right path with the -delta normalisation, by hand (known-answer 5)"""
    L, R = textured_pair(90, 1, 9, max_disp=10)
    C = np_sgm.cost_volume(L, R)[0].astype(np.int64)
    Lr = np.zeros_like(C)
    prev = np.zeros(64, np.int64)
    for x in range(C.shape[0]):
        m = prev.min()
        cand = np.minimum(prev, np.minimum(np.r_[32767, prev[:-1]] + 600, np.r_[prev[1:], 32767] + 600))
        Lr[x] = C[x] + np.minimum(cand, m + 2400) - (m + 2400)
        prev = Lr[x]
    assert np.array_equal(Lr[0], C[0] - 2400)                      # out-of-image predecessor: plain box cost
    assert np.array_equal(np_sgm.path(C[None], 600, 2400, -1, 0)[0], Lr)
    assert (Lr >= C - 2400).all() and (Lr <= C).all()


def test_hh_saturates(oracle):
    L, R = textured_pair(100, 20, 4)
    _, S = oracle.sgbm_raw(L, R, oracle.default_params(mode=1, P1=3000, P2=12000), want_S=True)
    assert S.max() == 32767


def test_median_matches_scipy(oracle):
    from scipy.ndimage import median_filter
    rng = np.random.default_rng(1)
    img = (rng.integers(-1, 64, (50, 70)) * 16).astype(np.int16)
    assert np.array_equal(oracle.median3x3(img), median_filter(img, size=3, mode="nearest"))


def test_speckle_threshold_and_label_crosscheck(oracle):
    from scipy.ndimage import label
    img = np.full((60, 80), -16, np.int16)
    img[2:12, 2:12] = 160            # 100 px: removed
    img[20:30, 20:30] = 320
    img[30, 20] = 320                # 101 px: kept
    out = oracle.filter_speckles(img)
    assert (out[2:12, 2:12] == -16).all() and (out[20:30, 20:30] == 320).all() and out[30, 20] == 320
    # constant-valued random blobs: components == scipy's 4-connected labels
    rng = np.random.default_rng(2)
    mask = rng.random((60, 80)) < 0.55
    img = np.where(mask, 48, -16).astype(np.int16)
    lab, n = label(mask)
    sizes = np.bincount(lab.ravel())
    want = np.where(mask & (sizes[lab] > 100), 48, -16)
    assert np.array_equal(oracle.filter_speckles(img), want)


def test_lanczos_taps(oracle):
    a, b = oracle.lanczos4_taps(0.25), oracle.lanczos4_taps(0.75)
    assert a.sum() == 2048 and np.array_equal(a, b[::-1])
    assert np.array_equal(oracle.lanczos4_taps(0.0), [0, 0, 0, 2048, 0, 0, 0, 0])
    assert a.argmax() == 3


def test_split_sbs(oracle):
    flat = np.full((4, 64, 3), 77, np.uint8)
    l, r = oracle.split_sbs(flat, True)
    assert l.shape == (4, 64, 3) and (l == 77).all() and (r == 77).all()      # taps sum to 2048: flat stays flat
    rng = np.random.default_rng(3)
    sbs = rng.integers(0, 256, (5, 40, 3), dtype=np.uint8)
    l, r = oracle.split_sbs(sbs, False)
    assert np.array_equal(l, sbs[:, :20]) and np.array_equal(r, sbs[:, 20:])
    with pytest.raises(ValueError):
        oracle.split_sbs(np.zeros((4, 7, 3), np.uint8))
    g = oracle.bgr_to_gray(np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], np.uint8))
    assert g.tolist() == [[29, 150, 76, 255]]


def test_depth_conversions(oracle):
    d = np.array([[-16, 0, 1, 16, 1008]], np.int16)
    dep = oracle.disp_to_depth(d)
    assert dep.tolist() == [[0.0, 0.0, 0.0625, 1.0, 63.0]]
    u = oracle.depth_to_u16(dep)
    ref = ((dep - dep.min()) / (dep.max() - dep.min()) * 65535).astype(np.uint16)       # depth.py:401
    assert np.array_equal(u, ref)
    assert (oracle.depth_to_u16(np.full((3, 3), 2.5, np.float32)) == 0).all()
    rng = np.random.default_rng(4)
    big = rng.uniform(0, 63, (64, 64)).astype(np.float32)
    assert np.array_equal(oracle.depth_to_u16(big), ((big - big.min()) / (big.max() - big.min()) * 65535).astype(np.uint16))


def test_guided_filter_vs_numpy_and_properties(oracle):
    rng = np.random.default_rng(5)
    depth = rng.uniform(0, 60, (9, 12)).astype(np.float32)
    guide = rng.integers(0, 256, (18, 24), dtype=np.uint8)
    q = oracle.guided_upscale(depth, guide, 3, 1e-3)
    p = oracle.bilinear_resize(depth, 24, 18)
    assert np.allclose(q, np_sgm.guided(depth, guide, 3, 1e-3, p), rtol=1e-10, atol=1e-10)
    # affine in the guide, eps -> 0: reproduced
    I = guide.astype(np.float64) / 255.0
    pa = (I * 40 + 5).astype(np.float32)
    qa = oracle.guided_upscale(pa, guide, 3, 1e-12)                 # scale 1: bilinear is the identity
    assert np.allclose(qa, pa, atol=1e-4)
    # eps -> inf: a -> 0, q -> box(box(p))
    qi = oracle.guided_upscale(depth, guide, 3, 1e12)
    const = oracle.guided_upscale(depth, np.full_like(guide, 9), 3, 1e-3)
    assert np.allclose(qi, const, atol=1e-6)
    # bilinear x2 of a constant is the constant
    assert np.allclose(oracle.bilinear_resize(np.full((4, 5), 3.5, np.float32), 10, 8), 3.5)


def test_corr_one_hot(oracle):
    h, w = 3, 20
    fl = np.zeros((64, h, w), np.float32)
    fr = np.zeros((64, h, w), np.float32)
    fl[5] = 1.0
    fr[5] = np.arange(w, dtype=np.float32)[None, :]
    out = oracle.corr_lookup(fl, fr, np.zeros((2, h, w), np.float32), 1, 0)
    for k in range(9):
        assert np.allclose(out[k], np.clip(np.arange(w) + k - 4, 0, w - 1)[None, :] / 64.0)
    # integer flow shifts the sampled features; outside samples are zero
    flow = np.zeros((2, h, w), np.float32)
    flow[0] = 2.0
    out = oracle.corr_lookup(fl, fr, flow, 1, 0)
    want = np.arange(w) + 2.0
    want[want > w - 1] = 0.0
    assert np.allclose(out[4], want[None, :] / 64.0)


def test_golden_vectors_pin_the_oracle(oracle):
    for name in ("sgbm_160x96.npz", "sgbm_320x180.npz"):
        z = np.load(os.path.join(GOLD, name))
        assert np.array_equal(oracle.sgbm_raw(z["left"], z["right"]), z["raw"])
        assert np.array_equal(oracle.sgbm_compute(z["left"], z["right"]), z["disp"])
        assert np.array_equal(oracle.sgbm_compute(z["left"], z["right"], oracle.default_params(mode=1)), z["disp_hh"])
    z = np.load(os.path.join(GOLD, "prepost_192x64.npz"))
    gl, gr = oracle.sbs_to_gray(z["sbs"], True)
    assert np.array_equal(gl, z["left_gray"]) and np.array_equal(gr, z["right_gray"])
    assert np.array_equal(oracle.depth_to_u16(z["depth"]), z["u16"])
    z = np.load(os.path.join(GOLD, "guided_96x54.npz"))
    assert np.allclose(oracle.guided_upscale(z["depth"], z["guide"], 8, 1e-3), z["q"], rtol=1e-12, atol=1e-12)
    z = np.load(os.path.join(GOLD, "corr_128x6x20.npz"))
    assert np.allclose(oracle.corr_lookup(z["fl"], z["fr"], z["flow"], 2, 0), z["out_1x9"], rtol=1e-6, atol=1e-6)
    z = np.load(os.path.join(GOLD, "blend_200x60.npz"))
    for tag in ("small", "big"):
        assert np.array_equal(oracle.resize_linear_f32(z[f"mono_{tag}"], 200, 60), z[f"resized_{tag}"])
        assert np.array_equal(oracle.mono_blend(z["disp16"], z[f"mono_{tag}"]), z[f"blend_{tag}"])
        r = z[f"resized_{tag}"]
        nb = 0.7 * (z["disp16"].astype(np.float32) / 16.0) + 0.3 * ((r - r.min()) / (r.max() - r.min()) * 64)
        nb[nb <= 0] = 0
        assert np.array_equal(z[f"blend_{tag}"], nb.astype(np.float32))          # == depth.py:359-374 written in NumPy float32


def test_mono_blend_known_answers(oracle):
    """depth.py:344-374 restatement: resize properties, the max == min bypass, the clamp, the weights"""
    rng = np.random.default_rng(2)
    # resize: constants stay constant, same size is the identity, a horizontal ramp stays linear in the interior,
    # exact 2x decimation is the 2x2 mean (linear taps at phase 0.5)
    assert np.all(oracle.resize_linear_f32(np.full((7, 9), 2.5, np.float32), 31, 17) == 2.5)
    m = rng.random((12, 16)).astype(np.float32)
    assert np.array_equal(oracle.resize_linear_f32(m, 16, 12), m)
    ramp = np.tile(np.arange(10, dtype=np.float32), (4, 1))
    up = oracle.resize_linear_f32(ramp, 40, 4)
    assert np.allclose(up[:, 2:-2], ((np.arange(40) + 0.5) / 4 - 0.5)[None, 2:-2], atol=1e-5) and up[0, 0] == 0 and up[0, -1] == 9
    half = oracle.resize_linear_f32(m, 8, 6)
    assert np.allclose(half, m.reshape(6, 2, 8, 2).mean(axis=(1, 3)), atol=1e-6)
    # blend
    d16 = (rng.integers(-1, 64, (12, 16)) * 16).astype(np.int16)
    flat = oracle.mono_blend(d16, np.full((5, 5), 3.0, np.float32))
    assert np.array_equal(flat, oracle.disp_to_depth(d16))                      # max == min: stereo only (depth.py:359, 365)
    b = oracle.mono_blend(d16, m)
    lo, hi = m.min(), m.max()
    want = 0.7 * (d16.astype(np.float32) / 16.0) + 0.3 * ((m - lo) / (hi - lo) * 64)
    want[want <= 0] = 0
    assert np.array_equal(b, want.astype(np.float32)) and b.min() >= 0 and b.max() <= 0.7 * 63 + 0.3 * 64 + 1e-4
    inv = d16 == -16
    assert np.array_equal(b[inv] > 0, (0.3 * ((m - lo) / (hi - lo) * 64) > 0.7)[inv])   # invalid (-1.0) pixels: the mono term decides


def test_unsqueeze_against_a_float64_lanczos4_and_bt601(oracle):
    """independent restatement in floating point (no shared code, no fixed-point tables): cv2.resize(INTER_LANCZOS4) maps
    output x to source fx = (x + 0.5) / 2 - 0.5, weights L(t) = sinc(t) sinc(t / 4) on the 8 taps floor(fx) - 3 .. + 4,
    normalised, borders replicated; cvtColor luma = 0.299 R + 0.587 G + 0.114 B.  The oracle's 2^11 fixed-point taps and 15-bit luma
    must agree with it within one level everywhere (phases, tap order and border handling pinned; the last-bit rounding is not)"""
    rng = np.random.default_rng(11)
    from scipy.ndimage import gaussian_filter
    sbs = np.clip(gaussian_filter(rng.uniform(0, 255, (6, 96, 3)), (0, 1.2, 0)) * 1.3 - 30, 0, 255).astype(np.uint8)
    left, right = oracle.split_sbs(sbs, True)

    def lanczos4_x2(half):                                   # half: [H, w, 3] uint8 -> [H, 2w, 3] float64
        H, w, _ = half.shape
        out = np.zeros((H, 2 * w, 3))
        for x in range(2 * w):
            fx = (x + 0.5) * 0.5 - 0.5
            x0 = int(np.floor(fx))
            t = fx - x0
            taps = np.array([np.sinc(t + 3 - i) * np.sinc((t + 3 - i) / 4.0) for i in range(8)])
            taps /= taps.sum()
            idx = np.clip(np.arange(x0 - 3, x0 + 5), 0, w - 1)
            out[:, x] = np.tensordot(taps, half[:, idx].astype(np.float64), axes=(0, 1))
        return out
    for got, half in ((left, sbs[:, :48]), (right, sbs[:, 48:])):
        want = np.clip(lanczos4_x2(half), 0, 255)
        assert np.abs(got.astype(np.float64) - want).max() <= 1.0
    gray = oracle.bgr_to_gray(left)
    want = left[..., 2] * 0.299 + left[..., 1] * 0.587 + left[..., 0] * 0.114
    assert np.abs(gray.astype(np.float64) - want).max() <= 1.0

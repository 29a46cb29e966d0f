"""GPU parity of the guided-filter upscaler (upscale.py:21-73 re-specified) against the
float64 oracle.  Tolerance (BASELINE.json north_star): 1e-3 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def _rel_err(got, want):
    scale = np.maximum(np.abs(want), 1e-6 * max(float(np.abs(want).max()), 1e-30))
    return np.abs(got - want) / scale


def _case(seed, Wlo, Hlo, scale=2):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    depth = gaussian_filter(rng.uniform(0, 63, (Hlo, Wlo)), 3.0).astype(np.float32)
    depth[rng.random(depth.shape) < 0.05] = 0.0
    guide = np.clip(gaussian_filter(rng.uniform(0, 255, (Hlo * scale, Wlo * scale)), 2.0) * 1.5 - 60, 0, 255).astype(np.uint8)
    return depth, guide


@pytest.mark.parametrize("Wlo,Hlo,r,eps", [(96, 54, 8, 1e-3), (131, 77, 8, 1e-3), (160, 90, 4, 1e-2), (100, 60, 16, 1e-4)])
def test_guided_upscale_matches_oracle(native, oracle, Wlo, Hlo, r, eps):
    depth, guide = _case(Wlo + Hlo, Wlo, Hlo)
    want = oracle.guided_upscale(depth, guide, r, eps)
    got = native.guided_upscale(native.to_device(depth), native.to_device(guide), r, eps).cpu().numpy().astype(np.float64)
    err = _rel_err(got, want)
    assert err.max() <= RTOL, f"max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


def test_non_integer_scale(native, oracle):
    depth, _ = _case(3, 100, 60)
    rng = np.random.default_rng(4)
    guide = rng.integers(0, 256, (150, 230), dtype=np.uint8)
    want = oracle.guided_upscale(depth, guide, 8, 1e-3)
    got = native.guided_upscale(native.to_device(depth), native.to_device(guide), 8, 1e-3).cpu().numpy()
    assert _rel_err(got.astype(np.float64), want).max() <= RTOL


@pytest.mark.parametrize("Wg,Hg,r", [(233, 151, 8), (301, 97, 4), (257, 33, 8), (19, 21, 8)])
def test_odd_guide_sizes(native, oracle, Wg, Hg, r):
    """odd widths take the marching kernel's scalar-store path (pixel pairs straddle row alignment), odd heights a
    half-filled last row pair; a guide narrower than the window clips every box"""
    depth, _ = _case(Wg, 90, 50)
    rng = np.random.default_rng(Wg * 7 + Hg)
    guide = rng.integers(0, 256, (Hg, Wg), dtype=np.uint8)
    want = oracle.guided_upscale(depth, guide, r, 1e-3)
    got = native.guided_upscale(native.to_device(depth), native.to_device(guide), r, 1e-3).cpu().numpy()
    assert _rel_err(got.astype(np.float64), want).max() <= RTOL


def test_constant_guide_is_double_box_of_p(native, oracle):
    """I const => a = 0, b = box(p), q = box(box(p))   (SURVEY 8c known-answer 11)"""
    depth, _ = _case(9, 80, 50)
    guide = np.full((100, 160), 77, np.uint8)
    got = native.guided_upscale(native.to_device(depth), native.to_device(guide), 8, 1e-3).cpu().numpy()
    p = oracle.bilinear_resize(depth, 160, 100)
    from scipy.ndimage import uniform_filter

    def box(a, r=8):
        ones = np.ones_like(a)
        s = uniform_filter(a, 2 * r + 1, mode="constant") * (2 * r + 1) ** 2
        c = uniform_filter(ones, 2 * r + 1, mode="constant") * (2 * r + 1) ** 2
        return s / c

    want = box(box(p))
    assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max())


def test_affine_in_guide_reproduced(native):
    """p = alpha*I + beta with eps -> 0 gives q ~= p"""
    rng = np.random.default_rng(12)
    guide = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    p = (guide.astype(np.float32) / 255.0) * 40.0 + 5.0
    got = native.guided_upscale(native.to_device(p), native.to_device(guide), 4, 1e-9).cpu().numpy()   # scale 1: bilinear is identity
    assert np.abs(got - p).max() < 5e-2


def test_full_4k_size(native, oracle):
    from video_3d_pipeline import synthetic as syn
    depth = syn.gt_disparity(1920, 1080).astype(np.float32)
    guide = syn.guide_frame(1920, 1080, 0)
    got = native.guided_upscale(native.to_device(depth), native.to_device(guide), 8, 1e-3).cpu().numpy()
    assert got.shape == (2160, 3840) and np.isfinite(got).all()
    want = oracle.guided_upscale(depth, guide, 8, 1e-3)
    assert _rel_err(got.astype(np.float64), want).max() <= RTOL


def test_rejects_bad_radius(native):
    d = native.to_device(np.zeros((10, 10), np.float32))
    g = native.to_device(np.zeros((20, 20), np.uint8))
    with pytest.raises(native.NativeError):
        native.guided_upscale(d, g, 40, 1e-3)


def test_batch_with_strided_guides(native, oracle):
    """the multi-GPU guide round buffer is [B, world, H, W]: frames of one rank are strided views"""
    import torch
    depths, guides = zip(*[_case(40 + i, 64, 40) for i in range(3)])
    d = native.to_device(np.stack(depths))
    buf = torch.zeros((3, 2, 80, 128), dtype=torch.uint8, device="cuda")
    buf[:, 1] = native.to_device(np.stack(guides))
    got = native.guided_upscale_batch(d, buf[:, 1], 8, 1e-3).cpu().numpy().astype(np.float64)
    for i in range(3):
        want = oracle.guided_upscale(depths[i], guides[i], 8, 1e-3)
        assert _rel_err(got[i], want).max() <= RTOL


@pytest.mark.parametrize("cols", [256, 512])
@pytest.mark.parametrize("Wg,Hg,r,band", [(640, 360, 8, 270), (233, 151, 8, 16), (301, 97, 4, 40), (257, 33, 8, 8), (19, 21, 8, 270),
                                          (500, 301, 8, 64), (3840, 2160, 8, 270)])
def test_fused_kernel_equals_two_sweeps_bit_for_bit(native, oracle, Wg, Hg, r, band, cols):
    """k_gff (a/b handed stage-1 -> stage-2 waves through LDS, the default) performs the two-sweep kernels' arithmetic (window
    sums re-associated as pair sums since round 3): the outputs agree to the last float32 bit or one, for every band height (warm-up rows, ragged last band, bands shorter than
    the window), odd sizes (scalar stores, half-filled row pairs) and strips that end mid-image; and both meet the oracle"""
    import torch
    depth, _ = _case(Wg + band, max(Wg // 2, 8), max(Hg // 2, 8))
    rng = np.random.default_rng(Wg + 3 * Hg)
    guide = rng.integers(0, 256, (Hg, Wg), dtype=np.uint8)
    d, g = native.to_device(depth), native.to_device(guide)
    try:
        native.set_option("gf_fused", 0)
        two = native.guided_upscale(d, g, r, 1e-3)
        native.set_option("gf_fused", 1)
        native.set_option("gf_band", band)
        native.set_option("gf_cols", cols)                     # 512-column strips: 16 waves, one workgroup per CU
        one = native.guided_upscale(d, g, r, 1e-3)
    finally:
        native.set_option("gf_fused", 1)
        native.set_option("gf_band", 432)                     # the default
        native.set_option("gf_cols", 256)
    # round 3: the fused kernel forms a window as aligned pair sums minus one column, the sweeps as a running sum of columns -- the
    # same f64 value up to rounding (~1e-16 relative), which the float32 output absorbs except for rare last-bit flips
    diff = (one.double() - two.double()).abs() / two.double().abs().clamp_min(1e-6 * float(two.abs().max()))
    assert float(diff.max()) <= 2e-7, f"{int((one != two).sum())} pixels differ, max rel {float(diff.max()):.3e}"
    if Wg * Hg <= 700 * 400:
        want = oracle.guided_upscale(depth, guide, r, 1e-3)
        assert _rel_err(one.cpu().numpy().astype(np.float64), want).max() <= RTOL


@pytest.mark.parametrize("r,fused,tiled", [(8, 1, 0), (8, 0, 0), (4, 1, 0), (16, 1, 0), (8, 1, 1)])
def test_disp16_input_equals_depth_input_bit_for_bit(native, r, fused, tiled):
    """v3d_guided_upscale_disp16_batch applies depth.py:341 `/16` and depth.py:374 `<= 0 -> 0` inside the filter's loads:
    identical bits to v3d_disp_to_depth followed by v3d_guided_upscale_batch, in every kernel family (fused strips,
    two sweeps, LDS tiles), invalid (-16) and zero disparities included"""
    import torch
    rng = np.random.default_rng(100 + r)
    disp = rng.integers(-16, 64 * 16, (2, 90, 160)).astype(np.int16)
    disp[rng.random(disp.shape) < 0.1] = -16
    disp[:, :, :8] = 0
    guide = rng.integers(0, 256, (2, 180, 320), dtype=np.uint8)
    d16, g = native.to_device(disp), native.to_device(guide)
    try:
        native.set_option("gf_fused", fused)
        native.set_option("gf_tiled", tiled)
        via_depth = native.guided_upscale_batch(native.disp_to_depth(d16), g, r, 1e-3)
        direct = native.guided_upscale_batch(d16, g, r, 1e-3)
    finally:
        native.set_option("gf_fused", 1)
        native.set_option("gf_tiled", 0)
    assert torch.equal(direct, via_depth)
    assert float(direct.abs().max()) > 1.0


def test_a_frames_bits_do_not_depend_on_the_batch_it_shares(native):
    """the band partition of the fused kernel is fixed by the frame height, not chosen per launch: the same frame filtered alone
    and as one of five comes out bit-identical (option gf_band = 0, the per-launch optimum, does not promise that)"""
    import torch
    rng = np.random.default_rng(77)
    disp = rng.integers(-16, 64 * 16, (1, 540, 960)).astype(np.int16)
    guide = rng.integers(0, 256, (1, 1080, 1920), dtype=np.uint8)
    d1, g1 = native.to_device(disp), native.to_device(guide)
    alone = native.guided_upscale_batch(d1, g1, 8, 1e-3)
    five = native.guided_upscale_batch(d1.expand(5, -1, -1).contiguous(), g1.expand(5, -1, -1).contiguous(), 8, 1e-3)
    for i in range(5):
        assert torch.equal(five[i], alone[0])


@pytest.mark.parametrize("Wlo,Hlo,r,band", [(320, 180, 8, 432), (117, 75, 8, 16), (150, 48, 4, 40), (9, 10, 8, 432), (500, 350, 8, 100)])
def test_integer_stage1_is_bit_identical_to_the_f64_stage1(native, oracle, Wlo, Hlo, r, band):
    """int16 disparity + exact 2x: stage 1 of the fused kernel runs in exact integers (sum of 256 p and of g * 256 p fit int32);
    the four window sums are the values the f64 path holds, so the output has the same bits -- and meets the oracle"""
    import torch
    rng = np.random.default_rng(Wlo * 3 + Hlo)
    d = rng.integers(-16, 1024, (2, Hlo, Wlo)).astype(np.int16)
    d[rng.random(d.shape) < 0.1] = -16
    g = rng.integers(0, 256, (2, 2 * Hlo, 2 * Wlo), dtype=np.uint8)
    dd, gg = native.to_device(d), native.to_device(g)
    try:
        native.set_option("gf_band", band)
        native.set_option("gf_int1", 0)
        a = native.guided_upscale_batch(dd, gg, r, 1e-3)
        native.set_option("gf_int1", 1)
        b = native.guided_upscale_batch(dd, gg, r, 1e-3)
    finally:
        native.set_option("gf_int1", 1)
        native.set_option("gf_band", 432)
    assert torch.equal(a, b)
    want = oracle.guided_upscale(oracle.disp_to_depth(d[0]), g[0], r, 1e-3)
    assert _rel_err(b[0].cpu().numpy().astype(np.float64), want).max() <= RTOL

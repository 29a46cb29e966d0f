"""Host logic of the drop-in surface (depth.py / upscale.py mirror) on CPU.  The HIP backend needs a GPU,
so these tests inject a stand-in backend built on the oracle: they exercise the plumbing (cache naming,
streaming, batching, PNG formats, CLI, error behaviour), not the kernels."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O


class OracleStereoBackend:
    """test-only stand-in for HipStereoBackend"""

    def split_sbs(self, f, unsqueeze):
        return O.split_sbs(f, unsqueeze)

    def pairs_to_disparity(self, pairs, monos=None):
        disps = [O.sgbm_compute(O.bgr_to_gray(l), O.bgr_to_gray(r)) for l, r in pairs]
        if monos is None:
            return [O.disp_to_depth(d) for d in disps]
        return [O.mono_blend(d, np.asarray(m, np.float32)) for d, m in zip(disps, monos)]

    def sbs_to_disparity(self, frames, unsqueeze, mono_provider=None):
        out = []
        for f in frames:
            l, r = O.sbs_to_gray(f, unsqueeze)
            d = O.sgbm_compute(l, r)
            if mono_provider is None:
                out.append(O.disp_to_depth(d))
            else:
                left_rgb = O.split_sbs(f, unsqueeze)[0][..., ::-1]
                out.append(O.mono_blend(d, np.asarray(mono_provider([left_rgb])[0], np.float32)))
        return np.stack(out)

    def normalise_u16(self, depth):
        return O.depth_to_u16(np.asarray(depth, np.float32))


class _FakeTensor:
    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def numpy(self):
        return self.a


class OracleUpscaleBackend:
    device = "cpu"

    class torch:                                   # the tiny subset upscale.py touches
        @staticmethod
        def is_tensor(x):
            return False

    def to_luma(self, frame):
        return frame if frame.ndim == 2 else O.bgr_to_gray(frame)

    def upscale(self, depth_lo, guide, r, eps):
        if isinstance(guide, _FakeTensor):
            guide = guide.a
        elif hasattr(guide, "numpy"):              # a CPU torch tensor handed back by the guide exchange
            guide = guide.numpy()
        g = guide if guide.ndim == 2 else O.bgr_to_gray(guide)
        return _FakeTensor(O.guided_upscale(np.asarray(depth_lo, np.float32), g, r, eps).astype(np.float32))

    def upscale_u16(self, depth_lo, guide, r, eps):
        return np.clip(np.rint(self.upscale(depth_lo, guide, r, eps).a), 0, 65535).astype(np.uint16)

    def flat_guide(self, h, w):
        return np.full((h, w), 128, np.uint8)


@pytest.fixture()
def clip(tmp_path):
    from video_3d_pipeline import synthetic as syn
    frames = np.stack([syn.sbs_frame(192, 48, i) for i in range(5)])
    p = tmp_path / "clip.npy"
    np.save(p, frames)
    return str(p), frames


def test_both_class_names_are_exported():
    import video_3d_pipeline as v
    from video_3d_pipeline.depth import HybridStereoDepthExtractor, IGEVStereoDepthExtractor
    assert IGEVStereoDepthExtractor is HybridStereoDepthExtractor          # run_pipeline.py:12 / reference __init__.py:6
    assert v.SimpleDepthUpscaler and v.VideoAligner and v.get_video_info


def test_no_gpu_fails_loudly(tmp_path):
    import torch
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="CUDA not available but requested"):        # depth.py:43-44
        HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"))
    with pytest.raises(RuntimeError):
        HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), device="cpu")


def test_cache_path_format_is_byte_identical(tmp_path):
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "c"), backend=OracleStereoBackend())
    p = ex.get_cache_path("movie.mp4", 3, 17)
    key = "movie.mp4_3_17_Intel/dpt-large_True"                                        # depth.py:119
    assert p == tmp_path / "c" / f"depth_{hashlib.md5(key.encode()).hexdigest()[:16]}"
    assert p.is_dir() and not ex.is_cached(p, 2)
    for i in range(2):
        (p / f"depth_{i:06d}.png").write_bytes(b"x")
    assert ex.is_cached(p, 2) and not ex.is_cached(p, 3)


def test_split_and_batch_numpy_surface(tmp_path, clip):
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    _, frames = clip
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), stereo_only=True,
                                    backend=OracleStereoBackend())
    left, right = ex.split_sbs_frame(frames[0], unsqueeze=True)
    assert left.shape == frames[0].shape and right.shape == frames[0].shape
    l2, r2 = ex.split_sbs_frame(frames[0], unsqueeze=False)
    assert l2.shape == (48, 96, 3)
    with pytest.raises(ValueError, match="SBS frame width must be even"):               # depth.py:254-255
        ex.split_sbs_frame(np.zeros((4, 7, 3), np.uint8))
    out = ex.process_frame_batch([(left, right)])
    assert len(out) == 1 and out[0].dtype == np.float32 and out[0].shape == (48, 192) and out[0].min() >= 0
    assert (out[0][:, :64] == 0).all()
    pp = ex.preprocess_frame_pair(left, right)
    assert np.array_equal(pp["stereo_pair"]["left"], left[..., ::-1])                   # BGR -> RGB
    assert ex.process_frame_batch([]) == []


def test_neural_guidance_falls_back_to_stereo_only(tmp_path, capsys):
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), backend=OracleStereoBackend())
    assert not ex.stereo_only
    ex.load_model()
    assert ex.stereo_only and ex.model_loaded                                           # depth.py:107-114
    assert "falling back to stereo-only" in capsys.readouterr().out


def test_process_video_sbs_writes_normalised_png16(tmp_path, clip):
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.utils import read_png16
    path, frames = clip
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), batch_size=2,
                                    stereo_only=True, backend=OracleStereoBackend())
    out = ex.process_video_sbs(path, start_frame=1, max_frames=3)
    files = sorted(os.listdir(out))
    assert files == [f"depth_{i:06d}.png" for i in range(3)]
    l, r = O.sbs_to_gray(frames[1], True)
    depth = O.disp_to_depth(O.sgbm_compute(l, r))
    want = ((depth - depth.min()) / (depth.max() - depth.min()) * 65535).astype(np.uint16)   # depth.py:401
    got = read_png16(out / "depth_000000.png")
    assert got.dtype == np.uint16 and np.array_equal(got, want)
    # cached: a second call must not recompute
    ex.backend = None
    assert ex.process_video_sbs(path, start_frame=1, max_frames=3) == out
    with pytest.raises(ValueError, match="Could not read video info"):                   # depth.py:420
        ex.process_video_sbs(str(tmp_path / "missing.mp4"))


def test_depth_cli_exit_codes(tmp_path, capsys):
    import torch
    from video_3d_pipeline import depth
    rc = depth.main([str(tmp_path / "nope.mp4"), "--work-dir", str(tmp_path / "w"), "--stereo-only"])
    assert rc == 1 and "Error:" in capsys.readouterr().out                                # depth.py:534-536
    if not torch.cuda.is_available():
        rc = depth.main([str(tmp_path / "nope.mp4"), "--work-dir", str(tmp_path / "w"), "--device", "cpu"])
        assert rc == 1
    with pytest.raises(SystemExit):
        depth.main(["--definitely-not-a-flag"])


def test_video_info_and_frame_sources(tmp_path, clip):
    from video_3d_pipeline.utils import get_video_info, iter_frames, write_png16, read_png16, create_work_directory
    path, frames = clip
    info = get_video_info(path)
    assert info["width"] == 192 and info["height"] == 48 and info["frames"] == 5 and info["fps"] > 0
    assert get_video_info(str(tmp_path / "none.mp4")) is None                             # utils.py:36-38
    got = list(iter_frames(path, 2, 2))
    assert len(got) == 2 and np.array_equal(got[0], frames[2])
    d = create_work_directory(str(tmp_path / "frames"))
    from PIL import Image
    for i in range(3):
        Image.fromarray(frames[i][..., ::-1]).save(d / f"frame_{i:06d}.png")
    (d / "info.json").write_text(json.dumps({"fps": 24.0}))
    info = get_video_info(str(d))
    assert info["frames"] == 3 and info["fps"] == 24.0 and info["width"] == 192
    assert np.array_equal(next(iter_frames(str(d), 1, 1)), frames[1])
    a = (np.arange(12, dtype=np.uint16) * 5000).reshape(3, 4)
    write_png16(tmp_path / "x.png", a)
    assert np.array_equal(read_png16(tmp_path / "x.png"), a)


def test_upscaler_flow_with_manifest(tmp_path):
    from video_3d_pipeline.upscale import SimpleDepthUpscaler
    from video_3d_pipeline.utils import write_png16, read_png16
    from video_3d_pipeline import synthetic as syn
    ddir = tmp_path / "depth_abc"
    ddir.mkdir()
    rng = np.random.default_rng(0)
    lows = [(rng.uniform(0, 65535, (20, 32))).astype(np.uint16) for _ in range(2)]
    for i, a in enumerate(lows):
        write_png16(ddir / f"depth_{i:06d}.png", a)
    guides = np.stack([np.repeat(syn.guide_frame(32, 20, i)[..., None], 3, axis=2) for i in range(2)])
    v4k = tmp_path / "v4k.npy"
    np.save(v4k, guides)
    up = SimpleDepthUpscaler(use_nvenc=True, backend=OracleUpscaleBackend())
    out = up.process_depth_upscaling(str(ddir), str(v4k), output_path=str(tmp_path / "final.mp4"))
    assert os.path.exists(out)
    man = json.loads(open(out).read())
    assert man["count"] == 2 and man["width"] == 64 and man["height"] == 40
    q = read_png16(os.path.join(man["frames_dir"], "depth4k_000001.png"))
    want = O.guided_upscale(lows[1].astype(np.float32), O.bgr_to_gray(guides[1]), 8, 1e-3)
    assert q.shape == (40, 64) and np.abs(q.astype(np.float64) - np.clip(np.rint(want), 0, 65535)).max() <= 1
    # skip-if-exists (upscale.py:105-107) and the error paths
    assert up.process_depth_upscaling(str(ddir), str(v4k), output_path=out) == out
    with pytest.raises(ValueError, match="No depth maps found"):                          # upscale.py:38
        up.upscale_depth_maps_ffmpeg(str(tmp_path), 64, 40, str(tmp_path / "o.mp4"))
    with pytest.raises(ValueError, match="Could not read video info"):
        up.process_depth_upscaling(str(ddir), str(tmp_path / "missing.mp4"))
    q1 = up.upscale_frame(lows[0].astype(np.float32), guides[0])
    assert q1.shape == (40, 64) and q1.dtype == np.float32
    # alignment offset: depth_000000 guided by 4K frame 1
    out2 = up.process_depth_upscaling(str(ddir), str(v4k), output_path=str(tmp_path / "shifted.mp4"), guide_start_frame=1)
    man2 = json.loads(open(out2).read())
    q0 = read_png16(os.path.join(man2["frames_dir"], "depth4k_000000.png"))
    want0 = O.guided_upscale(lows[0].astype(np.float32), O.bgr_to_gray(guides[1]), 8, 1e-3)
    assert np.abs(q0.astype(np.float64) - np.clip(np.rint(want0), 0, 65535)).max() <= 1


def test_guide_count_comes_from_the_decoder_not_the_container(tmp_path, capsys, monkeypatch):
    """the container's frame count is a hint.  A 4K clip that ends early degrades the remaining depth frames to
    a flat guide with a warning (no exception at the last round); one that runs longer than promised is simply used."""
    from video_3d_pipeline import upscale, synthetic as syn
    from video_3d_pipeline.utils import write_png16, read_png16, get_video_info
    ddir = tmp_path / "depth_x"
    ddir.mkdir()
    rng = np.random.default_rng(1)
    lows = [(rng.uniform(0, 65535, (20, 32))).astype(np.uint16) for _ in range(3)]
    for i, a in enumerate(lows):
        write_png16(ddir / f"depth_{i:06d}.png", a)
    guides = np.stack([np.repeat(syn.guide_frame(32, 20, i)[..., None], 3, axis=2) for i in range(3)])
    flat = np.full((40, 64), 128, np.uint8)

    def run(n_in_clip, promised, tag):
        v4k = tmp_path / f"v_{tag}.npy"
        np.save(v4k, guides[:n_in_clip])
        info = dict(get_video_info(str(v4k)), frames=promised)
        monkeypatch.setattr(upscale, "get_video_info", lambda p: info)
        up = upscale.SimpleDepthUpscaler(backend=OracleUpscaleBackend())
        out = up.upscale_depth_maps_ffmpeg(str(ddir), 64, 40, str(tmp_path / f"o_{tag}.mp4"), video_4k_path=str(v4k))
        fr = json.loads(open(out).read())["frames_dir"]
        return up, [read_png16(os.path.join(fr, f"depth4k_{i:06d}.png")).astype(np.float64) for i in range(3)]

    def want(i, g):
        return np.clip(np.rint(O.guided_upscale(lows[i].astype(np.float32), g, 8, 1e-3)), 0, 65535)
    up, got = run(1, 3, "short")                               # container promises 3, decoder delivers 1
    assert up.last_flat_guides == 2 and "ended after 1 frames (container promised 3)" in capsys.readouterr().out
    assert np.abs(got[0] - want(0, O.bgr_to_gray(guides[0]))).max() <= 1
    assert all(np.abs(got[i] - want(i, flat)).max() <= 1 for i in (1, 2))
    up, got = run(3, 1, "long")                                # container promises 1, decoder delivers 3: all three used
    assert up.last_flat_guides == 0 and "delivered 3 guide frames, the container promised 1" in capsys.readouterr().out
    assert all(np.abs(got[i] - want(i, O.bgr_to_gray(guides[i]))).max() <= 1 for i in range(3))


def test_upscale_cli_error_exit(tmp_path, capsys):
    from video_3d_pipeline import upscale
    rc = upscale.main([str(tmp_path), str(tmp_path / "none.mp4")])
    assert rc == 1 and "Error:" in capsys.readouterr().out


def test_aligner_stub_is_explicit():
    from video_3d_pipeline.align import VideoAligner
    with pytest.raises(RuntimeError, match="skip-alignment"):
        VideoAligner("a", "b").find_alignment(300)


def test_config0_geometry_plumbing(tmp_path):
    """BASELINE.json configs[0]: a 960x540 synthetic SBS clip through the depth plumbing on CPU (stand-in backend;
    the reference's own CPU-runnable case -- a parity/plumbing case, not a bench line).  8 frames keep it quick."""
    from video_3d_pipeline import synthetic as syn
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.utils import read_png16
    frames = np.stack([syn.sbs_frame(960, 540, i) for i in range(2)] * 4)
    clip = tmp_path / "clip960.npy"
    np.save(clip, frames)
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), batch_size=4,
                                    stereo_only=True, backend=OracleStereoBackend())
    out = ex.process_video_sbs(str(clip))
    files = sorted(os.listdir(out))
    assert len(files) == 8 and files[0] == "depth_000000.png"
    a = read_png16(out / "depth_000000.png")
    assert a.shape == (540, 960) and a.max() == 65535 and a.min() == 0           # per-frame min-max normalisation
    assert np.array_equal(a, read_png16(out / "depth_000002.png"))                # identical input frame -> identical PNG


def test_png16_encoder_roundtrip_and_pool(tmp_path):
    """the zlib-based writer produces PNGs any reader decodes to the same samples (depth.py:406 / upscale.py:43 format);
    the writer pool writes them all and reports a failed write at close()"""
    from PIL import Image
    from video_3d_pipeline.utils import encode_png16, read_png16, PngWriterPool
    rng = np.random.default_rng(7)
    cases = [rng.integers(0, 65536, (37, 53), dtype=np.uint16), np.full((1, 1), 65535, np.uint16),
             np.zeros((5, 1), np.uint16), (np.arange(300 * 7) % 65536).astype(np.uint16).reshape(7, 300)]
    for k, a in enumerate(cases):
        pth = tmp_path / f"c{k}.png"
        pth.write_bytes(encode_png16(a))
        with Image.open(pth) as im:
            assert im.mode in ("I;16", "I;16B", "I") and im.size == (a.shape[1], a.shape[0])
            assert np.array_equal(np.asarray(im).astype(np.uint16), a)
        assert np.array_equal(read_png16(pth), a)
    with pytest.raises(ValueError):
        encode_png16(np.zeros((2, 2, 3), np.uint16))
    imgs = [rng.integers(0, 65536, (64, 80), dtype=np.uint16) for _ in range(40)]
    with PngWriterPool(workers=4, max_pending=6) as pool:
        for k, a in enumerate(imgs):
            pool.submit(tmp_path / f"p{k:03d}.png", a)
    for k, a in enumerate(imgs):
        assert np.array_equal(read_png16(tmp_path / f"p{k:03d}.png"), a)
    pool = PngWriterPool(workers=2)
    pool.submit(tmp_path / "ok.png", imgs[0])
    pool.submit(tmp_path / "no_such_dir" / "x.png", imgs[0])
    with pytest.raises(OSError):
        pool.close()
    assert (tmp_path / "ok.png").exists()


def test_png16_decoder_paths(tmp_path):
    """own files take the single-call decoder; rows mixing none / sub / up take the row loop; Pillow-written files
    (adaptive filters), 8-bit and RGB PNGs fall back to Pillow -- all give the same samples; prefetch keeps order"""
    import struct, zlib
    from PIL import Image
    from video_3d_pipeline.utils import _decode_png16_fast, encode_png16, read_png16, prefetch_map
    rng = np.random.default_rng(3)
    a = rng.integers(0, 65536, (23, 31), dtype=np.uint16)
    assert np.array_equal(_decode_png16_fast(encode_png16(a)), a)
    # hand-built file: row r uses filter r % 3
    be = a.astype(">u2").view(np.uint8).reshape(23, 62)
    raw = np.zeros((23, 63), np.uint8)
    for r in range(23):
        f = r % 3
        raw[r, 0] = f
        if f == 0: raw[r, 1:] = be[r]
        elif f == 1: raw[r, 1:3] = be[r, :2]; raw[r, 3:] = be[r, 2:] - be[r, :-2]
        else: raw[r, 1:] = be[r] - (be[r - 1] if r else 0)
    def chunk(t, b): return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    mixed = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 31, 23, 16, 0, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw.tobytes())) + chunk(b"IEND", b"")
    assert np.array_equal(_decode_png16_fast(mixed), a)
    (tmp_path / "mixed.png").write_bytes(mixed)
    with Image.open(tmp_path / "mixed.png") as im:
        assert np.array_equal(np.asarray(im).astype(np.uint16), a)          # an independent decoder agrees with the file
    smooth = (np.add.outer(np.arange(40), np.arange(50)) * 300).astype(np.uint16)
    Image.fromarray(smooth).save(tmp_path / "pil.png")                      # libpng-style adaptive filtering
    assert np.array_equal(read_png16(tmp_path / "pil.png"), smooth)
    Image.fromarray((smooth >> 8).astype(np.uint8)).save(tmp_path / "g8.png")
    assert _decode_png16_fast((tmp_path / "g8.png").read_bytes()) is None
    assert np.array_equal(read_png16(tmp_path / "g8.png"), (smooth >> 8).astype(np.uint16))
    assert _decode_png16_fast(b"not a png") is None
    assert list(prefetch_map(lambda v: v * v, range(23), workers=3, lookahead=4)) == [v * v for v in range(23)]
    assert list(prefetch_map(lambda v: v, [], workers=2)) == []


def test_lockstep_timeout_switches_mode_instead_of_failing(capsys):
    """depth.py backend: a lock-step time-out flag makes the batch recompute with per-direction launches"""
    from video_3d_pipeline.depth import HipStereoBackend

    class FakeMatcher:
        def __init__(self, errs): self.errs, self.off = errs, False
        def sync_errors(self): return 0 if self.off else self.errs
        def set_lockstep(self, on): self.off = not on

    healthy, sick = FakeMatcher(0), FakeMatcher(5)
    assert HipStereoBackend._lockstep_ok(healthy) is True and healthy.off is False
    assert HipStereoBackend._lockstep_ok(sick) is False and sick.off is True
    assert "over-subscribed" in capsys.readouterr().out
    assert HipStereoBackend._lockstep_ok(sick) is True          # after the switch the handle reports healthy


def test_iter_frames_stride_offset(clip):
    """rank r of a world of w decodes frames r, r + w, ... of the requested range and nothing else"""
    from video_3d_pipeline.utils import iter_frames
    path, frames = clip
    got = list(iter_frames(path, 1, 4, stride=2, offset=1))          # range [1, 5): frames 2 and 4
    assert len(got) == 2 and np.array_equal(got[0], frames[2]) and np.array_equal(got[1], frames[4])
    assert [len(list(iter_frames(path, 0, None, stride=3, offset=o))) for o in range(3)] == [2, 2, 1]   # ceil / floor split of 5
    assert len(list(iter_frames(path, 0, 5))) == 5
    with pytest.raises(ValueError):
        list(iter_frames(path, 0, 5, stride=2, offset=2))


def _numpy_blend(disp16, mono_resized):
    """depth.py:341, 359-374 transcribed in NumPy float32 (the resize is the oracle's)"""
    disparity = disp16.astype(np.float32) / 16.0
    m = mono_resized
    if m.max() > m.min():
        mono_normalized = ((m - m.min()) / (m.max() - m.min()) * 64)
        combined = 0.7 * disparity + 0.3 * mono_normalized
    else:
        combined = disparity
    combined[combined <= 0] = 0
    return combined.astype(np.float32)


def test_neural_guidance_provider_blend(tmp_path, clip, capsys):
    """f-4: a monocular-depth provider handed to the extractor is blended in exactly as depth.py:344-374 does"""
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.utils import read_png16
    path, frames = clip
    rng = np.random.default_rng(5)
    seen = []

    def provider(left_rgb_frames):
        seen.extend(left_rgb_frames)
        return [rng.random((24, 40)).astype(np.float32) * 9 + 1 for _ in left_rgb_frames]

    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), batch_size=2,
                                    backend=OracleStereoBackend(), mono_provider=provider)
    pairs = [ex.split_sbs_frame(f, True) for f in frames[:2]]
    rng = np.random.default_rng(5)
    out = ex.process_frame_batch(pairs)
    assert not ex.stereo_only and "supplied monocular depth provider" in capsys.readouterr().out
    assert len(seen) == 2 and np.array_equal(seen[0], pairs[0][0][..., ::-1])          # the left view, as RGB (depth.py:274)
    rng = np.random.default_rng(5)
    for (l, r), got in zip(pairs, out):
        mono = rng.random((24, 40)).astype(np.float32) * 9 + 1
        d16 = O.sgbm_compute(O.bgr_to_gray(l), O.bgr_to_gray(r))
        want = _numpy_blend(d16, O.resize_linear_f32(mono, d16.shape[1], d16.shape[0]))
        assert got.dtype == np.float32 and np.array_equal(got, want)
        assert (got[:, :64] > 0).any()                  # 0.7 * (-1) + 0.3 * mono: invalid columns are filled by the mono term
    # stereo_only / --no-neural wins over a provider (depth.py:344: `not self.stereo_only`)
    ex2 = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w2"), cache_dir=str(tmp_path / "w2"), stereo_only=True,
                                     backend=OracleStereoBackend(), mono_provider=provider)
    plain = ex2.process_frame_batch(pairs[:1])[0]
    assert np.array_equal(plain, O.disp_to_depth(O.sgbm_compute(O.bgr_to_gray(pairs[0][0]), O.bgr_to_gray(pairs[0][1]))))
    # a failing provider degrades to stereo-only for that batch with the reference's warning (depth.py:367-369)
    ex3 = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w3"), cache_dir=str(tmp_path / "w3"),
                                     backend=OracleStereoBackend(), mono_provider=lambda fr: 1 / 0)
    assert np.array_equal(ex3.process_frame_batch(pairs[:1])[0], plain)
    assert "Neural guidance failed, using stereo only" in capsys.readouterr().out
    # the streaming path blends too
    outdir = ex.process_video_sbs(path, max_frames=2)
    a = read_png16(outdir / "depth_000000.png")
    assert a.shape == (48, 192) and a.max() == 65535


def test_load_model_local_directory_and_fallback(tmp_path, capsys):
    """f-4 hook: DPT weights in a LOCAL directory load (nothing is downloaded); a bare model name that is not in the
    HF cache falls back to stereo-only with the reference's warning (depth.py:107-114)"""
    pytest.importorskip("transformers")
    import torch
    from transformers import DPTConfig, DPTForDepthEstimation, DPTImageProcessor
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    cfg = DPTConfig(hidden_size=32, num_hidden_layers=4, num_attention_heads=2, intermediate_size=64, image_size=64, patch_size=16,
                    backbone_out_indices=[0, 1, 2, 3], neck_hidden_sizes=[16, 32, 64, 64], fusion_hidden_size=32,
                    reassemble_factors=[4, 2, 1, 0.5], is_hybrid=False)
    torch.manual_seed(0)
    d = tmp_path / "tiny_dpt"
    DPTForDepthEstimation(cfg).save_pretrained(d)
    DPTImageProcessor(size={"height": 64, "width": 64}).save_pretrained(d)
    ex = HybridStereoDepthExtractor(model_checkpoint=str(d), work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"),
                                    device="cpu", backend=OracleStereoBackend())
    ex.load_model()
    assert ex.model_loaded and not ex.stereo_only and ex.model is not None and "Model loaded successfully" in capsys.readouterr().out
    rng = np.random.default_rng(1)
    pair = (rng.integers(0, 255, (40, 160, 3), dtype=np.uint8),) * 2
    got = ex.process_frame_batch([pair])[0]
    mono = ex._dpt_provider([np.ascontiguousarray(pair[0][..., ::-1])])[0].numpy()
    d16 = O.sgbm_compute(O.bgr_to_gray(pair[0]), O.bgr_to_gray(pair[1]))
    assert mono.shape == (64, 64) and np.array_equal(got, O.mono_blend(d16, mono))
    ex2 = HybridStereoDepthExtractor(model_checkpoint="Intel/dpt-large", work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"),
                                     device="cpu", backend=OracleStereoBackend())
    ex2.load_model()
    assert ex2.stereo_only and "falling back to stereo-only mode" in capsys.readouterr().out


def test_world_without_process_group_fails_loudly(tmp_path, clip, monkeypatch):
    """WORLD_SIZE > 1 from the environment but no process group must never shard silently"""
    import torch.distributed as dist
    from video_3d_pipeline import sharding
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.upscale import SimpleDepthUpscaler
    if dist.is_initialized():
        pytest.skip("a process group is live")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(RuntimeError, match="not initialised"):
        sharding.barrier()
    with pytest.raises(RuntimeError, match="not initialised"):
        sharding.broadcast_guide_round(None, (4, 4), "cpu")
    path, _ = clip
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), stereo_only=True,
                                    backend=OracleStereoBackend())
    with pytest.raises(RuntimeError, match="not initialised"):
        ex.process_video_sbs(path, max_frames=2)
    (tmp_path / "d").mkdir()
    from video_3d_pipeline.utils import write_png16
    write_png16(tmp_path / "d" / "depth_000000.png", np.zeros((4, 4), np.uint16))
    up = SimpleDepthUpscaler(backend=OracleUpscaleBackend())
    with pytest.raises(RuntimeError, match="not initialised"):
        up.upscale_depth_maps_ffmpeg(str(tmp_path / "d"), 8, 8, str(tmp_path / "o.mp4"), video_4k_path=path)

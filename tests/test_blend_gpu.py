"""GPU parity of the hybrid blend (depth.py:344-374: resize INTER_LINEAR + min-max + 0.7/0.3 + clamp) against the
oracle and the golden vectors.  Bar: BIT-EXACT float32 -- every operation is a single rounded f32 op in the reference's
order, so even the north_star's 1e-3 / the verdict's 1e-6 relative bar is met with zero difference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _d16(rng, H, W):
    d = (rng.integers(0, 64 * 16, (H, W))).astype(np.int16)
    d[rng.random((H, W)) < 0.25] = -16
    d[:, :min(64, W)] = -16
    return d


@pytest.mark.parametrize("W,H,mw,mh", [(192, 108, 384, 384), (192, 108, 48, 48), (320, 180, 320, 180), (203, 77, 97, 333),
                                       (1920, 1080, 384, 384), (70, 1, 5, 3), (257, 33, 1, 1), (64, 64, 128, 128)])
def test_blend_bit_exact(native, oracle, W, H, mw, mh):
    rng = np.random.default_rng(W * 7 + H + mw)
    d16 = _d16(rng, H, W)
    mono = (rng.random((mh, mw)).astype(np.float32) * 30 - 4)
    want = oracle.mono_blend(d16, mono)
    got = native.mono_blend(native.to_device(d16), native.to_device(mono)).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == (H, W)
    bad = got != want
    assert not bad.any(), f"{bad.sum()} of {bad.size} differ, max abs {np.abs(got - want).max()}"


def test_blend_flat_mono_and_batch(native, oracle):
    rng = np.random.default_rng(3)
    H, W, n = 90, 250, 3
    d16 = np.stack([_d16(rng, H, W) for _ in range(n)])
    flat = native.mono_blend(native.to_device(d16[0]), native.to_device(np.full((9, 9), 2.0, np.float32))).cpu().numpy()
    assert np.array_equal(flat, oracle.disp_to_depth(d16[0]))                    # max == min: stereo only (depth.py:359, 365)
    monos = rng.random((n, 40, 56)).astype(np.float32)
    monos[1] *= 1e-3                                                             # per-frame min / max
    got = native.mono_blend(native.to_device(d16), native.to_device(monos)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], oracle.mono_blend(d16[i], monos[i])), i
    with pytest.raises(native.NativeError):
        native.mono_blend(native.to_device(d16), native.to_device(monos[:2]))


def test_blend_golden(native):
    z = np.load(os.path.join(GOLD, "blend_200x60.npz"))
    for tag in ("small", "big"):
        got = native.mono_blend(native.to_device(z["disp16"]), native.to_device(z[f"mono_{tag}"])).cpu().numpy()
        assert np.array_equal(got, z[f"blend_{tag}"]), tag


def test_extractor_with_injected_provider(native, oracle, tmp_path):
    """through the reference-shaped surface on the HIP backend: process_frame_batch (depth.py:297) and the streaming
    process_video_sbs, with a provider returning device tensors of differing sizes"""
    from video_3d_pipeline import synthetic as syn
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.utils import read_png16
    frames = np.stack([syn.sbs_frame(256, 72, i) for i in range(3)])
    clip = tmp_path / "clip.npy"
    np.save(clip, frames)
    calls = []

    def provider(left_rgb_frames):
        out = []
        for rgb in left_rgb_frames:
            calls.append(rgb)
            g = torch.from_numpy(rgb.astype(np.float32).mean(axis=2)).cuda()       # a luma "network": deterministic in the input
            out.append(g[::2, ::3].contiguous() if len(calls) % 2 else g.contiguous())
        return out

    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), batch_size=2, mono_provider=provider)
    pairs = [ex.split_sbs_frame(f, True) for f in frames]
    out = ex.process_frame_batch(pairs)
    assert not ex.stereo_only and len(calls) == 3
    calls2 = []
    for k, ((l, r), got) in enumerate(zip(pairs, out)):
        rgb = np.ascontiguousarray(l[..., ::-1])
        g = rgb.astype(np.float32).mean(axis=2)
        mono = g[::2, ::3] if (k + 1) % 2 else g
        d16 = oracle.sgbm_compute(oracle.bgr_to_gray(l), oracle.bgr_to_gray(r))
        assert np.array_equal(got, oracle.mono_blend(d16, np.ascontiguousarray(mono))), k
    calls.clear()
    outdir = ex.process_video_sbs(str(clip))
    assert len(calls) == 3
    l, r = oracle.split_sbs(frames[0], True)
    assert np.array_equal(calls[0], l[..., ::-1])                                # the provider saw the Lanczos-unsqueezed left view, RGB
    g = calls[0].astype(np.float32).mean(axis=2)[::2, ::3]
    d16 = oracle.sgbm_compute(*oracle.sbs_to_gray(frames[0], True))
    want = oracle.depth_to_u16(oracle.mono_blend(d16, np.ascontiguousarray(g)))
    assert np.array_equal(read_png16(outdir / "depth_000000.png"), want)


def test_local_dpt_directory_on_gpu(native, oracle, tmp_path):
    """f-4 hook end to end: a (random-init, tiny) DPT saved to a local directory is loaded with local_files_only, runs
    on the GPU, and its prediction is blended by v3d_mono_blend exactly as depth.py:344-374 does"""
    pytest.importorskip("transformers")
    from transformers import DPTConfig, DPTForDepthEstimation, DPTImageProcessor
    from video_3d_pipeline import synthetic as syn
    from video_3d_pipeline.depth import IGEVStereoDepthExtractor
    cfg = DPTConfig(hidden_size=32, num_hidden_layers=4, num_attention_heads=2, intermediate_size=64, image_size=64, patch_size=16,
                    backbone_out_indices=[0, 1, 2, 3], neck_hidden_sizes=[16, 32, 64, 64], fusion_hidden_size=32,
                    reassemble_factors=[4, 2, 1, 0.5], is_hybrid=False)
    torch.manual_seed(0)
    d = tmp_path / "tiny_dpt"
    DPTForDepthEstimation(cfg).save_pretrained(d)
    DPTImageProcessor(size={"height": 64, "width": 64}).save_pretrained(d)
    ex = IGEVStereoDepthExtractor(model_checkpoint=str(d), work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), batch_size=2)
    frame = syn.sbs_frame(256, 72, 1)
    pair = ex.split_sbs_frame(frame, True)
    got = ex.process_frame_batch([pair])[0]
    assert not ex.stereo_only and next(ex.model.parameters()).is_cuda
    mono = ex._dpt_provider([np.ascontiguousarray(pair[0][..., ::-1])])[0].cpu().numpy()
    d16 = oracle.sgbm_compute(oracle.bgr_to_gray(pair[0]), oracle.bgr_to_gray(pair[1]))
    assert mono.shape == (64, 64) and mono.max() > mono.min()
    assert np.array_equal(got, oracle.mono_blend(d16, mono))

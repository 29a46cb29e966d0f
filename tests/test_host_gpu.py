"""The reference-shaped Python surface running on the real HIP backend (depth.py / upscale.py mirror)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def clip(tmp_path):
    from video_3d_pipeline import synthetic as syn
    frames = np.stack([syn.sbs_frame(256, 72, i) for i in range(5)])
    p = tmp_path / "clip.npy"
    np.save(p, frames)
    return str(p), frames


def test_extractor_numpy_surface_matches_oracle(native, oracle, tmp_path, clip):
    from video_3d_pipeline.depth import IGEVStereoDepthExtractor
    _, frames = clip
    ex = IGEVStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), unsqueeze_sbs=True, batch_size=8)
    left, right = ex.split_sbs_frame(frames[0], unsqueeze=True)
    wl, wr = oracle.split_sbs(frames[0], True)
    assert np.array_equal(left, wl) and np.array_equal(right, wr)
    pairs = [ex.split_sbs_frame(f, True) for f in frames[:3]]
    out = ex.process_frame_batch(pairs)
    assert ex.stereo_only                                   # neural guidance fell back, like depth.py:107-114
    for (l, r), d in zip(pairs, out):
        want = oracle.disp_to_depth(oracle.sgbm_compute(oracle.bgr_to_gray(l), oracle.bgr_to_gray(r)))
        assert d.dtype == np.float32 and np.array_equal(d, want)
    with pytest.raises(ValueError, match="SBS frame width must be even"):
        ex.split_sbs_frame(np.zeros((4, 7, 3), np.uint8))


def test_process_video_sbs_and_upscale_end_to_end(native, oracle, tmp_path, clip):
    from video_3d_pipeline import synthetic as syn
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.upscale import SimpleDepthUpscaler
    from video_3d_pipeline.utils import read_png16
    path, frames = clip
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), batch_size=2, stereo_only=True)
    out = ex.process_video_sbs(path, max_frames=4)
    assert sorted(os.listdir(out)) == [f"depth_{i:06d}.png" for i in range(4)]
    for i in (0, 3):
        l, r = oracle.sbs_to_gray(frames[i], True)
        want = oracle.depth_to_u16(oracle.disp_to_depth(oracle.sgbm_compute(l, r)))
        assert np.array_equal(read_png16(out / f"depth_{i:06d}.png"), want)
    # upscale against a synthetic "4K" clip at 2x
    guides = np.stack([np.repeat(syn.guide_frame(256, 72, i)[..., None], 3, axis=2) for i in range(4)])
    v4k = tmp_path / "v4k.npy"
    np.save(v4k, guides)
    up = SimpleDepthUpscaler(use_nvenc=True)
    res = up.process_depth_upscaling(str(out), str(v4k), output_path=str(tmp_path / "depth_4k_final.mp4"))
    man = json.loads(open(res).read())
    assert man["count"] == 4 and (man["width"], man["height"]) == (512, 144)
    q = read_png16(os.path.join(man["frames_dir"], "depth4k_000002.png")).astype(np.float64)
    lo = read_png16(out / "depth_000002.png").astype(np.float32)
    want = oracle.guided_upscale(lo, oracle.bgr_to_gray(guides[2]), 8, 1e-3)
    assert np.abs(q - np.clip(np.rint(want), 0, 65535)).max() <= 1
    d4 = up.upscale_frame(lo, guides[2])
    assert d4.shape == (144, 512) and np.abs(d4 - want).max() <= 1e-3 * max(1.0, np.abs(want).max())


def test_cli_runs_on_gpu(native, tmp_path, clip, capsys):
    from video_3d_pipeline import depth, upscale
    path, _ = clip
    rc = depth.main([path, "--work-dir", str(tmp_path / "cli"), "--max-frames", "2", "--stereo-only", "--batch-size", "2"])
    assert rc == 0 and "Success" in capsys.readouterr().out
    dirs = [d for d in os.listdir(tmp_path / "cli") if d.startswith("depth_")]
    assert len(dirs) == 1 and len(os.listdir(tmp_path / "cli" / dirs[0])) == 2
    rc = upscale.main([str(tmp_path / "cli" / dirs[0]), path, "--output", str(tmp_path / "o.mp4")])
    assert rc == 0


def test_bench_line_contract(native):
    """bench.py prints ONE JSON line with the keys the driver reads (metric/value/unit/..., roofline, cpu_baseline)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 0 and d["lockstep_timeouts"] == 0 and d["lockstep_recomputed"] is False
    assert d["parity_check"] is None                                   # --no-cpu-baseline: no oracle leg, no stamp
    assert d["config"]["distinct_frames_per_gpu"] == 4
    e = d["e2e"]
    assert e["value"] > 0 and e["unit"] == "frames/s" and e["batch4"]["latency_p50_ms_per_batch"] > 0 and e["lockstep_timeouts"] == 0
    x = d["extra"]
    assert x["sgbm_only"]["value"] > 0 and x["hh"]["value"] > 0 and x["corr"]["value"] > 0 and x["hh"]["lockstep_timeouts"] == 0


def test_config0_64_frames_960x540_on_the_hip_backend(native, oracle, tmp_path):
    """BASELINE.json configs[0]: a 64-frame 960x540 synthetic SBS clip through `process_video_sbs` -- here on the HIP
    backend (batch loop of depth.py:448-470 at the reference's batch size 8); every PNG equals the oracle's"""
    from concurrent.futures import ThreadPoolExecutor
    from video_3d_pipeline import synthetic as syn
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.utils import read_png16
    base = [syn.sbs_frame(960, 540, i) for i in range(4)]
    frames = np.stack([base[i % 4] for i in range(64)])
    clip = tmp_path / "clip960.npy"
    np.save(clip, frames)
    ex = HybridStereoDepthExtractor(work_dir=str(tmp_path / "w"), cache_dir=str(tmp_path / "w"), batch_size=8, stereo_only=True)
    out = ex.process_video_sbs(str(clip))
    assert sorted(os.listdir(out)) == [f"depth_{i:06d}.png" for i in range(64)] and ex.last_decoded_frames == 64

    def want(i):
        l, r = oracle.sbs_to_gray(base[i], True)
        return oracle.depth_to_u16(oracle.disp_to_depth(oracle.sgbm_compute(l, r)))
    with ThreadPoolExecutor(4) as pool:
        wants = list(pool.map(want, range(4)))
    for i in range(64):
        a = read_png16(out / f"depth_{i:06d}.png")
        assert a.shape == (540, 960) and np.array_equal(a, wants[i % 4]), i


def test_backend_follows_the_current_device(native):
    """a bare "cuda" resolves to the CURRENT device for tensors, workspace and kernels alike"""
    import torch
    from video_3d_pipeline.depth import HipStereoBackend
    from video_3d_pipeline.upscale import HipUpscaleBackend
    cur = torch.cuda.current_device()
    assert native.resolve_device("cuda").index == cur and native.resolve_device(None).index == cur
    assert native.resolve_device("cuda:0").index == 0 and native.resolve_device(0).index == 0
    with pytest.raises(native.NativeError):
        native.resolve_device("cpu")
    if torch.cuda.device_count() < 2:
        assert HipStereoBackend("cuda").device.index == cur and HipUpscaleBackend("cuda").device.index == cur
        return
    from conftest import textured_pair
    torch.cuda.set_device(1)
    try:
        b = HipStereoBackend("cuda")
        assert b.device.index == 1
        L, R = textured_pair(200, 40, 1)
        bgr = lambda g: np.repeat(g[..., None], 3, axis=2)
        out = b.pairs_to_disparity([(bgr(L), bgr(R))])
        assert b._matcher.device.index == 1 and out[0].shape == (40, 200)
    finally:
        torch.cuda.set_device(cur)

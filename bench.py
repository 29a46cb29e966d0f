#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: 1080p SBS frame -> SGBM disparity -> guided-filter
3840x2160 depth (BASELINE.json metric; workload = configs[2], the configuration the metric is quoted on).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of `--batch` synthetic SBS frames (default: what ONE co-resident
lock-step k_vdd launch holds at 1080p, 34 frames on 256 CUs; the reference's own batch_size is 8, depth.py:27 -- the same
path at batch 8 is reported under `e2e.batch8` and `extra.batch8_hbm_resident`) already resident in HBM:
v3d_sbs_to_gray -> v3d_sgbm_compute_batch -> v3d_guided_upscale_disp16_batch (depth.py:341/374's `/16` + clamp inside its loads)
against the 4K guide -> float32 4K depth in HBM.  Frames shard round-robin over ranks (weak scaling: every rank runs a full batch per step); the only
collective is the 4K guide round from rank 0 (RCCL), double-buffered on a side stream and ordered BEHIND the lock-step
SGM pass of the step it overlaps (v3d_sgbm_stream_wait_lockstep).  Rank 0 prints ONE JSON line:

  value / ms_per_step  HBM-resident whole-job rate (the contract's timed region)
  roofline             the dominant kernel (largest summed duration, HIP events on the launch stream)
  cpu_baseline         the CPU oracle (a port of the reference's OpenCV path) on one frame + one frame per host core
  parity_check         the timed region's own output (frame 0 of the last step) against that oracle; non-zero exit on mismatch
  e2e                  SURVEY 8(d) config (3) as specified: pinned host SBS + guide in -> kernels -> pinned host f32 4K depth
                       out, three streams, double-buffered; frames/s and per-batch latency, at the bench batch and at batch 8
  extra                (N = 1, --workload all, the default) configs[1] disparity only, configs[3] correlation lookup, MODE_HH
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "video-3d-pipeline_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

W, H, D = 1920, 1080, 64
SCALE = 2
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy
GF_R, GF_EPS = 8, 1e-3
N_DISTINCT = 8                  # distinct synthetic frames per rank, cycled through the batch


def alg_bytes_per_frame():
    """SURVEY.md 8(d) algorithmic bytes per frame, split over this build's launches so the shares sum
    to the survey's 1 291 161 600 B (SGBM) + 190 771 200 B (guided upscale).  V = cost-volume bytes."""
    P = W * H
    V = (W - D) * H * D * 2
    sg = {
        "prefilter": 2 * P,                       # gray in
        "cost": V,                                # C write
        "chain_v2": 2 * V / 3, "chain_d1": 2 * V / 3, "chain_d3": 2 * V / 3,   # K_v: C read + S write, 3 launches
        "chain_h0": V, "chain_h4_wta": V + 2 * P,                              # K_h: C read + S read, + disp16 out
        "chain_v2r": 0, "chain_d1r": 0, "chain_d3r": 0, "lrcheck": 0, "median": 0, "speckles": 0,     # (8-path extras: no share)
    }
    P4 = P * SCALE * SCALE
    gf = {"guided_sweep1+2": 2 * P4 + 4 * P + 4 * P4 * 2 * 2 + 4 * P4}          # 190 771 200 B (survey figure)
    return sg, gf


def cpu_baseline(sbs, guide):
    """the oracle (kind 'port': C restatement of the OpenCV path depth.py drives) on ONE frame of the same
    workload, single thread; plus real OpenCV if the box happens to have it (it does not in this image).
    Returns (json object, oracle disp16 of the frame, oracle f64 4K depth of the frame) -- the last two feed parity_check."""
    from oracle import oracle as O
    O.lib()
    t0 = time.perf_counter()
    gl, gr = O.sbs_to_gray(sbs, True)
    t1 = time.perf_counter()
    disp = O.sgbm_compute(gl, gr)
    t2 = time.perf_counter()
    depth = O.disp_to_depth(disp)
    q = O.guided_upscale(depth, guide, GF_R, GF_EPS)
    t3 = time.perf_counter()
    total = t3 - t0
    out = {"value": 1.0 / total, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "1 frame of the same workload (1920x1080 SBS -> 3840x2160 depth), oracle/liboracle.so, 1 thread",
           "seconds": {"sbs_to_gray": t1 - t0, "sgbm": t2 - t1, "guided_upscale": t3 - t2}}
    # all host cores: one frame per thread (frames are independent; ctypes releases the GIL inside liboracle.so).
    # MODE_SGBM itself is a serial scan, so frame-level parallelism is how a CPU deployment would scale.
    try:
        from concurrent.futures import ThreadPoolExecutor
        nthr = max(1, min(len(os.sched_getaffinity(0)), 16))
        if nthr > 1:
            def one(_):
                l, r = O.sbs_to_gray(sbs, True)
                O.guided_upscale(O.disp_to_depth(O.sgbm_compute(l, r)), guide, GF_R, GF_EPS)
            t6 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(one, range(nthr)))
            t7 = time.perf_counter()
            out["all_cores"] = {"value": nthr / (t7 - t6), "unit": "frames/s", "cores": nthr,
                                "sample": f"{nthr} frames, one per thread"}
    except Exception as e:      # the single-thread figure above stays the reported baseline
        out["all_cores"] = f"not measured: {e}"
    try:
        import cv2
        st = cv2.StereoSGBM_create(minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
                                   uniquenessRatio=10, speckleWindowSize=100, speckleRange=32)
        t4 = time.perf_counter()
        ref = st.compute(gl, gr)
        t5 = time.perf_counter()
        out["opencv"] = {"version": cv2.__version__, "threads": cv2.getNumThreads(), "sgbm_seconds": t5 - t4,
                         "oracle_mismatch_px": int((ref != disp).sum())}
    except ImportError:
        out["opencv"] = "unavailable on this box: parity and CPU baseline are vs this repo's restatement"
    return out, disp, q


def bench_corr(args, N, dev):
    """BASELINE configs[3]: CREStereo-style correlation lookup, bf16 in / f32 accumulate on MFMA, 1080p/4 features"""
    h, w, C, G = 270, 480, 256, 4
    fl = torch.randn((h, w, C), device=dev).to(torch.bfloat16)
    fr = torch.randn((h, w, C), device=dev).to(torch.bfloat16)
    flow = torch.rand((2, h, w), device=dev) * 4 - 2
    steps = max(args.steps, 20)
    for _ in range(max(args.warmup, 3)):
        N.corr_lookup(fl, fr, flow, G, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        N.corr_lookup(fl, fr, flow, G, 0)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / steps
    alg = 2 * h * w * C * 2 + 2 * h * w * 4 + G * 9 * h * w * 4           # fl + fr bf16, flow, out f32 (~142 MB form A)
    flops = 2.0 * h * w * C * 9
    # the kernels that ran, from the options in force (v3d_get_option), not from memory
    if N.get_option("corr_gather"):
        kernel = "k_corr_gather<0>"
    elif N.get_option("corr_fused"):
        kernel = "k_corr_fused0"
    else:
        kernel = "k_corr_warp + k_corr<0>"
    # MFMA counters cannot be read in-process: taken from the newest committed rocprofv3 --pmc pass of this workload
    # (tools/collect_profiles.sh -> profiles/rNN_corr_mfma.json), like `traffic` of the headline kernel
    mfma, msrc = None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_corr_mfma.json")))
    if files:
        try:
            mfma, msrc = json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
        except (OSError, ValueError):
            mfma = None
    # MFMA work actually issued: per 16-pixel tile and group two 16x16x32 MFMAs per 32 channels over 2 x 16 warped positions
    # (32 x 16 x 64 MACs for 9 x 16 x 64 wanted ones); dense bf16 peak 2.5 PFLOP/s (MI355X_MICROARCH.md)
    issued = 2.0 * 32 * 16 * 64 * G * ((w + 15) // 16) * h
    roof = {"bound": "hbm", "kernel": kernel, "achieved": alg / (ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "useful_gflops": flops / (ms * 1e-3) / 1e9, "issued_mfma_gflops": issued / (ms * 1e-3) / 1e9,
            "issued_frac_of_bf16_peak": issued / (ms * 1e-3) / 2.5e15,
            "mfma_util": (mfma or {}).get("mfma_util"), "mfma_counters": mfma, "mfma_source": msrc,
            "note": "AI ~ 4 flop/B: HBM/L2 bound; MFMA only removes the VALU bottleneck"}
    if mfma and kernel in (mfma.get("kernels") or {}):
        roof["traffic"] = mfma["kernels"][kernel].get("traffic_bytes")
    return {"metric": "corr_lookups_per_s", "value": steps / el, "unit": "lookups/s", "n_gpus": 1,
            "steps": steps, "warmup": max(args.warmup, 3), "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 in / f32 accumulate (MFMA 16x16x32)", "data": "synthetic",
            "config": {"workload": "configs[3]: correlation lookup 270x480x256, 4 groups x 9 offsets (1x9)"},
            "roofline": roof, "cpu_baseline": None}


class _RawSinkPool:
    """stands where utils.PngWriterPool stands: the 16-bit maps go to disk as raw little-endian samples (no zlib), so the rate
    of the file-to-file path shows what the GPU side and the decode sustain next to the PNG-bound figure"""

    def __init__(self):
        self.count = 0

    def submit(self, path, img_u16):
        np.ascontiguousarray(img_u16).tofile(str(path) + ".raw")
        self.count += 1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def bench_cli(base_sbs, base_guide, n_depth=68, n_up=16):
    """The PRODUCT path a caller of the reference gets (run_pipeline.py:63-98), file to file: synthetic .npy clip ->
    HybridStereoDepthExtractor.process_video_sbs -> depth_%06d.png dir -> SimpleDepthUpscaler.process_depth_upscaling ->
    4K 16-bit PNG sequence.  batch_size 8 is what run_pipeline.py:67 / depth.py:27 pass; the device pass is decoupled from
    it (depth.py here: compute_batch_size).  The PNG figures are bound by zlib on the host cores; the raw-sink figures show
    the same path without the compression."""
    import contextlib
    import io
    import shutil
    import tempfile
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from video_3d_pipeline.upscale import SimpleDepthUpscaler
    work = tempfile.mkdtemp(prefix="v3d_cli_")
    out = {"what": "synthetic 1080p SBS .npy clip -> process_video_sbs -> 16-bit PNG dir -> process_depth_upscaling (4K guide clip) -> 4K 16-bit PNGs",
           "frames_depth": n_depth, "frames_upscale": n_up, "host_threads": len(os.sched_getaffinity(0))}
    try:
        nd = len(base_sbs)
        clip = os.path.join(work, "sbs.npy")
        np.save(clip, np.stack([base_sbs[i % nd] for i in range(n_depth)]))
        clip4k = os.path.join(work, "v4k.npy")
        np.save(clip4k, np.stack([np.repeat(base_guide[i % nd][..., None], 3, axis=2) for i in range(n_up)]))
        sink = io.StringIO()
        depth_dir = None
        for label, bs, raw in (("batch8", 8, False), ("batch34", 34, False), ("batch8_raw_sink", 8, True)):
            with contextlib.redirect_stdout(sink):
                ex = HybridStereoDepthExtractor(work_dir=os.path.join(work, label), cache_dir=os.path.join(work, label), stereo_only=True, batch_size=bs)
                if raw:
                    ex.writer_pool_factory = _RawSinkPool
                ex.process_video_sbs(clip, max_frames=min(n_depth, 34), force_reprocess=True)        # warm-up: workspace, first launches
                t0 = time.perf_counter()
                d = ex.process_video_sbs(clip, force_reprocess=True)
                t1 = time.perf_counter()
            out["depth_" + label] = {"value": n_depth / (t1 - t0), "unit": "frames/s", "batch_size": bs, "frames_per_device_pass": getattr(ex, "last_pass_frames", None)}
            if not raw and depth_dir is None:
                depth_dir = str(d)
            ex.backend._matcher.close() if getattr(ex.backend, "_matcher", None) is not None else None
            del ex
        ddir = os.path.join(work, "d_up")
        os.makedirs(ddir)
        for i, f in enumerate(sorted(os.listdir(depth_dir))[:n_up]):
            shutil.copy(os.path.join(depth_dir, f), os.path.join(ddir, f"depth_{i:06d}.png"))
        for label, raw in (("upscale", False), ("upscale_raw_sink", True)):
            with contextlib.redirect_stdout(sink):
                up = SimpleDepthUpscaler()
                if raw:
                    up.writer_pool_factory = _RawSinkPool
                up.process_depth_upscaling(ddir, clip4k, output_path=os.path.join(work, label + "_w.json"), force_reprocess=True)
                t0 = time.perf_counter()
                up.process_depth_upscaling(ddir, clip4k, output_path=os.path.join(work, label + ".json"), force_reprocess=True)
                t1 = time.perf_counter()
            out[label] = {"value": n_up / (t1 - t0), "unit": "frames/s"}
        out["batch8_vs_batch34"] = out["depth_batch8"]["value"] / out["depth_batch34"]["value"]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def latest_traffic():
    """HBM traffic per kernel from the newest committed rocprofv3 --pmc passes (profiles/rNN_traffic.json)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError):
        return None, None


class HotPath:
    """device buffers + the matcher of one rank; step() enqueues one pass over a batch on the current stream"""

    def __init__(self, N, dev, B, mode):
        self.N, self.dev, self.B = N, dev, B
        Hh, Wh = H * SCALE, W * SCALE
        self.lg = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        self.rg = torch.empty_like(self.lg)
        self.disp = torch.empty((B, H, W), dtype=torch.int16, device=dev)
        self.depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        self.out4k = torch.empty((B, Hh, Wh), dtype=torch.float32, device=dev)
        self.matcher = N.StereoSGBM(W, H, B, device=dev, mode=mode)
        self.gf_ev = []

    def step(self, sbs, guides, out4k=None, upscale=True, timed=False):
        N, n = self.N, sbs.shape[0]
        N.sbs_to_gray_batch(sbs, True, (self.lg[:n], self.rg[:n]))
        self.matcher.compute(self.lg[:n], self.rg[:n], self.disp[:n])
        if not upscale:
            N.disp_to_depth(self.disp[:n], self.depth[:n])       # configs[1]: the float32 depth map is the output (depth.py:341, 374)
            return
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # the filter takes the int16 disparity itself: `/16` and `<= 0 -> 0` happen in its loads, the float32 1080p plane is never
        # written or re-read (bit-identical to v3d_disp_to_depth + v3d_guided_upscale_batch: tests/test_guided_gpu.py)
        N.guided_upscale_batch(self.disp[:n], guides, GF_R, GF_EPS, self.out4k[:n] if out4k is None else out4k)
        if timed:
            e1.record()
            self.gf_ev.append((e0, e1))


def run_e2e(hp, h_sbs, h_gui, nb, steps, warmup, ref0=None):
    """SURVEY 8(d) config (3) as specified: SBS frames + 4K guides start in PINNED HOST memory, the float32 4K depth ends in
    pinned host memory.  Three streams (H2D, compute, D2H), double-buffered device and host tensors: step s+1's inputs
    upload and step s-1's result downloads while step s computes.  Per-step events give a real per-batch latency
    (first H2D byte -> last D2H byte) next to the throughput."""
    dev = hp.dev
    Hh, Wh = H * SCALE, W * SCALE
    h_sbs, h_gui = h_sbs[:nb], h_gui[:nb]
    h_out = [torch.empty((nb, Hh, Wh), dtype=torch.float32).pin_memory() for _ in range(2)]
    d_sbs = [torch.empty((nb, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_gui = [torch.empty((nb, Hh, Wh), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_out = [torch.empty((nb, Hh, Wh), dtype=torch.float32, device=dev) for _ in range(2)]
    s_in, s_out, s_cmp = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)

    def run(nsteps):
        in_ready, cmp_done, out_done = [None, None], [None, None], [None, None]
        t_in, t_done = [], []

        def upload(s):
            k = s & 1
            with torch.cuda.stream(s_in):
                if cmp_done[k] is not None:
                    s_in.wait_event(cmp_done[k])           # step s-2 has consumed this input slot
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(s_in)
                t_in.append(e0)
                d_sbs[k].copy_(h_sbs, non_blocking=True)
                d_gui[k].copy_(h_gui, non_blocking=True)
                e = torch.cuda.Event()
                e.record(s_in)
                in_ready[k] = e
        upload(0)
        for s in range(nsteps):
            k = s & 1
            if s + 1 < nsteps:
                upload(s + 1)
            s_cmp.wait_event(in_ready[k])
            if out_done[k] is not None:
                s_cmp.wait_event(out_done[k])              # step s-2's result has left this output slot
            hp.step(d_sbs[k], d_gui[k], out4k=d_out[k])
            e = torch.cuda.Event()
            e.record(s_cmp)
            cmp_done[k] = e
            with torch.cuda.stream(s_out):
                s_out.wait_event(e)
                h_out[k].copy_(d_out[k], non_blocking=True)
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record(s_out)
                out_done[k] = e2
                t_done.append(e2)
        torch.cuda.synchronize()
        return t_in, t_done

    run(warmup)
    t0 = time.perf_counter()
    t_in, t_done = run(steps)
    el = time.perf_counter() - t0
    lat = [a.elapsed_time(b) for a, b in zip(t_in, t_done)]
    gaps = [t_done[i].elapsed_time(t_done[i + 1]) for i in range(len(t_done) - 1)]
    mb = (h_sbs.numel() + h_gui.numel() + h_out[0].numel() * 4) / nb / 1e6
    fps = nb * steps / el
    # the leg's own output, stamped: frame 0 of the last step, as it arrived in pinned host memory, against the HBM-resident
    # leg's frame 0 (same input frame, same kernels: the bits must agree; that frame is checked against the oracle)
    stamp = None
    if ref0 is not None:
        stamp = bool(torch.equal(h_out[(steps - 1) & 1][0], ref0))
    return {"value": fps, "unit": "frames/s", "frames_per_step": nb, "steps": steps, "output_equals_resident_leg": stamp,
            "latency_p50_ms_per_batch": statistics.median(lat), "latency_max_ms_per_batch": max(lat),
            "p50_ms_per_step": statistics.median(gaps) if gaps else el / steps * 1e3,
            "p50_ms_per_frame": (statistics.median(gaps) if gaps else el / steps * 1e3) / nb,
            "link_MB_per_frame": mb, "link_GBps": fps * mb / 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per step per GPU; 0 (default) = what ONE lock-step k_vdd launch holds co-resident at this width (the handle's "
                         "vdd_frames_per_launch_dpl8: 34 at 1080p on 256 CUs).  The reference's --batch-size is 8 (list chunking, depth.py:448-461)")
    ap.add_argument("--guide-exchange", choices=["auto", "broadcast", "scatter", "none"], default="scatter",
                    help="how rank 0 hands out the 4K guide rounds: scatter each rank's own frames (default: 8.3 MB per frame and "
                         "rank), broadcast whole rounds (north_star's variant: world x the bytes, 2 GB per 30-frame step at world 8), "
                         "or auto = broadcast if one step's rounds hide behind a compute step, else scatter.  Whatever the choice, one "
                         "round of `world` frames is timed both ways during warm-up and reported (config.guide_exchange_probe)")
    ap.add_argument("--workload", choices=["all", "full", "sgbm", "corr"], default="all",
                    help="all (default) = the headline line of `full` + `extra` {sgbm_only, corr, hh, batch8} at N = 1; full = BASELINE "
                         "configs[2] only; sgbm = configs[1] (disparity only) as the line; corr = configs[3] (bf16 MFMA correlation lookup)")
    ap.add_argument("--sgbm-mode", choices=["sgbm", "hh"], default="sgbm",
                    help="sgbm = OpenCV MODE_SGBM, 5 paths (what depth.py:315-325 gets by default); hh = MODE_HH, 8 paths")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (then no parity_check either)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive pipeline leg")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--no-cli", action="store_true", help="skip the file-to-file product-path leg (extra.cli)")
    ap.add_argument("--dump-frame0", default=None, metavar="NPZ",
                    help="tests: rank 0 writes frame 0 of its last timed step (disp16 + 4K depth) to this .npz")
    ap.add_argument("--test-inject-lockstep-timeout", action="store_true",
                    help="tests only: make every lock-step workgroup of the first timed region report a time-out (drives the recovery path)")
    args = ap.parse_args()

    from video_3d_pipeline import _native as N, sharding, synthetic as syn
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    import torch.distributed as dist
    if args.dist_backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)         # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    N.lib()
    if args.workload == "corr":
        print(json.dumps(bench_corr(args, N, dev)))
        return 0
    if world > 1:
        sharding.init_process_group(args.dist_backend)
    B = args.batch
    if B <= 0:
        probe = N.StereoSGBM(W, H, 1, device=dev)
        B = max(1, probe.get_option("vdd_frames_per_launch_dpl8"))
        probe.close()
    full = args.workload in ("all", "full")

    # ---- synthetic inputs: N_DISTINCT distinct frames per rank, cycled through the batch, resident in HBM ----
    nd = min(B, N_DISTINCT)
    from concurrent.futures import ThreadPoolExecutor

    def synth(i):
        return syn.sbs_frame(W, H, nd * rank + i), syn.guide_frame(W, H, nd * rank + i, SCALE)
    with ThreadPoolExecutor(min(nd, max(1, len(os.sched_getaffinity(0)) // max(world, 1)))) as ex:
        base = list(ex.map(synth, range(nd)))
    base_sbs, base_guide = [b[0] for b in base], [b[1] for b in base]
    h_sbs = torch.from_numpy(np.stack([base_sbs[i % nd] for i in range(B)]))
    h_gui = torch.from_numpy(np.stack([base_guide[i % nd] for i in range(B)]))
    sbs = h_sbs.to(dev)
    guide_own = h_gui.to(dev)
    Hh, Wh = H * SCALE, W * SCALE
    # guide rounds: round j holds the guide frames of global frames j*world .. j*world+world-1 (rank r owns slot r)
    mode = {"v": args.guide_exchange}
    exch_info = {}
    scatter_cache = {}
    guide_buf, side, rounds_src = None, None, None
    if world > 1 and args.guide_exchange != "none" and full:
        rounds_src = guide_own[:, None].expand(B, world, Hh, Wh).contiguous() if rank == 0 else None
        side = torch.cuda.Stream(device=dev)

        def alloc_bufs():
            return [torch.empty((B, world, Hh, Wh), dtype=torch.uint8, device=dev) if mode["v"] == "broadcast"
                    else torch.empty((B, Hh, Wh), dtype=torch.uint8, device=dev) for _ in range(2)]

    hp = HotPath(N, dev, B, 1 if args.sgbm_mode == "hh" else 0)
    matcher = hp.matcher

    def exchange(slot):
        """enqueue the guide exchange for the NEXT step on the side stream, ordered behind the lock-step SGM pass of the
        step just enqueued on the main stream: RCCL's workgroups then never take CU slots that pass was sized with (they
        overlap the horizontal pass, the post-filters and the upscale instead); the next step waits for the exchange."""
        if guide_buf is None:
            return None
        matcher.stream_wait_lockstep(side)
        with torch.cuda.stream(side):
            if mode["v"] == "broadcast":
                # the root sends straight from the decoded round (no staging copy), the others receive into the slot
                dist.broadcast(rounds_src if rank == 0 else guide_buf[slot], src=0)
            else:
                # per-rank chunks are laid out once (the decoder of a real pipeline writes them in place): the root's
                # timed steps carry no 2 GB repacking copy
                if rank == 0 and "chunks" not in scatter_cache:
                    scatter_cache["chunks"] = [rounds_src[:, r].contiguous() for r in range(world)]
                dist.scatter(guide_buf[slot], scatter_cache.get("chunks"), src=0)
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def my_guides(slot):
        if guide_buf is None:
            return guide_own
        if mode["v"] == "broadcast":
            return (rounds_src if rank == 0 else guide_buf[slot])[:, rank]
        return guide_buf[slot]

    lockstep = {"raised": 0}

    def run(nsteps, timed):
        """nsteps passes.  A lock-step time-out (V3D_ERR_LOCKSTEP: k_vdd_guard has raised the host flag, every later compute
        on the handle is refused) must not unwind past the loop: the other ranks are still posting one collective per step,
        so this rank keeps posting its own and only stops computing; the caller then switches ALL ranks together."""
        evs = []
        pending = exchange(0)
        for s in range(nsteps):
            slot = s & 1
            if pending is not None:
                torch.cuda.current_stream().wait_event(pending)
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
            if not lockstep["raised"]:
                try:
                    hp.step(sbs, my_guides(slot), upscale=full, timed=timed)
                except N.LockstepTimeout:
                    lockstep["raised"] = 1
            pending = exchange(1 - slot) if s + 1 < nsteps else None          # after the step: its k_vdd event is the one to order behind
        if timed:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        return evs

    if side is not None:
        # one round of `world` guide frames (north_star: "RCCL carrying only the 4K guide broadcast") timed as a broadcast and
        # as a scatter, outside the timed region: the absolute cost of the collective on this node, whatever mode the steps use
        one = torch.empty((world, Hh, Wh), dtype=torch.uint8, device=dev)
        mine = torch.empty((Hh, Wh), dtype=torch.uint8, device=dev)
        parts = [one[r] for r in range(world)] if rank == 0 else None
        for name, fn in (("broadcast", lambda: dist.broadcast(one, src=0)), ("scatter", lambda: dist.scatter(mine, parts, src=0))):
            fn(); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            exch_info[name + "_one_round_ms"] = float(tt[0]) * 1e3
        exch_info["round_bytes"] = {"broadcast": world * Hh * Wh, "scatter_per_rank": Hh * Wh}
        del one, mine, parts
        if args.guide_exchange == "auto":
            # time one compute step and one full-round broadcast; keep the broadcast only if it hides behind the step
            mode["v"] = "broadcast"
            guide_buf = alloc_bufs()
            hp.step(sbs, guide_own); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter(); hp.step(sbs, guide_own); torch.cuda.synchronize(); t_step = time.perf_counter() - t0
            exchange(0); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter(); exchange(0); torch.cuda.synchronize(); dist.barrier(); t_bc = time.perf_counter() - t0
            tt = torch.tensor([t_step, t_bc], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_step, t_bc = float(tt[0]), float(tt[1])
            exch_info.update({"step_ms": t_step * 1e3, "broadcast_step_rounds_ms": t_bc * 1e3})
            if t_bc > 0.5 * t_step:              # the exchange window is the part of a step behind k_vdd (about 60 % of it)
                mode["v"] = "scatter"
                guide_buf = None
                torch.cuda.empty_cache()
        if guide_buf is None:
            guide_buf = alloc_bufs()

    def timed_region():
        run(args.warmup, False)
        torch.cuda.synchronize()
        matcher.profile(True)
        hp.gf_ev.clear()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs = run(args.steps, True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return evs, time.perf_counter() - t0

    if args.test_inject_lockstep_timeout:
        matcher.set_option("vdd_spin_limit", -1)          # test hook: every k_vdd workgroup reports a time-out
    evs, elapsed = timed_region()
    # a lock-step pass that timed out (GPU shared with another job) invalidated its outputs on the device: such a run is
    # not a measurement.  Do what depth.py does -- switch the handle to per-direction launches -- and time again.
    timeouts = max(matcher.sync_errors(), lockstep["raised"])
    lockstep["raised"] = 0
    if args.test_inject_lockstep_timeout:
        matcher.set_option("vdd_spin_limit", 0)
    recomputed = False
    if world > 1:
        t = torch.tensor([timeouts], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        any_timeouts = int(t.item())
    else:
        any_timeouts = timeouts
    if any_timeouts > 0:
        matcher.set_lockstep(False)
        recomputed = True
        evs, elapsed = timed_region()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rc = 0
    if rank == 0:
        step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
        calls, stage_ms = matcher.read_profile()
        matcher.profile(False)
        sg, gf = alg_bytes_per_frame()
        if stage_ms.get("chain_h0", 0.0) / max(calls, 1) < 0.02:     # both horizontal paths run inside the fused last launch
            sg["chain_h4_wta"] += sg["chain_h0"]
            sg["chain_h0"] = 0
        if max(stage_ms.get("chain_d1", 0.0), stage_ms.get("chain_d3", 0.0)) / max(calls, 1) < 0.02:
            sg["chain_v2"] += sg["chain_d1"] + sg["chain_d3"]        # k_vdd: all three top-down paths in the "chain_v2" slot
            sg["chain_d1"] = sg["chain_d3"] = 0
        if stage_ms.get("chain_v2", 0.0) / max(calls, 1) < 0.02:     # vertical path rode inside k_cost
            sg["cost"] += sg["chain_v2"]
            sg["chain_v2"] = 0
        kernels = {}
        for name, total in stage_ms.items():
            if calls and total > 0:
                kernels[name] = {"avg_ms": total / calls, "alg_bytes": sg.get(name, 0) * B}
        gfa = 0.0
        if full:
            gfa = sum(a.elapsed_time(b) for a, b in hp.gf_ev) / max(len(hp.gf_ev), 1) if hp.gf_ev else 1e-9
            kernels["guided_upscale"] = {"avg_ms": gfa, "alg_bytes": gf["guided_sweep1+2"] * B}    # one launch (fused) or a launch pair over the batch
        dom = max((k for k in kernels if not k.startswith("guided")), key=lambda k: kernels[k]["avg_ms"])
        dk = kernels[dom]
        achieved = dk["alg_bytes"] / (dk["avg_ms"] * 1e-3) / 1e9
        traffic, tsrc = None, None          # PMC counters cannot be read in-process: taken from the committed rocprofv3 --pmc passes
        tfile, tname = latest_traffic()
        if tfile and dom in tfile.get("kernels", {}):
            traffic = int(tfile["kernels"][dom]["traffic_bytes"] * B / tfile.get("_frames_per_launch", 30))
            tsrc = tname + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)"
        sgbm_ms = sum(v["avg_ms"] for k, v in kernels.items() if not k.startswith("guided"))
        sgbm_alg = sum(sg.values()) * B
        full_alg = sgbm_alg + (gf["guided_sweep1+2"] * B if full else 0)
        p50_step = statistics.median(step_ms)
        res = {
            "metric": "1080p_sbs_to_4k_depth_frames_per_s" if full else "1080p_sbs_to_disparity_frames_per_s",
            "value": world * B * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 (SGM) + f64 (guided filter sums; f32 depth in/out)", "data": "synthetic",
            "config": {"workload": ("configs[2]: full depth.py + upscale.py hot path, 1920x1080 SBS -> 3840x2160 guided-filter depth" if full
                                    else "configs[1]: 1920x1080 SBS -> disparity (SBS split + SGBM + depth), no upscale"),
                       "frames_per_step_per_gpu": B, "distinct_frames_per_gpu": nd, "reference_batch_size": 8,
                       "numDisparities": D, "sgbm_mode": "MODE_HH (8 paths)" if args.sgbm_mode == "hh" else "MODE_SGBM (5 paths)",
                       "guided_radius": GF_R, "guided_eps": GF_EPS, "parallelism": f"frames round-robin over {world} GPU(s)",
                       "guide_exchange": (mode["v"] if side is not None else "local"), "guide_exchange_probe": exch_info,
                       "guide_exchange_order": "side stream, behind the step's lock-step SGM pass" if side is not None else None,
                       "timed_region": "HBM-resident inputs and outputs (the contract); host-link-inclusive figures under e2e"},
            "p50_ms_per_step": p50_step,
            "p50_ms_per_frame": p50_step / B,
            "p50_note": "amortised: median step time / frames per step; per-batch latencies are under e2e",
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                         "avg_launch_ms": dk["avg_ms"], "alg_bytes_per_launch": dk["alg_bytes"],
                         "sgbm_all_kernels": {"alg_bytes_per_batch": sgbm_alg, "ms_per_batch": sgbm_ms,
                                              "achieved": sgbm_alg / (sgbm_ms * 1e-3) / 1e9,
                                              "frac": sgbm_alg / (sgbm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "whole_step": {"alg_bytes_per_step": full_alg, "achieved": full_alg / (p50_step * 1e-3) / 1e9,
                                        "frac": full_alg / (p50_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "stage_ms_per_launch": {k: round(v["avg_ms"], 4) for k, v in kernels.items()}},
        }
        if full:
            res["roofline"]["guided_upscale"] = {"alg_bytes_per_frame": gf["guided_sweep1+2"], "ms_per_frame": gfa / B,
                                                 "achieved": gf["guided_sweep1+2"] / (gfa / B * 1e-3) / 1e9,
                                                 "frac": gf["guided_sweep1+2"] / (gfa / B * 1e-3) / 1e9 / HBM_PEAK_GBS}
        res["lockstep_timeouts"] = timeouts           # of the first attempt; > 0 means the timed region above is the per-direction rerun
        res["lockstep_recomputed"] = recomputed
        if recomputed:
            res["lockstep_timeouts_after_switch"] = matcher.sync_errors()

        if args.dump_frame0:
            np.savez(args.dump_frame0, disp=hp.disp[0].cpu().numpy(), q=hp.out4k[0].cpu().numpy() if full else np.zeros(0, np.float32))
        # ---- the timed region's own output against the oracle (frame 0 of the last step is base frame 0) ----
        if world == 1 and not args.no_cpu_baseline:
            cb, want_disp, want_q = cpu_baseline(base_sbs[0], base_guide[0])
            res["cpu_baseline"] = cb
            got_disp = hp.disp[0].cpu().numpy()
            bad = int((got_disp != want_disp).sum())
            pc = {"frame": 0, "disp_mismatch_px": bad, "disp_px": int(want_disp.size)}
            # frames i and i + nd of the batch carry the same content: data-dependent kernels (CCL, atomicMin keys) must agree
            pc["batch_repeats_identical"] = all(bool(torch.equal(hp.disp[i], hp.disp[i % nd])) for i in range(nd, B))
            ok = bad == 0 and pc["batch_repeats_identical"]
            if full:
                got_q = hp.out4k[0].cpu().numpy().astype(np.float64)
                rel = np.abs(got_q - want_q) / np.maximum(np.abs(want_q), 1e-6 * max(float(np.abs(want_q).max()), 1e-30))
                pc["guided_max_rel"] = float(rel.max())
                pc["guided_tolerance"] = 1e-3
                pc["guided_repeats_identical"] = all(bool(torch.equal(hp.out4k[i], hp.out4k[i % nd])) for i in range(nd, B))
                ok = ok and pc["guided_max_rel"] <= 1e-3 and pc["guided_repeats_identical"]
            pc["ok"] = bool(ok)
            res["parity_check"] = pc
            if not ok:
                rc = 3
        else:
            res["cpu_baseline"] = None
            res["parity_check"] = None

        # ---- SURVEY 8(d) config (3) as specified: host in -> host out, pipelined ----
        def guarded(fn):
            """a leg behind the headline: a lock-step time-out (another tenant took CU slots) switches the handle to
            per-direction launches and runs the leg again, and the leg says so -- never a traceback, never a silent number"""
            try:
                r = fn()
                if hp.matcher.sync_errors() == 0:
                    return r
            except N.LockstepTimeout:
                torch.cuda.synchronize()
            hp.matcher.set_lockstep(False)
            r = fn()
            if isinstance(r, dict):
                r["lockstep_recomputed"] = True
            return r

        if full and world == 1 and not args.no_e2e:
            h_sbs_p, h_gui_p = h_sbs.pin_memory(), h_gui.pin_memory()
            e2e_steps = max(4, min(args.steps, 12))
            ref0 = hp.out4k[0].cpu()                  # the resident leg's frame 0 (checked against the oracle above)
            e2e = {"what": "pinned host SBS + 4K guide -> H2D -> hot path -> D2H -> pinned host f32 4K depth; 3 streams, double-buffered",
                   "batch%d" % B: guarded(lambda: run_e2e(hp, h_sbs_p, h_gui_p, B, e2e_steps, 2, ref0))}
            if B > 8:
                e2e["batch8"] = guarded(lambda: run_e2e(hp, h_sbs_p, h_gui_p, 8, 2 * e2e_steps, 2, ref0))     # the reference's batch size (depth.py:27)
            e2e["output_equals_resident_leg"] = all(v.get("output_equals_resident_leg") is not False for v in e2e.values() if isinstance(v, dict))
            if res.get("parity_check") is not None:
                res["parity_check"]["e2e_output_equals_resident_leg"] = e2e["output_equals_resident_leg"]
                res["parity_check"]["ok"] = bool(res["parity_check"]["ok"] and e2e["output_equals_resident_leg"])
            if not e2e["output_equals_resident_leg"]:
                rc = 3
            e2e["value"] = e2e["batch%d" % B]["value"]
            e2e["unit"] = "frames/s"
            e2e["p50_ms_per_frame"] = e2e["batch%d" % B]["p50_ms_per_frame"]
            e2e["link_GBps"] = e2e["batch%d" % B]["link_GBps"]
            e2e["lockstep_timeouts"] = matcher.sync_errors()
            res["e2e"] = e2e

        # ---- the other single-GPU configs of BASELINE.json, so that the driver's line carries them too ----
        if args.workload == "all" and world == 1:
            extra = {}
            st = max(3, min(args.steps, 10))

            def rate(fn, n_frames, nsteps):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(nsteps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / nsteps
                return {"value": n_frames / ms * 1e3, "unit": "frames/s", "ms_per_step": ms, "frames_per_step": n_frames, "steps": nsteps}
            r = guarded(lambda: rate(lambda: hp.step(sbs, guide_own, upscale=False), B, st))
            r["config"] = "configs[1]: 1920x1080 SBS -> disparity (SBS split + SGBM + depth), no upscale"
            r["alg_frac_of_8TBps"] = sum(alg_bytes_per_frame()[0].values()) * B / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            extra["sgbm_only"] = r
            if B > 8:
                r = guarded(lambda: rate(lambda: hp.step(sbs[:8], guide_own[:8]), 8, st))
                r["config"] = "configs[2] at the reference's batch size 8 (depth.py:27), HBM-resident"
                extra["batch8_hbm_resident"] = r
            extra["lockstep_timeouts"] = matcher.sync_errors()
            matcher.close()
            hp.matcher = None
            if args.sgbm_mode == "sgbm":
                hh = N.StereoSGBM(W, H, B, device=dev, mode=1)
                hp.matcher = hh
                r = guarded(lambda: rate(lambda: hp.step(sbs, guide_own), B, st))
                r["config"] = "configs[2] with MODE_HH (8 SGM paths)"
                r["lockstep_timeouts"] = hh.sync_errors()
                extra["hh"] = r
                hh.close()
                hp.matcher = None
            if not args.no_cli:
                extra["cli"] = bench_cli(base_sbs, base_guide)
            c = bench_corr(args, N, dev)
            extra["corr"] = {"value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"], "config": c["config"]["workload"],
                             "roofline": c["roofline"], "dtype": c["dtype"]}
            res["extra"] = extra
        print(json.dumps(res))
    if hp.matcher is not None:
        hp.matcher.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())

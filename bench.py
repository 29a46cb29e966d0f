#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: 1080p SBS frame -> SGBM disparity -> guided-filter
3840x2160 depth (BASELINE.json metric; workload = configs[2], the configuration the metric is quoted on).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of `--batch` (default 8 = the reference's
batch_size, depth.py:27) synthetic SBS frames already resident in HBM: v3d_sbs_to_gray ->
v3d_sgbm_compute_batch -> v3d_disp_to_depth -> v3d_guided_upscale against the 4K guide -> float32 4K depth
in HBM.  Frames shard round-robin over ranks (weak scaling: every rank runs a full batch per step); the
only collective is the 4K guide round broadcast from rank 0 (RCCL), double-buffered on a side stream.
Rank 0 prints ONE JSON line.  `roofline` = the dominant kernel (largest summed duration, HIP events on the
launch stream); `cpu_baseline` = the CPU oracle (a port of the reference's OpenCV path) on one frame.
"""
import argparse
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
    workload, single thread; plus real OpenCV if the box happens to have it (it does not in this image)."""
    from oracle import oracle as O
    O.lib()
    t0 = time.perf_counter()
    gl, gr = O.sbs_to_gray(sbs, True)
    t1 = time.perf_counter()
    disp = O.sgbm_compute(gl, gr)
    t2 = time.perf_counter()
    depth = O.disp_to_depth(disp)
    O.guided_upscale(depth, guide, 8, 1e-3)
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
                O.guided_upscale(O.disp_to_depth(O.sgbm_compute(l, r)), guide, 8, 1e-3)
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
    return out


def bench_corr(args, N):
    """BASELINE configs[3]: CREStereo-style correlation lookup, bf16 in / f32 accumulate on MFMA, 1080p/4 features"""
    h, w, C, G = 270, 480, 256, 4
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    fl = torch.randn((h, w, C), device=dev).to(torch.bfloat16)
    fr = torch.randn((h, w, C), device=dev).to(torch.bfloat16)
    flow = torch.rand((2, h, w), device=dev) * 4 - 2
    for _ in range(args.warmup):
        N.corr_lookup(fl, fr, flow, G, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        N.corr_lookup(fl, fr, flow, G, 0)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / args.steps
    alg = 2 * h * w * C * 2 + 2 * h * w * 4 + G * 9 * h * w * 4           # fl + fr bf16, flow, out f32 (~142 MB form A)
    flops = 2.0 * h * w * C * 9
    print(json.dumps({"metric": "corr_lookups_per_s", "value": args.steps / el, "unit": "lookups/s", "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "bf16 in / f32 accumulate (MFMA 16x16x32)", "data": "synthetic",
                      "config": {"workload": "configs[3]: correlation lookup 270x480x256, 4 groups x 9 offsets (1x9)"},
                      "roofline": {"bound": "hbm", "kernel": "k_corr_warp + k_corr<0>", "achieved": alg / (ms * 1e-3) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                   "useful_gflops": flops / (ms * 1e-3) / 1e9,
                                   "note": "AI ~ 4 flop/B: HBM/L2 bound; MFMA only removes the VALU bottleneck"},
                      "cpu_baseline": None}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=30,
                    help="frames per step per GPU (the reference's --batch-size; 30 = what one lock-step k_vdd launch holds co-resident)")
    ap.add_argument("--guide-exchange", choices=["auto", "broadcast", "scatter", "none"], default="auto",
                    help="how rank 0 hands out the 4K guide rounds: broadcast the whole round (north_star), scatter each rank's "
                         "frames (world x fewer bytes), or auto = broadcast if it hides behind one compute step, else scatter")
    ap.add_argument("--workload", choices=["full", "sgbm", "corr"], default="full",
                    help="full = BASELINE configs[2] (default, the headline metric); sgbm = configs[1] (disparity only); "
                         "corr = configs[3] (bf16 MFMA correlation lookup, 270x480x256 features)")
    ap.add_argument("--sgbm-mode", choices=["sgbm", "hh"], default="sgbm",
                    help="sgbm = OpenCV MODE_SGBM, 5 paths (what depth.py:315-325 gets by default); hh = MODE_HH, 8 paths")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    args = ap.parse_args()

    from video_3d_pipeline import _native as N, sharding, synthetic as syn
    if args.workload == "corr":
        return bench_corr(args, N)

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
    if world > 1:
        sharding.init_process_group(args.dist_backend)
    N.lib()
    B = args.batch

    # ---- synthetic inputs (two distinct frames, tiled to the batch), resident in HBM ----
    base_sbs = [syn.sbs_frame(W, H, 2 * rank + i) for i in range(2)]
    base_guide = [syn.guide_frame(W, H, 2 * rank + i, SCALE) for i in range(2)]
    sbs = torch.from_numpy(np.stack([base_sbs[i % 2] for i in range(B)])).to(dev)
    guide_own = torch.from_numpy(np.stack([base_guide[i % 2] for i in range(B)])).to(dev)
    Hh, Wh = H * SCALE, W * SCALE
    # guide rounds: round j holds the guide frames of global frames j*world .. j*world+world-1 (rank r owns slot r)
    mode = {"v": args.guide_exchange}
    exch_info = {}
    scatter_cache = {}
    if world > 1 and args.guide_exchange != "none":
        rounds_src = torch.from_numpy(np.stack([base_guide[i % 2] for i in range(B)])).to(dev)      # [B,Hh,Wh]
        rounds_src = rounds_src[:, None].expand(B, world, Hh, Wh).contiguous() if rank == 0 else None
        side = torch.cuda.Stream(device=dev)
        guide_buf = None

        def alloc_bufs():
            return [torch.empty((B, world, Hh, Wh), dtype=torch.uint8, device=dev) if mode["v"] == "broadcast"
                    else torch.empty((B, Hh, Wh), dtype=torch.uint8, device=dev) for _ in range(2)]
    else:
        rounds_src, guide_buf, side = None, None, None

    lg = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    rg = torch.empty_like(lg)
    disp = torch.empty((B, H, W), dtype=torch.int16, device=dev)
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    out4k = torch.empty((B, Hh, Wh), dtype=torch.float32, device=dev)
    matcher = N.StereoSGBM(W, H, B, device=local, mode=1 if args.sgbm_mode == "hh" else 0)

    def exchange(slot):
        """enqueue the guide exchange for the NEXT step on the side stream"""
        if guide_buf is None:
            return None
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

    gf_ev = []

    def step(guides, timed):
        N.sbs_to_gray_batch(sbs, True, (lg, rg))
        matcher.compute(lg, rg, disp)
        N.disp_to_depth(disp, depth)
        if args.workload == "sgbm":
            return
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        N.guided_upscale_batch(depth, guides, 8, 1e-3, out4k)
        if timed:
            e1.record()
            gf_ev.append((e0, e1))

    def run(nsteps, timed):
        evs = []
        pending = exchange(0)
        for s in range(nsteps):
            slot = s & 1
            if pending is not None:
                torch.cuda.current_stream().wait_event(pending)
            nxt = exchange(1 - slot) if s + 1 < nsteps else None
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
            step(my_guides(slot), timed)
            pending = nxt
        if timed:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        return evs

    if world > 1 and args.guide_exchange != "none":
        if args.guide_exchange == "auto":
            # time one compute step and one full-round broadcast; keep the broadcast only if it hides behind the step
            mode["v"] = "broadcast"
            guide_buf = alloc_bufs()
            step(guide_own, False); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter(); step(guide_own, False); torch.cuda.synchronize(); t_step = time.perf_counter() - t0
            exchange(0); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter(); exchange(0); torch.cuda.synchronize(); dist.barrier(); t_bc = time.perf_counter() - t0
            tt = torch.tensor([t_step, t_bc], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_step, t_bc = float(tt[0]), float(tt[1])
            exch_info = {"step_ms": t_step * 1e3, "broadcast_ms": t_bc * 1e3}
            if t_bc > 0.8 * t_step:
                mode["v"] = "scatter"
                guide_buf = None
                torch.cuda.empty_cache()
        if guide_buf is None:
            guide_buf = alloc_bufs()
    run(args.warmup, False)
    torch.cuda.synchronize()
    matcher.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = run(args.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
        calls, stage_ms = matcher.read_profile()
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
        gfa = sum(a.elapsed_time(b) for a, b in gf_ev) / max(len(gf_ev), 1) if gf_ev else 1e-9
        kernels["guided_sweep1+2"] = {"avg_ms": gfa, "alg_bytes": gf["guided_sweep1+2"] * B}    # launch pair over the batch
        dom = max((k for k in kernels if not k.startswith("guided")), key=lambda k: kernels[k]["avg_ms"])
        dk = kernels[dom]
        achieved = dk["alg_bytes"] / (dk["avg_ms"] * 1e-3) / 1e9
        traffic = None                     # PMC counters cannot be read in-process: taken from the committed rocprofv3 --pmc passes
        try:
            tfile = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            tj = tfile["kernels"]
            if dom in tj:
                traffic = int(tj[dom]["traffic_bytes"] * B / tfile.get("_frames_per_launch", 30))
        except (OSError, KeyError, ValueError):
            pass
        sgbm_ms = sum(v["avg_ms"] for k, v in kernels.items() if not k.startswith("guided"))
        sgbm_alg = sum(sg.values()) * B
        res = {
            "metric": "1080p_sbs_to_4k_depth_frames_per_s" if args.workload == "full" else "1080p_sbs_to_disparity_frames_per_s",
            "value": world * B * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 (SGM) + f64 (guided filter sums; f32 depth in/out)", "data": "synthetic",
            "config": {"workload": ("configs[2]: full depth.py + upscale.py hot path, 1920x1080 SBS -> 3840x2160 guided-filter depth" if args.workload == "full"
                                    else "configs[1]: 1920x1080 SBS -> disparity (SBS split + SGBM + depth), no upscale"),
                       "frames_per_step_per_gpu": B, "numDisparities": D, "sgbm_mode": "MODE_HH (8 paths)" if args.sgbm_mode == "hh" else "MODE_SGBM (5 paths)",
                       "guided_radius": 8, "guided_eps": 1e-3, "parallelism": f"frames round-robin over {world} GPU(s)",
                       "guide_exchange": (mode["v"] if world > 1 else "local"), "guide_exchange_probe": exch_info},
            "p50_ms_per_frame": statistics.median(step_ms) / B,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)",
                         "avg_launch_ms": dk["avg_ms"], "alg_bytes_per_launch": dk["alg_bytes"],
                         "sgbm_all_kernels": {"alg_bytes_per_batch": sgbm_alg, "ms_per_batch": sgbm_ms,
                                              "achieved": sgbm_alg / (sgbm_ms * 1e-3) / 1e9,
                                              "frac": sgbm_alg / (sgbm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "guided_upscale": {"alg_bytes_per_frame": gf["guided_sweep1+2"], "ms_per_frame": gfa / B,
                                            "achieved": gf["guided_sweep1+2"] / (gfa / B * 1e-3) / 1e9},
                         "stage_ms_per_launch": {k: round(v["avg_ms"], 4) for k, v in kernels.items()}},
        }
        res["lockstep_timeouts"] = matcher.sync_errors()            # must be 0: k_vdd's bounded neighbour waits never tripped
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(base_sbs[0], base_guide[0])
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    matcher.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

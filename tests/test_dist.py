"""Multi-GPU path rehearsed on CPU: world_size-2 gloo processes.  Frames shard round-robin
(frame i -> rank i mod world) with no data-path collective; the single exchange is the
guide-round broadcast."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, tmp, clip, backend="gloo", hip=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "video-3d-pipeline_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from video_3d_pipeline import sharding
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from test_host import OracleStereoBackend, OracleUpscaleBackend
    from video_3d_pipeline.upscale import SimpleDepthUpscaler

    sharding.init_process_group(backend)
    assert sharding.rank_world() == (rank, world)
    gpu = backend == "nccl" or hip                             # the product's HIP backends (RCCL: one GPU per rank; gloo + hip: ranks share GPU 0)
    dev = torch.device("cuda", torch.cuda.current_device()) if gpu else torch.device("cpu")
    if backend == "nccl":
        assert torch.cuda.current_device() == rank             # init_process_group selected LOCAL_RANK's GPU
    # 1. guide round broadcast: rank 0 owns the frames, every rank gets its own slot back
    H, W = 6, 10
    rounds = [np.full((H, W), 10 * (r + 1), np.uint8) for r in range(world)] if rank == 0 else None
    mine = sharding.broadcast_guide_round(rounds, (H, W), dev)
    assert mine is not None and int(mine[0, 0]) == 10 * (rank + 1) and mine.shape == (H, W)
    tail = [np.full((H, W), 7, np.uint8)] + [None] * (world - 1) if rank == 0 else None      # ragged last round
    mine = sharding.broadcast_guide_round(tail, (H, W), dev)
    assert (mine is not None) == (rank == 0)
    # 1b. the persistent double-buffered exchange: rounds posted one ahead, ragged tail, one collective per round
    ex = sharding.GuideRoundExchange((H, W), dev)
    nfr = 2 * world + 1                                        # two full rounds + one frame
    def rnd(b):
        return [np.full((H, W), 1 + b + r, np.uint8) if b + r < nfr else None for r in range(world)] if rank == 0 else None
    ex.post(rnd(0))
    for b in range(0, nfr, world):
        if b + world < nfr:
            ex.post(rnd(b + world))
        g = ex.take()
        i = b + rank
        assert (g is not None) == (i < nfr)
        if g is not None:
            assert g.dtype == torch.uint8 and int(g[0, 0]) == 1 + i and int(g[-1, -1]) == 1 + i
    # 2. sharded depth extraction writes disjoint frame sets into one cache dir; each rank decodes ONLY its own frames
    ex = HybridStereoDepthExtractor(work_dir=os.path.join(tmp, "w"), cache_dir=os.path.join(tmp, "w"), batch_size=2,
                                    stereo_only=True, backend=None if gpu else OracleStereoBackend())
    if gpu:
        assert ex.backend.device.index == torch.cuda.current_device()      # a bare "cuda" resolved to this rank's GPU
    out = ex.process_video_sbs(clip, max_frames=5, force_reprocess=True)
    dist.barrier()
    if gpu:
        assert ex.backend._matcher.device.index == torch.cuda.current_device() and ex.backend._matcher.sync_errors() == 0
    assert ex.last_decoded_frames == len(range(rank, 5, world))          # ceil(5/2) on rank 0, floor on rank 1
    files = sorted(os.listdir(out))
    assert files == [f"depth_{i:06d}.png" for i in range(5)], files
    # 3. sharded guided upscale: rank 0 decodes the guide clip, rounds travel through the exchange, ragged last round,
    #    4 guide frames for 5 depth maps (the last one is beyond the 4K clip: flat guide by design)
    up = SimpleDepthUpscaler(backend=None if gpu else OracleUpscaleBackend())
    up.upscale_depth_maps_ffmpeg(str(out), 320, 48, os.path.join(tmp, "up.mp4"), video_4k_path=os.path.join(tmp, "guide.npy"))
    dist.barrier()
    assert sorted(os.listdir(os.path.join(tmp, "up_frames"))) == [f"depth4k_{i:06d}.png" for i in range(5)]
    dist.destroy_process_group()


def _world8_worker(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "video-3d-pipeline_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from video_3d_pipeline import sharding
    sharding.init_process_group("gloo")
    H, W = 4, 6
    dev = torch.device("cpu")
    for tail in range(1, world):                               # ragged last round of 1 .. 7 frames
        ex = sharding.GuideRoundExchange((H, W), dev, depth=4)
        nfr = 3 * world + tail
        nround = -(-nfr // world)

        def rnd(b):
            return [np.full((H, W), (3 + b + r) % 251, np.uint8) if b + r < nfr else None for r in range(world)] if rank == 0 else None
        for k in range(min(3, nround)):                        # three rounds posted ahead of the first take
            ex.post(rnd(k * world))
        for k in range(nround):
            if k + 3 < nround:
                ex.post(rnd((k + 3) * world))
            g = ex.take()
            i = k * world + rank
            assert (g is not None) == (i < nfr), (tail, k, rank)
            if g is not None:
                assert g.dtype == torch.uint8 and tuple(g.shape) == (H, W) and int(g[0, 0]) == (3 + i) % 251 and int(g[-1, -1]) == (3 + i) % 251
        # the validity bitmap of the last round as it arrived: bits 0 .. tail-1
        meta = ex._bufs[(nround - 1) % 4][world * H * W:].numpy()
        assert int(meta[0]) == (1 << tail) - 1 and not meta[1:].any(), (tail, meta)
    # a failing root: the aborted round makes EVERY rank raise, nobody waits in a collective
    ex = sharding.GuideRoundExchange((H, W), dev)
    ex.post([np.zeros((H, W), np.uint8)] * world if rank == 0 else None)
    assert ex.take() is not None
    ex.post(abort=True) if rank == 0 else ex.post()
    with pytest.raises(sharding.GuideExchangeAborted):
        ex.take()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_guide_exchange_world8_gloo():
    """the one collective at the world size it is meant for (8 ranks, CPU rehearsal): validity bitmap bytes, every ragged
    tail 1 .. 7, three rounds posted ahead (depth-4 ring), and the abort flag of a failing root"""
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_world8_worker, args=(8, port), nprocs=8, join=True)


def _abort_worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "video-3d-pipeline_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from video_3d_pipeline import sharding, upscale
    from test_host import OracleUpscaleBackend
    sharding.init_process_group("gloo")
    real = upscale.iter_frames

    def broken(path, start, count, **kw):                      # the guide decoder dies after two frames
        for k, f in enumerate(real(path, start, count, **kw)):
            if k == 2:
                raise IOError("decoder lost the stream")
            yield f
    upscale.iter_frames = broken
    up = upscale.SimpleDepthUpscaler(backend=OracleUpscaleBackend())
    with pytest.raises((IOError, sharding.GuideExchangeAborted)) as ei:
        up.upscale_depth_maps_ffmpeg(os.path.join(tmp, "d"), 64, 40, os.path.join(tmp, "up.mp4"), video_4k_path=os.path.join(tmp, "guide.npy"))
    assert isinstance(ei.value, IOError if rank == 0 else sharding.GuideExchangeAborted)
    dist.barrier()                                             # both ranks got here: nobody hung in the broadcast
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_guide_decoder_failure_raises_on_every_rank(tmp_path):
    """rank 0's guide decoder failing must not leave the other ranks blocked in the exchange"""
    sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
    from video_3d_pipeline import synthetic as syn
    from video_3d_pipeline.utils import write_png16
    d = tmp_path / "d"
    d.mkdir()
    rng = np.random.default_rng(5)
    for i in range(6):
        write_png16(d / f"depth_{i:06d}.png", rng.integers(0, 65535, (20, 32)).astype(np.uint16))
    guides = np.stack([np.repeat(syn.guide_frame(32, 20, i)[..., None], 3, axis=2) for i in range(6)])
    np.save(str(tmp_path / "guide.npy"), guides)
    port = 29800 + (os.getpid() % 2000)
    mp.spawn(_abort_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_round_robin_assignment():
    from video_3d_pipeline import sharding
    assert sharding.my_frames(10, 1, 4) == [1, 5, 9]
    owners = [[r for r in range(8) if sharding.owns(i, r, 8)] for i in range(20)]
    assert all(len(o) == 1 for o in owners) and [o[0] for o in owners] == [i % 8 for i in range(20)]
    assert sharding.rank_world() == (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def _cuda_exchange_worker(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "video-3d-pipeline_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from video_3d_pipeline import sharding
    sharding.init_process_group("gloo")                       # transport on the host; buffers, streams and events on the GPU
    dev = torch.device("cuda", 0)
    H, W = 216, 384
    ex = sharding.GuideRoundExchange((H, W), dev)
    nfr = 7 * world + 1

    def rnd(b):
        return [torch.full((H, W), (1 + b + r) % 251, dtype=torch.uint8, device=dev) if b + r < nfr else None
                for r in range(world)] if rank == 0 else None
    ex.post(rnd(0))
    for b in range(0, nfr, world):
        if b + world < nfr:
            ex.post(rnd(b + world))                           # one round ahead, on the exchange's side stream
        g = ex.take()
        i = b + rank
        assert (g is not None) == (i < nfr)
        if g is not None:
            assert g.is_cuda and int(g.to(torch.int64).sum().item()) == ((1 + i) % 251) * H * W, i
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_guide_exchange_streams_on_one_gpu():
    """the exchange's device path -- persistent double-buffered round buffers, side stream, events, pinned validity bytes,
    slot reuse behind the consumer's copy -- with two ranks sharing the one GPU (gloo carries the bytes): 8 rounds, ragged tail"""
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_cuda_exchange_worker, args=(2, port), nprocs=2, join=True)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_rccl(tmp_path):
    """the same scenario on two real GPUs over RCCL (backend nccl) with the product's HIP backends: guide exchange on a
    side stream, sharded extraction with one matcher per GPU, sharded guided upscale.  Skips on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _two_rank_scenario(tmp_path, "nccl")


@pytest.mark.timeout(180)
def test_two_rank_gloo(tmp_path):
    _two_rank_scenario(tmp_path, "gloo")


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_ranks_share_one_gpu(tmp_path):
    """the sharded product path with the real HIP backends on a one-GPU box: two ranks on GPU 0, gloo as transport --
    strided decode, one matcher per rank, the guide exchange on device buffers, sharded guided upscale"""
    _two_rank_scenario(tmp_path, "gloo", hip=True)


def _two_rank_scenario(tmp_path, backend, hip=False):
    sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
    from video_3d_pipeline import synthetic as syn
    frames = np.stack([syn.sbs_frame(160, 24, i) for i in range(5)])
    clip = str(tmp_path / "clip.npy")
    np.save(clip, frames)
    guides = np.stack([syn.guide_frame(160, 24, i) for i in range(4)])            # 320 x 48 luma, one frame short of the depth maps
    np.save(str(tmp_path / "guide.npy"), np.repeat(guides[..., None], 3, axis=3))
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), clip, backend, hip), nprocs=2, join=True)
    # sharded result == single-process result
    from oracle import oracle as O
    from video_3d_pipeline.utils import read_png16
    import glob
    d = glob.glob(str(tmp_path / "w" / "depth_*"))[0]
    for i in (0, 3):
        l, r = O.sbs_to_gray(frames[i], True)
        want = O.depth_to_u16(O.disp_to_depth(O.sgbm_compute(l, r)))
        assert np.array_equal(read_png16(os.path.join(d, f"depth_{i:06d}.png")), want)
    # sharded upscale == what one process computes: every frame guided by ITS 4K frame (never a flat guide inside the clip)
    for i in range(5):
        lo = read_png16(os.path.join(d, f"depth_{i:06d}.png")).astype(np.float32)
        g = O.bgr_to_gray(np.repeat(guides[i][..., None], 3, axis=2)) if i < 4 else np.full((48, 320), 128, np.uint8)
        want = np.clip(np.rint(O.guided_upscale(lo, g, 8, 1e-3)), 0, 65535)
        got = read_png16(str(tmp_path / "up_frames" / f"depth4k_{i:06d}.png")).astype(np.float64)
        assert np.abs(got - want).max() <= 1, (i, np.abs(got - want).max())       # rounding of the f32 result at .5
        if i < 4:                                                                  # ... and a flat guide would be far off
            flat = O.guided_upscale(lo, np.full((48, 320), 128, np.uint8), 8, 1e-3)
            assert np.abs(got - flat).max() > 50

"""Multi-GPU path rehearsed on CPU: world_size-2 gloo processes.  Frames shard round-robin
(frame i -> rank i mod world) with no data-path collective; the single exchange is the
guide-round broadcast."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, tmp, clip):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "video-3d-pipeline_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from video_3d_pipeline import sharding
    from video_3d_pipeline.depth import HybridStereoDepthExtractor
    from test_host import OracleStereoBackend

    sharding.init_process_group("gloo")
    assert sharding.rank_world() == (rank, world)
    # 1. guide round broadcast: rank 0 owns the frames, every rank gets its own slot back
    H, W = 6, 10
    rounds = [np.full((H, W), 10 * (r + 1), np.uint8) for r in range(world)] if rank == 0 else None
    mine = sharding.broadcast_guide_round(rounds, (H, W), torch.device("cpu"))
    assert mine is not None and int(mine[0, 0]) == 10 * (rank + 1) and mine.shape == (H, W)
    tail = [np.full((H, W), 7, np.uint8)] + [None] * (world - 1) if rank == 0 else None      # ragged last round
    mine = sharding.broadcast_guide_round(tail, (H, W), torch.device("cpu"))
    assert (mine is not None) == (rank == 0)
    # 2. sharded depth extraction writes disjoint frame sets into one cache dir
    ex = HybridStereoDepthExtractor(work_dir=os.path.join(tmp, "w"), cache_dir=os.path.join(tmp, "w"), batch_size=2,
                                    stereo_only=True, backend=OracleStereoBackend())
    out = ex.process_video_sbs(clip, max_frames=5, force_reprocess=True)
    dist.barrier()
    files = sorted(os.listdir(out))
    assert files == [f"depth_{i:06d}.png" for i in range(5)], files
    dist.destroy_process_group()


def test_round_robin_assignment():
    from video_3d_pipeline import sharding
    assert sharding.my_frames(10, 1, 4) == [1, 5, 9]
    owners = [[r for r in range(8) if sharding.owns(i, r, 8)] for i in range(20)]
    assert all(len(o) == 1 for o in owners) and [o[0] for o in owners] == [i % 8 for i in range(20)]
    assert sharding.rank_world() == (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


@pytest.mark.timeout(180)
def test_two_rank_gloo(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "video-3d-pipeline_amd"))
    from video_3d_pipeline import synthetic as syn
    frames = np.stack([syn.sbs_frame(160, 24, i) for i in range(5)])
    clip = str(tmp_path / "clip.npy")
    np.save(clip, frames)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), clip), nprocs=2, join=True)
    # sharded result == single-process result
    from oracle import oracle as O
    from video_3d_pipeline.utils import read_png16
    import glob
    d = glob.glob(str(tmp_path / "w" / "depth_*"))[0]
    for i in (0, 3):
        l, r = O.sbs_to_gray(frames[i], True)
        want = O.depth_to_u16(O.disp_to_depth(O.sgbm_compute(l, r)))
        assert np.array_equal(read_png16(os.path.join(d, f"depth_{i:06d}.png")), want)

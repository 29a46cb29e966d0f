"""Frame sharding across the GPUs of one node.

The reference is single-process (frames strictly serial, depth.py:329); frames are mutually
independent, so the only parallelism the path needs is data parallelism over frames:
frame i -> rank i mod world, one process per GPU, no data-path collective for depth.  The single
real exchange is the 4K guide: rank 0 decodes a round of `world` guide frames and broadcasts the
round over RCCL/xGMI (root -> 7 peers uses all 7 links in parallel); each rank keeps its own frame.
"""
import os

import numpy as np


def rank_world():
    """(rank, world) from torch.distributed if initialised, else from the torchrun environment"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def owns(frame_idx: int, rank: int, world: int) -> bool:
    return frame_idx % world == rank


def my_frames(n_frames: int, rank: int, world: int):
    return list(range(rank, n_frames, world))


def init_process_group(backend=None):
    """one process per GPU; backend 'nccl' is RCCL on ROCm, 'gloo' for CPU rehearsals"""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend, device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group(backend=backend)


def barrier():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
    except ImportError:
        pass


def broadcast_guide_round(round_frames, shape, device, src=0):
    """Rank `src` passes a list of `world` guide frames (HxW uint8 arrays, None-padded at the tail of
    the clip); every rank gets back its own frame as a device tensor (or None).  One broadcast of
    the whole round [world, H, W] u8, as BASELINE.json's north_star specifies."""
    import torch
    import torch.distributed as dist
    rank, world = rank_world()
    H, W = shape
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        f = round_frames[0] if round_frames else None
        if f is None or not isinstance(f, np.ndarray):          # already a device tensor (single-rank fast path)
            return f
        return torch.from_numpy(np.ascontiguousarray(f)).to(device)
    buf = torch.zeros((world, H, W), dtype=torch.uint8, device=device)
    valid = torch.zeros(world, dtype=torch.uint8, device=device)
    if rank == src:
        for i, f in enumerate(round_frames[:world]):
            if f is not None:
                buf[i] = torch.from_numpy(np.ascontiguousarray(f)).to(device)
                valid[i] = 1
    dist.broadcast(buf, src=src)
    dist.broadcast(valid, src=src)
    return buf[rank].clone() if int(valid[rank]) else None

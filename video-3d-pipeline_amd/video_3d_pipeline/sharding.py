"""Frame sharding across the GPUs of one node.

The reference is single-process (frames strictly serial, depth.py:329); frames are mutually
independent, so the only parallelism the path needs is data parallelism over frames:
frame i -> rank i mod world, one process per GPU, no data-path collective for depth.  The single
real exchange is the 4K guide: rank 0 decodes a round of `world` guide frames and broadcasts the
round over RCCL/xGMI (root -> 7 peers uses all 7 links in parallel); each rank keeps its own frame.
"""
import os
from collections import deque

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


def _initialized() -> bool:
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except ImportError:
        return False


def require_initialized(world: int) -> None:
    """A world of several ranks needs a live process group: with WORLD_SIZE > 1 only in the environment the barriers
    would be no-ops and the guide exchange would silently hand every rank but 0 a flat guide.  Fail loudly."""
    if world > 1 and not _initialized():
        raise RuntimeError(f"WORLD_SIZE={world} but torch.distributed is not initialised: call "
                           "video_3d_pipeline.sharding.init_process_group() first (the CLIs do)")


def owns(frame_idx: int, rank: int, world: int) -> bool:
    return frame_idx % world == rank


def my_frames(n_frames: int, rank: int, world: int):
    return list(range(rank, n_frames, world))


def init_process_group(backend=None):
    """one process per GPU; backend 'nccl' is RCCL on ROCm, 'gloo' for CPU rehearsals.  Sets the current device to
    LOCAL_RANK so that every later "cuda" resolves to this rank's GPU (see _native.resolve_device)."""
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
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend)


def barrier():
    rank, world = rank_world()
    require_initialized(world)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def total(value: int) -> int:
    """sum of a per-rank integer over the world (host-side bookkeeping, e.g. decoded-frame counts)"""
    rank, world = rank_world()
    require_initialized(world)
    if world == 1:
        return int(value)
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    return int(t.item())


class GuideExchangeAborted(RuntimeError):
    """the root of the guide exchange failed while decoding; raised by take() on every rank"""


class GuideRoundExchange:
    """The one collective of the path: rank `src` decodes a round of `world` 4K guide frames, one broadcast
    hands the round to every rank, rank r keeps frame r.

    * the luma never leaves the device on the root: `post` copies device tensors into the round buffer;
    * persistent, double-buffered `[world*H*W + 16]` uint8 round buffers -- no allocation per round;
    * ONE collective per round: which frames of the round exist (ragged end of the clip, or a decoder that ends before
      the container's frame count) travels in the 16 trailing bytes of the same buffer, and so does an ABORT flag:
      a root whose decoder fails posts an aborted round, every rank's `take()` raises `GuideExchangeAborted`, nobody
      is left blocked in the collective;
    * the collective runs on a side stream, so `post(round r+1)` overlaps the compute of round r; `take()` makes the
      current stream wait for the oldest posted round and returns this rank's frame (a private copy) or None.
    On the CPU (`gloo` rehearsals) the same calls run synchronously."""

    META = 16                  # trailing bytes: validity bitmap in bytes 0..14 (120 ranks), abort flag in byte 15
    ABORT = 15

    def __init__(self, shape, device, src=0, depth=2):
        import torch
        import torch.distributed as dist
        self.rank, self.world = rank_world()
        require_initialized(self.world)
        if self.world > 8 * self.ABORT:
            raise ValueError(f"world {self.world} too large for the {8 * self.ABORT}-bit validity bitmap")
        self.H, self.W = int(shape[0]), int(shape[1])
        self.device = torch.device(device)
        self.src, self.depth = src, depth
        self.cuda = self.device.type == "cuda"
        n = self.world * self.H * self.W
        self._bufs = [torch.zeros(n + self.META, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self._meta_host = [torch.zeros(self.META, dtype=torch.uint8).pin_memory() if self.cuda else None for _ in range(depth)]
        self._free_ev = [None] * depth            # slot may be overwritten once this event has passed (consumer copy done)
        self._side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._pending = deque()
        self._posted = 0
        self._dist = dist

    def post(self, round_frames=None, abort=False):
        """enqueue one round.  On `src`: a list of up to `world` entries, each a HxW uint8 device tensor / NumPy array or
        None (no such frame); other ranks pass nothing.  abort=True (on `src`): the round carries no frames and makes
        every rank's take() raise."""
        import torch
        # never enter another collective behind an aborted round: the root has left.  The youngest round already posted
        # must have arrived before the next one is issued (with the usual one-round-ahead pattern it has: take() of the
        # same round follows at once), and its abort flag is looked at first.
        if self._pending:
            pslot, pev = self._pending[-1]
            if self._round_aborted(pslot, pev):
                raise GuideExchangeAborted(f"rank {self.src} aborted the guide exchange (its decoder failed)")
        slot = self._posted % self.depth
        self._posted += 1
        buf = self._bufs[slot]
        n = self.world * self.H * self.W
        cur = torch.cuda.current_stream(self.device) if self.cuda else None
        if self.cuda and self._free_ev[slot] is not None:
            cur.wait_event(self._free_ev[slot])
            self._side.wait_event(self._free_ev[slot])
        if self.rank == self.src:
            frames = [] if abort else list(round_frames or [])[:self.world]
            valid = np.zeros(self.META, np.uint8)
            valid[self.ABORT] = 1 if abort else 0
            view = buf[:n].view(self.world, self.H, self.W)
            for i, f in enumerate(frames):
                if f is None:
                    continue
                t = f if torch.is_tensor(f) else torch.from_numpy(np.ascontiguousarray(f))
                if tuple(t.shape) != (self.H, self.W) or t.dtype != torch.uint8:
                    raise ValueError(f"guide frame {i}: expected uint8 {self.H}x{self.W}, got {t.dtype} {tuple(t.shape)}")
                view[i].copy_(t, non_blocking=True)
                valid[i // 8] |= 1 << (i % 8)
            buf[n:].copy_(torch.from_numpy(valid), non_blocking=False)
            if self.cuda:
                e = torch.cuda.Event()
                e.record(cur)
                self._side.wait_event(e)
        if self.cuda:
            with torch.cuda.stream(self._side):
                self._dist.broadcast(buf, src=self.src)
                self._meta_host[slot].copy_(buf[n:], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._side)
        else:
            self._dist.broadcast(buf, src=self.src)
            ev = None
        self._pending.append((slot, ev))

    def _round_aborted(self, slot, ev):
        if self.cuda:
            ev.synchronize()
            return bool(int(self._meta_host[slot].numpy()[self.ABORT]))
        n = self.world * self.H * self.W
        return bool(int(self._bufs[slot][n + self.ABORT]))

    def take(self):
        """this rank's frame of the oldest posted round as a private HxW uint8 tensor, or None if the clip had none"""
        import torch
        slot, ev = self._pending.popleft()
        buf = self._bufs[slot]
        n = self.world * self.H * self.W
        if self.cuda:
            ev.synchronize()                              # normally long past: the round was posted one round ago
            meta = self._meta_host[slot].numpy()
            torch.cuda.current_stream(self.device).wait_event(ev)
        else:
            meta = buf[n:].numpy()
        if int(meta[self.ABORT]):
            raise GuideExchangeAborted(f"rank {self.src} aborted the guide exchange (its decoder failed)")
        if not (int(meta[self.rank // 8]) >> (self.rank % 8)) & 1:
            return None
        out = buf[:n].view(self.world, self.H, self.W)[self.rank].clone()
        if self.cuda:
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(self.device))
            self._free_ev[slot] = e
        return out


def broadcast_guide_round(round_frames, shape, device, src=0):
    """One-shot form of GuideRoundExchange (kept for callers that exchange a single round): rank `src` passes a list of
    `world` guide frames (None-padded at the tail of the clip); every rank gets back its own frame as a device tensor
    (or None)."""
    import torch
    rank, world = rank_world()
    require_initialized(world)
    if world == 1:
        f = round_frames[0] if round_frames else None
        if f is None or torch.is_tensor(f):
            return f
        return torch.from_numpy(np.ascontiguousarray(f)).to(device)
    ex = GuideRoundExchange(shape, device, src=src, depth=1)
    ex.post(round_frames if rank == src else None)
    return ex.take()

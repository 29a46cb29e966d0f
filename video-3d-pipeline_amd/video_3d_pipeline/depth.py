"""Depth extraction from SBS stereoscopic video -- MI355X-native host side.

Mirror of reference src/video_3d_pipeline/depth.py (class, method names, argument meaning, error
behaviour, CLI flags) with the per-frame arithmetic moved from OpenCV-on-CPU to libv3d_hip.so:

    split_sbs_frame     depth.py:250-268  -> v3d_split_sbs / v3d_sbs_to_gray
    process_frame_batch depth.py:297-395  -> v3d_bgr_to_gray + v3d_sgbm_compute_batch + v3d_disp_to_depth
    save_depth_map      depth.py:397-406  -> v3d_depth_to_u16 + 16-bit PNG (Pillow)
    process_video_sbs   depth.py:408-476  -> streaming decode -> fused SBS batch path -> PNG cache
    main                depth.py:479-538  -> same argparse surface, exit code 0/1

There is no CPU compute path: without a GPU (or without libv3d_hip.so) construction fails loudly,
like the reference's `RuntimeError("CUDA not available but requested")` (depth.py:43-44).
Neural guidance (the reference's DPT blend, depth.py:344-371): the blend itself -- resize of the monocular map,
min-max to the disparity range, 0.7 / 0.3 mix, clamp -- runs on the GPU (v3d_mono_blend).  The monocular map comes
from a provider: `DPTForDepthEstimation` loaded from a LOCAL directory (or an already populated HF cache; nothing is
ever downloaded), or any callable handed to the constructor (`mono_provider`).  When neither is available the loader
falls back to stereo-only exactly like the reference does when loading fails (depth.py:107-114).
"""
import argparse
import hashlib
from collections import defaultdict
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

from .utils import PngWriterPool, create_work_directory, get_video_info, iter_frames, write_png16


class HipStereoBackend:
    """The product compute backend: PyTorch-ROCm buffers + libv3d_hip.so kernels."""

    def __init__(self, device: str = "cuda", sgbm_params: Optional[Dict] = None):
        import torch
        from . import _native
        if not torch.cuda.is_available():
            raise RuntimeError("CUDA not available but requested")
        self.torch = torch
        self.native = _native
        _native.lib()                                  # fail now, loudly, if the HIP library is missing
        # resolved ONCE: a bare "cuda" is the current device (what torch.cuda.set_device / LOCAL_RANK selected), and
        # tensors, the matcher's workspace and its kernels all live on that same index
        self.device = _native.resolve_device(device)
        self.sgbm_params = dict(sgbm_params or {})
        self._matcher = None
        self._geom = None

    def _get_matcher(self, W, H, n):
        g = self._geom
        if self._matcher is None or g[0] < W or g[1] < H or g[2] < n:
            if self._matcher is not None:
                self._matcher.close()
            self._matcher = self.native.StereoSGBM(W, H, n, device=self.device, **self.sgbm_params)
            self._geom = (W, H, n)
        return self._matcher

    @staticmethod
    def _lockstep_ok(matcher) -> bool:
        """The lock-step SGM kernel needs all its workgroups resident at once; if another process is hogging the GPU
        its bounded spins give up and raise a flag.  Then the batch is recomputed with one launch per direction
        (still on the GPU, ~2x the SGM time, same bits) and the handle stays in that mode -- never a wrong disparity,
        never a CPU path."""
        n = matcher.sync_errors()
        if n == 0:
            return True
        print(f"warning: SGM lock-step kernel timed out waiting for neighbour strips ({n} workgroups): the GPU is "
              "over-subscribed; recomputing this batch and continuing with one launch per direction")
        matcher.set_lockstep(False)
        return False

    def compute_batch_size(self, W: int, H: int, requested: int) -> int:
        """frames per device pass of the streaming path (process_video_sbs).  The caller's batch_size is list chunking
        in the reference (depth.py:448-461); here a pass should fill one lock-step SGM launch -- two workgroup slots per
        CU over the 128-column strips of a frame, 34 frames at 1080p -- because a smaller pass leaves CUs idle (batch 8:
        0.58 ms per frame, batch 34: 0.36).  Bounded by free HBM: the matcher's workspace is ~290 B per pixel."""
        torch = self.torch
        props = torch.cuda.get_device_properties(self.device)
        strips = max(1, -(-(W - 64) // 128))
        fill = max(1, (2 * props.multi_processor_count) // strips)
        free, _ = torch.cuda.mem_get_info(self.device)
        per_frame = 290 * W * H + 16 * W * H                     # workspace + staging / result tensors of this class
        fit = max(1, int(0.5 * free) // per_frame)
        if self._matcher is not None:                            # the workspace already allocated is not "free"
            fit = max(fit, self._geom[2])
        n = min(fill, fit)
        if requested > n:                                        # a caller asking for more gets whole launches
            n = min((requested // fill) * fill or requested, fit)
        return max(1, n)

    def split_sbs(self, sbs_frame: np.ndarray, unsqueeze: bool):
        d = self.native.to_device(sbs_frame, self.device)
        L, R = self.native.split_sbs(d, unsqueeze)
        return L.cpu().numpy(), R.cpu().numpy()

    def _depth_from(self, disp, monos, out=None):
        """disp16 [n,H,W] -> float32 depth: /16 + clamp (depth.py:341, 374), or, with monocular maps, the hybrid blend
        of depth.py:344-374 (maps of one shape go through one batched launch set)"""
        torch, nat = self.torch, self.native
        if monos is None:
            return nat.disp_to_depth(disp, out)
        if len(monos) != disp.shape[0]:
            raise ValueError(f"{len(monos)} monocular maps for {disp.shape[0]} frames")
        ms = [m if torch.is_tensor(m) else torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)) for m in monos]
        ms = [m.to(self.device, torch.float32).contiguous() for m in ms]
        if any(m.dim() != 2 for m in ms):
            raise ValueError("monocular depth maps must be 2-D")
        if out is None:
            out = torch.empty(disp.shape, dtype=torch.float32, device=self.device)
        if len({tuple(m.shape) for m in ms}) == 1:
            nat.mono_blend(disp, torch.stack(ms), 0.7, 0.3, out)
        else:
            for i, m in enumerate(ms):
                nat.mono_blend(disp[i], m, 0.7, 0.3, out[i])
        return out

    def pairs_to_disparity(self, pairs: List[Tuple[np.ndarray, np.ndarray]], monos=None) -> List[np.ndarray]:
        """BGR (left, right) pairs -> float32 disparity maps (>= 0), depth.py:337-341 + 374 (+ 344-371 with `monos`)"""
        torch, nat = self.torch, self.native
        n = len(pairs)
        H, W = pairs[0][0].shape[:2]
        lg = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        rg = torch.empty_like(lg)
        for i, (l, r) in enumerate(pairs):
            lg[i] = nat.bgr_to_gray(nat.to_device(l, self.device))
            rg[i] = nat.bgr_to_gray(nat.to_device(r, self.device))
        matcher = self._get_matcher(W, H, n)
        disp = matcher.compute(lg, rg)
        if not self._lockstep_ok(matcher):
            disp = matcher.compute(lg, rg)
            if matcher.sync_errors():
                raise RuntimeError("SGM kernels report time-outs with the lock-step pass off: device fault")
        depth = self._depth_from(disp, monos)
        out = depth.cpu().numpy()
        return [out[i] for i in range(n)]

    def _staging(self, key, shape, dtype, pinned):
        """reusable buffers: pinned host staging + device tensors (no allocation on the steady-state path)"""
        torch = self.torch
        bufs = self.__dict__.setdefault("_bufs", {})
        t = bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=dtype, pin_memory=True) if pinned else torch.empty(shape, dtype=dtype, device=self.device)
            bufs[key] = t
        return t

    def sbs_to_disparity(self, frames: List[np.ndarray], unsqueeze: bool, mono_provider=None):
        """fused path: SBS BGR frames -> device float32 disparity [n,H,W] (no BGR halves materialised).
        Frames are gathered into one pinned buffer and cross PCIe in a single asynchronous copy.
        mono_provider (neural guidance on): called with the left views as RGB arrays, its maps are blended in."""
        torch, nat = self.torch, self.native
        n = len(frames)
        H, W = frames[0].shape[:2]
        ow = W if unsqueeze else W // 2
        host = self._staging("sbs_host", (n, H, W, 3), torch.uint8, True)
        hview = host.numpy()
        for i, f in enumerate(frames):
            hview[i] = f
        dev = self._staging("sbs_dev", (n, H, W, 3), torch.uint8, False)
        dev.copy_(host, non_blocking=True)
        lg = self._staging("lg", (n, H, ow), torch.uint8, False)
        rg = self._staging("rg", (n, H, ow), torch.uint8, False)
        nat.sbs_to_gray_batch(dev, unsqueeze, (lg, rg))
        matcher = self._get_matcher(ow, H, n)
        disp = matcher.compute(lg, rg, self._staging("disp", (n, H, ow), torch.int16, False))
        if not self._lockstep_ok(matcher):
            disp = matcher.compute(lg, rg, self._staging("disp", (n, H, ow), torch.int16, False))
            if matcher.sync_errors():
                raise RuntimeError("SGM kernels report time-outs with the lock-step pass off: device fault")
        monos = None
        out = self._staging("depth", (n, H, ow), torch.float32, False)
        if mono_provider is not None:
            # a failing provider (DPT forward out of memory, bad shape ...) must not abort the clip -- nor, in a sharded run,
            # leave the other ranks waiting in the final barrier: warn and continue stereo-only like depth.py:367-369
            try:
                lefts = [nat.split_sbs(dev[i], unsqueeze)[0].flip(-1).cpu().numpy() for i in range(n)]     # left view, RGB (depth.py:274)
                monos = mono_provider(lefts)
                return self._depth_from(disp, monos, out)
            except Exception as e:
                print(f"    Warning: Neural guidance failed, using stereo only: {e}")
        return self._depth_from(disp, None, out)

    def depth_to_host(self, depth) -> np.ndarray:
        """device float32 [n,H,W] -> NumPy through a pinned buffer"""
        host = self._staging("depth_host", tuple(depth.shape), self.torch.float32, True)
        host.copy_(depth, non_blocking=True)
        self.torch.cuda.current_stream().synchronize()
        return host.numpy().copy()

    def normalise_u16(self, depth) -> np.ndarray:
        nat = self.native
        d = depth if self.torch.is_tensor(depth) else nat.to_device(np.asarray(depth, np.float32), self.device)
        return nat.depth_to_u16(d.contiguous()).cpu().numpy().view(np.uint16)


class HybridStereoDepthExtractor:
    """ GPU-accelerated depth extraction from SBS video using hybrid stereo matching + neural guidance """

    # where the 16-bit maps of process_video_sbs go: anything with PngWriterPool's submit(path, uint16 image) / context-
    # manager surface (bench.py swaps in a raw sink to time the path without zlib)
    writer_pool_factory = PngWriterPool

    def __init__(self,
                 model_checkpoint: str = "Intel/dpt-large",
                 work_dir: str = "temp_depth",
                 cache_dir: str = "temp_depth",
                 device: str = "cuda",
                 batch_size: int = 8,
                 use_neural_guidance: bool = True,
                 stereo_only: bool = False,
                 unsqueeze_sbs: bool = True,
                 backend=None,
                 mono_provider=None):
        """ mono_provider: optional callable(list of HxWx3 uint8 RGB left views) -> list of 2-D float32 monocular
        depth maps (NumPy arrays or device tensors, any size); takes the place of the DPT forward of depth.py:348-350 """

        self.device = device
        self.work_dir = create_work_directory(work_dir)
        self.cache_dir = create_work_directory(cache_dir)
        self.batch_size = batch_size
        self.model_checkpoint = model_checkpoint
        self.use_neural_guidance = use_neural_guidance
        self.stereo_only = stereo_only
        self.unsqueeze_sbs = unsqueeze_sbs

        # `backend` exists so host-logic tests can run without a GPU; the product always builds the HIP one
        if backend is None:
            if not str(device).startswith("cuda"):
                raise RuntimeError(f"device {device!r} requested, but this build only has the MI355X (HIP) path")
            backend = HipStereoBackend(device)
        self.backend = backend

        print(f"Initializing Hybrid Stereo depth extractor...")
        print(f"Device: {self.device}")
        print(f"Model: {self.model_checkpoint if not self.stereo_only else 'Stereo-only mode'}")
        print(f"Batch size: {self.batch_size}")
        print(f"Neural guidance: {self.use_neural_guidance and not self.stereo_only}")

        self.model = None
        self.processor = None
        self.mono_provider = mono_provider
        self.model_loaded = False
        self.max_vram_usage = 0.9
        self.memory_stats = defaultdict(float)

    def load_model(self):
        """ Load depth estimation model (depth.py:60-114).  Offline by construction: a local directory, a model already in
        the HF cache, or a provider handed to the constructor; anything else falls back like depth.py:107-114 """
        if self.model_loaded:
            return
        if self.stereo_only:
            print("Using stereo-only mode (no neural network)")
            self.model_loaded = True
            return
        if self.mono_provider is not None:
            print("Using the supplied monocular depth provider for neural guidance")
            self.model_loaded = True
            return
        print(f"Loading depth model: {self.model_checkpoint}")
        try:
            from transformers import DPTForDepthEstimation, DPTImageProcessor
            print("Loading DPT model for neural depth guidance")
            self.processor = DPTImageProcessor.from_pretrained(self.model_checkpoint, local_files_only=True)
            self.model = DPTForDepthEstimation.from_pretrained(self.model_checkpoint, local_files_only=True)
            dev = getattr(self.backend, "device", self.device)
            self.model = self.model.to(dev)
            self.model.eval()
            self.mono_provider = self._dpt_provider
            self.model_loaded = True
            print("✓ Model loaded successfully")
        except ImportError:
            print("Warning: transformers library not available, falling back to stereo-only mode")
            self.stereo_only = True
            self.model_loaded = True
        except Exception as e:
            print(f"Warning: Failed to load neural model, falling back to stereo-only mode: {e}")
            self.stereo_only = True
            self.model_loaded = True

    def _dpt_provider(self, left_rgb_frames):
        """ depth.py:283-293 + 346-350: DPT forward on the left view; the predicted depth stays on the device """
        import torch
        dev = next(self.model.parameters()).device
        out = []
        with torch.no_grad():
            for rgb in left_rgb_frames:
                inputs = self.processor(images=rgb, return_tensors="pt")
                inputs = {k: v.to(dev) for k, v in inputs.items()}
                out.append(self.model(**inputs).predicted_depth[0].float())
        return out

    def _guidance_provider(self):
        """ the provider when neural guidance is active (depth.py:344-345), else None """
        if self.use_neural_guidance and not self.stereo_only and self.mono_provider is not None:
            return self.mono_provider
        return None

    def get_cache_path(self, video_path: str, frame_start: int, frame_count: int) -> Path:
        """ Generate cache path for depth maps (key format identical to depth.py:119-120) """
        cache_key = f"{video_path}_{frame_start}_{frame_count}_{self.model_checkpoint}_{self.unsqueeze_sbs}"
        cache_hash = hashlib.md5(cache_key.encode()).hexdigest()[:16]
        cache_subdir = self.cache_dir / f"depth_{cache_hash}"
        cache_subdir.mkdir(exist_ok=True)
        return cache_subdir

    def is_cached(self, cache_path: Path, frame_count: int) -> bool:
        """ Check if depth maps are already cached """
        if not cache_path.exists():
            return False
        expected_files = [cache_path / f"depth_{i:06d}.png" for i in range(frame_count)]
        all_exist = all(f.exists() for f in expected_files)
        if all_exist:
            print(f"✓ Found cached depth maps: {cache_path}")
            return True
        return False

    def _frame_count(self, video_path, start_frame, max_frames):
        video_info = get_video_info(video_path)
        if not video_info:
            raise ValueError(f"Could not read video info: {video_path}")
        total_frames = video_info.get('frames', 0) or int(video_info['duration'] * video_info['fps'])
        if max_frames is None:
            n = total_frames - start_frame
        else:
            n = min(max_frames, total_frames - start_frame)
        return video_info, n

    def extract_frames_opencv(self, video_path: str, start_frame: int = 0, max_frames: int = None) -> List[np.ndarray]:
        """ Extract video frames (kept for API compatibility; process_video_sbs streams instead) """
        print(f"Extracting frames from {video_path}...")
        _, max_frames = self._frame_count(video_path, start_frame, max_frames)
        print(f"Extracting {max_frames} frames starting from frame {start_frame}")
        try:
            frames = list(iter_frames(video_path, start_frame, max_frames))
        except ValueError:
            raise
        except Exception as e:
            raise RuntimeError(f"Frame extraction failed: {e}")
        print(f"✓ Extracted {len(frames)} frames")
        return frames

    extract_frames_ffmpeg = extract_frames_opencv

    def split_sbs_frame(self, sbs_frame: np.ndarray, unsqueeze: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """ Split side-by-side frame into left and right images """
        height, width = sbs_frame.shape[:2]
        if width % 2 != 0:
            raise ValueError("SBS frame width must be even")
        return self.backend.split_sbs(np.ascontiguousarray(sbs_frame), unsqueeze)

    def preprocess_frame_pair(self, left_frame: np.ndarray, right_frame: np.ndarray) -> Dict:
        """ Preprocess frame pair for depth estimation (BGR -> RGB views; the provider does its own input processing) """
        if left_frame.shape[2] == 3:
            left_rgb, right_rgb = left_frame[..., ::-1], right_frame[..., ::-1]
        else:
            left_rgb, right_rgb = left_frame, right_frame
        return {'stereo_pair': {'left': left_rgb, 'right': right_rgb}}

    def process_frame_batch(self, frame_pairs: List[Tuple[np.ndarray, np.ndarray]]) -> List[np.ndarray]:
        """ Process batch of BGR frame pairs -> list of HxW float32 disparity maps (>= 0) """
        if not self.model_loaded:
            self.load_model()
        batch_size = len(frame_pairs)
        print(f"Processing batch of {batch_size} frame pairs...")
        if batch_size == 0:
            return []
        try:
            monos = None
            provider = self._guidance_provider()
            if provider is not None:
                try:
                    monos = provider([np.ascontiguousarray(l[..., ::-1]) for l, _ in frame_pairs])     # left views as RGB (depth.py:274)
                except Exception as e:                       # depth.py:367-369
                    print(f"    Warning: Neural guidance failed, using stereo only: {e}")
            depth_maps = self.backend.pairs_to_disparity(frame_pairs, monos) if monos is not None \
                else self.backend.pairs_to_disparity(frame_pairs)
        except Exception as e:
            print(f"Error processing frame batch: {e}")
            raise
        print(f"✓ Processed {len(depth_maps)} depth maps")
        return depth_maps

    def save_depth_map(self, depth_map: np.ndarray, output_path: Path):
        """ Save depth map as 16-bit PNG (per-frame min-max normalisation, depth.py:399-403) """
        write_png16(output_path, self.backend.normalise_u16(depth_map))

    def process_video_sbs(self,
                          video_path: str,
                          start_frame: int = 0,
                          max_frames: int = None,
                          force_reprocess: bool = False) -> Path:
        """ Process entire SBS video to extract depth maps """
        from . import sharding

        print(f"Processing SBS video: {video_path}")
        video_info, frame_count = self._frame_count(video_path, start_frame, max_frames)
        print(f"Video info: {video_info['width']}x{video_info['height']} @ {video_info['fps']:.1f}fps")
        print(f"Processing {frame_count} frames starting from frame {start_frame}")

        cache_path = self.get_cache_path(video_path, start_frame, frame_count)
        if not force_reprocess and self.is_cached(cache_path, frame_count):
            print("✓ Using cached depth maps")
            return cache_path
        if video_info['width'] % 2 != 0:
            raise ValueError("SBS frame width must be even")
        if not self.model_loaded:
            self.load_model()

        rank, world = sharding.rank_world()
        sharding.require_initialized(world)                  # WORLD_SIZE > 1 without a process group would race the cache dir
        processed_count = 0
        batch, batch_idx = [], []
        # frames per device pass: decoupled from batch_size (which the reference only uses to chunk its frame list,
        # depth.py:448-461) -- a pass fills one lock-step SGM launch whatever the caller's chunk size is
        ow = video_info['width'] if self.unsqueeze_sbs else video_info['width'] // 2
        sizer = getattr(self.backend, "compute_batch_size", None)
        pass_frames = sizer(ow, video_info['height'], self.batch_size) if sizer else self.batch_size
        self.last_pass_frames = pass_frames
        provider = self._guidance_provider()
        # PNG compression (zlib) costs ~20 ms per 1080p map on one core, the GPU path 0.5 ms: the maps of a batch go to
        # a bounded pool of writer threads and compress while the next batch is decoded and computed
        writers = self.writer_pool_factory()

        def flush():
            nonlocal processed_count
            if not batch:
                return
            if provider is not None:
                depth = self.backend.sbs_to_disparity(batch, self.unsqueeze_sbs, provider)
            else:
                depth = self.backend.sbs_to_disparity(batch, self.unsqueeze_sbs)
            for j, frame_idx in enumerate(batch_idx):
                writers.submit(cache_path / f"depth_{frame_idx:06d}.png", self.backend.normalise_u16(depth[j]))
                processed_count += 1
            print(f"✓ Queued batch depth maps ({processed_count} on rank {rank})")
            batch.clear()
            batch_idx.clear()

        # frame i -> rank i mod world (round-robin): every rank seeks to and decodes ONLY its own frames, so the decode
        # (the scaling limiter once the kernels are fast) is divided by the world size, not replicated
        decoded = 0
        with writers:
            for k, frame in enumerate(iter_frames(video_path, start_frame, frame_count, stride=world, offset=rank)):
                decoded += 1
                batch.append(frame)
                batch_idx.append(rank + k * world)
                if len(batch) == pass_frames:
                    flush()
            flush()
        self.last_decoded_frames = decoded
        if sharding.total(decoded) == 0:
            raise ValueError("No frames extracted from video")
        sharding.barrier()

        print(f"✓ Depth extraction complete: {cache_path}")
        print(f"  Processed {processed_count} frames")
        print(f"  Output directory: {cache_path}")
        return cache_path


# run_pipeline.py:12,63 and reference __init__.py:6 import this name
IGEVStereoDepthExtractor = HybridStereoDepthExtractor


def main(argv=None):
    """ Command line interface for depth extraction """
    parser = argparse.ArgumentParser(description='Extract depth maps from SBS stereoscopic video')
    parser.add_argument('video', help='Path to SBS video file')
    parser.add_argument('--start-frame', type=int, default=0, help='Starting frame number (default: 0)')
    parser.add_argument('--max-frames', type=int, default=None, help='Maximum number of frames to process (default: all)')
    parser.add_argument('--batch-size', type=int, default=8, help='Batch size for GPU processing (default: 8)')
    parser.add_argument('--model', default="Intel/dpt-large", help='Neural model checkpoint (default: Intel/dpt-large)')
    parser.add_argument('--work-dir', default='temp_depth', help='Working directory for output (default: temp_depth)')
    parser.add_argument('--force', action='store_true', help='Force reprocessing even if cached results exist')
    parser.add_argument('--device', default='cuda', help='Processing device (default: cuda)')
    parser.add_argument('--stereo-only', action='store_true', help='Use stereo matching only (no neural guidance)')
    parser.add_argument('--no-neural', action='store_true', help='Disable neural guidance (same as --stereo-only)')
    parser.add_argument('--no-unsqueeze', action='store_true', help='Skip SBS unsqueezing (keep squeezed aspect ratio)')
    args = parser.parse_args(argv)

    stereo_only = args.stereo_only or args.no_neural
    use_neural_guidance = not stereo_only
    unsqueeze_sbs = not args.no_unsqueeze

    try:
        from . import sharding
        sharding.init_process_group()            # no-op for one process; under torchrun: one rank per GPU (sets the device)
        extractor = HybridStereoDepthExtractor(
            model_checkpoint=args.model, work_dir=args.work_dir, cache_dir=args.work_dir, device=args.device,
            batch_size=args.batch_size, use_neural_guidance=use_neural_guidance, stereo_only=stereo_only,
            unsqueeze_sbs=unsqueeze_sbs)
        output_path = extractor.process_video_sbs(video_path=args.video, start_frame=args.start_frame,
                                                  max_frames=args.max_frames, force_reprocess=args.force)
        print(f"\n✓ Success! Depth maps saved to: {output_path}")
    except Exception as e:
        print(f"Error: {e}")
        return 1
    return 0


if __name__ == "__main__":
    exit(main())

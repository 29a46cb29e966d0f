"""Depth upscaling to the 4K frame -- MI355X-native host side.

Mirror of reference src/video_3d_pipeline/upscale.py (class / method names, argument meaning, skip-if-
exists and error behaviour, CLI flags).  The reference shells out to ffmpeg's `scale` filter + H.264
(upscale.py:47-63) and never looks at the 4K pixels; BASELINE.json re-specifies the step as
guided-filter joint upsampling with the 4K frame as guide, which is what
v3d_guided_upscale computes here.  `use_nvenc` is accepted for signature compatibility (NVENC is an
NVIDIA encoder); H.264 encoding is used only if an ffmpeg binary exists at run time, otherwise the
4K depth frames are written as 16-bit PNGs and `output_path` becomes a small JSON manifest.
"""
import argparse
import glob
import json
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np

from .utils import PngWriterPool, get_video_info, iter_frames, prefetch_map, read_png16

GUIDED_RADIUS = 8       # at 4K; the reference specifies nothing
GUIDED_EPS = 1e-3       # on [0,1]-scaled guide


class HipUpscaleBackend:
    def __init__(self, device: str = "cuda"):
        import torch
        from . import _native
        if not torch.cuda.is_available():
            raise RuntimeError("CUDA not available but requested")
        _native.lib()
        self.torch, self.native, self.device = torch, _native, _native.resolve_device(device)   # bare "cuda" = the current device

    def to_luma(self, frame_bgr):
        d = frame_bgr if self.torch.is_tensor(frame_bgr) else self.native.to_device(frame_bgr, self.device)
        return d if d.dim() == 2 else self.native.bgr_to_gray(d.contiguous())

    def upscale(self, depth_lo, guide, r, eps):
        nat = self.native
        d = depth_lo if self.torch.is_tensor(depth_lo) else nat.to_device(np.asarray(depth_lo, np.float32), self.device)
        g = self.to_luma(guide)
        return nat.guided_upscale(d.contiguous(), g.contiguous(), r, eps)

    def upscale_u16(self, depth_lo, guide, r, eps) -> np.ndarray:
        """guided upscale, rounded and clamped to the 16-bit range of the PNG sink"""
        q = self.upscale(depth_lo, guide, r, eps)
        # v3d_round_to_u16: the uint16 bit pattern leaves the device as 2 bytes per pixel
        return self.native.round_to_u16(q.contiguous()).cpu().numpy().view(np.uint16)

    def flat_guide(self, h, w):
        return self.torch.full((h, w), 128, dtype=self.torch.uint8, device=self.device)


class SimpleDepthUpscaler:
    """ Depth upscaling to the 4K frame (guided filter on the GPU) """

    writer_pool_factory = PngWriterPool       # sink of the 4K 16-bit maps (see HybridStereoDepthExtractor.writer_pool_factory)

    def __init__(self, use_nvenc: bool = True, radius: int = GUIDED_RADIUS, eps: float = GUIDED_EPS,
                 device: str = "cuda", backend=None):
        self.use_nvenc = use_nvenc
        self.radius, self.eps = radius, eps
        self.backend = backend if backend is not None else HipUpscaleBackend(device)
        print(f"Initializing Simple Depth Upscaler...")
        print(f"NVENC encoding: {self.use_nvenc}")

    def upscale_frame(self, depth_lo: np.ndarray, guide_4k: np.ndarray) -> np.ndarray:
        """NumPy surface: HxW float32 depth + 2Hx2W(x3) uint8 guide (BGR or luma) -> 2Hx2W float32"""
        return self.backend.upscale(depth_lo, guide_4k, self.radius, self.eps).cpu().numpy()

    def upscale_depth_maps_ffmpeg(self, depth_dir: str, target_width: int, target_height: int,
                                  output_path: str, fps: float = 23.976, video_4k_path: str = None,
                                  guide_start_frame: int = 0):
        """ Upscale the depth_%06d.png sequence to target_width x target_height (name kept from the reference).
        guide_start_frame: index of the 4K frame that belongs to depth_000000 -- the alignment offset that
        run_pipeline.py:45-50 computes and then drops; round(offset_seconds * fps), clamped at 0. """
        from . import sharding

        print(f"Processing depth upscaling...")
        print(f"Input: {depth_dir}")
        print(f"Output: {output_path}")
        print(f"Target: {target_width}x{target_height} @ {fps}fps")

        depth_files = sorted(glob.glob(os.path.join(depth_dir, "depth_*.png")))
        if not depth_files:
            raise ValueError(f"No depth maps found in {depth_dir}")
        print(f"Found {len(depth_files)} depth maps")

        rank, world = sharding.rank_world()
        sharding.require_initialized(world)           # never a silent single-rank fallback under torchrun
        frames_dir = Path(str(Path(output_path).with_suffix("")) + "_frames")
        frames_dir.mkdir(parents=True, exist_ok=True)
        n = len(depth_files)
        g0 = max(int(guide_start_frame), 0)
        # The DECODER decides how many guide frames exist; the container's frame count is only a hint (cv2's
        # CAP_PROP_FRAME_COUNT and duration * fps are both unreliable).  A clip that ends early degrades the remaining
        # depth frames to a flat guide (plain smoothing upsample) with a warning; one that runs longer is simply used.
        n_hint = None
        if video_4k_path:
            info = get_video_info(video_4k_path)
            if info:
                total = info.get('frames', 0) or int(info['duration'] * info['fps'])
                n_hint = max(0, min(n, total - g0))
            else:
                print(f"Warning: could not read video info of {video_4k_path}; decoding it anyway")
        # the guide video is decoded ONCE, by rank 0 (one rank: directly into the filter; several: into the round exchange)
        guides = iter_frames(video_4k_path, g0, n) if (video_4k_path and rank == 0) else None
        decoded = [0, False]                                  # rank 0: guide frames delivered so far, decoder ended

        def next_luma():
            """rank 0: the next guide frame as a device luma tensor (the BGR frame crosses PCIe once, the luma stays on
            the device), or None once the decoder has ended"""
            if decoded[1]:
                return None
            f = next(guides, None)
            if f is None:
                decoded[1] = True
                if n_hint is not None and decoded[0] < n_hint:
                    print(f"Warning: 4K guide video ended after {decoded[0]} frames (container promised {n_hint}); "
                          f"depth frames from {decoded[0]} on are upsampled with a flat guide")
                return None
            decoded[0] += 1
            return self.backend.to_luma(f)

        exchange = None
        if world > 1 and video_4k_path:
            exchange = sharding.GuideRoundExchange((target_height, target_width), self.backend.device)

        def post_round(base):
            """rank 0 decodes the guide frames of depth frames base .. base+world-1 and posts the round (one collective).
            Every rank posts the same ceil(n / world) rounds -- the count depends on the depth files only, which all
            ranks see; what the decoder really delivered travels in the round's validity bitmap.  A decoder failure on
            rank 0 is posted as an aborted round first, so the other ranks raise instead of waiting in the collective."""
            if base >= n:
                return False
            if rank != 0:
                exchange.post()
                return True
            try:
                frames = [next_luma() if base + r < n else None for r in range(world)]
            except Exception:
                exchange.post(abort=True)
                raise
            exchange.post(frames)
            return True

        # this rank's depth maps, decoded a few files ahead on reader threads (PNG inflate is the slowest host step)
        my_depth = prefetch_map(read_png16, [depth_files[i] for i in range(rank, n, world)])
        flat = 0
        with self.writer_pool_factory() as writers:                  # 4K 16-bit PNGs: ~80 ms of zlib each, compressed off the main thread
            posted = post_round(0) if exchange is not None else False
            for base in range(0, n, world):
                # one round ahead: the collective of round base+world runs on the exchange's side stream while this
                # round's filter runs on the main stream
                posted_next = post_round(base + world) if exchange is not None else False
                i = base + rank
                guide = None
                if exchange is not None:
                    if posted:
                        guide = exchange.take()           # raises GuideExchangeAborted on every rank if rank 0's decoder failed
                elif guides is not None and i < n:        # one rank
                    guide = next_luma()
                posted = posted_next
                if i >= n:
                    continue
                if guide is None:
                    flat += 1
                    guide = self.backend.flat_guide(target_height, target_width)    # beyond the 4K clip: flat guide == plain smoothing upsample
                d16 = next(my_depth).astype(np.float32)
                writers.submit(frames_dir / f"depth4k_{i:06d}.png", self.backend.upscale_u16(d16, guide, self.radius, self.eps))
        self.last_flat_guides = flat
        if rank == 0 and guides is not None and n_hint is not None and decoded[0] > n_hint:
            print(f"Note: the 4K video delivered {decoded[0]} guide frames, the container promised {n_hint}")
        sharding.barrier()

        if rank == 0:
            ffmpeg = shutil.which("ffmpeg")
            if ffmpeg and str(output_path).endswith(".mp4"):
                cmd = [ffmpeg, "-y", "-v", "error", "-r", str(fps), "-f", "image2", "-i", str(frames_dir / "depth4k_%06d.png"),
                       "-vcodec", "libx264", "-pix_fmt", "yuv420p", "-crf", "18", "-preset", "medium", "-r", str(fps), str(output_path)]
                res = subprocess.run(cmd, capture_output=True)
                if res.returncode != 0:
                    print("FFmpeg error:")
                    print(res.stderr.decode())
                    raise RuntimeError(f"FFmpeg processing failed: rc={res.returncode}")
            else:
                Path(output_path).write_text(json.dumps({
                    "format": "png16-sequence", "frames_dir": str(frames_dir), "pattern": "depth4k_%06d.png",
                    "count": n, "width": target_width, "height": target_height, "fps": fps,
                    "guided_radius": self.radius, "guided_eps": self.eps,
                    "note": "no ffmpeg binary on this host: 4K depth frames kept as 16-bit PNGs"}, indent=1))
        sharding.barrier()
        print(f"✓ Depth video saved: {output_path}")
        return output_path

    def process_depth_upscaling(self, depth_dir: str, video_4k_path: str, output_path: str = None,
                                force_reprocess: bool = False, guide_start_frame: int = 0) -> str:
        """ Main pipeline for depth upscaling """
        print(f"Processing depth upscaling...")
        print(f"Depth maps: {depth_dir}")
        print(f"4K video: {video_4k_path}")

        video_info = get_video_info(video_4k_path)
        if not video_info:
            raise ValueError(f"Could not read video info: {video_4k_path}")
        target_width, target_height, fps = video_info['width'], video_info['height'], video_info['fps']
        print(f"Target resolution: {target_width}x{target_height} @ {fps}fps")

        if output_path is None:
            depth_dir_name = Path(depth_dir).name
            output_path = f"depth_4k_{depth_dir_name}.mp4"
        output_path = Path(output_path)

        if output_path.exists() and not force_reprocess:
            print(f"✓ Using existing depth video: {output_path}")
            return str(output_path)

        result = self.upscale_depth_maps_ffmpeg(depth_dir=depth_dir, target_width=target_width, target_height=target_height,
                                                output_path=str(output_path), fps=fps, video_4k_path=video_4k_path,
                                                guide_start_frame=guide_start_frame)
        print(f"✓ Depth upscaling complete!")
        print(f"  Input: {depth_dir}")
        print(f"  Output: {result}")
        print(f"  Resolution: {target_width}x{target_height}")
        return result


def main(argv=None):
    """ Command line interface for depth upscaling """
    parser = argparse.ArgumentParser(description='Depth upscaling to the 4K frame (guided filter)')
    parser.add_argument('depth_dir', help='Directory containing depth maps')
    parser.add_argument('video_4k', help='Path to 4K 2D video (dimensions and guide frames)')
    parser.add_argument('--output', help='Output path for 4K depth video')
    parser.add_argument('--no-nvenc', action='store_true', help='Disable NVENC, use CPU encoding')
    parser.add_argument('--force', action='store_true', help='Force reprocessing even if output exists')
    parser.add_argument('--guide-start-frame', type=int, default=0,
                        help='4K frame that matches depth_000000 (alignment offset in frames; default 0)')
    args = parser.parse_args(argv)
    try:
        from . import sharding
        sharding.init_process_group()            # no-op for one process; under torchrun: one rank per GPU (sets the device)
        upscaler = SimpleDepthUpscaler(use_nvenc=not args.no_nvenc)
        output_path = upscaler.process_depth_upscaling(depth_dir=args.depth_dir, video_4k_path=args.video_4k,
                                                       output_path=args.output, force_reprocess=args.force,
                                                       guide_start_frame=args.guide_start_frame)
        print(f"\n✓ Success! 4K depth video: {output_path}")
        print(f"Ready for VisionDepth3D processing!")
    except Exception as e:
        print(f"Error: {e}")
        return 1
    return 0


if __name__ == "__main__":
    exit(main())

"""Shared helpers for the hot path (mirror of the two functions of reference utils.py the path uses:
get_video_info, utils.py:17-38, and create_work_directory, utils.py:292-296), plus the frame source /
PNG sink either side of it.

Unlike the reference module this one imports nothing heavy at module top (the reference drags in
cv2/librosa/soundfile/matplotlib, utils.py:3-14): decoders are probed lazily, in this order:
ffprobe/ffmpeg binaries, cv2, and -- always available -- the self-describing frame containers used by
the tests and the bench (`*.npy` / `*.npz` stacks of HxWx3 uint8 BGR frames, or a directory of
`frame_%06d.png`).
"""
import json
import os
import shutil
import subprocess
from fractions import Fraction
from pathlib import Path
from typing import Dict, Iterator, Optional

import numpy as np


def create_work_directory(base_path: str = "temp_pipeline") -> Path:
    """ Create and return working directory path """
    work_dir = Path(base_path)
    work_dir.mkdir(exist_ok=True)
    return work_dir


def _parse_rate(s) -> float:
    # the reference eval()s ffprobe's "24000/1001" (utils.py:32); parse it instead
    try:
        return float(Fraction(str(s)))
    except (ValueError, ZeroDivisionError):
        return float(s)


def _frame_dir_files(path: Path):
    return sorted(path.glob("frame_*.png"))


def _container_kind(video_path: str) -> Optional[str]:
    p = Path(video_path)
    if p.is_dir() and _frame_dir_files(p):
        return "pngdir"
    if p.suffix == ".npy" and p.exists():
        return "npy"
    if p.suffix == ".npz" and p.exists():
        return "npz"
    return None


def _open_stack(video_path: str):
    kind = _container_kind(video_path)
    if kind == "npy":
        return np.load(video_path, mmap_mode="r"), 23.976
    if kind == "npz":
        z = np.load(video_path)
        return z["frames"], float(z["fps"]) if "fps" in z.files else 23.976
    raise ValueError(video_path)


def get_video_info(video_path: str) -> Optional[Dict]:
    """ Get basic video information (width, height, fps, duration, frames); None if unreadable """
    try:
        kind = _container_kind(video_path)
        if kind in ("npy", "npz"):
            frames, fps = _open_stack(video_path)
            n, h, w = frames.shape[0], frames.shape[1], frames.shape[2]
            return {"width": int(w), "height": int(h), "fps": fps, "duration": n / fps, "frames": int(n)}
        if kind == "pngdir":
            from PIL import Image
            files = _frame_dir_files(Path(video_path))
            fps = 23.976
            meta = Path(video_path) / "info.json"
            if meta.exists():
                fps = float(json.loads(meta.read_text()).get("fps", fps))
            with Image.open(files[0]) as im:
                w, h = im.size
            return {"width": w, "height": h, "fps": fps, "duration": len(files) / fps, "frames": len(files)}
        if not os.path.exists(video_path):
            raise FileNotFoundError(video_path)
        ffprobe = shutil.which("ffprobe")
        if ffprobe:
            out = subprocess.run([ffprobe, "-v", "error", "-print_format", "json", "-show_streams", video_path],
                                 capture_output=True, check=True, text=True).stdout
            streams = json.loads(out).get("streams", [])
            vs = next((s for s in streams if s.get("codec_type") == "video"), None)
            if not vs:
                return None
            return {"width": int(vs["width"]), "height": int(vs["height"]), "fps": _parse_rate(vs["r_frame_rate"]),
                    "duration": float(vs["duration"]), "frames": int(vs.get("nb_frames", 0))}
        try:
            import cv2  # noqa: WPS433 (optional)
        except ImportError:
            raise RuntimeError("no decoder available (ffprobe and cv2 are both missing); "
                               "use a .npy/.npz frame stack or a frame_%06d.png directory")
        cap = cv2.VideoCapture(video_path)
        if not cap.isOpened():
            return None
        fps = cap.get(cv2.CAP_PROP_FPS) or 23.976
        n = int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
        info = {"width": int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), "height": int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT)),
                "fps": fps, "duration": n / fps, "frames": n}
        cap.release()
        return info
    except Exception as e:
        print(f"Error getting video info: {e}")
        return None


def iter_frames(video_path: str, start_frame: int = 0, max_frames: Optional[int] = None,
                stride: int = 1, offset: int = 0) -> Iterator[np.ndarray]:
    """Stream HxWx3 uint8 BGR frames (bounded memory; the reference loads the whole clip into a list,
    depth.py:160-176).  `max_frames` bounds the RANGE [start_frame, start_frame + max_frames); of that range only
    frames start_frame + offset + k*stride are produced (rank r of a world of w decodes stride = w, offset = r): the
    others are never converted, and in the indexed containers never read."""
    if stride < 1 or not 0 <= offset < stride:
        raise ValueError(f"bad stride/offset {stride}/{offset}")
    kind = _container_kind(video_path)
    if kind in ("npy", "npz"):
        frames, _ = _open_stack(video_path)
        end = frames.shape[0] if max_frames is None else min(frames.shape[0], start_frame + max_frames)
        for i in range(start_frame + offset, end, stride):
            yield np.ascontiguousarray(frames[i])
        return
    if kind == "pngdir":
        from PIL import Image
        files = _frame_dir_files(Path(video_path))
        end = len(files) if max_frames is None else min(len(files), start_frame + max_frames)
        for f in files[start_frame + offset:end:stride]:
            with Image.open(f) as im:
                rgb = np.asarray(im.convert("RGB"))
            yield np.ascontiguousarray(rgb[..., ::-1])
        return
    try:
        import cv2  # noqa: WPS433 (optional)
    except ImportError:
        cv2 = None
    if cv2 is not None:
        cap = cv2.VideoCapture(video_path)
        if not cap.isOpened():
            raise ValueError(f"Could not open video file: {video_path}")
        cap.set(cv2.CAP_PROP_POS_FRAMES, start_frame)
        n = 0
        while max_frames is None or n < max_frames:
            if n % stride == offset:
                ret, frame = cap.read()
                if not ret:
                    break
                yield frame
            elif not cap.grab():                  # another rank's frame: advance the decoder, skip retrieval + conversion
                break
            n += 1
        cap.release()
        return
    ffmpeg = shutil.which("ffmpeg")
    info = get_video_info(video_path)
    if not ffmpeg or not info:
        raise RuntimeError(f"Frame extraction failed: no decoder for {video_path}")
    cmd = [ffmpeg, "-v", "error", "-ss", str(start_frame / info["fps"]), "-i", video_path]
    if max_frames is not None:
        cmd += ["-frames:v", str(len(range(offset, max_frames, stride)))]
    if stride > 1:                                # only this rank's frames leave the decoder
        cmd += ["-vf", f"select='not(mod(n-{offset},{stride}))'", "-vsync", "0"]
    cmd += ["-f", "rawvideo", "-pix_fmt", "bgr24", "pipe:"]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE)
    size = info["width"] * info["height"] * 3
    try:
        while True:
            buf = proc.stdout.read(size)
            if not buf or len(buf) != size:
                break
            yield np.frombuffer(buf, np.uint8).reshape(info["height"], info["width"], 3)
    finally:
        proc.stdout.close()
        proc.wait()


# writer / reader threads of the PNG pools; None = from the host cores (at most 16 writers / 8 readers).  A module
# attribute, not an environment variable: the package reads none (tools set it directly)
IO_THREADS = None


def encode_png16(img_u16: np.ndarray, level: int = 1) -> bytes:
    """16-bit single-channel PNG as bytes (what cv2.imwrite produces for a uint16 array, depth.py:406).
    Written with zlib directly: filter type "sub" per row, deflate level 1.  zlib.compress releases the GIL, so a pool
    of writer threads scales with the host cores (Pillow's encoder holds it: 8 threads were no faster than one)."""
    import struct
    import zlib
    a = np.ascontiguousarray(img_u16, dtype=np.uint16)
    if a.ndim != 2:
        raise ValueError(f"expected a 2-D uint16 image, got shape {a.shape}")
    h, w = a.shape
    be = a.astype(">u2").view(np.uint8).reshape(h, 2 * w)            # PNG samples are big-endian
    raw = np.empty((h, 1 + 2 * w), np.uint8)
    raw[:, 0] = 1                                                    # filter "sub" on every row: byte minus the byte one
    raw[:, 1:3] = be[:, :2]                                          # pixel (2 bytes) to the left, mod 256 -- one array
    if w > 1:                                                        # op to encode, ONE cumulative sum to decode
        np.subtract(be[:, 2:], be[:, :-2], out=raw[:, 3:])

    def chunk(tag: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + chunk(b"IEND", b""))


def write_png16(path, img_u16: np.ndarray) -> None:
    with open(str(path), "wb") as f:
        f.write(encode_png16(img_u16))


class PngWriterPool:
    """Bounded pool of PNG writer threads: submit() returns at once unless `max_pending` images are already queued
    (back-pressure keeps host memory flat), close() waits for all of them and re-raises the first failure.  The
    encoder spends its time inside zlib with the GIL released, so frames of one batch compress in parallel while the
    next batch is decoded and computed (the step that bounds the end-to-end CLI once the kernels are fast)."""

    def __init__(self, workers: int = None, max_pending: int = None):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        if workers is None:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 4
            workers = IO_THREADS if IO_THREADS else max(1, min(16, ncpu))
        self.workers = max(1, workers)
        self._ex = ThreadPoolExecutor(self.workers, thread_name_prefix="v3d-png")
        self._slots = threading.Semaphore(max_pending if max_pending else 4 * self.workers)
        self._futures = []

    def _job(self, path, img):
        try:
            write_png16(path, img)
        finally:
            self._slots.release()

    def submit(self, path, img_u16: np.ndarray) -> None:
        self._slots.acquire()
        self._futures.append(self._ex.submit(self._job, path, img_u16))

    def close(self) -> None:
        first = None
        for f in self._futures:
            try:
                f.result()
            except Exception as e:              # keep draining so no writer is left running, then report
                first = first or e
        self._futures = []
        self._ex.shutdown(wait=True)
        if first is not None:
            raise first

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        else:                                   # an error is already propagating: still join the writers
            try:
                self.close()
            except Exception:
                pass
        return False


def _decode_png16_fast(data: bytes):
    """16-bit gray, non-interlaced PNG whose rows use only the filters none / sub / up (this package's own writer
    uses "sub") decoded with zlib + NumPy, both of which release the GIL -- reader threads scale, Pillow's decoder holds
    it.  Returns None for anything else (palette, 8-bit, interlaced, average / Paeth rows): the caller falls back."""
    import struct
    import zlib
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        return None
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    if hdr is None or hdr[2:] != (16, 0, 0, 0, 0) or not idat:
        return None
    w, h = hdr[0], hdr[1]
    try:
        raw = np.frombuffer(zlib.decompress(b"".join(idat), 15, h * (1 + 2 * w)), np.uint8)
    except zlib.error:
        return None
    if raw.size != h * (1 + 2 * w):
        return None
    raw = raw.reshape(h, 1 + 2 * w)
    ft, rows = raw[:, 0], raw[:, 1:]
    if (ft > 2).any():
        return None
    if (ft == 1).all():                                           # this package's writer: one GIL-free call
        out = np.cumsum(rows.reshape(h, w, 2), axis=1, dtype=np.uint8).reshape(h, 2 * w)
    else:
        out = np.empty_like(rows)
        prev = np.zeros(2 * w, np.uint8)
        for r in range(h):
            if ft[r] == 2:
                np.add(rows[r], prev, out=out[r])
            elif ft[r] == 0:
                out[r] = rows[r]
            else:                                                 # sub: running sum per byte lane (2 bytes per pixel)
                out[r] = np.cumsum(rows[r].reshape(w, 2), axis=0, dtype=np.uint8).reshape(-1)
            prev = out[r]
    return np.ascontiguousarray(out).view(">u2").astype(np.uint16)


def read_png16(path) -> np.ndarray:
    with open(str(path), "rb") as f:
        data = f.read()
    a = _decode_png16_fast(data)
    if a is not None:
        return a
    import io
    from PIL import Image
    with Image.open(io.BytesIO(data)) as im:
        a = np.asarray(im)
    if a.ndim == 3:
        a = a[..., 0]
    return a.astype(np.uint16) if a.dtype != np.uint16 else a


def prefetch_map(fn, items, workers: int = None, lookahead: int = None):
    """yield fn(item) for every item, in order, computing up to `lookahead` of them ahead on a thread pool"""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    if workers is None:
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 4
        workers = IO_THREADS if IO_THREADS else max(1, min(8, ncpu))
    lookahead = lookahead or 2 * workers
    with ThreadPoolExecutor(workers, thread_name_prefix="v3d-read") as ex:
        q, it = deque(), iter(items)
        for x in it:
            q.append(ex.submit(fn, x))
            if len(q) >= lookahead:
                break
        while q:
            r = q.popleft().result()
            nxt = next(it, None)
            if nxt is not None:
                q.append(ex.submit(fn, nxt))
            yield r

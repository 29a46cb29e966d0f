"""video_3d_pipeline -- MI355X-native drop-in for the per-frame hot path of
jabberjabberjabber/video-3d-pipeline (SBS frame -> SGBM disparity -> guided-filter 4K depth).

Exports both extractor names: the reference's package imports `IGEVStereoDepthExtractor`
(reference __init__.py:6, run_pipeline.py:12) while its depth.py defines `HybridStereoDepthExtractor`.
Importing the package pulls in no GPU code; the HIP library is loaded when an extractor/upscaler is built.
"""
__version__ = "0.1.0"

from .align import VideoAligner
from .depth import HybridStereoDepthExtractor, IGEVStereoDepthExtractor
from .upscale import SimpleDepthUpscaler
from .utils import get_video_info, create_work_directory

__all__ = [
    "VideoAligner",
    "HybridStereoDepthExtractor",
    "IGEVStereoDepthExtractor",
    "SimpleDepthUpscaler",
    "get_video_info",
    "create_work_directory",
]

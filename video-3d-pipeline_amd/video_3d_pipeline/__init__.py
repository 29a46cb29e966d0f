"""video_3d_pipeline -- MI355X-native drop-in for the per-frame hot path of
jabberjabberjabber/video-3d-pipeline (SBS frame -> SGBM disparity -> guided-filter 4K depth).

Exports both extractor names: the reference's package imports `IGEVStereoDepthExtractor`
(reference __init__.py:6, run_pipeline.py:12) while its depth.py defines `HybridStereoDepthExtractor`.
Names are resolved lazily (PEP 562): importing the package pulls in no GPU code and
`python -m video_3d_pipeline.depth` does not import its own module twice; the HIP library is loaded when an
extractor / upscaler is built.
"""
__version__ = "0.1.0"

_EXPORTS = {
    "VideoAligner": ("align", "VideoAligner"),
    "HybridStereoDepthExtractor": ("depth", "HybridStereoDepthExtractor"),
    "IGEVStereoDepthExtractor": ("depth", "IGEVStereoDepthExtractor"),
    "SimpleDepthUpscaler": ("upscale", "SimpleDepthUpscaler"),
    "get_video_info": ("utils", "get_video_info"),
    "create_work_directory": ("utils", "create_work_directory"),
}
__all__ = list(_EXPORTS)


def __getattr__(name):
    try:
        mod, attr = _EXPORTS[name]
    except KeyError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
    import importlib
    value = getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    globals()[name] = value
    return value


def __dir__():
    return sorted(list(globals()) + __all__)

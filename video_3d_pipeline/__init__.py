"""Path shim so that `python -m video_3d_pipeline.depth` / `.upscale` work from the repository root.
The real package lives in video-3d-pipeline_amd/video_3d_pipeline (a directory name with a hyphen
cannot be imported); this file only redirects the package search path there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "video-3d-pipeline_amd", "video_3d_pipeline")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))

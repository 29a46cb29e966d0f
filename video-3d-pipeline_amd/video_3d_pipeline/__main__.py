"""Main entry point: `python -m video_3d_pipeline` == the depth CLI (reference __main__.py:3-6)."""
from .depth import main

if __name__ == "__main__":
    exit(main())

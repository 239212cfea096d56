"""Alias module: `import ImageFusion` (Stitcher.py:11) -> the engine's ImageFusion mirror."""
from imagestitch_amd.fusion import ImageFusion  # noqa: F401

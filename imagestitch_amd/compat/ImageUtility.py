"""Alias module: `import ImageUtility as Utility` (Stitcher.py:10, ImageFusion.py:4) -> the engine's Method mirror."""
from imagestitch_amd.utility import Method, roi_rect  # noqa: F401

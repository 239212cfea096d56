"""Alias module: with this directory on sys.path the reference's own `Main.py` runs unchanged --
`from Stitcher import Stitcher` (Main.py:1) resolves to the MI355X engine's mirror of Stitcher.py.

    PYTHONPATH=<repo>:<repo>/imagestitch_amd/compat python Main.py
"""
from imagestitch_amd.stitcher import Stitcher, ImageFeature  # noqa: F401
from imagestitch_amd.utility import Method  # noqa: F401  (Stitcher.py:14 derives Stitcher from Utility.Method)

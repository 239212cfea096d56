"""imagestitch_amd -- MI355X-native engine for the VFSMS pairwise image-alignment hot path.

Drop-in for the reference's call surface (Keep-Passion/ImageStitch):

    from imagestitch_amd import Stitcher          # instead of: from Stitcher import Stitcher

Host code is Python (as in the reference); all arithmetic runs in hand-written HIP kernels for gfx950
behind the C ABI of include/vfsms.h (imagestitch_amd/lib/libvfsms.so, loaded with ctypes).
"""
from ._lib import Engine, VfsmsError, default_engine, load_library, LIB_PATH  # noqa: F401
from .utility import Method, roi_rect  # noqa: F401
from .fusion import ImageFusion  # noqa: F401
from .stitcher import Stitcher, ImageFeature, NpyBandWriter, PngBandWriter, TiffBandWriter, JpegBandWriter, band_writer_for  # noqa: F401

__all__ = ["Stitcher", "ImageFusion", "Method", "ImageFeature", "Engine", "VfsmsError", "default_engine",
           "load_library", "roi_rect", "NpyBandWriter", "PngBandWriter", "TiffBandWriter", "JpegBandWriter", "band_writer_for"]

"""Decode and ingest helpers of the sequence drivers: one decode per file for the registration plane and the mosaic tile, the decoder thread
pool, header-only size probes, the folder listing.  The way INTO the path; `stitcher.py` keeps the mirror of
`Stitcher.Stitcher` and takes these names from here.

Reference: cv2.imdecode(np.fromfile(...)) at /root/reference/Stitcher.py:68-69, 174-179, 382-403; glob at Stitcher.py:133-139.
"""
import os

import numpy as np

def _imread(path, color):
    """cv2.imdecode(np.fromfile(path), IMREAD_COLOR | IMREAD_GRAYSCALE) stand-in (Stitcher.py:68-69,382-384).
    Grayscale asks libjpeg for the luma plane directly like OpenCV does; colour is BGR."""
    from PIL import Image
    im = Image.open(path)
    if not color:
        im.draft("L", im.size)
        return np.asarray(im.convert("L"))
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def _imread_gray_pointer(path):
    """_imread(path, False) without the copy that holds the GIL: (owner, address, (rows, cols)) of the decoded luma plane.  Pillow decodes
    with the GIL released; its Arrow export (Pillow >= 11.2 with pyarrow) hands out the pixel block itself, so a pool of decoder threads
    scales until the cores run out instead of serialising on np.asarray's 4 MB copy (measured on the 256-thread host of the MI355X box:
    1.9 k tiles/s against 1.1 k).  Falls back to the numpy array."""
    from PIL import Image
    im = Image.open(path)
    im.draft("L", im.size)
    im.decodermaxblock = max(im.decodermaxblock, 1 << 24)    # the whole file in one read + decode call: fewer trips through the interpreter lock per tile
    im.load()
    if im.mode == "L" and hasattr(im, "__arrow_c_array__"):
        try:
            import pyarrow as pa
            arr = pa.array(im)
            buf = arr.buffers()[1]
            if buf is not None and buf.size == im.size[0] * im.size[1]:
                return (arr, im), buf.address, (im.size[1], im.size[0])
        except Exception:                                    # no pyarrow / not exportable: the copying path below
            pass
    a = np.ascontiguousarray(np.asarray(im.convert("L")))
    return a, a.ctypes.data, a.shape


class _PillowBlocks:
    """Pillow keeps 3-band images in 4-byte pixels: a 2048 x 2048 tile is 16.7 MB, more than one 16 MB storage block, and only an image in ONE
    block can be handed to the engine without a copy (_decode_once).  While the decoder pool runs, colour images are allocated as single
    blocks (process-wide switch, restored on exit).  (Measured and rejected: a cache of freed 32 MB blocks for the pool -- Image.core.
    set_blocks_max -- made the per-tile decode 20 % slower on the 256-thread host, gray and colour alike.)"""

    def __init__(self, color):
        self.color, self.saved = color, None

    def __enter__(self):
        if self.color:
            try:
                from PIL import Image
                self.saved = Image.core.get_use_block_allocator()
                Image.core.set_use_block_allocator(1)
            except Exception:                                # an older Pillow: the copying hand-over still works
                self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            try:
                from PIL import Image
                Image.core.set_use_block_allocator(self.saved)
            except Exception:
                pass
        return False


_POOLS = {}


def _decoder_pool(nthreads):
    """the decoder threads are kept between calls (a dataset after the other: starting sixteen threads costs a millisecond or two each time)"""
    from concurrent.futures import ThreadPoolExecutor
    pool = _POOLS.get(nthreads)
    if pool is None:
        pool = _POOLS[nthreads] = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="vfsms-decode")
    return pool


def _decode_once(path, want_color):
    """ONE decode of a file for both uses the reference makes of it (cv2.imdecode(..., 0) at Stitcher.py:68-69 for registration and, with
    isColorMode, cv2.imdecode(..., IMREAD_COLOR) at Stitcher.py:382-403 for the mosaic) -> (owner, (rows, cols), parts) with
    parts = ("src", address, stride_bytes, fmt): what vfsms_tile_fill_pair takes -- fmt 0 a gray plane, 1 / 2 the JPEG's own Y Cb Cr planes
          interleaved (libjpeg out_color_space = JCS_YCbCr: the colour conversion happens on the GPU, the Y plane IS the grayscale decode),
          or ("arrays", gray (h, w), bgr (h, w, 3) | None): other formats / colour spaces, both planes from the one loaded image."""
    from PIL import Image
    if not want_color:
        keep, addr, shape = _imread_gray_pointer(path)
        return keep, shape, ("src", addr, shape[1], 0)
    im = Image.open(path)
    im.decodermaxblock = max(im.decodermaxblock, 1 << 24)
    if im.format == "JPEG" and im.mode == "RGB":
        try:
            im.draft("YCbCr", im.size)
            im.load()
        except Exception:                                    # e.g. an Adobe RGB JPEG (no YCbCr planes): decode as it is
            im = Image.open(path)
    else:
        im.draft("L", im.size)
    im.load()
    shape = (im.size[1], im.size[0])
    if im.mode in ("L", "YCbCr"):
        spx = 1 if im.mode == "L" else 4                     # Pillow stores 3-band pixels in 4 bytes
        if hasattr(im, "__arrow_c_array__"):
            try:
                import pyarrow as pa
                arr = pa.array(im)
                buf = (arr.buffers()[1] if spx == 1 else arr.values.buffers()[1])
                if buf is not None and buf.size == shape[0] * shape[1] * spx:
                    return (arr, im), shape, ("src", buf.address, shape[1] * spx, 0 if spx == 1 else 2)
            except Exception:                                # no pyarrow / image in several blocks: the copying path below
                pass
        a = np.ascontiguousarray(np.asarray(im))
        return a, shape, ("src", a.ctypes.data, a.strides[0], 0 if spx == 1 else 1)
    gray = np.ascontiguousarray(np.asarray(im.convert("L")))
    bgr = np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    return None, shape, ("arrays", gray, bgr)


def _fill_from_jpeg(eng, path, gray_handle, color_handle):
    """JPEG files are decoded by the library itself when it can (vfsms_tile_fill_jpeg: the system's libjpeg-turbo writes into pinned staging
    memory that is reused from tile to tile, outside the interpreter lock; one decode, colour conversion on the GPU) -> True, both tiles
    filled.  False: not a JPEG, an engine without the entry point, a file this decoder does not take (CMYK, RGB-coded, damaged, ...), or
    VFSMS_NATIVE_JPEG=0 -- the tiles are still reserved and `_decode_once` (Pillow) decodes the file."""
    fill = getattr(eng, "tile_fill_jpeg", None)
    if fill is None or os.environ.get("VFSMS_NATIVE_JPEG", "1") == "0":
        return False
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"\xff\xd8":
        return False
    return bool(fill(gray_handle, color_handle, data))


def _ycc_to_bgr(ycc):
    """libjpeg's YCbCr -> RGB (jdcolor.c: 16-bit fixed-point tables), stored B G R: what cv2.imdecode(IMREAD_COLOR) yields from the planes
    `_decode_once` hands to the GPU.  Host-side twin of csrc/ingest_kernels.hip for the tiles that are not resident (lone tiles)."""
    y = ycc[..., 0].astype(np.int32); cb = ycc[..., 1].astype(np.int32) - 128; cr = ycc[..., 2].astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


def _imshape(path):
    """(rows, cols) of an image file from its header (no decode).  JPEG and PNG headers are read directly -- the batched path asks for the
    size of every file before the first decode starts, and ninety `Image.open` calls were 10-20 ms of interpreter time in front of the
    whole pipeline; anything else (or anything unexpected) goes through Pillow."""
    try:
        with open(path, "rb") as f:
            head = f.read(2048)
            if head[:2] == b"\xff\xd8" and b"\xff\xc0" not in head and b"\xff\xc2" not in head:
                head += f.read((1 << 18) - 2048)               # a long EXIF / ICC block in front of the frame header
        if head[:8] == b"\x89PNG\r\n\x1a\n" and head[12:16] == b"IHDR":
            return (int.from_bytes(head[20:24], "big"), int.from_bytes(head[16:20], "big"))
        if head[:2] == b"\xff\xd8":
            p, n = 2, len(head)
            while p + 9 < n:
                if head[p] != 0xFF:
                    break
                m = head[p + 1]
                if m == 0xFF:                                  # fill byte
                    p += 1
                    continue
                if 0xD0 <= m <= 0xD9 or m == 0x01:             # markers without a length
                    p += 2
                    continue
                seg = int.from_bytes(head[p + 2:p + 4], "big")
                if 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):      # SOFn: precision, height, width
                    h, w = int.from_bytes(head[p + 5:p + 7], "big"), int.from_bytes(head[p + 7:p + 9], "big")
                    if h > 0 and w > 0:
                        return (h, w)
                    break
                p += 2 + seg
    except OSError:
        pass
    from PIL import Image
    with Image.open(path) as im:
        return (im.size[1], im.size[0])


def _list_images(folder, extension):
    """glob(folder/*.ext): the reference relies on Windows semantics (case-insensitive, name order)."""
    ext = "." + extension.lower()
    names = [n for n in os.listdir(folder) if n.lower().endswith(ext)] if os.path.isdir(folder) else []
    return [os.path.join(folder, n) for n in sorted(names)]

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


class TileIngest:
    """The ingest pipeline of one file list (Stitcher._registerBatched).  The reference decodes the whole file list before the first pair is
    looked at, and each tile three times (Stitcher.py:68-69, 382-403).  Here every tile gets its device handle(s) up front
    (vfsms_tile_reserve) and a pool of decoder threads (the library's own JPEG decoder, or Pillow with the GIL released) fills them in path
    order; the native registrar starts at once and waits only for the tiles of the batch it is about to launch, so registration overlaps
    decoding and the decoded arrays never pile up on the host (a thread holds one tile at a time).  Tiles a previous segment of this file
    list decoded but did not use (they lay behind its registration break, flowStitchWithMutiple) are taken over as they are -- a file is
    decoded once even when the path breaks (the reference decodes the remaining list again after every break, Stitcher.py:96-127).

        job = TileIngest(stitcher, files, shapes, color, keep);  handles = job.start();  ...register...;  job.finish(table)

    `finish` always runs (the caller's `finally`; `failed = True` first when the registration raised): decodes that have not started are
    cancelled, running ones are waited for, every handle is released, handed to getStitchByOffset (`keep`: stitcher._resident) or parked for
    the next segment (stitcher._ingestCache); the decoder's error is raised unless the file lies behind the registration break."""

    def __init__(self, stitcher, fileList, shapes, color, keep):
        self.st, self.eng = stitcher, stitcher.engine
        self.fileList, self.shapes, self.color, self.keep = fileList, shapes, color, keep
        self.handles, self.chandles, self.futures, self.todo = [], [], [], []
        self.failed = False
        self._blocks = None

    def start(self):
        """reserve (or take over) a device tile per file and hand the files to the decoder pool -> gray tile handles, in file order"""
        import threading
        import time
        from . import stitcher as ST                      # (the decode helpers are looked up there at call time: tests swap them)
        st, eng, fileList, shapes, color = self.st, self.eng, self.fileList, self.shapes, self.color
        handles, chandles = self.handles, self.chandles
        if not hasattr(eng, "tile_reserve"):
            handles.extend(eng.tile_upload(ST._imread(f, False)) for f in fileList)
            return handles
        cache = st.__dict__.get("_ingestCache") or {}
        have = []
        for s, f in zip(shapes, fileList):                # one by one: a reserve that fails midway leaves nothing behind (finish)
            ent = cache.get(f)
            if ent is not None and tuple(ent[2]) == tuple(s) and (bool(ent[1]) or not color):
                del cache[f]
                handles.append(ent[0])
                if color:
                    chandles.append(ent[1])
                elif ent[1]:
                    eng.tile_free(ent[1])
                have.append(True)
                continue
            handles.append(eng.tile_reserve(s[0], s[1]))
            if color:
                chandles.append(eng.tile_reserve_color(s[0], s[1], 3))
            have.append(False)
        nthreads = st._decoderThreads(len(fileList))
        self._blocks = ST._PillowBlocks(color)
        self._blocks.__enter__()
        istats = st._ingestStats = dict(tiles=0, decode_s=0.0, fill_s=0.0, threads=nthreads)   # summed over the decoder threads
        istats_mu = threading.Lock()

        def ingest(k):
            hc = chandles[k] if color else 0
            try:
                t0 = time.perf_counter()
                if ST._fill_from_jpeg(eng, fileList[k], handles[k], hc):
                    with istats_mu:
                        istats["tiles"] += 1; istats["native"] = istats.get("native", 0) + 1; istats["decode_s"] += time.perf_counter() - t0
                    return
                owner, shape, parts = ST._decode_once(fileList[k], color)
                t1 = time.perf_counter()
                if tuple(shape) != tuple(shapes[k]):
                    raise ValueError("decoded size %s of %s differs from its header %s" % (shape, fileList[k], shapes[k]))
                if parts[0] == "src":
                    if hc or parts[3] != 0:
                        eng.tile_fill_pair(handles[k], hc, parts[1], parts[2], parts[3])
                    else:
                        eng.tile_fill_ptr(handles[k], parts[1], parts[2])
                else:
                    eng.tile_fill(handles[k], parts[1])
                    if hc:
                        eng.tile_fill(hc, parts[2])
                del owner
                with istats_mu:
                    istats["tiles"] += 1; istats["decode_s"] += t1 - t0; istats["fill_s"] += time.perf_counter() - t1
            except BaseException:
                for h in (handles[k], hc):                # the batch waiting for this tile fails instead of hanging
                    if h:
                        try:
                            eng.tile_fill(h, None)
                        except Exception:                 # already filled (the second fill of an "arrays" pair failed)
                            pass
                raise
        pool = ST._decoder_pool(nthreads)
        self.todo = [k for k in range(len(fileList)) if not have[k]]
        self.futures = [pool.submit(ingest, k) for k in self.todo]
        return handles

    def finish(self, table):
        """flowStitch discards everything behind a break (Stitcher.py:74-76) and the reference never opens those files: decodes that have not
        started are cancelled (their handles are given up so that they can be freed), and a file behind the last registered pair that fails
        to decode is not an error of this call"""
        st, eng, fileList, shapes, color = self.st, self.eng, self.fileList, self.shapes, self.color
        handles, chandles, failed = self.handles, self.chandles, self.failed
        keep = self.keep and not failed
        # tiles of this segment: 0 .. (leading registered pairs); the rest lies behind the break
        n_used = len(fileList)
        if table is not None and not failed:
            n_used = 1
            for row in table:
                if not row[0]:
                    break
                n_used += 1
            n_used = min(n_used, len(fileList))
        # (the incremental registrars return a FULL-length table with zero rows behind the break, so len(table) says nothing: what the
        #  registration looked at are the tiles of the leading registered pairs plus the B tile of the pair that failed)
        needed = len(fileList) if (failed or table is None) else min(n_used + 1, len(fileList))
        err, unfilled = None, set()
        for k, fu in zip(self.todo, self.futures):
            if fu.cancel():
                unfilled.add(k)
                for h in ([handles[k]] + ([chandles[k]] if color else [])):
                    try:
                        eng.tile_fill(h, None)
                    except Exception:
                        pass
        for k, fu in zip(self.todo, self.futures):        # every running decoder has finished with its handles before any is freed
            if fu.cancelled():
                continue
            try:
                fu.result()
            except BaseException as e:                     # noqa: PERF203
                unfilled.add(k)
                if k < needed:
                    err = err or e
        if self._blocks is not None:
            self._blocks.__exit__(None, None, None)
        stash = st.__dict__.get("_ingestCache") if (not failed and err is None) else None
        mine = list(range(len(handles)))

        def release(k):
            for h in ([handles[k]] + ([chandles[k]] if k < len(chandles) else [])):
                try:
                    eng.tile_free(h)
                except Exception:
                    # still reserved: its decoder never ran (a reserve further down the list failed before the pool started) -- give it
                    # up first, a reserved tile cannot be freed
                    try:
                        eng.tile_fill(h, None)
                        eng.tile_free(h)
                    except Exception:
                        if not failed and err is None:
                            raise
        if keep and err is None:
            # the mosaic is assembled from these very tiles: getStitchByOffset takes them over (and frees them)
            kept = chandles if color else handles
            st._resident = {fileList[k]: (kept[k], (shapes[k][0], shapes[k][1], 3) if color else shapes[k]) for k in range(n_used)}
            for k in range(n_used):
                if color:
                    eng.tile_free(handles[k])
            mine = list(range(n_used, len(handles)))
        for k in mine:
            if stash is not None and k >= n_used and k not in unfilled and fileList[k] not in stash and k < len(handles) and \
                    (not color or k < len(chandles)):
                stash[fileList[k]] = (handles[k], chandles[k] if color else 0, shapes[k])      # decoded, unused: the next segment takes it over
            else:
                release(k)
        if err is not None and not failed:
            raise err

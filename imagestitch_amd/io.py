"""Mosaic write-out: cv2.imwrite's stand-in and the band sinks (`Stitcher.mosaicSink` / `streamOutput`) that encode a mosaic while its bands
still leave the GPU -- JPEG in stripes on all host cores (the library's encoder), PNG, TIFF / BigTIFF, NPY.  The way OUT of the path (SURVEY
section 8 row f-2); `stitcher.py` takes these names from here.

Reference: cv2.imwrite at /root/reference/Stitcher.py:150-153, 196-199 (Main.py's `.jpg` results).
"""
import os

import numpy as np

def _imwrite(path, img):
    if img is None:                                          # streamed to Stitcher.mosaicSink instead
        return
    from PIL import Image
    img = np.asarray(img)
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    if path.lower().endswith((".jpg", ".jpeg")) and _imwrite_jpeg_stripes(path, img):
        return
    if img.ndim == 3:
        img = img[:, :, ::-1]
    kw = {"quality": 95} if path.lower().endswith((".jpg", ".jpeg")) else {}     # cv2.imwrite's JPEG default (Pillow's is 75)
    Image.fromarray(np.ascontiguousarray(img)).save(path, **kw)


def _imwrite_jpeg_stripes(path, img):
    """a .jpg result through the library's encoder (JpegBandWriter: the stripes of the image on all cores) -> True; False = not written
    (no libjpeg.so.8, VFSMS_NATIVE_JPEG=0, an image libjpeg cannot hold): Pillow writes it."""
    if os.environ.get("VFSMS_NATIVE_JPEG", "1") == "0" or img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3):
        return False
    if max(img.shape[:2]) > 65500 or min(img.shape[:2]) < 1:
        return False
    w = JpegBandWriter(path)
    try:
        w(0, img, img.shape)
    except _NoNativeJpeg:
        return False
    return True


class NpyBandWriter:
    """A `Stitcher.mosaicSink`: writes the bands of a mosaic into one .npy file through a memory map, so a mosaic larger than host
    memory can be assembled (set `stitcher.mosaicSink = NpyBandWriter(path)`; getStitchByOffset then returns None)."""

    transient_bands = True      # done with a band when the call returns

    def __init__(self, path):
        self.path, self._mm = path, None

    def __call__(self, row0, band, full_shape):
        if self._mm is None:
            d = os.path.dirname(self.path)
            if d and not os.path.exists(d):
                os.makedirs(d)
            self._mm = np.lib.format.open_memmap(self.path, mode="w+", dtype=np.uint8, shape=tuple(full_shape))
        self._mm[row0:row0 + band.shape[0]] = band
        if row0 + band.shape[0] >= full_shape[0]:
            self._mm.flush()
            self._mm = None


class PngBandWriter:
    """A `Stitcher.mosaicSink` that encodes the bands into ONE PNG as they leave the device (cv2.imwrite's job at Stitcher.py:174-179,
    without the whole mosaic in host memory): IHDR, then every band deflated into IDAT chunks by a streaming zlib compressor (filter 0
    on every row), IEND.  Bands are B G R like the canvas; the file is R G B."""

    transient_bands = True      # done with a band when the call returns

    def __init__(self, path, level=1):
        self.path, self.level, self._f, self._z = path, level, None, None

    @staticmethod
    def _chunk(f, tag, data):
        import struct
        import zlib
        f.write(struct.pack(">I", len(data))); f.write(tag); f.write(data)
        f.write(struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xffffffff))

    def __call__(self, row0, band, full_shape):
        import struct
        import zlib
        if self._f is None:
            d = os.path.dirname(self.path)
            if d and not os.path.exists(d):
                os.makedirs(d)
            self._f = open(self.path, "wb")
            self._f.write(b"\x89PNG\r\n\x1a\n")
            ch = full_shape[2] if len(full_shape) == 3 else 1
            self._chunk(self._f, b"IHDR", struct.pack(">IIBBBBB", full_shape[1], full_shape[0], 8, 2 if ch == 3 else 0, 0, 0, 0))
            self._z = zlib.compressobj(self.level)
        band = np.asarray(band)
        if band.ndim == 3:
            band = band[:, :, ::-1]
        rows = np.empty((band.shape[0], 1 + band.shape[1] * (band.shape[2] if band.ndim == 3 else 1)), np.uint8)
        rows[:, 0] = 0                                           # filter type None
        rows[:, 1:] = band.reshape(band.shape[0], -1)
        data = self._z.compress(rows.tobytes())
        if data:
            self._chunk(self._f, b"IDAT", data)
        if row0 + band.shape[0] >= full_shape[0]:
            data = self._z.flush()
            if data:
                self._chunk(self._f, b"IDAT", data)
            self._chunk(self._f, b"IEND", b"")
            self._f.close()
            self._f = self._z = None


class _NoNativeJpeg(Exception):
    pass


_ENCODER_POOL = {}


def _encoder_pool(nthreads):
    from concurrent.futures import ThreadPoolExecutor
    pool = _ENCODER_POOL.get(nthreads)
    if pool is None:
        pool = _ENCODER_POOL[nthreads] = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="vfsms-encode")
    return pool


class JpegBandWriter:
    """A `Stitcher.mosaicSink` for the reference's own output format (Main.py:21-51 writes every result as .jpg; cv2.imwrite at
    Stitcher.py:149, 175-179): the bands are cut into STRIPES whose height is a multiple of the MCU height, every stripe is encoded on a pool
    of threads by the library (vfsms_jpeg_encode: the system's libjpeg-turbo with cv2.imwrite's settings, quality 95, no interpreter lock)
    while the next band is still leaving the device, and the stripes are joined into ONE baseline JPEG as restart intervals
    (vfsms_jpeg_join).  The DCT coefficients -- hence the decoded pixels -- are those of cv2.imwrite's / Pillow's one-thread encode of the
    whole mosaic; the file is a few bytes per stripe longer (RSTn markers + one DRI segment).  Only the compressed stripes are kept in
    host memory.  Bands are B G R like the canvas."""

    def __init__(self, path, quality=95, threads=None, stripe_rows=None):
        self.path, self.quality = path, int(quality)
        self.threads = int(threads or min(os.cpu_count() or 4, 32))
        self.stripe_rows = stripe_rows
        self._reset()

    transient_bands = True      # done with a band's memory once the band after the next one has been handed over (Engine.canvas_download_bands)

    def _reset(self):
        self._futures, self._carry, self._rows_in, self._stripe, self._shape, self._band_end = [], None, 0, None, None, []

    def _encode(self, rows):
        from . import _lib
        out = _lib.jpeg_encode(rows, bgr=True, quality=self.quality)
        if out is None:
            raise _NoNativeJpeg("no libjpeg.so.8 on this host")
        return out

    def _submit(self, rows):
        if len(self._futures) >= 4 * self.threads:             # encoders far behind the band stream: do not pile the bands up in host memory
            self._futures[len(self._futures) - 4 * self.threads].result()
        self._futures.append(_encoder_pool(self.threads).submit(self._encode, rows))

    def __call__(self, row0, band, full_shape):
        band = np.asarray(band)
        if self._shape is None:
            rows, cols = int(full_shape[0]), int(full_shape[1])
            ch = int(full_shape[2]) if len(full_shape) == 3 else 1
            if ch not in (1, 3) or max(rows, cols) > 65500:
                raise ValueError("JPEG holds 1- or 3-channel images of at most 65500 pixels a side (this one: %s)" % (tuple(full_shape),))
            mcu = 16 if ch == 3 else 8
            per_row = (cols + mcu - 1) // mcu
            k = max(1, min((int(self.stripe_rows) if self.stripe_rows else 256) // mcu, 65535 // per_row))    # a stripe is one restart interval: <= 65535 MCUs
            self._stripe, self._shape = k * mcu, (rows, cols, ch)
        rows, cols, ch = self._shape
        assert row0 == self._rows_in and band.shape[1] == cols, "bands arrive in order"
        self._rows_in += band.shape[0]
        last = self._rows_in >= rows
        try:
            if self._carry is not None:
                band = np.concatenate([self._carry, band], 0)
                self._carry = None
            if not band.flags.c_contiguous:
                band = np.ascontiguousarray(band)
            S, n = self._stripe, band.shape[0]
            full = n if last else (n // S) * S
            for r in range(0, full, S):
                self._submit(band[r:min(r + S, full)])
            if full < n:
                self._carry = band[full:].copy()
            self._band_end.append(len(self._futures))
            if len(self._band_end) >= 2:                       # the band before this one may be overwritten after the next call: its stripes are done
                for f in self._futures[(self._band_end[-3] if len(self._band_end) >= 3 else 0):self._band_end[-2]]:
                    f.result()
            if not last:
                return
            from . import _lib
            parts = [f.result() for f in self._futures]
            data = _lib.jpeg_join(parts, S, rows)
            d = os.path.dirname(self.path)
            if d and not os.path.exists(d):
                os.makedirs(d)
            with open(self.path, "wb") as f:
                f.write(memoryview(data))
            self._reset()
        except BaseException:
            for f in self._futures:
                f.cancel()
            for f in self._futures:
                try:
                    f.result()
                except BaseException:                          # noqa: PERF203
                    pass
            self._reset()
            raise


class TiffBandWriter:
    """A `Stitcher.mosaicSink` for uncompressed baseline TIFF (BigTIFF beyond 4 GB): one strip per band, the directory written behind the
    last band.  R G B (or gray) 8-bit samples."""

    transient_bands = True      # done with a band when the call returns

    def __init__(self, path, force_big=False):
        self.path, self._f, self._strips = path, None, []
        self.force_big = force_big                               # BigTIFF whatever the size (tests: the > 4 GB layout on a small image)

    def __call__(self, row0, band, full_shape):
        import struct
        rows, cols = full_shape[0], full_shape[1]
        ch = full_shape[2] if len(full_shape) == 3 else 1
        big = self.force_big or rows * cols * ch + (1 << 20) >= (1 << 32)
        if self._f is None:
            d = os.path.dirname(self.path)
            if d and not os.path.exists(d):
                os.makedirs(d)
            self._f = open(self.path, "wb")
            self._f.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, 0) if big else struct.pack("<2sHI", b"II", 42, 0))
            self._strips, self._band_rows = [], band.shape[0]
        band = np.asarray(band)
        if band.ndim == 3:
            band = band[:, :, ::-1]
        self._strips.append((self._f.tell(), band.size))
        self._f.write(np.ascontiguousarray(band).tobytes())
        if row0 + band.shape[0] < rows:
            return
        f, n = self._f, len(self._strips)
        if f.tell() & 1:
            f.write(b"\0")
        fmt_off = "<%dQ" % n if big else "<%dI" % n
        off_pos = f.tell(); f.write(struct.pack(fmt_off, *[o for o, _c in self._strips]))
        cnt_pos = f.tell(); f.write(struct.pack(fmt_off, *[c for _o, c in self._strips]))
        # BitsPerSample 8, 8, 8: three SHORTs are 6 bytes -- more than classic TIFF's 4-byte value field (stored behind the strips, the field
        # holds the offset), but they FIT BigTIFF's 8-byte field and must then sit in it (a reader takes the field as the values)
        bps = 8 | 8 << 16 | 8 << 32
        if ch == 3 and not big:
            bps = f.tell(); f.write(struct.pack("<3H", 8, 8, 8)); f.write(b"\0\0")
        ifd = f.tell()
        ltype = 16 if big else 4                                 # LONG8 / LONG
        tags = [(256, ltype, 1, cols), (257, ltype, 1, rows), (258, 3, ch, bps if ch == 3 else 8), (259, 3, 1, 1),
                (262, 3, 1, 2 if ch == 3 else 1), (273, ltype, n, off_pos if n > 1 else self._strips[0][0]), (277, 3, 1, ch),
                (278, ltype, 1, self._band_rows), (279, ltype, n, cnt_pos if n > 1 else self._strips[0][1])]
        if big:
            f.write(struct.pack("<Q", len(tags)))
            for t, ty, c, v in tags:
                f.write(struct.pack("<HHQQ", t, ty, c, v))
            f.write(struct.pack("<Q", 0))
            f.seek(8); f.write(struct.pack("<Q", ifd))
        else:
            f.write(struct.pack("<H", len(tags)))
            for t, ty, c, v in tags:
                f.write(struct.pack("<HHII", t, ty, c, v))
            f.write(struct.pack("<I", 0))
            f.seek(4); f.write(struct.pack("<I", ifd))
        f.close()
        self._f = None


def _native_jpeg_encoder():
    """does the library encode JPEG on this host (libjpeg.so.8 present, not switched off)?"""
    if os.environ.get("VFSMS_NATIVE_JPEG", "1") == "0":
        return False
    try:
        from . import _lib
        return _lib.jpeg_encode(np.zeros((8, 8), np.uint8)) is not None
    except Exception:
        return False


def band_writer_for(path):
    """the streaming encoder for an output file name, or None when its format has none here (JPEG without libjpeg.so.8 on the host: written
    whole through Pillow)"""
    ext = os.path.splitext(path)[1].lower()
    if ext in (".jpg", ".jpeg"):
        return JpegBandWriter(path) if _native_jpeg_encoder() else None
    return PngBandWriter(path) if ext == ".png" else TiffBandWriter(path) if ext in (".tif", ".tiff") else NpyBandWriter(path) if ext == ".npy" else None

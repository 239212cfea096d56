"""Test doubles.  OracleEngine implements the Engine interface on top of the CPU oracle so that the HOST
logic of imagestitch_amd (state machines, layout, ROI arithmetic, dispatch) can be checked on a machine
without a GPU against the reference's golden outputs.  It lives under tests/ and is never shipped."""
import numpy as np


class OracleEngine:
    def __init__(self, oracle):
        self.O = oracle
        self._canvases = {}
        self._next = 1
        self.fuse_calls = []

    # operators ------------------------------------------------------------------------------------
    @staticmethod
    def surf_params(hessian=100.0, n_octaves=4, n_layers=3, extended=False, upright=False):
        class P:  # noqa
            pass
        p = P(); p.hessian_threshold = hessian; p.n_octaves = n_octaves; p.n_octave_layers = n_layers
        p.extended = int(extended); p.upright = int(upright)
        return p

    def surf_detect_describe(self, img, params=None, cap=None, full=False):
        p = params or self.surf_params()
        k, d = self.O.surf_detect_describe(np.ascontiguousarray(img), p.hessian_threshold, p.n_octaves, p.n_octave_layers,
                                           bool(p.extended), bool(p.upright))
        return np.stack([k["x"], k["y"]], 1).astype(np.float32).reshape(-1, 2), d

    @staticmethod
    def orb_params(*a, **k):
        return None

    def orb_detect_describe(self, img, params=None, cap=None, full=False):
        k, d = self.O.orb_detect_describe(np.ascontiguousarray(img))
        return np.stack([k["x"], k["y"]], 1).astype(np.float32).reshape(-1, 2), d

    def bf_hamming_matches(self, q, t, max_dist=-1):
        return self.O.bf_hamming_matches(q, t, max_dist)[0]

    def bf_l2_ratio_matches(self, q, t, ratio=0.75):
        return self.O.bf_l2_ratio_matches(q, t, ratio)

    def mode_offset(self, kA, kB, pairs, ev=3):
        return self.O.mode_offset(kA, kB, pairs, ev)

    def phase_correlate(self, a, b):
        return self.O.phase_correlate(np.ascontiguousarray(a), np.ascontiguousarray(b))

    def fuse_fade_i64(self, A, B, dx, dy, return_info=False):
        return self.O.fuse_fade(A, B, dx, dy, return_info=return_info)

    def fuse_ramps_i64(self, A, dx, dy, force_corner=False):
        assert force_corner
        wr, wc, info = self.O.corner_ramps(A)
        return (np.ones_like(wr), wr, np.ones_like(wc), wc), info

    # canvas (int64 / -1 like the reference) ----------------------------------------------------------
    def canvas_create(self, rows, cols, ch):
        h = self._next; self._next += 1
        self._canvases[h] = np.zeros((rows, cols, ch) if ch > 1 else (rows, cols), np.int64) - 1
        return h

    def canvas_free(self, h):
        del self._canvases[h]

    def canvas_paste(self, h, tile, y0, x0):
        self._canvases[h][y0:y0 + tile.shape[0], x0:x0 + tile.shape[1]] = tile

    def fuse_trig_i64(self, A, B, dx, dy, return_info=False):
        """ImageFusion.fuseByTrigonometric (ImageFusion.py:246-293) in numpy, as the reference evaluates it (CPU test double)."""
        import math
        imageA = np.array(A, np.int64); imageB = np.asarray(B, np.int64)
        row, col = imageA.shape[:2]
        tail = (1,) * (imageA.ndim - 2)
        if np.count_nonzero(imageA > -1) / imageA.size > 0.65:
            weightMatA = np.ones(imageA.shape, dtype=np.float64)
            if col <= row:
                k = np.arange(col, dtype=np.float64)
                weightMatA = weightMatA * ((k if dy >= 0 else (col - k)) * 1.0 / col).reshape((1, col) + tail)
            else:
                k = np.arange(row, dtype=np.float64)
                weightMatA = weightMatA * ((k if dx <= 0 else (row - k)) * 1.0 / row).reshape((row, 1) + tail)
        else:
            wr, wc, _info = self.O.corner_ramps(imageA)
            wB = (wr.reshape((row, 1) + tail) * wc.reshape((1, col) + tail)).astype(np.float32) * np.ones(imageA.shape, np.float32)
            weightMatA = np.float32(1) - wB
        weightMatA = np.power(np.sin(weightMatA * math.pi / 2), 2)
        weightMatB = 1 - weightMatA
        hole = imageA < 0
        imageA[hole] = imageB[hole]
        result = weightMatA * imageA + weightMatB * imageB
        result[result < 0] = 0
        result[result > 255] = 255
        return np.uint8(result)

    def canvas_fuse_tile(self, h, tile, y0, x0, roi, dx, dy, method=0):
        cv = self._canvases[h]
        ry0, rx0, ry1, rx1 = roi
        A = cv[ry0:ry1, rx0:rx1].copy()
        cv[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1]] = tile
        B = cv[ry0:ry1, rx0:rx1].copy()
        self.fuse_calls.append((A.shape, dx, dy))
        if A.size:
            cv[ry0:ry1, rx0:rx1] = self.fuse_trig_i64(A, B, dx, dy) if method == 1 else self.O.fuse_fade(A, B, dx, dy)

    def canvas_download(self, h, rows, cols, ch):
        cv = self._canvases[h].copy()
        cv[cv == -1] = 0
        return cv.astype(np.uint8)


class IngestOracleEngine(OracleEngine):
    """OracleEngine plus the fused-attempt, ingest (tile_reserve / tile_fill / tile_fill_pair) and resident-canvas entry points, so that the
    Stitcher's decode-once pipeline and its error paths run on a machine without a GPU.  Tiles are numpy arrays keyed by handle; a reserved
    tile is an Event the attempts wait on.  tile_fill_pair restates csrc/ingest_kernels.hip with stitcher._ycc_to_bgr."""

    def __init__(self, oracle, scripted=None):
        super().__init__(oracle)
        import threading
        self.tiles, self.ready, self.failed, self.live = {}, {}, set(), set()
        self.batches = 0
        self.scripted = scripted                              # optional: f(job index in call order) -> row, instead of the oracle chain
        self._mu = threading.Lock()

    # tiles -------------------------------------------------------------------------------------------
    def _new(self):
        with self._mu:
            h = self._next; self._next += 1
            self.live.add(h)
            return h

    def tile_upload(self, img):
        h = self._new(); self.tiles[h] = np.ascontiguousarray(img)
        return h

    def tile_reserve(self, h, w):
        import threading
        hd = self._new(); self.ready[hd] = threading.Event(); self.tiles[hd] = ("reserved", (h, w))
        return hd

    def tile_reserve_color(self, h, w, ch=3):
        import threading
        hd = self._new(); self.ready[hd] = threading.Event(); self.tiles[hd] = ("reserved", (h, w, ch))
        return hd

    def _deliver(self, hd, arr):
        assert hd in self.ready and not self.ready[hd].is_set(), "not a reserved tile"
        if arr is None:
            self.failed.add(hd)
        else:
            assert arr.shape == self.tiles[hd][1], (arr.shape, self.tiles[hd][1])
            self.tiles[hd] = np.array(arr, np.uint8)
        self.ready[hd].set()

    def tile_fill(self, hd, img):
        self._deliver(hd, img)

    def tile_fill_ptr(self, hd, address, stride):
        import ctypes
        h, w = self.tiles[hd][1]
        buf = np.frombuffer((ctypes.c_uint8 * (h * stride)).from_address(address), np.uint8).reshape(h, stride)[:, :w]
        self._deliver(hd, buf)

    SRC_GRAY8, SRC_YCC24, SRC_YCCX32 = 0, 1, 2

    def tile_fill_pair(self, gray, color, address, stride, fmt):
        import ctypes
        from imagestitch_amd.stitcher import _ycc_to_bgr
        if address is None:
            for hd in (gray, color):
                if hd:
                    self._deliver(hd, None)
            return
        h, w = self.tiles[gray or color][1][:2]
        spx = {0: 1, 1: 3, 2: 4}[fmt]
        buf = np.frombuffer((ctypes.c_uint8 * (h * stride)).from_address(address), np.uint8).reshape(h, stride)[:, :w * spx].reshape(h, w, spx)
        if gray:
            self._deliver(gray, buf[:, :, 0])
        if color:
            self._deliver(color, np.repeat(buf, 3, axis=2) if fmt == 0 else _ycc_to_bgr(buf[:, :, :3]))

    def tile_free(self, hd):
        with self._mu:
            assert hd in self.live, "freed twice / unknown handle"
            assert hd not in self.ready or self.ready[hd].is_set(), "tile_free: the tile is reserved and its decoder has not filled it yet"
            self.live.remove(hd)
            self.tiles.pop(hd, None)

    def _tile(self, hd):
        if hd in self.ready:
            assert self.ready[hd].wait(30), "a reserved tile was never filled"
            if hd in self.failed:
                raise RuntimeError("a reserved tile was never filled (its decoder reported a failure)")
        return self.tiles[hd]

    def set_keypoint_capacity(self, n):
        pass

    # fused attempts ------------------------------------------------------------------------------------
    def attempt_surf_batch(self, jobs, params=None, ratio=0.75, offset_evaluate=3):
        self.batches += 1
        out = np.zeros((len(jobs), 8), np.int32)
        for n, job in enumerate(jobs):
            ta, tb, ay0, ax0, by0, bx0, h, w = [int(v) for v in job]
            A, B = self._tile(ta), self._tile(tb)
            if self.scripted is not None:
                out[n] = self.scripted(A, B, job)
                continue
            a = np.ascontiguousarray(A[ay0:ay0 + h, ax0:ax0 + w]); b = np.ascontiguousarray(B[by0:by0 + h, bx0:bx0 + w])
            ka, da = self.surf_detect_describe(a); kb, db = self.surf_detect_describe(b)
            if len(ka) == 0 or len(kb) == 0:
                out[n] = [0, 0, 0, 0, len(ka), len(kb), 0, 0]; continue
            pairs = self.bf_l2_ratio_matches(da, db, ratio)
            st, off, votes = self.mode_offset(ka, kb, pairs, offset_evaluate) if len(pairs) else (False, [0, 0], 0)
            out[n] = [int(st), off[0], off[1], votes, len(ka), len(kb), len(pairs), 0]
        return out

    # resident canvas calls ---------------------------------------------------------------------------------
    def canvas_paste_tile(self, h, tile_handle, y0, x0):
        self.canvas_paste(h, self._tile(tile_handle), y0, x0)

    def canvas_fuse_tile_resident(self, h, tile_handle, y0, x0, roi, dx, dy, want_info=False, method=0):
        self.canvas_fuse_tile(h, self._tile(tile_handle), y0, x0, roi, dx, dy, method=method)

    def canvas_download_bands(self, h, rows, cols, ch, band_rows=4096, transient=False):
        """`transient`: like the engine's pinned band ring -- three buffers, a band is OVERWRITTEN (here: scribbled over first) when the
        third band after it is requested"""
        img = self.canvas_download(h, rows, cols, ch)
        ring = [np.empty((min(band_rows, rows),) + img.shape[1:], np.uint8) for _ in range(3)] if transient else None
        self.transient_bands_served = getattr(self, "transient_bands_served", 0)
        for k, r0 in enumerate(range(0, rows, band_rows)):
            if not transient:
                yield r0, img[r0:r0 + band_rows]
                continue
            buf = ring[k % 3]
            buf[:] = 0x5A
            n = min(band_rows, rows - r0)
            buf[:n] = img[r0:r0 + n]
            self.transient_bands_served += 1
            yield r0, buf[:n]


class NativeJpegEngine(IngestOracleEngine):
    """IngestOracleEngine plus vfsms_tile_fill_jpeg: the decode is the PRODUCT's own host decoder (vfsms_jpeg_decode in libvfsms.so -- no GPU
    involved), the device half (plane split + colour conversion) is the restatement of tile_fill_pair.  Counts what it decoded."""

    def __init__(self, oracle, scripted=None):
        super().__init__(oracle, scripted)
        self.native, self.refused = [], []

    def tile_fill_jpeg(self, gray, color, data):
        from imagestitch_amd import _lib
        from imagestitch_amd.stitcher import _ycc_to_bgr
        out = _lib.jpeg_decode(data, want_planes=bool(color))
        shape = self.tiles[gray or color][1][:2]
        if out is None or out.shape[:2] != tuple(shape):
            self.refused.append(len(data))
            return False
        self.native.append(len(data))
        if gray:
            self._deliver(gray, out if out.ndim == 2 else out[:, :, 0])
        if color:
            self._deliver(color, np.repeat(out[:, :, None], 3, axis=2) if out.ndim == 2 else _ycc_to_bgr(out))
        return True

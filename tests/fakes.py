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

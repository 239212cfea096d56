"""Scripted engine for the batched / sharded registrar tests: answers fused attempts from a truth table
instead of running kernels.  A pair's attempt succeeds iff (direction, i) is in its accept set."""
import numpy as np


class ScriptedAttemptEngine:
    def __init__(self, shape, roiRatio, accept, raw=(7, -3)):
        """accept: list over pairs of {(direction, i): (raw_dx, raw_dy)} (or set of (direction, i))."""
        self.shape, self.roiRatio, self.accept, self.raw = shape, roiRatio, accept, raw
        self.log = []

    @staticmethod
    def surf_params(*a, **k):
        return None

    def _decode(self, job):
        ta, tb, ay0, ax0, by0, bx0, h, w = [int(v) for v in job]
        H, W = self.shape
        assert tb == ta + 1                      # handles are global tile indices in these tests
        if w == W and h < H:
            direction = 1 if (ay0 == H - h and by0 == 0) else 3
            assert (ay0, by0) == ((H - h, 0) if direction == 1 else (0, H - h))
            i = int(round(h / (self.roiRatio * H)))
        else:
            direction = 2 if (ax0 == W - w and bx0 == 0) else 4
            assert (ax0, bx0) == ((W - w, 0) if direction == 2 else (0, W - w))
            i = int(round(w / (self.roiRatio * W)))
        return ta, direction, i

    def attempt_surf_batch(self, jobs, params=None, ratio=0.75, offset_evaluate=3):
        out = np.zeros((len(jobs), 8), np.int32)
        for n, job in enumerate(jobs):
            k, d, i = self._decode(job)
            self.log.append((k, d, i))
            acc = self.accept[k]
            ok = (d, i) in acc
            raw = acc[(d, i)] if ok and isinstance(acc, dict) else self.raw
            out[n] = [int(ok), raw[0], raw[1], 5 if ok else 1, 100, 100, 10, 0]
        return out


def random_truth(rng, n_pairs, roiRatio, p_fail=0.05, p_false=0.2):
    maxI = int(np.floor(0.5 / roiRatio) + 1) + 1
    accept = []
    d = int(rng.integers(1, 5))
    for k in range(n_pairs):
        if rng.random() < 0.25:
            d = int(rng.integers(1, 5))
        acc = {}
        if rng.random() >= p_fail:
            i0 = int(rng.integers(1, maxI))
            for i in range(i0, maxI):
                acc[(d, i)] = (int(rng.integers(-9, 9)), int(rng.integers(-9, 9)))
        if rng.random() < p_false:           # a wrong direction that also votes >= offsetEvaluate: order matters
            acc[(int(rng.integers(1, 5)), int(rng.integers(1, maxI)))] = (int(rng.integers(-9, 9)), int(rng.integers(-9, 9)))
        accept.append(acc)
    return accept


def serpentine_truth(rows, cols, roiRatio, seed=5):
    """The shooting path of BASELINE configs[4] (32 x 32 tiles, column-major serpentine: rows-1 pairs down, one to the right,
    rows-1 up, ...): pair k accepts its true direction at i = 1 (every i), plus an occasional late success at i = 2."""
    rng = np.random.default_rng(seed)
    maxI = int(np.floor(0.5 / roiRatio) + 1) + 1
    accept = []
    for c in range(cols):
        d_col = 1 if c % 2 == 0 else 3
        for r in range(rows - 1):
            i0 = 2 if rng.random() < 0.03 else 1
            accept.append({(d_col, i): (int(rng.integers(-8, 9)), int(rng.integers(-8, 9))) for i in range(i0, maxI)})
        if c < cols - 1:
            accept.append({(2, i): (int(rng.integers(-8, 9)), int(rng.integers(-8, 9))) for i in range(1, maxI)})
    return accept

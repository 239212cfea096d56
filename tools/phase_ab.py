"""tools/phase_ab.py -- the LDS transforms of the phase path (csrc/phase_kernels.hip, VFSMS_PHASE_LDS_FFT=1) against the rocFFT path (=0) and the
CPU oracle on strips of many shapes (both orientations, odd / 3- and 5-smooth paddings, tiny and tile-sized), then the time of a batch.
GPU box only:  python tools/phase_ab.py [batch] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import imagestitch_amd as isa
from imagestitch_amd.synthetic import SyntheticGrid


def textured(seed, shape):
    rng = np.random.default_rng(seed)
    h, w = shape
    base = rng.random((h // 4 + 2, w // 4 + 2))
    img = np.kron(base, np.ones((4, 4)))[:h, :w] * 160 + rng.random((h, w)) * 60
    return img.astype(np.uint8)


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    eng = isa.Engine(0)
    from oracle import oracle          # the checker (tools only; never the product path)
    oracle.build(); oracle.lib()
    shapes = [(409, 2048), (2048, 409), (128, 640), (640, 128), (97, 131), (625, 64), (64, 625), (80, 96), (16, 16), (5, 7), (37, 64), (300, 1000),
              (819, 4096), (4096, 819), (100, 100), (1, 64), (64, 1), (2, 2), (1500, 1500), (243, 250)]
    worst = 0.0
    if len(sys.argv) > 3 and sys.argv[3] == "t":
        shapes = []                                                # timing only (under rocprofv3)
    for k, shp in enumerate(shapes):
        a = textured(2 * k, shp)
        dy, dx = min(5, shp[0] // 3), -min(9, shp[1] // 3)
        b = np.roll(np.roll(a, dy, 0), dx, 1)
        b = (b.astype(np.int32) + (textured(2 * k + 1, shp) >> 4)).clip(0, 255).astype(np.uint8)
        res = {}
        for mode in ("1", "0"):
            os.environ["VFSMS_PHASE_LDS_FFT"] = mode
            res[mode] = eng.phase_correlate(a, b)
        (x1, y1), r1 = res["1"]; (x0, y0), r0 = res["0"]
        d = max(abs(x1 - x0), abs(y1 - y0)); dr = abs(r1 - r0)
        line = "%-12s lds (%.9f, %.9f) r %.12f | rocfft d=%.2e dr=%.2e" % (shp, x1, y1, r1, d, dr)
        if oracle is not None and shp[0] * shp[1] <= 1 << 21:
            (ox, oy), orr = oracle.phase_correlate(np.ascontiguousarray(a), np.ascontiguousarray(b))
            do = max(abs(x1 - ox), abs(y1 - oy)); dro = abs(r1 - orr)
            line += " | oracle d=%.2e dr=%.2e" % (do, dro)
            worst = max(worst, do, dro * 1e3)
        worst = max(worst, d, dr * 1e3)
        print(line, flush=True)
    print("worst difference (px; response x 1e3):", worst)
    assert worst < 1e-6, worst
    # timing: a batch of nb attempts of the bench's two strip shapes, resident tiles
    g = SyntheticGrid(2, 2, 2048, overlap=0.10)
    tiles = g.tiles(threads=4)
    hs = [eng.tile_upload(t) for t in tiles]
    T = 2048; r = int(0.2 * T)
    geoms = {"409x2048 (strip above / below)": (T - r, 0, 0, 0, r, T), "2048x409 (strip left / right)": (0, T - r, 0, 0, T, r)}
    for name, ge in geoms.items():
        jobs = [(hs[k % 3], hs[k % 3 + 1]) + ge for k in range(nb)]
        for mode in (("1",) if len(sys.argv) > 3 and sys.argv[3] == "t" else ("0", "1")):
            os.environ["VFSMS_PHASE_LDS_FFT"] = mode
            out = eng.attempt_phase_batch(jobs); eng.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = eng.attempt_phase_batch(jobs)
            eng.sync()
            dt = (time.perf_counter() - t0) / reps
            print("%-32s %s: %.3f ms per batch of %d = %.1f us per attempt   first row %s" % (name, "LDS transforms" if mode == "1" else "rocFFT        ", dt * 1e3, nb,
                                                                                             dt * 1e6 / nb, np.round(out[0], 6)), flush=True)


if __name__ == "__main__":
    main()

"""Does the host scale JPEG decoding beyond what one interpreter lock allows?  Decode-only throughput of N threads (one process) against N
worker PROCESSES (spawned, Pillow only) on the synthetic grid's JPEG tiles, gray and colour.  Run on the GPU box (no GPU needed)."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _work(args):
    path, color = args
    from PIL import Image
    im = Image.open(path)
    im.decodermaxblock = 1 << 24
    im.draft("YCbCr" if color else "L", im.size)
    im.load()
    return im.size


def main():
    from PIL import Image
    from concurrent.futures import ThreadPoolExecutor
    import multiprocessing as mp
    from imagestitch_amd.synthetic import SyntheticGrid
    g = SyntheticGrid(10, 9, 2048)
    with tempfile.TemporaryDirectory() as d:
        files = {False: [], True: []}
        for k, t in enumerate(g.tiles(range(90), threads=16)):
            f = os.path.join(d, "g%03d.jpg" % k); Image.fromarray(t).save(f, quality=90); files[False].append(f)
            ft = t.astype(np.float32)
            c = np.clip(np.stack([0.6 * ft + 30, ft, 255 - 0.7 * ft], -1), 0, 255).astype(np.uint8)
            f = os.path.join(d, "c%03d.jpg" % k); Image.fromarray(c).save(f, quality=90); files[True].append(f)
        print("host threads", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
        for color in (False, True):
            t0 = time.perf_counter(); _work((files[color][0], color)); print("color" if color else "gray", "one tile %.2f ms" % ((time.perf_counter() - t0) * 1e3))
            for n in (16, 32, 64):
                with ThreadPoolExecutor(n) as ex:
                    list(ex.map(_work, [(f, color) for f in files[color][:n]]))
                    t0 = time.perf_counter(); list(ex.map(_work, [(f, color) for f in files[color]] * 2)); dt = (time.perf_counter() - t0) / 2
                print("  threads   %3d: %.1f ms per 90 tiles (%.0f tiles/s)" % (n, dt * 1e3, 90 / dt))
            ctx = mp.get_context("spawn")
            for n in (16, 32, 64):
                with ctx.Pool(n) as pool:
                    pool.map(_work, [(f, color) for f in files[color][:n]])
                    t0 = time.perf_counter(); pool.map(_work, [(f, color) for f in files[color]] * 2, chunksize=1); dt = (time.perf_counter() - t0) / 2
                print("  processes %3d: %.1f ms per 90 tiles (%.0f tiles/s)" % (n, dt * 1e3, 90 / dt))


if __name__ == "__main__":
    main()

"""One dataset the way Main.py runs it (Main.py:14-51): a folder of colour JPEG tiles in, `stitching_result_1.jpg` out -- decode, registration,
mosaic assembly with fadeInAndFadeOut, download, JPEG encode -- through Stitcher.imageSetStitchWithMutiple, timed as a whole.  Run on the GPU
box:   python tools/e2e_dataset.py [rows cols tile]   -> one JSON line (also with the host codecs switched to Pillow, VFSMS_NATIVE_JPEG=0).
Not the bench metric (that is pairs/s of the registration path): this is what a user of the reference waits for."""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(engine=None):
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    import imagestitch_amd as isa
    from imagestitch_amd.synthetic import SyntheticGrid
    rows, cols, tile = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (10, 9, 2048)))
    g = SyntheticGrid(rows, cols, tile)
    out = {"workload": "%dx%d grid of %dx%d colour JPEG tiles -> imageSetStitchWithMutiple (SURF, fadeInAndFadeOut, isColorMode) -> stitching_result_1.jpg" % (rows, cols, tile, tile),
           "pairs": g.n_pairs, "host_cores": os.cpu_count()}
    with tempfile.TemporaryDirectory(prefix="vfsms_e2e_") as d:
        proj = os.path.join(d, "proj"); os.makedirs(os.path.join(proj, "1"))
        for k, t in enumerate(g.tiles(range(g.n_tiles), threads=min(16, os.cpu_count() or 1))):
            ft = t.astype(np.float32)
            c = np.clip(np.stack([0.6 * ft + 30, ft, 255 - 0.7 * ft], -1), 0, 255).astype(np.uint8)
            Image.fromarray(c).save(os.path.join(proj, "1", "t%03d.jpg" % k), quality=90)
        isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.featureMethod, isa.Stitcher.offsetEvaluate = 1, 0.2, "surf", 3
        isa.Stitcher.isColorMode, isa.Stitcher.fuseMethod = True, "fadeInAndFadeOut"
        s = isa.Stitcher(); s.isPrintLog = False
        if engine is not None:                                 # (dry runs of this script on a machine without a GPU: tests/fakes.py)
            s._engine = engine
        # where the time goes (main-thread wall clock inside the named calls; the decoder and encoder pools run beside them)
        from imagestitch_amd import stitcher as ST
        phase = {}

        def timed(obj, attr, key):
            fn = getattr(obj, attr)

            def wrapper(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    phase[key] = phase.get(key, 0.0) + time.perf_counter() - t0
            setattr(obj, attr, wrapper)
        timed(s, "_registerBatched", "ingest_and_registration")
        timed(s, "getStitchByOffset", "mosaic_assembly_and_download")
        timed(ST, "_imwrite", "imwrite")
        eng = s.engine
        for attr in ("canvas_download", "canvas_download_rows"):
            if hasattr(eng, attr):
                timed(eng, attr, "of_which_download")
        sys.stdout = open(os.devnull, "w")                     # (the reference prints a line per dataset)
        try:
            # "default" = the Stitcher as it comes since round 5: results streamed to the encoder through the pinned band ring
            for name, env, stream, pinned, reps in (("default_streamed_pinned_bands", "1", True, "1", 4), ("whole_mosaic_download", "1", False, "0", 3),
                                                    ("streamed_pageable_bands", "1", True, "0", 3), ("default_streamed_pinned_bands_again", "1", True, "1", 3),
                                                    ("pillow_codecs", "0", False, "0", 1)):
                os.environ["VFSMS_NATIVE_JPEG"], os.environ["VFSMS_PINNED_BANDS"] = env, pinned
                s.streamOutput = stream
                times = []
                for r in range(reps):
                    isa.Stitcher.direction = 1; s.direction = 1
                    o = os.path.join(d, "out_%s_%d" % (name, r)) + os.sep
                    phase.clear()
                    t0 = time.perf_counter()
                    s.imageSetStitchWithMutiple(proj, o, 1, s.calculateOffsetForFeatureSearchIncre, startNum=1, fileExtension="jpg", outputfileExtension="jpg")
                    times.append(time.perf_counter() - t0)
                res = os.path.join(o, "stitching_result_1.jpg")
                with Image.open(res) as im:
                    size = im.size
                out[name] = {"seconds": [round(t, 3) for t in times], "best_s": round(min(times), 3), "result_px": [size[1], size[0]],
                             "result_MB": round(os.path.getsize(res) / 1e6, 1), "pairs_per_s_whole_job": round(g.n_pairs / min(times), 1),
                             "last_run_phases_s": {k: round(v, 3) for k, v in phase.items()}}
        finally:
            sys.stdout = sys.__stdout__
    print(json.dumps(out))


if __name__ == "__main__":
    main()

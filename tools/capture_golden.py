"""Generate tests/golden/* by RUNNING THE REFERENCE'S OWN PYTHON (container only).

    python tools/capture_golden.py

Imports /root/reference's Stitcher / ImageFusion / ImageUtility under the shims of tools/refshim.py and
records inputs + outputs of every pure-numpy function on the hot path (SURVEY.md section 8c list):
ROI slicing, mode vote, the incremental-ROI / direction state machines (with scripted fake operators),
getStitchByOffset layout + paste + fuse dispatch, fuseByFadeInAndFadeOut / getWeightsMatrix, and the
segment-restart driver.  Also extracts the only numeric ground truth in the reference (the 89-offset list
in the comment at Stitcher.py:87) and crops a few real demo strips whose expected offsets that list gives.

Fixtures are DATA (arrays in, arrays out).  No reference source text is stored.
"""
import ast
import json
import os
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import refshim  # noqa: E402

_phase_script = {}


def _phase_stub(a, b):
    _phase_script["calls"].append((a.shape, b.shape, str(a.dtype)))
    k = len(_phase_script["calls"])
    if k == _phase_script["success_at"]:
        return _phase_script["value"], 0.5
    return _phase_script["value"], 0.1


cv2, RS, RF, RU = refshim.install(phase_correlate=_phase_stub)
rng = np.random.default_rng(20190158)


# ------------------------------------------------------------------------------------------ ROI
def cap_roi():
    m = RU.Method()
    cases = []
    for shape in [(1936, 2584), (2048, 2048), (1024, 1280), (97, 131)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        base = img.__array_interface__["data"][0]
        for direction in (1, 2, 3, 4):
            for order in ("first", "second"):
                for ratio in (0.1, 0.2, 2 * 0.2, 3 * 0.2, 3 * 0.1, 6 * 0.1):
                    r = m.getROIRegionForIncreMethod(img, direction=direction, order=order, searchRatio=ratio)
                    off = r.__array_interface__["data"][0] - base
                    cases.append(dict(shape=list(shape), direction=direction, order=order, ratio=ratio,
                                      out_shape=list(r.shape), row0=int(off // shape[1]), col0=int(off % shape[1])))
    json.dump(cases, open(os.path.join(OUT, "roi_cases.json"), "w"))
    print("roi cases", len(cases))


# ------------------------------------------------------------------------------------------ mode
def cap_mode():
    m = RU.Method()
    m.isPrintLog = False
    store = {}
    exp = []
    ncase = 0
    for case in range(240):
        na, nb = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        kind = case % 8
        kA = (rng.random((na, 2)) * 400).astype(np.float32)
        kB = (rng.random((nb, 2)) * 400).astype(np.float32)
        M = int(rng.integers(0, 80)) if kind != 7 else 0
        pairs = np.stack([rng.integers(0, nb, M), rng.integers(0, na, M)], 1).astype(np.int32) if M else np.zeros((0, 2), np.int32)
        if kind in (1, 2, 3) and M:
            # plant a true shift so several matches agree (and near-integer / negative fractions occur)
            shift = np.array([rng.integers(-30, 30) + rng.choice([0.0, 0.999, -0.999, 0.5]), rng.integers(-30, 30) + 0.25], np.float32)
            for k in range(0, M, 2):
                kA[pairs[k, 1]] = kB[pairs[k, 0]] + shift
        if kind == 4 and M:   # everything identical -> all (0,0) votes dropped
            for k in range(M):
                kA[pairs[k, 1]] = kB[pairs[k, 0]] + np.float32(0.4)
        if kind == 5 and M > 4:  # engineered tie: two offsets with equal counts, order decides
            kB[:] = 100
            q = pairs[:, 1]
            for k in range(M):
                pairs[k, 1] = k % na
            kA[:] = 100
            for k in range(min(M, na)):
                kA[k] = 100 + (3 if k % 2 == 0 else 7)
        ev = int(rng.choice([1, 3, 3, 10]))
        (st, off) = m.getOffsetByMode(kA, kB, [tuple(int(v) for v in p) for p in pairs], offsetEvaluate=ev)
        store["c%d_kpsA" % ncase] = kA; store["c%d_kpsB" % ncase] = kB; store["c%d_pairs" % ncase] = pairs
        exp.append([int(ev), int(bool(st)), int(off[0]), int(off[1])])
        ncase += 1
    store["expected"] = np.array(exp, np.int32)
    np.savez_compressed(os.path.join(OUT, "mode_cases.npz"), **store)
    print("mode cases", ncase)


# ------------------------------------------------------------------------------------------ state machines
class ScriptedStitcher(RS.Stitcher):
    """Reference Stitcher with the three operators replaced by scripted fakes (control flow only)."""

    def __init__(self, success_at, raw):
        self.trace = []
        self.success_at = success_at
        self.raw = raw
        self.pending = []

    def detectAndDescribe(self, image, featureMethod):
        self.pending.append(list(image.shape))
        return (np.zeros((1, 2), np.float32), np.zeros((1, 64), np.float32))

    def matchDescriptors(self, featuresA, featuresB):
        return [(0, 0)]

    def getOffsetByMode(self, kpsA, kpsB, matches, offsetEvaluate=10):
        self.trace.append(self.pending[-2:])
        ok = len(self.trace) == self.success_at
        return (ok, list(self.raw))


def cap_state_machine():
    cases = []
    for shape in [(1936, 2584), (2048, 2048), (1024, 1280)]:
        imgA = np.zeros(shape, np.uint8); imgB = np.zeros(shape, np.uint8)
        for roiRatio in (0.1, 0.2):
            for ini in (1, 2, 3, 4):
                for incre in (-1, 0, 1):
                    maxI = int(np.floor(0.5 / roiRatio) + 1) + 1
                    max_att = (maxI - 1) * (1 if incre == 0 else 4)
                    for success_at in list(range(1, max_att + 1)) + [0]:
                        if shape != (1936, 2584) and success_at not in (0, 1, 2, 5, max_att):
                            continue
                        raw = [7, -3]
                        s = ScriptedStitcher(success_at, raw)
                        s.isPrintLog = False
                        s.roiRatio = roiRatio; s.direction = ini; s.directIncre = incre
                        s.featureMethod = "surf"; s.offsetCaculate = "mode"; s.isEnhance = False
                        st, off = s.calculateOffsetForFeatureSearchIncre([imgA, imgB])
                        rec = dict(kind="feature", shape=list(shape), roiRatio=roiRatio, ini=ini, incre=incre,
                                   success_at=success_at, raw=raw, trace=s.trace, status=bool(st),
                                   offset=(list(map(int, off)) if st else off), final_direction=int(s.direction))
                        cases.append(rec)
                        # phase variant: cv2.phaseCorrelate stub returns ((x, y), response)
                        _phase_script.update(calls=[], success_at=success_at, value=(-3.7, 7.9))
                        p = RS.Stitcher()
                        p.isPrintLog = False
                        p.roiRatio = roiRatio; p.direction = ini; p.directIncre = incre
                        st, off = p.calculateOffsetForPhaseCorrleateIncre([imgA, imgB])
                        cases.append(dict(kind="phase", shape=list(shape), roiRatio=roiRatio, ini=ini, incre=incre,
                                          success_at=success_at, raw=[-3.7, 7.9],
                                          trace=[[list(c[0]), list(c[1])] for c in _phase_script["calls"]],
                                          dtype=_phase_script["calls"][0][2], status=bool(st),
                                          offset=(list(map(int, off)) if st else off), final_direction=int(p.direction)))
    json.dump(cases, open(os.path.join(OUT, "state_machine.json"), "w"))
    print("state machine cases", len(cases))


def cap_feature_search_cache():
    """calculateOffsetForFeatureSearch (Stitcher.py:260-304): which images get described, cache reuse/invalidate."""
    class S(RS.Stitcher):
        def __init__(self, script):
            self.script = list(script); self.described = []; self.k = 0

        def detectAndDescribe(self, image, featureMethod):
            self.described.append(int(image[0, 0]))
            return (np.full((1, 2), image[0, 0], np.float32), np.full((1, 64), image[0, 0], np.float32))

        def matchDescriptors(self, fa, fb):
            self.matched = (int(fa[0, 0]), int(fb[0, 0]))
            return [(0, 0)]

        def getOffsetByMode(self, kpsA, kpsB, matches, offsetEvaluate=10):
            ok = self.script[self.k]; self.k += 1
            return (ok, [5, -6])
    out = []
    for script in ([True, True, True], [True, False, True, True], [False, True]):
        RS.Stitcher.tempImageFeature.isBreak = True
        s = S(script); s.isPrintLog = False; s.isEnhance = False; s.offsetCaculate = "mode"
        steps = []
        for k, _ in enumerate(script):
            A = np.full((8, 8), 10 + k, np.uint8); B = np.full((8, 8), 11 + k, np.uint8)
            n0 = len(s.described)
            st, off = s.calculateOffsetForFeatureSearch([A, B])
            steps.append(dict(described=s.described[n0:], matched=list(s.matched), status=bool(st),
                              offset=(list(off) if st else off), isBreak=bool(s.tempImageFeature.isBreak)))
        out.append(dict(script=script, steps=steps))
    RS.Stitcher.tempImageFeature.isBreak = True
    json.dump(out, open(os.path.join(OUT, "feature_search_cache.json"), "w"))
    print("feature-search cache cases", len(out))


# ------------------------------------------------------------------------------------------ fuse
def _mk_corner(r, c, ch, which, rng, t_r, t_c, black=False):
    """int64 canvas ROI with an L-shaped occupied region: `which` = empty corner quadrant id as in getWeightsMatrix."""
    shape = (r, c) if ch == 1 else (r, c, ch)
    A = rng.integers(1 if not black else 0, 256, shape).astype(np.int64)
    if which == 2:   # bottom-right empty: A occupies top strip (t_r rows) and left strip (t_c cols)
        A[t_r:, t_c:] = -1
    elif which == 3:  # top-right empty
        A[:r - t_r, t_c:] = -1
    elif which == 0:  # top-left empty
        A[:r - t_r, :c - t_c] = -1
    elif which == 1:  # bottom-left empty
        A[t_r:, :c - t_c] = -1
    return A


def cap_fuse():
    f = RF.ImageFusion()
    f.isPrintLog = False
    store = {}
    meta = []
    n = 0

    def add(A, B, dx, dy, color):
        nonlocal n
        f.isColorMode = color
        try:
            out = f.fuseByFadeInAndFadeOut([A.copy(), B.copy()], dx, dy)
        except (IndexError, ZeroDivisionError):
            return
        store["f%d_A" % n] = A; store["f%d_B" % n] = B; store["f%d_out" % n] = out
        meta.append([int(dx), int(dy), int(color)])
        n += 1

    # strip mode: full or >65 % occupied, both aspect classes, all sign combinations, gray + colour
    for (r, c) in [(6, 4), (4, 6), (5, 5), (40, 13), (13, 40), (64, 48), (1, 9), (9, 1)]:
        for ch in (1, 3):
            shape = (r, c) if ch == 1 else (r, c, ch)
            for dx, dy in [(5, 3), (-5, -3), (0, 0), (7, -2), (-7, 2)]:
                A = rng.integers(0, 256, shape).astype(np.int64)
                B = rng.integers(0, 256, shape).astype(np.int64)
                if r * c >= 25:   # sprinkle < 35 % empties into A
                    mask = rng.random((r, c)) < 0.2
                    A[mask] = -1
                add(A, B, dx, dy, ch == 3)
    A = np.full((6, 4), 100, np.int64); B = np.full((6, 4), 200, np.int64)
    for dx, dy in [(1, 1), (1, -1)]:
        add(A, B, dx, dy, False)
    A = np.full((4, 6), 100, np.int64); B = np.full((4, 6), 200, np.int64)
    for dx, dy in [(1, 1), (-1, 1), (0, 1)]:
        add(A, B, dx, dy, False)
    # corner mode: the four L-shapes, several thicknesses incl. degenerate 1-2 px strips, gray + colour
    for (r, c) in [(24, 30), (31, 23), (48, 64), (12, 12)]:
        for which in (0, 1, 2, 3):
            for (t_r, t_c) in [(3, 4), (1, 1), (2, 1), (1, 3), (5, 2)]:
                for ch in (1, 3):
                    A = _mk_corner(r, c, ch, which, rng, t_r, t_c)
                    shape = A.shape
                    B = rng.integers(0, 256, shape).astype(np.int64)
                    add(A, B, int(rng.integers(-9, 9)), int(rng.integers(-9, 9)), ch == 3)
    # corner mode with black (0) valid pixels (quadrant test uses > 0) and an all-empty A
    for which in (0, 1, 2, 3):
        A = _mk_corner(20, 26, 1, which, rng, 4, 5, black=True)
        A[A > 0] = np.where(rng.random(np.count_nonzero(A > 0)) < 0.5, 0, A[A > 0])
        add(A, rng.integers(0, 256, A.shape).astype(np.int64), 3, 3, False)
    store["meta"] = np.array(meta, np.int32)
    np.savez_compressed(os.path.join(OUT, "fuse_cases.npz"), **store)
    print("fuse cases", n)


# ------------------------------------------------------------------------------------------ getStitchByOffset
def _write_png(path, arr):
    from PIL import Image
    if arr.ndim == 3:
        Image.fromarray(arr[:, :, ::-1]).save(path)   # stored RGB so the stub's BGR flip restores arr
    else:
        Image.fromarray(arr).save(path)


def cap_stitch():
    store = {}
    meta = []
    n = 0
    tmp = tempfile.mkdtemp()
    th, tw = 64, 80
    # column-major serpentine 3x3 with jitter; plus a line scan to the left (direction 4) and negative sums
    serp = [[52, 2], [50, -1], [3, 66], [-51, 1], [-49, -2], [-2, 65], [50, 2], [52, -3]]
    line4 = [[1, -60], [-2, -58], [0, -61]]
    mixed = [[-30, 40], [60, -70], [-50, -20], [20, 75]]
    for name, offs in (("serp", serp), ("line4", line4), ("mixed", mixed)):
        ntile = len(offs) + 1
        for color in (False, True):
            tiles = [rng.integers(0, 256, (th, tw, 3) if color else (th, tw), dtype=np.uint8) for _ in range(ntile)]
            tiles[1][:7, :9] = 0   # some true black pixels: matters for average/max/min (0 treated as empty)
            files = []
            for k, t in enumerate(tiles):
                p = os.path.join(tmp, "%s_%d_%d.png" % (name, color, k)); _write_png(p, t); files.append(p)
            for fm in ("notFuse", "average", "maximum", "minimum", "fadeInAndFadeOut", "trigonometric"):
                s = RS.Stitcher()
                msgs = []
                s.printAndWrite = lambda c, msgs=msgs: msgs.append(c)
                RS.Stitcher.isColorMode = color   # getStitchByOffset reads both self.isColorMode and Stitcher.isColorMode
                s.fuseMethod = fm
                ol = [list(o) for o in offs]
                res = s.getStitchByOffset(files, ol)
                rect = [m for m in msgs if "rectified" in m][0]
                rect = ast.literal_eval(rect[rect.index("["):])
                store["s%d_tiles" % n] = np.stack(tiles)
                store["s%d_offsets" % n] = np.array(offs, np.int32)
                store["s%d_rect" % n] = np.array(rect, np.int32)
                store["s%d_out" % n] = res
                meta.append([int(color), ["notFuse", "average", "maximum", "minimum", "fadeInAndFadeOut", "trigonometric"].index(fm), int(len(ol) == ntile)])
                n += 1
    RS.Stitcher.isColorMode = True
    store["meta"] = np.array(meta, np.int32)
    np.savez_compressed(os.path.join(OUT, "stitch_cases.npz"), **store)
    print("stitch cases", n)


def cap_flow():
    """flowStitchWithMutiple segmenting (Stitcher.py:96-127) with a scripted offset method; notFuse, gray."""
    tmp = tempfile.mkdtemp()
    th, tw = 20, 24
    out = []
    store = {}
    n = 0
    for script in ([1, 1, 1, 1], [1, 0, 1, 1], [0, 1, 1], [1, 1, 0], [0, 0, 0], [1, 0, 0, 1, 1]):
        ntile = len(script) + 1
        tiles = [np.full((th, tw), 10 * (k + 1), np.uint8) for k in range(ntile)]
        files = []
        for k, t in enumerate(tiles):
            p = os.path.join(tmp, "flow_%d_%d.png" % (n, k)); _write_png(p, t); files.append(p)
        s = RS.Stitcher()
        msgs = []
        s.printAndWrite = lambda c, msgs=msgs: msgs.append(c)
        RS.Stitcher.isColorMode = False
        s.fuseMethod = "notFuse"

        def method(images, script=script):
            a = int(images[0][0, 0]) // 10 - 1   # pair index from the tile's fill value
            return (True, [15, 3]) if script[a] else (False, "  The two images can not match")
        res = s.flowStitchWithMutiple(files, method)
        for k, r in enumerate(res):
            store["w%d_res%d" % (n, k)] = r
        out.append(dict(script=script, nres=len(res), shapes=[list(r.shape) for r in res],
                        breaks=[m for m in msgs if "stitching Break" in m or "can not be stitched" in m].__len__()))
        n += 1
    RS.Stitcher.isColorMode = True
    json.dump(out, open(os.path.join(OUT, "flow_cases.json"), "w"))
    np.savez_compressed(os.path.join(OUT, "flow_cases.npz"), **store)
    print("flow cases", n)


# ------------------------------------------------------------------------------------------ real data pins
def cap_real():
    src = open(os.path.join(refshim.REF, "Stitcher.py"), encoding="utf-8-sig").read().splitlines()[86]
    gold = ast.literal_eval(src[src.index("["):])
    assert len(gold) == 89
    json.dump(dict(source="Stitcher.py:87 (commented-out offsetList; entry k = offset of tile k+2 relative to tile k+1)",
                   offsets=gold), open(os.path.join(OUT, "dendritic_offsets.json"), "w"))
    from PIL import Image

    def load(path):
        im = Image.open(path); im.draft("L", im.size)
        return np.asarray(im.convert("L"))
    m = RU.Method()
    store = {}
    meta = []
    d = os.path.join(refshim.REF, "demoImages", "dendriticCrystal", "1")
    # (tile a, tile b, direction, column/row crop) ; expected = gold[a-1]
    for n, (a, b, direction, lo, hi) in enumerate([(4, 5, 1, 600, 1900), (16, 17, 3, 0, 1300), (15, 16, 2, 300, 1500)]):
        A = load(os.path.join(d, "1-%03d.jpg" % a)); B = load(os.path.join(d, "1-%03d.jpg" % b))
        ra = m.getROIRegionForIncreMethod(A, direction=direction, order="first", searchRatio=0.2)
        rb = m.getROIRegionForIncreMethod(B, direction=direction, order="second", searchRatio=0.2)
        if direction in (1, 3):
            ra, rb = ra[:, lo:hi], rb[:, lo:hi]
        else:
            ra, rb = ra[lo:hi, :], rb[lo:hi, :]
        store["r%d_roiA" % n] = np.ascontiguousarray(ra); store["r%d_roiB" % n] = np.ascontiguousarray(rb)
        meta.append([a, b, direction, A.shape[0], A.shape[1]] + gold[a - 1])
    store["meta"] = np.array(meta, np.int32)
    np.savez_compressed(os.path.join(OUT, "real_strips.npz"), **store)
    print("real strips", len(meta), os.path.getsize(os.path.join(OUT, "real_strips.npz")) // 1024, "KiB")


def _oracle_search(O, roi_rect, A, B, d0, roiRatio=0.2, directIncre=1, offsetEvaluate=3, method="surf"):
    """Stitcher.calculateOffsetForFeatureSearchIncre (Stitcher.py:306-367) with the oracle's SURF + BF-L2 + ratio + mode vote
    (method "surf") or ORB + BF-Hamming 1-NN + mode vote (method "orb": ImageUtility.py:260, 297-302 -- every query votes, no ratio
    test, no distance threshold) as the operators -> (status, [dx, dy], direction, i, attempts log)."""
    def rot(d):
        d += directIncre
        return 1 if d == 5 else 4 if d == 0 else d
    log = []
    maxI = int(np.floor(0.5 / roiRatio) + 1) + 1
    for i in range(1, maxI):
        d = d0
        while True:
            ra = roi_rect(A.shape, d, "first", i * roiRatio); rb = roi_rect(B.shape, d, "second", i * roiRatio)
            a = np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]])
            b = np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]])
            if method == "orb":
                ka, da = O.orb_detect_describe(a); kb, db = O.orb_detect_describe(b)
            else:
                ka, da = O.surf_detect_describe(a); kb, db = O.surf_detect_describe(b)
            st, off, votes, nm = False, [0, 0], 0, 0
            if len(ka) and len(kb):
                pairs = O.bf_hamming_matches(da, db)[0] if method == "orb" else O.bf_l2_ratio_matches(da, db, 0.75)
                nm = len(pairs)
                st, off, votes = O.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, offsetEvaluate)
            log.append([d, i, int(st), int(off[0]), int(off[1]), int(votes), len(ka), len(kb), nm])
            if st:
                off = list(off); H, W = A.shape; Hb, Wb = B.shape
                if d == 1: off[0] += H - int(i * roiRatio * H)
                elif d == 2: off[1] += W - int(i * roiRatio * W)
                elif d == 3: off[0] -= Hb - int(i * roiRatio * Hb)
                elif d == 4: off[1] -= Wb - int(i * roiRatio * Wb)
                return True, [int(off[0]), int(off[1])], d, i, log
            d = rot(d)
            if d == d0:
                break
    return False, [0, 0], d0, 0, log


def cap_real_path():
    """The widest pin to the reference this container allows (cv2 is absent, the demo JPEGs and Stitcher.py:87 are present):

    1. dendritic_path_oracle.json -- the oracle (SURF + BF-L2 + ratio + mode vote behind the reference's incremental search,
       direction threaded from pair to pair exactly as Stitcher.py:252,361 does) run over the WHOLE dendriticCrystal shooting
       path on the real 1936 x 2584 tiles, tiles 003..090 (001-002 touch the missing blob): every pair's offset, accepted
       (direction, i), votes and the full attempt log, beside the reference's own value from Stitcher.py:87.
    2. real_path_strips.npz / .json -- around each of the five serpentine turns (tiles t-2 .. t+3), the ROI strips the
       accepted attempts read, cropped (640 columns for the row strips, 640 rows for the column strips) and stored with
       their position in the frame.  The parity tests rebuild 1936 x 2584 frames (zeros elsewhere) and register each
       neighbourhood through the grid registrar; expected rows = the oracle on those same rebuilt frames (stored), each within
       +-1 px of Stitcher.py:87 and with the (direction, i) of the full-tile run.  Candidates tried before the accepted one
       see blank strips in the rebuilt frames; item 1 records that they fail on the real tiles too."""
    from PIL import Image
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from imagestitch_amd.utility import roi_rect
    O.build()
    src = open(os.path.join(refshim.REF, "Stitcher.py"), encoding="utf-8-sig").read().splitlines()[86]
    gold = ast.literal_eval(src[src.index("["):])
    d = os.path.join(refshim.REF, "demoImages", "dendriticCrystal", "1")
    cache = {}

    def load(t):
        if t not in cache:
            im = Image.open(os.path.join(d, "1-%03d.jpg" % t)); im.draft("L", im.size)
            cache[t] = np.asarray(im.convert("L"))
            for old in [k for k in cache if k < t - 1]:
                del cache[old]
        return cache[t]

    # ---- 1. whole path on the real tiles
    rows, direction, d_in = [], 1, {}
    for a in range(3, 90):
        d_in[a] = direction
        st, off, dn, i, log = _oracle_search(O, roi_rect, load(a), load(a + 1), direction)
        g = gold[a - 1]
        assert st and abs(off[0] - g[0]) <= 1 and abs(off[1] - g[1]) <= 1, (a, off, g)
        rows.append(dict(a=a, b=a + 1, gold=g, oracle=off, direction=dn, i=i, votes=log[-1][5], incoming_direction=direction, attempts=log))
        direction = dn
        print("path", a, a + 1, g, off, dn, i, log[-1][5], flush=True)
    exact = sum(r["gold"] == r["oracle"] for r in rows)
    json.dump(dict(source="oracle (oracle/vfsms_oracle.c) on the reference's demoImages/dendriticCrystal/1 tiles 003..090, decoded with Pillow "
                          "(draft L); gold = Stitcher.py:87; attempts = [direction, i, status, raw dx, raw dy, votes, nA, nB, matches]",
                   pairs=len(rows), exact=exact, within_one=len(rows), rows=rows),
              open(os.path.join(OUT, "dendritic_path_oracle.json"), "w"))
    print("real path: %d pairs, %d exact, all within 1 px" % (len(rows), exact))

    # ---- 2. cropped strips around the turns
    by_a = {r["a"]: r for r in rows}
    CROP = 640
    store, meta = {}, []
    for turn in (15, 30, 45, 60, 75):
        tiles = list(range(turn - 2, turn + 4))
        full = {}
        for t in tiles:
            im = Image.open(os.path.join(d, "1-%03d.jpg" % t)); im.draft("L", im.size)
            full[t] = np.asarray(im.convert("L"))
        H, W = full[tiles[0]].shape
        frames = {t: np.zeros((H, W), np.uint8) for t in tiles}
        strips = []
        for a in tiles[:-1]:
            r = by_a[a]
            dd, ii = r["direction"], r["i"]
            assert ii == 1
            for t, order in ((a, "first"), (a + 1, "second")):
                y0, x0, h, w = roi_rect((H, W), dd, order, 0.2)
                if dd in (1, 3):
                    x0 += (w - CROP) // 2; w = CROP
                else:
                    y0 += (h - CROP) // 2; h = CROP
                img = full[t]
                frames[t][y0:y0 + h, x0:x0 + w] = img[y0:y0 + h, x0:x0 + w]
                key = "n%d_t%d_%d" % (turn, t, len([s for s in strips if s["tile"] == t]))
                store[key] = np.ascontiguousarray(img[y0:y0 + h, x0:x0 + w])
                strips.append(dict(tile=t, key=key, y0=y0, x0=x0))
        # expected rows: the oracle on the rebuilt frames, direction threaded
        direction = d_in[tiles[0]]
        exp = []
        for a in tiles[:-1]:
            st, off, dn, i, log = _oracle_search(O, roi_rect, frames[a], frames[a + 1], direction)
            g = gold[a - 1]
            assert st and abs(off[0] - g[0]) <= 1 and abs(off[1] - g[1]) <= 1 and (dn, i) == (by_a[a]["direction"], by_a[a]["i"]), (a, off, g, dn, i)
            exp.append(dict(a=a, gold=g, offset=off, direction=dn, i=i, votes=log[-1][5], nA=log[-1][6], nB=log[-1][7], matches=log[-1][8]))
            direction = dn
            print("nbhd", turn, a, g, off, dn, i, log[-1][5:], flush=True)
        meta.append(dict(turn=turn, tiles=tiles, shape=[H, W], incoming_direction=d_in[tiles[0]], strips=strips, expected=exp))
    np.savez_compressed(os.path.join(OUT, "real_path_strips.npz"), **store)
    json.dump(dict(source="crops of the reference's dendriticCrystal tiles (Pillow draft-L decode); expected = oracle on the rebuilt frames, "
                          "gold = Stitcher.py:87", neighbourhoods=meta), open(os.path.join(OUT, "real_path_strips.json"), "w"))
    print("real path strips", os.path.getsize(os.path.join(OUT, "real_path_strips.npz")) // 1024, "KiB")


def cap_real_path_orb():
    """The ORB leg against the only vector the reference holds.  Stitcher.py:87 lists TRUE offsets of the dendriticCrystal path (it is
    not a list of SURF outputs), so ORB + BF-Hamming + mode vote behind the same incremental search must land on it too wherever it
    accepts the right direction:

    1. dendritic_path_oracle_orb.json -- the oracle's ORB (cv2.ORB_create(5000, 1.2, 8, 31, 0, 2, 0, 31, 20), ImageUtility.py:31-39,260)
       + Hamming 1-NN (no ratio, no threshold: :297-302) + mode vote (offsetEvaluate 3) over tiles 003..090 on the real 1936 x 2584
       tiles, direction threaded as Stitcher.py:361 does.  Nothing is asserted while capturing: every pair's outcome is recorded,
       with `within_one` and, for the others, what happened (a falsely accepted direction is the reference's own fragility: three
       equal votes out of ~5000 unconditional matches).
    2. real_path_strips.json gains `expected_orb` per neighbourhood: the oracle's ORB rows on the rebuilt frames of the 25 committed
       pairs (tests compare the HIP path with them row for row, and with Stitcher.py:87 within +-1 px)."""
    from PIL import Image
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from imagestitch_amd.utility import roi_rect
    O.build()
    src = open(os.path.join(refshim.REF, "Stitcher.py"), encoding="utf-8-sig").read().splitlines()[86]
    gold = ast.literal_eval(src[src.index("["):])
    d = os.path.join(refshim.REF, "demoImages", "dendriticCrystal", "1")

    def load(t):
        im = Image.open(os.path.join(d, "1-%03d.jpg" % t)); im.draft("L", im.size)
        return np.asarray(im.convert("L"))

    surf_rows = {r["a"]: r for r in json.load(open(os.path.join(OUT, "dendritic_path_oracle.json")))["rows"]}
    rows, direction = [], 1
    prev = None
    for a in range(3, 90):
        A = prev if prev is not None else load(a)
        B = load(a + 1)
        st, off, dn, i, log = _oracle_search(O, roi_rect, A, B, direction, method="orb")
        g = gold[a - 1]
        ok = bool(st and abs(off[0] - g[0]) <= 1 and abs(off[1] - g[1]) <= 1)
        true_dir = surf_rows[a]["direction"]
        note = "" if ok else ("not registered" if not st else
                              ("accepted direction %d (i = %d) with %d votes; the true direction is %d" % (dn, i, log[-1][5], true_dir)
                               if dn != true_dir else "true direction, offset off by more than 1 px"))
        rows.append(dict(a=a, b=a + 1, gold=g, oracle=off, status=bool(st), direction=dn, i=i, votes=log[-1][5], incoming_direction=direction,
                         within_one=ok, note=note, attempts=log))
        print("orb path", a, a + 1, g, off, st, dn, i, log[-1][5], "OK" if ok else note, flush=True)
        if st:
            direction = dn
        prev = B
    json.dump(dict(source="oracle ORB (oracle/vfsms_oracle_orb.c) + BF-Hamming 1-NN + mode vote (offsetEvaluate 3) on the reference's "
                          "demoImages/dendriticCrystal/1 tiles 003..090 (Pillow draft-L decode); gold = Stitcher.py:87; attempts = "
                          "[direction, i, status, raw dx, raw dy, votes, nA, nB, matches]",
                   pairs=len(rows), within_one=sum(r["within_one"] for r in rows), exact=sum(r["gold"] == r["oracle"] for r in rows),
                   rows=rows), open(os.path.join(OUT, "dendritic_path_oracle_orb.json"), "w"))
    print("orb real path: %d pairs, %d within 1 px, %d exact" % (len(rows), sum(r["within_one"] for r in rows), sum(r["gold"] == r["oracle"] for r in rows)))

    # ---- the 25 committed neighbourhood pairs, rebuilt frames
    meta = json.load(open(os.path.join(OUT, "real_path_strips.json")))
    z = np.load(os.path.join(OUT, "real_path_strips.npz"))
    for nb in meta["neighbourhoods"]:
        H, W = nb["shape"]
        frames = {t: np.zeros((H, W), np.uint8) for t in nb["tiles"]}
        for s_ in nb["strips"]:
            arr = z[s_["key"]]
            frames[s_["tile"]][s_["y0"]:s_["y0"] + arr.shape[0], s_["x0"]:s_["x0"] + arr.shape[1]] = arr
        direction = nb["incoming_direction"]
        exp = []
        for a in nb["tiles"][:-1]:
            st, off, dn, i, log = _oracle_search(O, roi_rect, frames[a], frames[a + 1], direction, method="orb")
            g = gold[a - 1]
            exp.append(dict(a=a, gold=g, status=bool(st), offset=off, direction=dn, i=i, votes=log[-1][5], nA=log[-1][6], nB=log[-1][7], matches=log[-1][8],
                            within_one=bool(st and abs(off[0] - g[0]) <= 1 and abs(off[1] - g[1]) <= 1)))
            print("orb nbhd", nb["turn"], a, g, off, st, dn, i, log[-1][5:], flush=True)
            if st:
                direction = dn
        nb["expected_orb"] = exp
    json.dump(meta, open(os.path.join(OUT, "real_path_strips.json"), "w"))


def cap_real_full_strips():
    """real_full_strips.npz / .json -- configs[1] at its REAL load: the full-width ROI strips of two dendriticCrystal pairs, uncropped.  Pair
    4 -> 5 (a column pair: direction 1, 387 x 2584 strips, 13.3 k / 13.5 k SURF keypoints) and pair 15 -> 16 (a serpentine turn: direction 2,
    1936 x 516 strips, 13.5 k / 12.8 k).  Expected = the attempt rows of the oracle's path runs (dendritic_path_oracle.json /
    dendritic_path_oracle_orb.json: [status, raw dx, raw dy, votes, nA, nB, matches] of the accepted attempt), re-derived here from the
    strips alone and asserted equal; gold = Stitcher.py:87.  The 640-px crops of real_path_strips carry 2.6 k keypoints per strip: the
    BF / describe load of the real dataset goes through the HIP path only with these."""
    from PIL import Image
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from imagestitch_amd.utility import roi_rect
    O.build()
    d = os.path.join(refshim.REF, "demoImages", "dendriticCrystal", "1")

    def load(t):
        im = Image.open(os.path.join(d, "1-%03d.jpg" % t)); im.draft("L", im.size)
        return np.asarray(im.convert("L"))
    surf_rows = {r["a"]: r for r in json.load(open(os.path.join(OUT, "dendritic_path_oracle.json")))["rows"]}
    orb_rows = {r["a"]: r for r in json.load(open(os.path.join(OUT, "dendritic_path_oracle_orb.json")))["rows"]}
    store, meta = {}, []
    for a, dd in ((4, 1), (15, 2)):
        A, B = load(a), load(a + 1)
        ra = roi_rect(A.shape, dd, "first", 0.2); rb = roi_rect(B.shape, dd, "second", 0.2)
        sa = np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]); sb = np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]])
        store["p%d_a" % a] = sa; store["p%d_b" % a] = sb
        exp = {}
        for method, rows in (("surf", surf_rows), ("orb", orb_rows)):
            want = [att for att in rows[a]["attempts"] if att[0] == dd and att[1] == 1][0]
            if method == "orb":
                ka, da = O.orb_detect_describe(sa); kb, db = O.orb_detect_describe(sb)
                pairs = O.bf_hamming_matches(da, db)[0]
            else:
                ka, da = O.surf_detect_describe(sa); kb, db = O.surf_detect_describe(sb)
                pairs = O.bf_l2_ratio_matches(da, db, 0.75)
            st, off, votes = O.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
            row = [int(st), int(off[0]), int(off[1]), int(votes), len(ka), len(kb), len(pairs)]
            assert row == want[2:], (a, method, row, want)
            exp[method] = row
            print("full strips", a, dd, method, row, flush=True)
        meta.append(dict(a=a, b=a + 1, direction=dd, i=1, tile_shape=list(A.shape), roi_first=list(map(int, ra)), roi_second=list(map(int, rb)),
                         gold=surf_rows[a]["gold"], expected_surf=exp["surf"], expected_orb=exp["orb"]))
    np.savez_compressed(os.path.join(OUT, "real_full_strips.npz"), **store)
    json.dump(dict(source="full-width ROI strips (roiRatio 0.2, accepted direction) of demoImages/dendriticCrystal/1 pairs 004-005 and 015-016, Pillow "
                          "draft-L decode; expected_* = [status, raw dx, raw dy, votes, nA, nB, matches] of the oracle on these strips == the rows of "
                          "dendritic_path_oracle*.json; gold = Stitcher.py:87 (full offset incl. the axis correction)", pairs=meta),
              open(os.path.join(OUT, "real_full_strips.json"), "w"), indent=1)
    print("real full strips", os.path.getsize(os.path.join(OUT, "real_full_strips.npz")) // 1024, "KiB")


def cap_demo_strips():
    """BASELINE configs[0] / configs[3]: ROI strips of the iron pair (direction 1) and of the first zirconCL pairs (direction 4)
    at roiRatio 0.2.  cv2 is not installable here, so the expected offsets are produced by the oracle (oracle/): these
    fixtures pin the HIP path to the oracle on real micrographs and guard the oracle against regressions; they are NOT
    reference-generated values."""
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    O.build()

    def load(path):
        im = Image.open(path); im.draft("L", im.size)
        return np.asarray(im.convert("L"))
    m = RU.Method()
    store = {}
    meta = []
    cases = [("iron", "1.jpg", "2.jpg", 1)]
    z = sorted(os.listdir(os.path.join(refshim.REF, "demoImages", "zirconCL", "1")))
    cases += [("zirconCL", z[k], z[k + 1], 4) for k in range(3)]
    for n, (ds, fa, fb, direction) in enumerate(cases):
        A = load(os.path.join(refshim.REF, "demoImages", ds, "1", fa)); B = load(os.path.join(refshim.REF, "demoImages", ds, "1", fb))
        ra = np.ascontiguousarray(m.getROIRegionForIncreMethod(A, direction=direction, order="first", searchRatio=0.2))
        rb = np.ascontiguousarray(m.getROIRegionForIncreMethod(B, direction=direction, order="second", searchRatio=0.2))
        (px, py), resp = O.phase_correlate(ra, rb)
        ka, da = O.surf_detect_describe(ra); kb, db = O.surf_detect_describe(rb)
        pairs = O.bf_l2_ratio_matches(da, db, 0.75)
        st, off, votes = O.mode_offset(np.stack([ka["x"], ka["y"]], 1), np.stack([kb["x"], kb["y"]], 1), pairs, 3)
        store["d%d_roiA" % n] = ra; store["d%d_roiB" % n] = rb
        meta.append(dict(dataset=ds, a=fa, b=fb, direction=direction, shape=list(A.shape), roi=list(ra.shape),
                         phase_xy=[px, py], phase_response=resp, phase_int=[int(py), int(px)],
                         surf=dict(status=int(st), offset=[int(off[0]), int(off[1])], votes=int(votes), nA=len(ka), nB=len(kb), matches=len(pairs))))
        print(ds, fa, fb, ra.shape, "phase", (px, py), resp, "surf", st, off, votes, len(ka), len(kb), len(pairs))
    np.savez_compressed(os.path.join(OUT, "demo_strips.npz"), **store)
    json.dump(dict(source="oracle-generated (cv2 not installable): regression pin of the oracle + parity target of the HIP path",
                   cases=meta), open(os.path.join(OUT, "demo_strips.json"), "w"), indent=1)
    print("demo strips", os.path.getsize(os.path.join(OUT, "demo_strips.npz")) // 1024, "KiB")


def cap_phase_independent():
    """BASELINE configs[3] in full, and an independent check of the phase-correlation oracle: the direction-4 ROI strips (roiRatio 0.2)
    of ALL 24 zirconCL tiles (23 pairs; demo_strips holds 3) + for each pair and for the iron strip pair the result of tests/phase_numpy.py
    -- a second float64 restatement of cv2.phaseCorrelate (numpy rfft2 / irfft2) that shares no code with the C oracle -- next to the
    oracle's.  Both are restatements (cv2 is not installable here); what this pins is that the oracle's per-bin rules were not
    mis-transcribed in a way a second transcription would not repeat."""
    from PIL import Image
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    import phase_numpy as PN
    from imagestitch_amd.utility import roi_rect
    O.build()

    def load(path):
        im = Image.open(path); im.draft("L", im.size)
        return np.asarray(im.convert("L"))
    zdir = os.path.join(refshim.REF, "demoImages", "zirconCL", "1")
    z = sorted(os.listdir(zdir))
    store, rows = {}, []
    tiles = [load(os.path.join(zdir, f)) for f in z]
    H, W = tiles[0].shape
    ra = roi_rect((H, W), 4, "first", 0.2); rb = roi_rect((H, W), 4, "second", 0.2)
    for k, T in enumerate(tiles):
        store["t%d_first" % k] = np.ascontiguousarray(T[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]])
        store["t%d_second" % k] = np.ascontiguousarray(T[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]])
    cases = [("zirconCL", z[k], z[k + 1], store["t%d_first" % k], store["t%d_second" % (k + 1)]) for k in range(len(z) - 1)]
    iron = np.load(os.path.join(OUT, "demo_strips.npz"))
    cases.append(("iron", "1.jpg", "2.jpg", iron["d0_roiA"], iron["d0_roiB"]))
    for ds, fa, fb, a, b in cases:
        (ox, oy), orr = O.phase_correlate(a, b)
        (nx, ny), nr, pk = PN.phase_correlate(a, b)
        assert int(ox) == int(nx) and int(oy) == int(ny) and abs(ox - nx) < 1e-9 and abs(oy - ny) < 1e-9 and abs(orr - nr) < 1e-12, (ds, fa, ox, nx, oy, ny, orr, nr)
        rows.append(dict(dataset=ds, a=fa, b=fb, roi=list(a.shape), numpy_xy=[float(nx), float(ny)], numpy_response=float(nr), peak=list(pk),
                         oracle_xy=[ox, oy], oracle_response=orr, offset_int=[int(oy), int(ox)], accepted=bool(orr > 0.15)))
        print(ds, fa, fb, a.shape, (nx, ny, nr), "|d|", abs(ox - nx), abs(oy - ny), abs(orr - nr))
    np.savez_compressed(os.path.join(OUT, "zirconcl_strips.npz"), **store)
    json.dump(dict(source="direction-4 ROI strips (roiRatio 0.2) of the reference's demoImages/zirconCL/1 tiles (Pillow draft-L decode), sorted by name; "
                          "numpy_* = tests/phase_numpy.py, oracle_* = oracle/vfsms_oracle.c at capture time; offset_int = [int(y), int(x)] as "
                          "Stitcher.py:231-232 truncates; accepted = response > phaseResponseThreshold 0.15 (Stitcher.py:30, 235)",
                   roi_first=list(ra), roi_second=list(rb), shape=[H, W], tiles=z, rows=rows),
              open(os.path.join(OUT, "phase_independent.json"), "w"), indent=1)
    print("zirconCL strips", os.path.getsize(os.path.join(OUT, "zirconcl_strips.npz")) // 1024, "KiB; pairs", len(rows))


def _opt_dft(n):
    m = max(int(n), 1)
    while True:
        k = m
        for q in (2, 3, 5):
            while k % q == 0:
                k //= q
        if k == 1:
            return m
        m += 1


def cap_phase87():
    """The phase leg against the reference-held vector (Stitcher.py:87 lists the TRUE offsets of the dendriticCrystal path): for every pair,
    the oracle's cv2.phaseCorrelate restatement on the two ROI strips of the ACCEPTED (direction, i) of dendritic_path_oracle.json, as
    Stitcher.calculateOffsetForPhaseCorrleateIncre slices them (Stitcher.py:224-229).  With the sign the feature path uses (phaseSignFix:
    offset = -[int(y), int(x)] + axis correction, Stitcher.py:244-251) the result must be the gold offset -- modulo the padded strip
    (the DFT is circular: a strip shift s and s - M are the same peak), wherever the correlation peak is the overlap's (response gate).
    Two records per pair: the FULL strips (387 x 2584 / 1936 x 516; rows only -- the images are the reference's) and, for the 25 pairs
    of real_path_strips.npz, the committed 640-px crops (what the CPU and GPU parity tests run)."""
    from PIL import Image
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from imagestitch_amd.utility import roi_rect
    O.build()
    path = json.load(open(os.path.join(OUT, "dendritic_path_oracle.json")))["rows"]
    d = os.path.join(refshim.REF, "demoImages", "dendriticCrystal", "1")

    def load(t):
        im = Image.open(os.path.join(d, "1-%03d.jpg" % t)); im.draft("L", im.size)
        return np.asarray(im.convert("L"))

    def record(a, b, H, W, dd, ii, gold):
        """phase-correlate strips a, b cut for (direction dd, growth ii) from H x W tiles -> row"""
        (x, y), resp = O.phase_correlate(a, b)
        corr = [0, 0]                                          # Stitcher.py:244-251 for the accepted direction
        if dd == 1: corr[0] = H - int(ii * 0.2 * H)
        elif dd == 2: corr[1] = W - int(ii * 0.2 * W)
        elif dd == 3: corr[0] = -(H - int(ii * 0.2 * H))
        elif dd == 4: corr[1] = -(W - int(ii * 0.2 * W))
        M, N = _opt_dft(a.shape[0]), _opt_dft(a.shape[1])
        fixed = [-int(y) + corr[0], -int(x) + corr[1]]        # phaseSignFix = True
        asref = [int(y) + corr[0], int(x) + corr[1]]          # the reference as written (mirrored)
        res = [(fixed[0] - gold[0] + M // 2) % M - M // 2, (fixed[1] - gold[1] + N // 2) % N - N // 2]
        return dict(direction=dd, i=ii, gold=gold, roi=list(a.shape), padded=[M, N], phase_xy=[x, y], response=resp, axis_correction=corr,
                    offset_sign_fixed=fixed, offset_as_written=asref, residual_mod_padded=res,
                    in_strip_shift=[gold[0] - corr[0], gold[1] - corr[1]], accepted=bool(resp > 0.15))
    rows = []
    tile = {}
    for r in path:
        a_, b_ = r["a"], r["b"]
        for t in (a_, b_):
            if t not in tile:
                tile[t] = load(t)
        for old in [k for k in tile if k < a_]:
            del tile[old]
        A, B = tile[a_], tile[b_]
        H, W = A.shape
        ra = roi_rect((H, W), r["direction"], "first", r["i"] * 0.2); rb = roi_rect((H, W), r["direction"], "second", r["i"] * 0.2)
        sa = np.ascontiguousarray(A[ra[0]:ra[0] + ra[2], ra[1]:ra[1] + ra[3]]); sb = np.ascontiguousarray(B[rb[0]:rb[0] + rb[2], rb[1]:rb[1] + rb[3]])
        row = record(sa, sb, H, W, r["direction"], r["i"], r["gold"])
        row.update(a=a_, b=b_)
        rows.append(row)
        print("phase87 full", a_, r["direction"], r["gold"], row["offset_sign_fixed"], row["residual_mod_padded"], "%.3f" % row["response"], flush=True)
    # the committed crops
    meta = json.load(open(os.path.join(OUT, "real_path_strips.json")))["neighbourhoods"]
    z = np.load(os.path.join(OUT, "real_path_strips.npz"))
    crops = []
    for nb in meta:
        H, W = nb["shape"]
        by_tile = {}
        for s_ in nb["strips"]:
            by_tile.setdefault(s_["tile"], []).append(s_)
        for e in nb["expected"]:
            a_ = e["a"]
            # the strip of tile a cut as "first" and of tile a + 1 cut as "second" for the accepted direction: the crop whose origin matches
            ra = roi_rect((H, W), e["direction"], "first", e["i"] * 0.2); rb = roi_rect((H, W), e["direction"], "second", e["i"] * 0.2)

            def pick(t, rect):
                for s_ in by_tile[t]:
                    arr = z[s_["key"]]
                    if rect[0] <= s_["y0"] and s_["y0"] + arr.shape[0] <= rect[0] + rect[2] and rect[1] <= s_["x0"] and s_["x0"] + arr.shape[1] <= rect[1] + rect[3]:
                        return s_["key"], arr
                raise KeyError((t, rect))
            ka, sa = pick(a_, ra); kb, sb = pick(a_ + 1, rb)
            assert sa.shape == sb.shape
            row = record(np.ascontiguousarray(sa), np.ascontiguousarray(sb), H, W, e["direction"], e["i"], e["gold"])
            row.update(a=a_, b=a_ + 1, key_a=ka, key_b=kb)
            crops.append(row)
            print("phase87 crop", a_, e["direction"], e["gold"], row["offset_sign_fixed"], row["residual_mod_padded"], "%.3f" % row["response"], flush=True)
    ok_full = [r for r in rows if r["accepted"] and max(abs(v) for v in r["residual_mod_padded"]) <= 1]
    ok_crop = [r for r in crops if r["accepted"] and max(abs(v) for v in r["residual_mod_padded"]) <= 1]
    json.dump(dict(source="oracle phase correlation (oracle/vfsms_oracle.c) of the ROI strips of the accepted (direction, i) of every dendriticCrystal pair "
                          "(tiles 003..090, Pillow draft-L decode) against gold = Stitcher.py:87; offset_sign_fixed = -[int(y), int(x)] + axis correction "
                          "(Stitcher.phaseSignFix), residual_mod_padded = (offset_sign_fixed - gold) wrapped to the padded strip size; "
                          "full = whole strips (rows only), crops = the committed 640-px crops of real_path_strips.npz",
                   full=rows, crops=crops, full_accepted=sum(r["accepted"] for r in rows), full_within_one=len(ok_full),
                   crops_accepted=sum(r["accepted"] for r in crops), crops_within_one=len(ok_crop)),
              open(os.path.join(OUT, "dendritic_phase87.json"), "w"), indent=1)
    print("phase87: full %d pairs, %d accepted, %d within 1 px mod padded; crops %d, %d accepted, %d within" %
          (len(rows), sum(r["accepted"] for r in rows), len(ok_full), len(crops), sum(r["accepted"] for r in crops), len(ok_crop)))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["roi", "mode", "sm", "cache", "fuse", "stitch", "flow", "real", "realpath", "demo"]
    fns = dict(roi=cap_roi, mode=cap_mode, sm=cap_state_machine, cache=cap_feature_search_cache,
               fuse=cap_fuse, stitch=cap_stitch, flow=cap_flow, real=cap_real, realpath=cap_real_path, realpath_orb=cap_real_path_orb,
               demo=cap_demo_strips, phase2=cap_phase_independent, phase87=cap_phase87, realfull=cap_real_full_strips)
    for w in which:
        fns[w]()

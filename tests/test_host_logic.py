"""CPU tests of the host-side mirror (imagestitch_amd.{utility,stitcher,fusion}) against golden vectors
captured from the reference's own Python (tools/capture_golden.py).  Operators are scripted fakes or the
oracle-backed engine of tests/fakes.py: this file checks control flow, ROI / layout arithmetic, dispatch and
return conventions, not kernels."""
import json
import os

import numpy as np
import pytest

import imagestitch_amd as isa
from imagestitch_amd import stitcher as st_mod
from fakes import OracleEngine, IngestOracleEngine


def test_roi_rect_matches_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "roi_cases.json")))
    m = isa.Method()
    for c in cases:
        y0, x0, h, w = isa.roi_rect(c["shape"], c["direction"], c["order"], c["ratio"])
        assert [h, w] == c["out_shape"] and (y0, x0) == (c["row0"], c["col0"]), c
    img = np.arange(97 * 131, dtype=np.uint32).astype(np.uint8).reshape(97, 131)
    v = m.getROIRegionForIncreMethod(img, direction=2, order="first", searchRatio=0.2)
    assert np.shares_memory(v, img) and v.shape == (97, 26) and v[0, 0] == img[0, 131 - 26]


class Scripted(isa.Stitcher):
    """operators replaced by scripted fakes, exactly like the capture harness did to the reference"""

    def __init__(self, success_at, raw):
        self.trace, self.pending = [], []
        self.success_at, self.raw = success_at, raw

    def detectAndDescribe(self, image, featureMethod):
        self.pending.append(list(image.shape))
        return (np.zeros((1, 2), np.float32), np.zeros((1, 64), np.float32))

    def matchDescriptors(self, fa, fb):
        return [(0, 0)]

    def getOffsetByMode(self, kpsA, kpsB, matches, offsetEvaluate=10):
        self.trace.append(self.pending[-2:])
        return (len(self.trace) == self.success_at, list(self.raw))


class ScriptedPhase(isa.Stitcher):
    def __init__(self, success_at, value):
        self.calls, self.success_at, self.value = [], success_at, value

    def _phaseCorrelate(self, a, b):
        self.calls.append([list(a.shape), list(b.shape)])
        return self.value, (0.5 if len(self.calls) == self.success_at else 0.1)


def test_incremental_state_machines_match_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "state_machine.json")))
    assert len(cases) > 1000
    imgs = {}
    for c in cases:
        shape = tuple(c["shape"])
        if shape not in imgs:
            imgs[shape] = (np.zeros(shape, np.uint8), np.zeros(shape, np.uint8))
        A, B = imgs[shape]
        if c["kind"] == "feature":
            s = Scripted(c["success_at"], c["raw"])
        else:
            s = ScriptedPhase(c["success_at"], tuple(c["raw"]))
        s.isPrintLog = False
        s.roiRatio, s.direction, s.directIncre = c["roiRatio"], c["ini"], c["incre"]
        if c["kind"] == "feature":
            status, off = s.calculateOffsetForFeatureSearchIncre([A, B])
            trace = s.trace
        else:
            status, off = s.calculateOffsetForPhaseCorrleateIncre([A, B])
            trace = s.calls
        assert status == c["status"], c
        assert (list(off) if status else off) == c["offset"], c
        assert trace == c["trace"], c
        assert s.direction == c["final_direction"], c


def test_feature_search_cache_matches_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "feature_search_cache.json")))

    class S(isa.Stitcher):
        def __init__(self, script):
            self.script, self.described, self.k = list(script), [], 0

        def detectAndDescribe(self, image, featureMethod):
            self.described.append(int(image[0, 0]))
            return (np.full((1, 2), image[0, 0], np.float32), np.full((1, 64), image[0, 0], np.float32))

        def matchDescriptors(self, fa, fb):
            self.matched = (int(fa[0, 0]), int(fb[0, 0]))
            return [(0, 0)]

        def getOffsetByMode(self, kpsA, kpsB, matches, offsetEvaluate=10):
            ok = self.script[self.k]; self.k += 1
            return (ok, [5, -6])
    for case in cases:
        isa.Stitcher.tempImageFeature.isBreak = True
        s = S(case["script"]); s.isPrintLog = False
        for k, step in enumerate(case["steps"]):
            A = np.full((8, 8), 10 + k, np.uint8); B = np.full((8, 8), 11 + k, np.uint8)
            n0 = len(s.described)
            status, off = s.calculateOffsetForFeatureSearch([A, B])
            assert s.described[n0:] == step["described"] and list(s.matched) == step["matched"]
            assert status == step["status"] and (list(off) if status else off) == step["offset"]
            assert bool(s.tempImageFeature.isBreak) == step["isBreak"]
    isa.Stitcher.tempImageFeature.isBreak = True


FUSE_NAMES = ["notFuse", "average", "maximum", "minimum", "fadeInAndFadeOut", "trigonometric"]


def _write_tiles(tmp_path, tiles, tag):
    from PIL import Image
    files = []
    for k, t in enumerate(tiles):
        p = os.path.join(str(tmp_path), "%s_%d.png" % (tag, k))
        Image.fromarray(t[:, :, ::-1] if t.ndim == 3 else t).save(p)
        files.append(p)
    return files


def test_get_stitch_by_offset_matches_reference(golden_dir, oracle, tmp_path):
    g = np.load(os.path.join(golden_dir, "stitch_cases.npz"))
    eng = OracleEngine(oracle)
    for n, (color, fm, _) in enumerate(g["meta"]):
        tiles = list(g["s%d_tiles" % n])
        files = _write_tiles(tmp_path, tiles, "s%d" % n)
        s = isa.Stitcher()
        s._engine = eng
        msgs = []
        s.printAndWrite = lambda c, msgs=msgs: msgs.append(c)
        s.isColorMode = bool(color)
        isa.Stitcher.isColorMode = bool(color)
        s.fuseMethod = FUSE_NAMES[fm]
        offs = [list(map(int, o)) for o in g["s%d_offsets" % n]]
        res = s.getStitchByOffset(files, offs)
        assert offs[0] == [0, 0] and len(offs) == len(tiles)         # the caller's list is mutated, as in the reference
        rect = [m for m in msgs if "rectified" in m][0]
        assert rect == "  The rectified offsetList is " + str([list(map(int, r)) for r in g["s%d_rect" % n]])
        assert res.shape == g["s%d_out" % n].shape, (n, FUSE_NAMES[fm])
        assert np.array_equal(res, g["s%d_out" % n]), (n, FUSE_NAMES[fm], color)
    isa.Stitcher.isColorMode = True


def test_flow_stitch_with_multiple_matches_reference(golden_dir, oracle, tmp_path):
    meta = json.load(open(os.path.join(golden_dir, "flow_cases.json")))
    g = np.load(os.path.join(golden_dir, "flow_cases.npz"))
    eng = OracleEngine(oracle)
    for n, case in enumerate(meta):
        script = case["script"]
        tiles = [np.full((20, 24), 10 * (k + 1), np.uint8) for k in range(len(script) + 1)]
        files = _write_tiles(tmp_path, tiles, "w%d" % n)
        s = isa.Stitcher(); s._engine = eng
        msgs = []
        s.printAndWrite = lambda c, msgs=msgs: msgs.append(c)
        s.isColorMode = False; isa.Stitcher.isColorMode = False
        s.fuseMethod = "notFuse"

        def method(images, script=script):
            a = int(images[0][0, 0]) // 10 - 1
            return (True, [15, 3]) if script[a] else (False, st_mod.CANNOT_MATCH)
        res = s.flowStitchWithMutiple(files, method)
        assert len(res) == case["nres"]
        for k, r in enumerate(res):
            assert np.array_equal(r, g["w%d_res%d" % (n, k)]), (n, k)
        assert len([m for m in msgs if "stitching Break" in m or "can not be stitched" in m]) == case["breaks"]
    isa.Stitcher.isColorMode = True


def test_fusion_dispatch_and_inplace_fill(golden_dir, oracle):
    """ImageFusion.fuseByFadeInAndFadeOut fills A's holes from B in place (ImageFusion.py:240) and fuseImage
    pre-processes the non-fade modes (Stitcher.py:498-504)."""
    f = isa.ImageFusion(); f._engine = OracleEngine(oracle)
    A = np.array([[-1, 10, 20, 30]] * 6, np.int64); B = np.full((6, 4), 200, np.int64)
    out = f.fuseByFadeInAndFadeOut([A, B], 1, 1)
    assert A[0, 0] == 200 and out.dtype == np.uint8 and out.shape == (6, 4)
    s = isa.Stitcher(); s._engine = f._engine; s.isColorMode = False
    s.fuseMethod = "maximum"
    A = np.array([[-1, 0, 7]], np.int64); B = np.array([[5, 6, 0]], np.int64)
    assert s.fuseImage([A, B], 0, 0).tolist() == [[5, 6, 7]]
    s.fuseMethod = "optimalSeamLine"
    with pytest.raises(NotImplementedError):
        s.fuseImage([A, B], 0, 0)


def test_dead_reference_paths_behave_like_the_reference():
    s = isa.Stitcher()
    with pytest.raises(AttributeError):
        s.calculateOffsetForPhaseCorrleate(["a", "b"])     # Stitcher.py:195 dereferences the undefined self.phase
    assert s.directionIncrease(4) == 1
    s.directIncre = -1
    assert s.directionIncrease(1) == 4
    assert s.getOffsetByMode([], [], []) == (False, [0, 0])


def test_flow_stitch_batched_registration_equals_pair_by_pair(oracle, tmp_path):
    """flowStitch routes the stock incremental searches through grid.GridRegistrar (all tiles on the device, speculative fused
    batches).  On a 2 x 3 serpentine of synthetic tiles the offsets, the threaded self.direction, the log lines and the mosaic
    must equal the pair-by-pair loop (batchRegistration = False), for the SURF and the phase-correlation search."""
    from imagestitch_amd.synthetic import SyntheticGrid

    class FusedOracleEngine(OracleEngine):
        """OracleEngine plus the fused-attempt entry points, evaluated operator by operator with the oracle."""
        def __init__(self, o):
            super().__init__(o); self.tiles = {}; self.batches = 0

        def tile_upload(self, img):
            self.tiles[len(self.tiles) + 1] = np.ascontiguousarray(img); return len(self.tiles)

        def tile_free(self, h):
            pass

        def set_keypoint_capacity(self, n):
            pass

        def _rois(self, job):
            ta, tb, ay0, ax0, by0, bx0, h, w = [int(v) for v in job]
            return (np.ascontiguousarray(self.tiles[ta][ay0:ay0 + h, ax0:ax0 + w]), np.ascontiguousarray(self.tiles[tb][by0:by0 + h, bx0:bx0 + w]))

        def attempt_surf_batch(self, jobs, params=None, ratio=0.75, offset_evaluate=3):
            self.batches += 1
            out = np.zeros((len(jobs), 8), np.int32)
            for n, job in enumerate(jobs):
                a, b = self._rois(job)
                ka, da = self.surf_detect_describe(a); kb, db = self.surf_detect_describe(b)
                if len(ka) == 0 or len(kb) == 0:
                    out[n] = [0, 0, 0, 0, len(ka), len(kb), 0, 0]; continue
                pairs = self.bf_l2_ratio_matches(da, db, ratio)
                st, off, votes = self.mode_offset(ka, kb, pairs, offset_evaluate) if len(pairs) else (False, [0, 0], 0)
                out[n] = [int(st), off[0], off[1], votes, len(ka), len(kb), len(pairs), 0]
            return out

        def attempt_phase_batch(self, jobs):
            self.batches += 1
            rows = []
            for job in jobs:
                a, b = self._rois(job)
                (x, y), r = self.phase_correlate(a, b)
                rows.append([x, y, r])
            return np.array(rows, np.float64)

    g = SyntheticGrid(2, 3, 256, overlap=0.2)
    files = _write_tiles(tmp_path, g.tiles(threads=1), "bf")
    old = (isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod)
    try:
        isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod = 1, 0.3, False, "surf"
        for which in ("calculateOffsetForFeatureSearchIncre", "calculateOffsetForPhaseCorrleateIncre"):
            runs = []
            for batched in (False, True):
                eng = FusedOracleEngine(oracle)
                s = isa.Stitcher(); s._engine = eng; s.batchRegistration = batched
                s.direction = 1; s.fuseMethod = "notFuse"
                msgs = []
                s.printAndWrite = lambda c, msgs=msgs: msgs.append(c)
                (status, mosaic) = s.flowStitch(list(files), getattr(s, which))
                runs.append((status, mosaic, s.direction, [m for m in msgs if "offset of stitching" in m or "stitching " in m], eng.batches))
            a, b = runs
            assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3], (which, a[0], b[0], a[3], b[3])
            assert np.array_equal(a[1], b[1])
            if which == "calculateOffsetForFeatureSearchIncre":
                assert a[0][0] is True and a[0][1] == 5          # all five pairs registered, through two turns
                assert b[4] < a[4]                                # fewer, larger fused batches than pair-by-pair attempts
    finally:
        isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod = old


def test_phase_sign_fix_is_opt_in(oracle, golden_dir):
    """The reference-as-written phase search mirrors the offset; phaseSignFix = True is an opt-in that is NOT the
    reference's behaviour.  On the iron pair: default [1400, 0] (what the reference would return), fixed [1698, 0] (the SURF
    search gives [1699, 0], the true offset is [1699, -1])."""
    c = json.load(open(os.path.join(golden_dir, "demo_strips.json")))["cases"][0]
    g = np.load(os.path.join(golden_dir, "demo_strips.npz"))
    H, W = c["shape"]
    A = np.zeros((H, W), np.uint8); B = np.zeros((H, W), np.uint8)
    A[H - g["d0_roiA"].shape[0]:, :] = g["d0_roiA"]; B[:g["d0_roiB"].shape[0], :] = g["d0_roiB"]
    old = (isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio)
    try:
        isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio = 1, 0, 0.2
        s = isa.Stitcher(); s._engine = OracleEngine(oracle); s.isPrintLog = False
        assert s.phaseSignFix is False
        assert s.calculateOffsetForPhaseCorrleateIncre([A, B]) == (True, [1400, 0])
        s.phaseSignFix = True
        assert s.calculateOffsetForPhaseCorrleateIncre([A, B]) == (True, [1698, 0])
    finally:
        isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio = old


def test_tile_cache_is_lru_and_never_evicts_the_current_job():
    """One anchor tile registered against many others through calculateOffsetForFeatureSearchIncre: the anchor's device handle
    must stay valid (refreshed on every hit, resolved together with its partner before anything is evicted), at most
    _TILE_CACHE tiles stay resident, and releaseTiles frees all of them."""
    import imagestitch_amd as isa

    class Eng:
        def __init__(self):
            self.live, self.next, self.uploads = set(), 1, 0
        def tile_upload(self, img):
            h = self.next; self.next += 1; self.live.add(h); self.uploads += 1
            return h
        def tile_free(self, h):
            assert h in self.live, "freed twice"
            self.live.remove(h)
        @staticmethod
        def surf_params(*a, **k):
            return None
        def attempt_surf_batch(self, jobs, params, ratio, ev):
            for j in jobs:
                assert j[0] in self.live and j[1] in self.live, "unknown tile handle"
            return np.array([[1, 5, -3, 9, 10, 10, 9, 0]] * len(jobs), np.int32)
    eng = Eng()
    st = isa.Stitcher(); st._engine = eng; st.isPrintLog = False
    st.direction = 1
    anchor = np.zeros((100, 120), np.uint8)
    others = [np.full((100, 120), k, np.uint8) for k in range(1, 9)]
    for o in others:
        assert st.calculateOffsetForFeatureSearchIncre([anchor, o])[0] is True
        assert len(eng.live) <= st._TILE_CACHE
    assert eng.uploads == 1 + len(others)                       # the anchor was uploaded once
    assert st.calculateOffsetForFeatureSearchIncre([others[0], anchor])[0] is True      # evicted long ago: uploaded again
    st.releaseTiles()
    assert not eng.live


def test_npy_band_writer_reassembles_the_mosaic(tmp_path):
    """Stitcher.mosaicSink protocol: (row0, band, full_shape) calls in row order -> one .npy equal to the whole image (gray and colour)."""
    import os
    import imagestitch_amd as isa
    rng = np.random.default_rng(9)
    for shape in ((53, 40), (31, 17, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        path = os.path.join(str(tmp_path), "sub", "m%d.npy" % len(shape))
        sink = isa.NpyBandWriter(path)
        for r0 in range(0, shape[0], 7):
            sink(r0, img[r0:r0 + 7], shape)
        assert np.array_equal(np.load(path), img)


def test_flow_stitch_releases_tiles_when_an_operator_raises(tmp_path):
    """A pair method that raises in the middle of a path must not leave the tiles the earlier pairs cached in HBM
    (flowStitch frees them on the way out, as it does at its normal end)."""
    import imagestitch_amd as isa

    class Eng:
        def __init__(self):
            self.live, self.next, self.calls = set(), 1, 0
        def tile_upload(self, img):
            h = self.next; self.next += 1; self.live.add(h)
            return h
        def tile_free(self, h):
            assert h in self.live, "freed twice"
            self.live.remove(h)
        @staticmethod
        def surf_params(*a, **k):
            return None
        def attempt_surf_batch(self, jobs, params, ratio, ev):
            self.calls += 1
            if self.calls == 2:
                raise RuntimeError("device error in the second pair")
            return np.array([[1, 5, -3, 9, 10, 10, 9, 0]] * len(jobs), np.int32)
    files = _write_tiles(tmp_path, [np.full((64, 80), 40 * k, np.uint8) for k in range(1, 4)], "raise")
    eng = Eng()
    st = isa.Stitcher(); st._engine = eng; st.isPrintLog = False
    st.batchRegistration = False                                 # the pair-by-pair loop is the one under test
    st.direction = 1
    with pytest.raises(RuntimeError):
        st.flowStitch(files, st.calculateOffsetForFeatureSearchIncre)
    assert eng.calls == 2 and not eng.live


# ---- ingest pipeline: one decode per file, both planes; error paths (CPU, IngestOracleEngine) ---------------------------------------------
def _colour_tiles(g):
    """colour versions of a synthetic grid's tiles: three differently weighted planes, so that Cb / Cr are not flat"""
    out = []
    for t in g.tiles(threads=1):
        f = t.astype(np.float32)
        out.append(np.clip(np.stack([0.6 * f + 30, f, 255 - 0.7 * f], -1), 0, 255).astype(np.uint8))
    return out


def _write_jpegs(tmp_path, tiles, tag, quality=92):
    from PIL import Image
    files = []
    for k, t in enumerate(tiles):
        p = os.path.join(str(tmp_path), "%s_%02d.jpg" % (tag, k))
        Image.fromarray(t).save(p, quality=quality)           # (RGB order in the file; the engine's tiles are B G R like cv2's)
        files.append(p)
    return files


def test_decode_once_planes_equal_the_two_decodes(tmp_path):
    """The one decode of the ingest pipeline against the two the reference makes (Stitcher.py:68-69, 382-403): the Y plane of the JPEG's
    YCbCr decode IS its grayscale decode, and jdcolor's fixed-point conversion of the same planes (stitcher._ycc_to_bgr, the host twin
    of csrc/ingest_kernels.hip) IS its colour decode, byte for byte -- on 4:2:0 and 4:4:4 files, odd sizes, and a grayscale file."""
    from PIL import Image
    from imagestitch_amd import stitcher as ST
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((211, 149), Image.BICUBIC))
    for k, kw in enumerate((dict(quality=90), dict(quality=95, subsampling=0), dict(quality=60, subsampling=2))):
        p = os.path.join(str(tmp_path), "c%d.jpg" % k)
        Image.fromarray(img).save(p, **kw)
        owner, shape, parts = ST._decode_once(p, True)
        assert parts[0] == "src" and parts[3] in (1, 2) and shape == (149, 211)
        spx = 3 if parts[3] == 1 else 4
        import ctypes
        buf = np.frombuffer((ctypes.c_uint8 * (shape[0] * parts[2])).from_address(parts[1]), np.uint8).reshape(shape[0], parts[2])
        ycc = buf[:, :shape[1] * spx].reshape(shape[0], shape[1], spx)[:, :, :3]
        assert np.array_equal(ycc[:, :, 0], ST._imread(p, False))
        assert np.array_equal(ST._ycc_to_bgr(ycc), ST._imread(p, True))
        del owner
    p = os.path.join(str(tmp_path), "g.jpg")
    Image.fromarray(img[:, :, 1]).save(p, quality=90)
    owner, shape, parts = ST._decode_once(p, True)
    assert parts[0] == "src" and parts[3] == 0
    p = os.path.join(str(tmp_path), "c.png")
    Image.fromarray(img).save(p)
    owner, shape, parts = ST._decode_once(p, True)
    assert parts[0] == "arrays" and np.array_equal(parts[1], ST._imread(p, False)) and np.array_equal(parts[2], ST._imread(p, True))


def test_colour_mosaic_decodes_every_file_exactly_once(oracle, tmp_path):
    """Main.py:14's default isColorMode = True through flowStitch: the ingest pipeline decodes each file ONCE (counted) for the
    registration plane and the B G R mosaic tile, no file is read again for the mosaic, nothing stays in HBM, and the mosaic equals the
    pair-by-pair run that decodes gray and colour separately like the reference (Stitcher.py:68-69, 382-403)."""
    from imagestitch_amd.synthetic import SyntheticGrid
    from imagestitch_amd import stitcher as ST
    g = SyntheticGrid(2, 2, 256, overlap=0.25)
    files = _write_jpegs(tmp_path, _colour_tiles(g), "col")
    old = (isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod)
    counts = {"once": 0, "imread": 0}
    real_once, real_imread = ST._decode_once, ST._imread

    def once(path, color):
        counts["once"] += 1
        return real_once(path, color)

    def imread(path, color):
        counts["imread"] += 1
        return real_imread(path, color)

    class HostOnly(IngestOracleEngine):
        """no ingest entry points: the Stitcher decodes gray for the pairs and colour for the mosaic, like the reference"""
        @property
        def tile_reserve(self):
            raise AttributeError("tile_reserve")
    try:
        isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod = 1, 0.3, True, "surf"
        runs = []
        for batched in (True, False):
            for fuse in ("fadeInAndFadeOut", "notFuse"):
                eng = IngestOracleEngine(oracle) if batched else HostOnly(oracle)
                s = isa.Stitcher(); s._engine = eng; s.batchRegistration = batched; s.isPrintLog = False
                s.direction = 1; s.fuseMethod = fuse
                counts["once"] = counts["imread"] = 0
                ST._decode_once, ST._imread = once, imread
                try:
                    (status, mosaic) = s.flowStitch(list(files), s.calculateOffsetForFeatureSearchIncre)
                finally:
                    ST._decode_once, ST._imread = real_once, real_imread
                assert status == (True, 3) and mosaic.ndim == 3
                if batched:
                    assert counts == {"once": len(files), "imread": 0}, counts
                assert not eng.live, eng.live                          # every tile handle was released
                runs.append(mosaic)
        assert np.array_equal(runs[0], runs[2]) and np.array_equal(runs[1], runs[3])
        assert runs[0].std() > 10
    finally:
        isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = old


def test_library_jpeg_decoder_equals_pillow_and_refuses_what_it_does_not_take(tmp_path):
    """vfsms_jpeg_decode (csrc/jpeg_host.cpp: the system's libjpeg-turbo behind a self-declared ABI) against Pillow's decode of the same
    bytes -- the grayscale decode and the Y Cb Cr planes, for the chroma subsamplings, progressive files and a grayscale file -- byte for
    byte; files it must hand back to the caller (truncated, CMYK, not a JPEG) come back as None, never as a wrong image."""
    import io
    from PIL import Image, ImageFilter
    from imagestitch_amd import _lib
    rng = np.random.default_rng(5)
    base = np.asarray(Image.fromarray((rng.random((203, 331, 3)) * 255).astype(np.uint8)).filter(ImageFilter.GaussianBlur(1.5)))
    if _lib.jpeg_decode(_jpeg_bytes(base), False) is None:
        pytest.skip("no libjpeg.so.8 on this host: the Stitcher decodes with Pillow")
    blobs = [(ss, prog, _jpeg_bytes(base, subsampling=ss, progressive=prog)) for ss in (0, 1, 2) for prog in (False, True)]
    for ss, prog, b in blobs:
        im = Image.open(io.BytesIO(b)); im.draft("YCbCr", im.size); im.load()
        assert im.mode == "YCbCr"
        planes = _lib.jpeg_decode(b, True)
        assert planes.shape == (203, 331, 3) and np.array_equal(planes, np.asarray(im)), (ss, prog)
        im = Image.open(io.BytesIO(b)); im.draft("L", im.size); im.load()
        gray = _lib.jpeg_decode(b, False)
        assert gray.shape == (203, 331) and np.array_equal(gray, np.asarray(im)), (ss, prog)
        assert np.array_equal(gray, planes[:, :, 0])                       # IMREAD_GRAYSCALE of a JPEG is its Y plane
    g = _jpeg_bytes(base[:, :, 1])
    for want in (False, True):                                             # a grayscale file has one plane, whatever is asked for
        out = _lib.jpeg_decode(g, want)
        assert out.ndim == 2 and np.array_equal(out, np.asarray(Image.open(io.BytesIO(g))))
    b = blobs[0][2]
    assert _lib.jpeg_decode(b[:len(b) // 2], True) is None                 # truncated: libjpeg would pad it with gray
    assert _lib.jpeg_decode(b"not a jpeg at all" * 20, False) is None
    bio = io.BytesIO(); Image.fromarray(base).convert("CMYK").save(bio, "JPEG")
    assert _lib.jpeg_decode(bio.getvalue(), True) is None                  # four components
    bio = io.BytesIO(); Image.fromarray(base).save(bio, "PNG")
    assert _lib.jpeg_decode(bio.getvalue(), False) is None
    # many threads at once, one buffer each (the decoder pool's use): same bytes as alone
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(lambda k: _lib.jpeg_decode(blobs[k % len(blobs)][2], True), range(32)))
    ref = [_lib.jpeg_decode(bb[2], True) for bb in blobs]
    assert all(np.array_equal(o, ref[k % len(blobs)]) for k, o in enumerate(outs))


def test_raw_420_planes_and_the_restated_fancy_upsampler_equal_the_library():
    """Round 6: with a colour tile wanted and a 4:2:0 file, vfsms_tile_fill_jpeg stops the host behind the IDCT (jpeg_read_raw_data) and the
    DEVICE upsamples the chroma (k_ingest_420 restates jdsample.c's h2v2 fancy upsampling).  The restatement is held here, on the CPU, to the
    library's own upsampled planes: the raw Y plane == the Y of the full decode, and numpy's copy of the kernel's arithmetic --
    colsum(c) = 3 near + far, out(2c) = (3 colsum(c) + colsum(c - 1) + 8) >> 4, out(2c + 1) = (3 colsum(c) + colsum(c + 1) + 7) >> 4,
    neighbours clamped to the image's first / last sample row and column -- on the raw Cb / Cr planes == the library's, byte for byte: odd
    sizes, one-iMCU-row images, progressive files, the bench's 2048 x 2048.  Other samplings are refused (the full decode takes them)."""
    import io
    from PIL import Image
    from imagestitch_amd import _lib
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (41, 57, 3), dtype=np.uint8)
    if _lib.jpeg_decode(_jpeg_bytes(base), False) is None:
        pytest.skip("no libjpeg.so.8 on this host: the Stitcher decodes with Pillow")

    def fancy(plane, H, W):
        dh, dw = (H + 1) // 2, (W + 1) // 2
        p = plane[:dh, :dw].astype(np.int32)
        up = np.vstack([p[:1], p[:-1]]); dn = np.vstack([p[1:], p[-1:]])
        rows = np.empty((2 * dh, dw), np.int32); rows[0::2] = 3 * p + up; rows[1::2] = 3 * p + dn
        left = np.hstack([rows[:, :1], rows[:, :-1]]); right = np.hstack([rows[:, 1:], rows[:, -1:]])
        out = np.empty((2 * dh, 2 * dw), np.int32)
        out[:, 0::2] = (3 * rows + left + 8) >> 4; out[:, 1::2] = (3 * rows + right + 7) >> 4
        return out[:H, :W].astype(np.uint8)
    for (w, h), kw in (((613, 407), dict(quality=90)), ((333, 7), dict(quality=70)), ((1021, 767), dict(quality=85, progressive=True)),
                       ((64, 48), dict(quality=95)), ((5, 3), dict(quality=90)), ((2048, 2048), dict(quality=90))):
        img = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC))
        b = io.BytesIO(); Image.fromarray(img).save(b, "JPEG", subsampling=2, **kw); data = b.getvalue()
        full = _lib.jpeg_decode(data, True)
        raw = _lib.jpeg_decode_raw420(data)
        assert raw is not None and full is not None, (w, h)
        Y, Cb, Cr, H, W = raw
        assert (H, W) == (h, w) and Y.shape == ((h + 15) // 16 * 16, (w + 15) // 16 * 16)
        assert np.array_equal(Y[:H, :W], full[:, :, 0]), (w, h)
        assert np.array_equal(fancy(Cb, H, W), full[:, :, 1]) and np.array_equal(fancy(Cr, H, W), full[:, :, 2]), (w, h)
    for ss in (0, 1):                                                          # 4:4:4 and 4:2:2: not this path's
        assert _lib.jpeg_decode_raw420(_jpeg_bytes(np.asarray(Image.fromarray(base).resize((64, 48))), subsampling=ss)) is None
    assert _lib.jpeg_decode_raw420(_jpeg_bytes(base[:, :, 0])) is None         # a grayscale file
    b = _jpeg_bytes(np.asarray(Image.fromarray(base).resize((320, 200))), subsampling=2)
    assert _lib.jpeg_decode_raw420(b[:len(b) // 2]) is None                     # truncated


def test_library_jpeg_decoder_on_damaged_files():
    """Files are untrusted input to a C decoder running in the decoder threads: flipped bytes, cuts and splices in baseline / progressive /
    restart-marker files.  Every outcome is either a refusal (None: the caller's decoder takes over) or exactly Pillow's decode of the same
    damaged bytes -- never a crash, never a silently different image (600 mutations here; 6000 gave 1821 equal decodes, 4179 refusals, no
    difference)."""
    import io, warnings
    from PIL import Image, ImageFilter
    from imagestitch_amd import _lib
    rng = np.random.default_rng(1)
    base = np.asarray(Image.fromarray((rng.random((97, 131, 3)) * 255).astype(np.uint8)).filter(ImageFilter.GaussianBlur(1.2)))
    seeds = [_jpeg_bytes(base), _jpeg_bytes(base, quality=75, subsampling=0), _jpeg_bytes(base, quality=80, progressive=True),
             _jpeg_bytes(base, quality=85, optimize=True, restart_marker_blocks=4)]
    if _lib.jpeg_decode(seeds[0], False) is None:
        pytest.skip("no libjpeg.so.8 on this host")
    equal = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(600):
            b = bytearray(seeds[it % len(seeds)])
            if it % 3 == 0:
                for _ in range(rng.integers(1, 4)):
                    b[rng.integers(2, len(b))] = rng.integers(0, 256)
            elif it % 3 == 1:
                b = b[:rng.integers(4, len(b))]
            else:
                i = rng.integers(2, len(b) - 8)
                b[i:i + rng.integers(1, 8)] = bytes(rng.integers(0, 256, rng.integers(0, 12), dtype=np.uint8))
            b, planes = bytes(b), bool(it & 1)
            out = _lib.jpeg_decode(b, planes)
            if out is None:
                continue
            im = Image.open(io.BytesIO(b)); im.draft("YCbCr" if planes else "L", im.size); im.load()
            assert np.array_equal(np.asarray(im), out), it
            equal += 1
    assert equal > 50                                          # (damage in a comment / quantisation table / chroma tail still decodes)


def _jpeg_bytes(arr, **kw):
    import io
    from PIL import Image
    bio = io.BytesIO()
    Image.fromarray(arr).save(bio, "JPEG", quality=kw.pop("quality", 90), **kw)
    return bio.getvalue()


def test_jpeg_files_are_decoded_by_the_library_once_and_pillow_takes_the_rest(oracle, tmp_path):
    """The ingest pipeline hands JPEG files to vfsms_tile_fill_jpeg: one decode per file inside the library, `_decode_once` (Pillow) is not
    called at all; a file the library refuses (here: a PNG, and a CMYK JPEG) goes through `_decode_once` -- once; VFSMS_NATIVE_JPEG=0 sends
    everything through Pillow.  The three runs give the same mosaic as the engine without the entry point."""
    from PIL import Image
    from imagestitch_amd.synthetic import SyntheticGrid
    from imagestitch_amd import stitcher as ST
    from imagestitch_amd import _lib
    from fakes import NativeJpegEngine
    if _lib.jpeg_decode(_jpeg_bytes(np.zeros((16, 16), np.uint8)), False) is None:
        pytest.skip("no libjpeg.so.8 on this host")
    g = SyntheticGrid(2, 2, 256, overlap=0.25)
    tiles = _colour_tiles(g)
    files = _write_jpegs(tmp_path, tiles, "nat")
    mixed = list(files)
    mixed[1] = os.path.join(str(tmp_path), "nat_1.png")
    Image.fromarray(np.asarray(Image.open(files[1]).convert("RGB"))).save(mixed[1])          # the decoded JPEG, losslessly: same pixels
    old = (isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod)
    counts = {"once": 0}
    real_once = ST._decode_once

    def once(path, color):
        counts["once"] += 1
        return real_once(path, color)
    env_old = os.environ.get("VFSMS_NATIVE_JPEG")
    try:
        isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = 1, 0.3, "surf", "fadeInAndFadeOut"
        for color in (True, False):
            isa.Stitcher.isColorMode = color
            runs = []
            for flist, env, cls, want_native, want_once in ((files, "1", NativeJpegEngine, 4, 0), (files, "0", NativeJpegEngine, 0, 4),
                                                            (files, "1", IngestOracleEngine, 0, 4), (mixed, "1", NativeJpegEngine, 3, 1)):
                os.environ["VFSMS_NATIVE_JPEG"] = env
                eng = cls(oracle)
                s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False; s.direction = 1
                counts["once"] = 0
                ST._decode_once = once
                try:
                    (status, mosaic) = s.flowStitch(list(flist), s.calculateOffsetForFeatureSearchIncre)
                finally:
                    ST._decode_once = real_once
                assert status == (True, 3)
                assert len(getattr(eng, "native", [])) == want_native and counts["once"] == want_once, (env, cls.__name__, getattr(eng, "native", None), counts)
                assert not eng.live, eng.live
                runs.append(mosaic)
            assert all(np.array_equal(runs[0], r) for r in runs[1:3])
            # (the PNG holds the JPEG's RGB pixels, not its planes: the mixed run differs by the rounding of RGB -> gray / YCC -> RGB -> BGR only)
            assert runs[3].shape == runs[0].shape and np.abs(runs[3].astype(int) - runs[0].astype(int)).max() <= 4
    finally:
        if env_old is None:
            os.environ.pop("VFSMS_NATIVE_JPEG", None)
        else:
            os.environ["VFSMS_NATIVE_JPEG"] = env_old
        isa.Stitcher.direction, isa.Stitcher.directIncre, isa.Stitcher.roiRatio, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = old


def test_ingest_error_paths_free_every_handle(oracle, tmp_path):
    """A file that cannot be decoded: the batch waiting for its tile fails, flowStitch raises the decoder's error and every reserved handle
    (gray and colour) is released.  A corrupt file BEHIND a registration break is never an error -- the reference stops at the break and
    does not open it (Stitcher.py:64-79) -- and its decode is cancelled or ignored."""
    from imagestitch_amd.synthetic import SyntheticGrid
    g = SyntheticGrid(1, 5, 128, overlap=0.25)
    files = _write_jpegs(tmp_path, _colour_tiles(g), "err")
    old = (isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod)
    try:
        isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = "surf", "notFuse"
        for color in (True, False):
            isa.Stitcher.isColorMode = color
            bad = list(files)
            bad[2] = os.path.join(str(tmp_path), "corrupt_%d.jpg" % color)
            data = open(files[2], "rb").read()
            open(bad[2], "wb").write(data[:len(data) // 3])                # header intact (the size is read from it), entropy data cut
            accept = lambda A, B, job: [1, 5, -3, 9, 10, 10, 9, 0]
            eng = IngestOracleEngine(oracle, scripted=accept)
            s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False; s.direction = 1; s.decodeThreads = 2
            with pytest.raises(Exception):
                s.flowStitch(list(bad), s.calculateOffsetForFeatureSearchIncre)
            assert not eng.live, eng.live
            # the same corrupt file behind a break at pair 0: not an error, nothing leaks, the lone first tile is the result
            calls = []
            refuse = lambda A, B, job: (calls.append(1), [0, 0, 0, 0, 10, 10, 0, 0])[1]
            eng = IngestOracleEngine(oracle, scripted=refuse)
            s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False; s.direction = 1; s.decodeThreads = 1
            (status, mosaic) = s.flowStitch([files[0], files[1], files[3], bad[2]], s.calculateOffsetForFeatureSearchIncre)
            assert status == (False, 0) and mosaic.shape[:2] == (128, 128) and not eng.live
            # ... and when its decode had already STARTED (one decoder thread per file, the refused attempt takes its time): the incremental
            # registrar's table is full length with zero rows behind the break -- what counts is the leading registered pairs
            import time as _time
            slow_refuse = lambda A, B, job: (_time.sleep(0.3), [0, 0, 0, 0, 10, 10, 0, 0])[1]
            eng = IngestOracleEngine(oracle, scripted=slow_refuse)
            s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False; s.direction = 1; s.decodeThreads = 4
            (status, mosaic) = s.flowStitch([files[0], files[1], files[3], bad[2]], s.calculateOffsetForFeatureSearchIncre)
            assert status == (False, 0) and mosaic.shape[:2] == (128, 128) and not eng.live
    finally:
        isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = old


def test_png_and_tiff_band_writers_round_trip(tmp_path):
    """Stitcher.mosaicSink encoders behind the band stream (cv2.imwrite's place, Stitcher.py:174-179): bands of a gray and of a B G R mosaic
    go in, Pillow reads the same pixels back (R G B in the file), for band heights that do and do not divide the image."""
    from PIL import Image
    rng = np.random.default_rng(21)
    for shape in ((157, 203), (90, 131, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for ext in ("png", "tif"):
            for band in (32, 157, 500):
                path = os.path.join(str(tmp_path), "sub", "m%d_%d.%s" % (len(shape), band, ext))
                sink = isa.band_writer_for(path)
                for r0 in range(0, shape[0], band):
                    sink(r0, img[r0:r0 + band], shape)
                back = np.asarray(Image.open(path))
                want = img[:, :, ::-1] if img.ndim == 3 else img
                assert back.shape == want.shape and np.array_equal(back, want), (shape, ext, band)


def test_bigtiff_layout_is_readable_for_gray_and_colour(tmp_path):
    """The BigTIFF branch of TiffBandWriter (mosaics >= 4 GB: configs[4]'s 32 x 32 grid of 4096^2 colour tiles is 40 GB) forced on a small
    image: 8-byte offsets, LONG8 directory, and BitsPerSample 8, 8, 8 INLINE in the 8-byte value field (three SHORTs fit it; an offset there
    is read as the values and the file is unreadable).  Pillow reads the same pixels back, gray and colour, one strip and several."""
    from PIL import Image
    rng = np.random.default_rng(23)
    for shape in ((61, 47), (53, 38, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for band in (16, 200):
            path = os.path.join(str(tmp_path), "big%d_%d.tif" % (len(shape), band))
            sink = isa.TiffBandWriter(path, force_big=True)
            for r0 in range(0, shape[0], band):
                sink(r0, img[r0:r0 + band], shape)
            assert open(path, "rb").read(4) == b"II\x2b\x00"                      # BigTIFF magic 43
            back = np.asarray(Image.open(path))
            want = img[:, :, ::-1] if img.ndim == 3 else img
            assert back.shape == want.shape and np.array_equal(back, want), (shape, band)


def _restart_layout(data):
    """(restart interval of the DRI segment or 0, number of RSTn markers in the entropy-coded data) of a baseline JPEG"""
    dri, p = 0, 2
    while data[p + 1] != 0xDA:
        seg = int.from_bytes(data[p + 2:p + 4], "big")
        if data[p + 1] == 0xDD:
            dri = int.from_bytes(data[p + 4:p + 6], "big")
        p += 2 + seg
    body = bytes(data[p:])
    return dri, sum(body.count(bytes([0xFF, 0xD0 + k])) for k in range(8))


def test_jpeg_band_writer_writes_what_cv2_imwrite_writes(tmp_path):
    """The reference writes every mosaic as .jpg (Main.py:21-51 -> cv2.imwrite at Stitcher.py:149, 175-179: libjpeg defaults, quality 95).
    vfsms_jpeg_encode of a whole image is that file BYTE FOR BYTE (Pillow's quality-95 save is the stand-in for cv2.imwrite: same libjpeg
    settings); the band writer's file -- stripes encoded on a thread pool, joined as restart intervals -- decodes to exactly the same pixels,
    for colour and gray, band heights that do not divide anything, stripes of 16 to 256 rows, and an image smaller than a stripe; the
    restart layout is what the header promises."""
    import io
    from PIL import Image, ImageFilter
    from imagestitch_amd import _lib
    if _lib.jpeg_encode(np.zeros((8, 8), np.uint8)) is None:
        pytest.skip("no libjpeg.so.8 on this host: .jpg results are written through Pillow")
    rng = np.random.default_rng(8)
    for shape in ((700, 531, 3), (333, 1001), (40, 57, 3), (512, 256, 3)):
        small = rng.integers(0, 256, (shape[0] // 6 + 2, shape[1] // 6 + 2) + shape[2:], dtype=np.uint8)
        img = np.asarray(Image.fromarray(small).resize((shape[1], shape[0]), Image.BICUBIC).filter(ImageFilter.GaussianBlur(0.7)))
        assert img.shape == shape
        bgr = np.ascontiguousarray(img[:, :, ::-1]) if img.ndim == 3 else img
        ref = io.BytesIO(); Image.fromarray(img).save(ref, "JPEG", quality=95)
        want = np.asarray(Image.open(io.BytesIO(ref.getvalue())))
        assert _lib.jpeg_encode(bgr, bgr=True, quality=95).tobytes() == ref.getvalue()
        if img.ndim == 3:
            assert _lib.jpeg_encode(img, bgr=False, quality=95).tobytes() == ref.getvalue()
        for band, stripe in ((4096, None), (100, 64), (37, 16), (shape[0], 256)):
            path = os.path.join(str(tmp_path), "j", "m%d_%d_%d.jpg" % (shape[0], band, stripe or 0))
            sink = isa.JpegBandWriter(path, stripe_rows=stripe, threads=4)
            for r0 in range(0, shape[0], band):
                sink(r0, bgr[r0:r0 + band], shape)
            data = open(path, "rb").read()
            back = np.asarray(Image.open(io.BytesIO(data)))
            assert back.shape == want.shape and np.array_equal(back, want), (shape, band, stripe)
            mcu = 16 if img.ndim == 3 else 8
            S = max(1, (stripe or 256) // mcu) * mcu
            n_stripes = -(-shape[0] // S)
            dri, rst = _restart_layout(data)
            assert (dri, rst) == ((0, 0) if n_stripes == 1 else (-(-shape[1] // mcu) * (S // mcu), n_stripes - 1)), (shape, band, stripe, dri, rst)
            if n_stripes == 1:
                assert data == ref.getvalue()
            got = _lib.jpeg_decode(data, True)                             # and through the library's own decoder: the planes of the one-thread file
            assert got is not None and np.array_equal(got, _lib.jpeg_decode(ref.getvalue(), True))
    assert isinstance(isa.band_writer_for("x.jpg"), isa.JpegBandWriter)
    # bands that live in a ring of three reused buffers (Engine.canvas_download_bands(transient=True): pinned memory): the writer is done
    # with a band before the band after the next one overwrites it, also with slow encoders
    import time
    shape = (1500, 300, 3)
    img = np.asarray(Image.fromarray(rng.integers(0, 256, (100, 20, 3), dtype=np.uint8)).resize((300, 1500), Image.BICUBIC))
    bgr = np.ascontiguousarray(img[:, :, ::-1])
    ref = io.BytesIO(); Image.fromarray(img).save(ref, "JPEG", quality=95)
    for band in (128, 100):
        path = os.path.join(str(tmp_path), "ring_%d.jpg" % band)
        sink = isa.JpegBandWriter(path, stripe_rows=32, threads=2)
        assert sink.transient_bands
        real = sink._encode
        sink._encode = lambda rows, real=real: (time.sleep(0.002), real(rows))[1]
        ring = [np.empty((band,) + shape[1:], np.uint8) for _ in range(3)]
        for k, r0 in enumerate(range(0, shape[0], band)):
            buf = ring[k % 3]
            buf[:] = 255 - buf                                              # whatever was there is gone now
            n = min(band, shape[0] - r0)
            buf[:n] = bgr[r0:r0 + n]
            sink(r0, buf[:n], shape)
        assert np.array_equal(np.asarray(Image.open(path)), np.asarray(Image.open(io.BytesIO(ref.getvalue())))), band
    # a stripe that is not a whole number of MCU rows cannot be a restart interval; neither can streams of different images
    a, b = _lib.jpeg_encode(bgr[:24]), _lib.jpeg_encode(bgr[24:48])
    with pytest.raises(Exception):
        _lib.jpeg_join([a, b], 24, 48)
    with pytest.raises(Exception):
        _lib.jpeg_join([_lib.jpeg_encode(bgr[:32]), _lib.jpeg_encode(bgr[32:64, :100])], 32, 64)
    with pytest.raises(Exception):
        _lib.jpeg_join([_lib.jpeg_encode(bgr[:32]), _lib.jpeg_encode(bgr[32:64], quality=80)], 32, 64)
    with pytest.raises(ValueError):
        isa.JpegBandWriter(os.path.join(str(tmp_path), "big.jpg"))(0, np.zeros((4, 8), np.uint8), (70000, 8))


def test_imwrite_jpg_goes_through_the_stripe_encoder(tmp_path):
    """_imwrite (cv2.imwrite's place) for .jpg: the library's encoder on all cores when the host has libjpeg-turbo, Pillow with quality 95
    under VFSMS_NATIVE_JPEG=0 -- the same decoded pixels either way (B G R in, R G B in the file)."""
    from PIL import Image
    from imagestitch_amd import stitcher as ST
    rng = np.random.default_rng(4)
    img = np.asarray(Image.fromarray(rng.integers(0, 256, (90, 70, 3), dtype=np.uint8)).resize((420, 540), Image.BICUBIC))
    old = os.environ.get("VFSMS_NATIVE_JPEG")
    try:
        outs = []
        for env in ("1", "0"):
            os.environ["VFSMS_NATIVE_JPEG"] = env
            for arr, tag in ((img, "c"), (np.ascontiguousarray(img[:, :, 1]), "g")):
                p = os.path.join(str(tmp_path), "w", "%s%s.jpg" % (tag, env))
                ST._imwrite(p, arr)
                outs.append(np.asarray(Image.open(p)))
        assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3])
        right, wrong = np.abs(outs[0].astype(int) - img[:, :, ::-1]).mean(), np.abs(outs[0].astype(int) - img).mean()
        assert right < 8 and wrong > 4 * right and outs[1].shape == (540, 420), (right, wrong)       # (4:2:0 at quality 95 on a noisy image)
    finally:
        if old is None:
            os.environ.pop("VFSMS_NATIVE_JPEG", None)
        else:
            os.environ["VFSMS_NATIVE_JPEG"] = old


def test_streamed_output_names_and_bytes_equal_the_whole_image_write(oracle, tmp_path):
    """streamOutput: imageSetStitchWithMutiple encodes every mosaic band by band while it leaves the canvas; the files carry the reference's
    names (stitching_result_<i>[_<j>].<ext>, Stitcher.py:174-179) and the same pixels as the whole-image write -- also across a
    registration break (two segments, the part files renamed at the end)."""
    from PIL import Image
    from imagestitch_amd.synthetic import SyntheticGrid
    g = SyntheticGrid(1, 4, 128, overlap=0.25)
    proj = tmp_path / "p"; (proj / "1").mkdir(parents=True)
    for k, t in enumerate(_colour_tiles(g)):
        Image.fromarray(t).save(str(proj / "1" / ("t%02d.png" % k)))
    old = (isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod)
    try:
        isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = True, "surf", "notFuse"
        for breaks, oext in ((False, "png"), (True, "png"), (False, "jpg"), (True, "jpg")):      # (.jpg: Main.py's own output format)
            outs = {}
            for stream in (False, True):
                eng = IngestOracleEngine(oracle, scripted=(lambda A, B, job: [1, 96, 0, 9, 10, 10, 9, 0]) if not breaks else _break_at_second_pair())
                s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False; s.direction = 2; isa.Stitcher.direction = 2
                s.streamOutput = stream; s.mosaicBandRows = 50
                os.environ["VFSMS_PINNED_BANDS"] = "1" if (stream and breaks) else "0"      # the engine's ring of reused band buffers
                out = tmp_path / ("o%d%d%s" % (breaks, stream, oext))
                s.imageSetStitchWithMutiple(str(proj), str(out) + os.sep, 1, s.calculateOffsetForFeatureSearchIncre, fileExtension="png", outputfileExtension=oext)
                names = sorted(n for n in os.listdir(str(out)) if not n.startswith("."))
                outs[stream] = {n: np.asarray(Image.open(str(out / n))) for n in names if n.endswith("." + oext)}
                assert not [n for n in os.listdir(str(out)) if n.startswith(".stitching_part")] and not eng.live
                assert (getattr(eng, "transient_bands_served", 0) > 0) == (stream and breaks)
            assert sorted(outs[False]) == sorted(outs[True]) and len(outs[True]) == (2 if breaks else 1), (breaks, sorted(outs[True]))
            for n in outs[False]:
                assert np.array_equal(outs[False][n], outs[True][n]), n
    finally:
        os.environ.pop("VFSMS_PINNED_BANDS", None)
        isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = old


def _break_at_second_pair():
    """scripted attempts: every attempt of the pair (tile 1, tile 2) fails -- a registration break in the middle of four tiles"""
    seen = {}

    def script(A, B, job):
        key = (int(job[0]), int(job[1]))
        order = seen.setdefault(key, len(seen))
        return [0, 0, 0, 0, 10, 10, 0, 0] if order == 1 else [1, 96, 0, 9, 10, 10, 9, 0]
    return script


def test_reserve_failure_midway_leaks_nothing(oracle, tmp_path):
    """HBM exhausted while the ingest pipeline reserves its tiles (a long file list): the handles reserved so far -- still unfilled, which
    vfsms_tile_free refuses -- are given up and freed, and the error reaches the caller."""
    from imagestitch_amd.synthetic import SyntheticGrid
    g = SyntheticGrid(1, 4, 128, overlap=0.25)
    files = _write_jpegs(tmp_path, _colour_tiles(g), "oom")

    class Exhausted(IngestOracleEngine):
        def tile_reserve(self, h, w):
            if len(self.live) >= 3:
                raise MemoryError("hipMalloc: out of memory")
            return super().tile_reserve(h, w)
    old = (isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod)
    try:
        isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod, isa.Stitcher.isColorMode = "surf", "notFuse", True
        eng = Exhausted(oracle, scripted=lambda A, B, job: [1, 5, -3, 9, 10, 10, 9, 0])
        s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False; s.direction = 1
        with pytest.raises(MemoryError):
            s.flowStitch(list(files), s.calculateOffsetForFeatureSearchIncre)
        assert not eng.live, eng.live
    finally:
        isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = old


def test_fast_header_reader_equals_pillow(tmp_path):
    """_imshape reads JPEG / PNG sizes straight from the header (ninety Image.open calls were 10-20 ms in front of the ingest pipeline); it
    must agree with Pillow on baseline and progressive JPEGs, JPEGs with a long metadata block in front of the frame header, gray and colour
    PNGs, and fall back to Pillow for anything else."""
    from PIL import Image
    from imagestitch_amd import stitcher as ST
    rng = np.random.default_rng(3)
    cases = []
    for k, (shape, kw, ext) in enumerate((((37, 53), {}, "jpg"), ((211, 149, 3), dict(progressive=True), "jpg"), ((64, 80, 3), dict(quality=95, subsampling=0), "jpeg"),
                                          ((33, 65), {}, "png"), ((20, 31, 3), {}, "png"), ((29, 41), {}, "tif"), ((29, 41), {}, "bmp"))):
        p = os.path.join(str(tmp_path), "h%d.%s" % (k, ext))
        Image.fromarray(rng.integers(0, 256, shape, dtype=np.uint8)).save(p, **kw)
        cases.append(p)
    big = os.path.join(str(tmp_path), "exif.jpg")
    Image.fromarray(rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)).save(big, icc_profile=bytes(40000))     # a 40 KB APP2 block before SOF
    cases.append(big)
    for p in cases:
        with Image.open(p) as im:
            want = (im.size[1], im.size[0])
        assert ST._imshape(p) == want, p


def test_decode_once_on_the_reference_demo_tiles():
    """The same identity on the reference's own micrographs, where they are present (this container; the GPU box has no /root/reference and
    no `-m gpu` test reads it): one file per demo dataset -- 4:2:0 colour JPEGs from the microscope cameras and the grayscale zircon scans --
    decoded ONCE to planes gives the grayscale decode (Y) and the colour decode (jdcolor arithmetic) byte for byte."""
    import ctypes
    import glob
    from imagestitch_amd import stitcher as ST
    root = "/root/reference/demoImages"
    if not os.path.isdir(root):
        pytest.skip("the reference checkout is not present here")
    seen = 0
    for d in sorted(glob.glob(os.path.join(root, "*", "1"))):
        files = sorted(f for f in glob.glob(os.path.join(d, "*")) if f.lower().endswith((".jpg", ".jpeg")))
        if not files:
            continue
        p = files[len(files) // 2]
        owner, shape, parts = ST._decode_once(p, True)
        assert parts[0] == "src" and shape == ST._imshape(p)
        spx = {0: 1, 1: 3, 2: 4}[parts[3]]
        buf = np.frombuffer((ctypes.c_uint8 * (shape[0] * parts[2])).from_address(parts[1]), np.uint8).reshape(shape[0], parts[2])
        px = buf[:, :shape[1] * spx].reshape(shape[0], shape[1], spx)
        assert np.array_equal(px[:, :, 0], ST._imread(p, False)), p
        want = ST._imread(p, True)
        got = np.repeat(px[:, :, :1], 3, 2) if parts[3] == 0 else ST._ycc_to_bgr(px[:, :, :3])
        assert np.array_equal(got, want), p
        del owner
        seen += 1
    assert seen >= 4


def test_a_registration_break_does_not_decode_a_file_twice(oracle, tmp_path):
    """flowStitchWithMutiple restarts behind every pair that cannot be registered (Stitcher.py:96-127) and the reference decodes the remaining
    list again each time.  Here the tiles the first segment's pipeline had already decoded behind the break wait in HBM for the segment
    that uses them: every file is decoded exactly once across the break, a trailing lone tile comes back from its device copy, and nothing
    stays in HBM -- gray and colour."""
    from imagestitch_amd.synthetic import SyntheticGrid
    from imagestitch_amd import stitcher as ST
    g = SyntheticGrid(1, 6, 128, overlap=0.25)
    files = _write_jpegs(tmp_path, _colour_tiles(g), "brk")
    old = (isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod)
    counts = {"once": 0, "imread": 0}
    real_once, real_imread = ST._decode_once, ST._imread

    def once(path, color):
        counts["once"] += 1
        return real_once(path, color)

    def imread(path, color):
        counts["imread"] += 1
        return real_imread(path, color)

    try:
        isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = "surf", "notFuse"
        for color in (True, False):
            isa.Stitcher.isColorMode = color
            for nfiles, bad, want_results in ((6, (2,), 2), (4, (2,), 2), (5, (0, 3), 3)):
                sub = files[:nfiles]
                ref_tiles = [ST._imread(f, False) for f in sub]

                def script(A, B, job, ref_tiles=ref_tiles, bad=bad):
                    ka = [i for i, t in enumerate(ref_tiles) if t.shape == np.asarray(A).shape and np.array_equal(t, A)][0]
                    return [0, 0, 0, 0, 10, 10, 0, 0] if ka in bad else [1, 96, 0, 9, 10, 10, 9, 0]
                eng = IngestOracleEngine(oracle, scripted=script)
                s = isa.Stitcher(); s._engine = eng; s.isPrintLog = False; s.direction = 2; isa.Stitcher.direction = 2
                counts["once"] = counts["imread"] = 0
                ST._decode_once, ST._imread = once, imread
                try:
                    results = s.flowStitchWithMutiple(list(sub), s.calculateOffsetForFeatureSearchIncre)
                finally:
                    ST._decode_once, ST._imread = real_once, real_imread
                assert len(results) == want_results, (color, nfiles, bad, len(results))
                assert counts["once"] == nfiles and counts["imread"] == 0, (color, nfiles, bad, counts)
                assert not eng.live and "_ingestCache" not in s.__dict__
                assert all(r is not None and r.ndim == (3 if color else 2) for r in results)
                if nfiles == 4:                                     # tiles 0-2 | tile 3 alone: the lone tile equals its decode
                    assert np.array_equal(results[1], ST._imread(sub[3], color))
    finally:
        isa.Stitcher.direction, isa.Stitcher.isColorMode, isa.Stitcher.featureMethod, isa.Stitcher.fuseMethod = old


def test_bench_says_when_its_pmc_evidence_is_of_another_build(tmp_path, monkeypatch):
    """bench.py reads the PMC-derived roofline fields from the newest profiles/*_pmc_summary.txt; tools/profile_round.sh stamps that file with
    the content hash of the kernel sources it profiled (tools/build_id.py) and bench.pmc_build() compares: the same hash -> pmc_stale False,
    another hash or a summary without a stamp -> True.  (What keeps `traffic`, `valu_busy_frac_pmc`, ... from silently describing an older
    kernel.)"""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    from tools.build_id import build_id, src_sha256
    mine = build_id()
    assert mine["src_sha256"] == src_sha256() and len(mine["src_sha256"]) == 16
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "r97_pmc_summary.txt").write_text("# build: head=abc lib_sha256=%s src_sha256=%s\nk_describe launches=1 INSTS_VALU=1e9\n" % (mine["lib_sha256"], mine["src_sha256"]))
    info = bench.pmc_build()
    assert info["pmc_stale"] is False and info["pmc_build"]["src_sha256"] == mine["src_sha256"] and info["pmc_summary"].endswith("r97_pmc_summary.txt")
    (prof / "r98_pmc_summary.txt").write_text("# build: head=abc lib_sha256=0 src_sha256=0000000000000000\nk_describe launches=1 INSTS_VALU=1e9\n")
    assert bench.pmc_build()["pmc_stale"] is True                       # the newest summary is of other sources
    (prof / "r99_pmc_summary.txt").write_text("k_describe launches=1 INSTS_VALU=1e9\n")
    assert bench.pmc_build()["pmc_stale"] is True                       # no stamp at all (summaries of rounds 1-4)

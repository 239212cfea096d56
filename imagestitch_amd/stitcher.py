"""Sequence drivers, pair registration and mosaic assembly -- host-side mirror of the reference's
`Stitcher.Stitcher` (/root/reference/Stitcher.py; every method cites the lines it mirrors).

Drop-in for Main.py: same class attributes (direction, directIncre, fuseMethod, ...), same method names and
return conventions (`(True, [dx, dy])` / `(False, "  The two images can not match")`), same log lines.
What differs is where the work happens: registration attempts run as fused device-resident batches
(SURF + BF-L2 + ratio + mode vote, or FP64 phase correlation) and the mosaic lives in a u8 + validity canvas
in HBM; see imagestitch_amd/csrc.
"""
import copy
import os
import time

import numpy as np

from . import utility as Utility
from . import fusion as ImageFusion
from .utility import roi_rect

CANNOT_MATCH = "  The two images can not match"


class ImageFeature():
    """Stitcher.py:14-18: features of the second image of the previous pair (full-image line scans)."""
    isBreak = True
    kps = None
    feature = None


class ResidentFeatures:
    """What Stitcher.tempImageFeature holds when the stock operators run: the keypoints + descriptors of a tile stay in HBM under
    an engine handle (vfsms_features_*); `kps` and `feature` of the cache point at this object, which downloads the arrays only if
    somebody actually reads them (np.asarray(obj.kps) / obj.descriptors())."""

    def __init__(self, engine, handle, n, dim):
        self.engine, self.handle, self.n, self.dim = engine, handle, n, dim
        self._host = None

    def __len__(self):
        return self.n

    def arrays(self):
        if self._host is None:
            self._host = self.engine.features_download(self.handle, self.n, self.dim)
        return self._host

    def __array__(self, dtype=None, copy=None):
        return self.arrays()[0]

    def descriptors(self):
        return self.arrays()[1]

    def release(self):
        if self.handle is not None:
            try:
                self.engine.features_free(self.handle)
            except Exception:
                pass
            self.handle = None


from .ingest import (_imread, _imread_gray_pointer, _PillowBlocks, _decoder_pool, _decode_once, _fill_from_jpeg, _ycc_to_bgr, _imshape,  # noqa: F401
                     _list_images, TileIngest)
from .io import (_imwrite, _imwrite_jpeg_stripes, NpyBandWriter, PngBandWriter, JpegBandWriter, TiffBandWriter, _NoNativeJpeg,  # noqa: F401
                 _native_jpeg_encoder, band_writer_for)


def _join(base, *parts):
    return os.path.join(base.replace("\\", os.sep), *[str(p) for p in parts])


class Stitcher(Utility.Method):
    isColorMode = True
    direction = 1               # 1: A above B; 2: A left of B; 3: A below B; 4: A right of B
    directIncre = 1             # rotation step of the direction search: 1, 0 or -1
    fuseMethod = "notFuse"
    phaseResponseThreshold = 0.15
    # cv2.phaseCorrelate(a, b) reports the shift of b's content relative to a's (b - a); the feature path votes a - b, and the
    # reference adds both with the same sign (Stitcher.py:244-251), so its phase offsets come out mirrored (SURVEY 8a row G:
    # iron gives [1400, 0] where the true offset is [1699, -1]).  The default reproduces the reference as written; True negates
    # the raw shift before the axis correction (iron -> [1698, 0]).  The batched registrar follows the reference only.
    phaseSignFix = False
    tempImageFeature = ImageFeature()

    imageFusion = ImageFusion.ImageFusion()

    # ------------------------------------------------------------------------------------------------
    def directionIncrease(self, direction):
        """Stitcher.py:36-47: rotate within [1, 4]."""
        direction += self.directIncre
        if direction == 5:
            direction = 1
        if direction == 0:
            direction = 4
        return direction

    # ------------------------------------------------------------------------------------------------
    def flowStitch(self, fileList, caculateOffsetMethod):
        """Stitcher.py:49-94 -> ((status, endfileIndex), stitchImage)."""
        self.printAndWrite("Stitching the directory which have " + str(fileList[0]))
        fileNum = len(fileList)
        offsetList = []
        describtion = ""
        startTime = time.time()
        status = True
        endfileIndex = 0
        imageB = None
        batched = self._registerBatched(fileList, caculateOffsetMethod)
        if batched is not None:
            (status, endfileIndex, offsetList, describtion) = batched
        try:
            for fileIndex in (range(0, fileNum - 1) if batched is None else ()):
                self.printAndWrite("stitching " + str(fileList[fileIndex]) + " and " + str(fileList[fileIndex + 1]))
                imageA = imageB if imageB is not None else _imread(fileList[fileIndex], False)   # decoded once per tile
                imageB = _imread(fileList[fileIndex + 1], False)
                if caculateOffsetMethod == self.calculateOffsetForPhaseCorrleate:
                    (status, offset) = self.calculateOffsetForPhaseCorrleate([fileList[fileIndex], fileList[fileIndex + 1]])
                else:
                    (status, offset) = caculateOffsetMethod([imageA, imageB])
                if status == False:
                    describtion = "  " + str(fileList[fileIndex]) + " and " + str(fileList[fileIndex + 1]) + " can not be stitched"
                    break
                else:
                    offsetList.append(offset)
                    endfileIndex = fileIndex + 1
        except Exception:
            self.releaseTiles()                     # an operator that raises must not leave the pair's tiles in HBM
            raise
        self.releaseTiles()
        endTime = time.time()
        self.printAndWrite("The time of registering is " + str(endTime - startTime) + "s")
        self.printAndWrite("start stitching")
        startTime = time.time()
        stitchImage = self.getStitchByOffset(fileList, offsetList)
        endTime = time.time()
        self.printAndWrite("The time of fusing is " + str(endTime - startTime) + "s")
        if status == False:
            self.printAndWrite(describtion)
        return ((status, endfileIndex), stitchImage)

    pathHint = None              # optional PREDICTION of the accepted direction of every pair (a scan pattern the operator knows, e.g. from grid.serpentine_directions); the speculation prior of the FIRST dataset -- later ones use what the previous one taught (GridRegistrar.path_memory).  Results never depend on it.
    streamOutput = True          # imageSetStitch*: encode JPEG / PNG / TIFF / NPY results band by band as they leave the device -- the mosaic is never whole in host memory and the encoder works while later bands are still copied (default since round 5; False: download the whole mosaic, then write it)
    batchRegistration = True     # let flowStitch register a whole file list in fused device batches when the stock search is used
    decodeThreads = 0            # decoder threads of the ingest pipeline (0: one per host core, at most 32 with the library's own JPEG decoder, 16 with Pillow -- beyond that its Python-side work and the registrar's own host thread get in each other's way; _decoderThreads)

    def _streamTo(self, paths, pattern=None):
        """install a mosaicSink that opens one streaming encoder per mosaic: `paths` names the files (pattern is None) or collects the
        part files made from `pattern % n`; returns None (and installs nothing) when the format has no band encoder"""
        if "mosaicSink" in self.__dict__:                    # the user streams the mosaics somewhere else (stitcher.mosaicSink = NpyBandWriter(...)): leave it alone
            return None
        probe = pattern % 0 if pattern else paths[0]
        if band_writer_for(probe) is None:
            return None
        state = {"w": None, "n": 0}

        def sink(row0, band, full_shape):
            if state["w"] is None:
                path = (pattern % state["n"]) if pattern else paths[state["n"]]
                if pattern:
                    paths.append(path)
                state["n"] += 1
                state["w"] = band_writer_for(path)
            state["w"](row0, band, full_shape)
            if row0 + band.shape[0] >= full_shape[0]:
                state["w"] = None
        sink.transient_bands = bool(getattr(band_writer_for(probe), "transient_bands", False))
        self.mosaicSink = sink
        return sink

    def _decoderThreads(self, n_files):
        """size of the decoder pool: `decodeThreads`, or one per host core up to 32 when the library decodes JPEGs itself (measured on the
        256-thread host of the MI355X box, colour 2048 x 2048: 1354 tiles/s at 16 threads, 1804 at 32; Pillow: 1094 at 16 and no more beyond)"""
        native = getattr(self.engine, "tile_fill_jpeg", None) is not None and os.environ.get("VFSMS_NATIVE_JPEG", "1") != "0"
        return max(1, min(int(self.decodeThreads or min(os.cpu_count() or 4, 32 if native else 16)), n_files, 64))

    def _batchedMethod(self, caculateOffsetMethod, n_files):
        """which fused path serves `caculateOffsetMethod` ("surf" / "orb" / "phase": incremental ROI search; "surf_full" / "orb_full": whole-tile
        features, the line scans of Main.py:29-51), or None: a custom method or operators, the switch off, fewer than two files"""
        fn, owner = getattr(caculateOffsetMethod, "__func__", None), getattr(caculateOffsetMethod, "__self__", None)
        if not self.batchRegistration or owner is not self or n_files < 2:
            return None
        if fn is Stitcher.calculateOffsetForFeatureSearchIncre and self._usesStockOperators():
            return self.featureMethod
        if fn is Stitcher.calculateOffsetForPhaseCorrleateIncre and not self.phaseSignFix:
            return "phase"
        if fn is Stitcher.calculateOffsetForFeatureSearch and self._usesStockOperators() and self.offsetCaculate == "mode":
            if self.featureMethod == "surf" and hasattr(self.engine, "features_surf_batch"):
                return "surf_full"
            if self.featureMethod == "orb" and not self.isEnhance and hasattr(self.engine, "attempt_orb_batch"):
                return "orb_full"
        return None

    def _makeRegistrar(self, method, n_files):
        """the GridRegistrar of one file list, primed with what the previous dataset taught this stitcher (accepted directions, same number of
        tiles: GridRegistrar.path_memory; Main.py runs its datasets through ONE Stitcher with one setting) or with the operator's pathHint"""
        from .grid import GridRegistrar
        params = None if method == "phase" else (self._orbParams() if method in ("orb", "orb_full") else self._surfParams())
        reg = GridRegistrar(self.engine, method="surf" if method == "surf_full" else "orb" if method == "orb_full" else method, roiRatio=self.roiRatio,
                            searchRatio=self.searchRatio, offsetEvaluate=self.offsetEvaluate, directIncre=self.directIncre, surfParams=params,
                            phaseResponseThreshold=self.phaseResponseThreshold, window=48,
                            enhance=self._enhanceSpec() if method in ("surf", "surf_full") else (0, 0.0, 0))
        reg.orbMaxDistance = self.orbMaxDistance if self.isGPUAvailable else -1
        reg.path_memory = self.__dict__.get("_pathMemory")
        reg.path_suspect = bool(self.__dict__.get("_pathSuspect", False))
        return reg

    def _operatorHint(self, reg, n_pairs):
        """the operator's pathHint as a CALLER'S hint of register() -- never installed as the registrar's memory, so it is not put on trial
        (GridRegistrar._learn tries memories only) -- and only while the registrar has no memory of its own for this path length; once a
        learned memory was dropped for mispredicting twice in a row, the next path runs cold as _learn promises (the hint is not slipped
        back in its place)."""
        m = reg.path_memory
        if (m is not None and len(m) == n_pairs) or self.pathHint is None or len(self.pathHint) != n_pairs:
            return None
        if self.__dict__.get("_pathMemoryDropped", False):
            return None
        return [int(d) for d in self.pathHint]

    def _registerBatched(self, fileList, caculateOffsetMethod):
        """The pair loop of flowStitch (Stitcher.py:64-79) through grid.GridRegistrar when `caculateOffsetMethod` is this
        object's own calculateOffsetForFeatureSearchIncre / calculateOffsetForPhaseCorrleateIncre with the stock operators:
        all tiles go to the GPU once (ingest.TileIngest: reserved handles filled by a pool of decoder threads while the registrar already
        works) and the pairs are registered in speculative fused batches whose selected results equal the pair-by-pair search (same
        offsets, same self.direction threading, same log lines).  Returns None when the sequential loop has to run (custom method or
        operators, tiles of different sizes, switch off) else (status, endfileIndex, offsetList, description of the break)."""
        method = self._batchedMethod(caculateOffsetMethod, len(fileList))
        if method is None:
            return None
        eng = self.engine
        shapes = [_imshape(f) for f in fileList]             # from the file headers: nothing is decoded yet
        if any(s != shapes[0] for s in shapes):
            return None
        reg = self._makeRegistrar(method, len(fileList))
        device_fuse = (self.fuseMethod in ("notFuse", "fadeInAndFadeOut", "trigonometric") and hasattr(eng, "canvas_fuse_tile_resident")) or \
                      (self.fuseMethod in ("average", "maximum", "minimum") and hasattr(eng, "canvas_blend_tile_resident"))
        # the tiles the mosaic is assembled from stay in HBM: the registration planes themselves for gray mosaics, and for colour mosaics
        # (Main.py:14's default) the B G R tiles the SAME decode produced -- every file is decoded exactly once (Stitcher.py:68-69, 382-403)
        color = bool(self.isColorMode) and device_fuse and hasattr(eng, "tile_fill_pair")
        keep = device_fuse and (color or not self.isColorMode) and len(set(fileList)) == len(fileList)
        job = TileIngest(self, fileList, shapes, color, keep)
        table = None
        try:
            handles = job.start()
            if method == "surf_full":
                table = self._fullImageTable(handles)
            elif method == "orb_full":
                table = self._fullImageTableOrb(handles, shapes)
            else:
                had_memory = reg.path_memory is not None
                table, _d = reg.register(handles, shapes, self.direction, stop_on_fail=True, hint=self._operatorHint(reg, len(shapes) - 1))
                # (what this path taught, incl. "nothing": a memory that mispredicted twice in a row is dropped, GridRegistrar._learn)
                self._pathMemory, self._pathSuspect = reg.path_memory, reg.path_suspect
                self._pathMemoryDropped = had_memory and reg.path_memory is None
        except BaseException:
            job.failed = True
            raise
        finally:
            job.finish(table)
        offsetList, endfileIndex, status, describtion = [], 0, True, ""
        for k, row in enumerate(table):
            self.printAndWrite("stitching " + str(fileList[k]) + " and " + str(fileList[k + 1]))
            if not row[0]:
                status = False
                describtion = "  " + str(fileList[k]) + " and " + str(fileList[k + 1]) + " can not be stitched"
                break
            if method not in ("surf_full", "orb_full"):
                self.direction = int(row[3])
            self.printAndWrite("  The offset of stitching: dx is " + str(int(row[1])) + " dy is " + str(int(row[2])))
            offsetList.append([int(row[1]), int(row[2])])
            endfileIndex = k + 1
        return (status, endfileIndex, offsetList, describtion)

    def _fullImageTable(self, handles):
        """calculateOffsetForFeatureSearch (Stitcher.py:260-304) over consecutive resident tiles, batched: every tile is described once
        (vfsms_features_surf_batch: 16 tiles per fused launch sequence -- the reference's B -> A feature reuse, Stitcher.py:278-290, taken
        to its end), the N - 1 matches + mode votes are ONE batch.  Rows as the incremental registrar's: (status, dx, dy, 0, 0, votes);
        like the pair loop, nothing behind the first pair that cannot be matched is reported (flowStitch breaks there)."""
        eng = self.engine
        feats, counts = eng.features_surf_batch(handles, self._surfParams(), self._enhanceSpec())
        try:
            rows = eng.features_match_offset_batch(feats[:-1], feats[1:], self.searchRatio, self.offsetEvaluate)
        finally:
            for f in feats:
                if f:
                    eng.features_free(f)
        table = []
        for r in rows:
            table.append([int(r[0]), int(r[1]), int(r[2]), 0, 0, int(r[3])])
            if not r[0]:
                break
        self.tempImageFeature.isBreak = True                  # the scan owns no cached set afterwards
        return table

    def _fullImageTableOrb(self, handles, shapes, chunk=16):
        """calculateOffsetForFeatureSearch (Stitcher.py:260-304) with featureMethod = "orb" over consecutive resident tiles: the N - 1
        whole-tile attempts (ORB of both tiles, BF-Hamming 1-NN, mode vote) as fused batches of `chunk` pairs.  A tile is described as B of
        one pair and again as A of the next -- the reference reuses B's features (Stitcher.py:278-290), which are the same numbers -- so the
        rows are those of the pair loop; nothing behind the first pair that cannot be matched is reported (flowStitch breaks there)."""
        eng = self.engine
        max_dist = self.orbMaxDistance if self.isGPUAvailable else -1
        table = []
        for c0 in range(0, len(handles) - 1, chunk):
            jobs = [(handles[k], handles[k + 1], 0, 0, 0, 0, shapes[k][0], shapes[k][1]) for k in range(c0, min(c0 + chunk, len(handles) - 1))]
            for r in eng.attempt_orb_batch(jobs, self._orbParams(), max_dist, self.offsetEvaluate):
                ok = bool(r[0]) and r[4] > 0 and r[5] > 0          # an image without keypoints: featuresX is None, status stays False
                table.append([int(ok), int(r[1]), int(r[2]), 0, 0, int(r[3])])
                if not ok:
                    self.tempImageFeature.isBreak = True
                    return table
        self.tempImageFeature.isBreak = True                  # the scan owns no cached set afterwards
        return table

    def flowStitchWithMutiple(self, fileList, caculateOffsetMethod):
        """Stitcher.py:96-127: restart after every registration break; a trailing lone tile is its own result."""
        result = []
        totalNum = len(fileList)
        startNum = 0
        self._ingestCache = {}                                # tiles decoded behind a break wait here for the segment that uses them
        try:
            while 1:
                (status, stitchResult) = self.flowStitch(fileList[startNum: totalNum], caculateOffsetMethod)
                result.append(stitchResult)
                self.tempImageFeature.isBreak = True
                startNum = startNum + status[1] + 1
                if startNum == totalNum:
                    break
                if startNum == (totalNum - 1):
                    result.append(self._loneTile(fileList[startNum]))
                    break
                self.printAndWrite("stitching Break, start from " + str(fileList[startNum]) + " again")
        finally:
            for g, c, _shape in self.__dict__.pop("_ingestCache", {}).values():
                for h in (g, c):
                    if h:
                        try:
                            self.engine.tile_free(h)
                        except Exception:
                            pass
        return result

    def _loneTile(self, path):
        """the trailing tile that is a result of its own (Stitcher.py:119-121): from the device copy a broken segment left behind, else decoded"""
        ent = (self.__dict__.get("_ingestCache") or {}).get(path)
        eng = self.engine
        if ent is not None and hasattr(eng, "canvas_paste_tile") and (bool(ent[1]) or not self.isColorMode):
            h, ch = (ent[1], 3) if self.isColorMode else (ent[0], 1)
            cv = eng.canvas_create(ent[2][0], ent[2][1], ch)
            try:
                eng.canvas_paste_tile(cv, h, 0, 0)
                return eng.canvas_download(cv, ent[2][0], ent[2][1], ch)
            finally:
                eng.canvas_free(cv)
        return _imread(path, self.isColorMode)

    def imageSetStitch(self, projectAddress, outputAddress, fileNum, caculateOffsetMethod, startNum=1, fileExtension="jpg", outputfileExtension="jpg"):
        """Stitcher.py:129-151."""
        for i in range(startNum, fileNum + 1):
            fileList = _list_images(_join(projectAddress, i), fileExtension)
            outDir = outputAddress.replace("\\", os.sep)
            if not os.path.exists(outDir):
                os.makedirs(outDir)
            Stitcher.outputAddress = outDir if outDir.endswith(os.sep) else outDir + os.sep
            outPath = os.path.join(outDir, "stitching_result_" + str(i) + "." + outputfileExtension)
            # streamed results are encoded under a hidden name and renamed when the mosaic is complete: an exception in the middle of
            # an assembly must not leave a truncated file under the result's name
            partPath = os.path.join(outDir, ".stitching_part_" + str(i) + "." + outputfileExtension)
            streamed = self._streamTo([partPath]) if self.streamOutput else None
            done = False
            try:
                (status, result) = self.flowStitch(fileList, caculateOffsetMethod)
                done = True
            finally:
                if streamed is not None:
                    self.__dict__.pop("mosaicSink", None)
                    if not done and os.path.exists(partPath):
                        os.remove(partPath)
            self.tempImageFeature.isBreak = True
            if result is None and streamed is not None:
                os.replace(partPath, outPath)
            else:
                if streamed is not None and os.path.exists(partPath):
                    os.remove(partPath)
                _imwrite(outPath, result)
            if status == False:
                self.printAndWrite("stitching Failed")

    def imageSetStitchWithMutiple(self, projectAddress, outputAddress, fileNum, caculateOffsetMethod, startNum=1, fileExtension="jpg", outputfileExtension="jpg"):
        """Stitcher.py:153-182 (the entry point Main.py:20-51 uses)."""
        for i in range(startNum, fileNum + 1):
            startTime = time.time()
            fileAddress = _join(projectAddress, i)
            fileList = _list_images(fileAddress, fileExtension)
            outDir = outputAddress.replace("\\", os.sep)
            if not os.path.exists(outDir):
                os.makedirs(outDir)
            Stitcher.outputAddress = outDir if outDir.endswith(os.sep) else outDir + os.sep   # printAndWrite appends the file name
            parts = []
            streamed = self._streamTo(parts, os.path.join(outDir, ".stitching_part_" + str(i) + "_%d." + outputfileExtension)) if self.streamOutput else None
            done = False
            try:
                result = self.flowStitchWithMutiple(fileList, caculateOffsetMethod)
                done = True
            finally:
                if streamed is not None:
                    self.__dict__.pop("mosaicSink", None)
                    if not done:                              # no part file outlives a failed dataset
                        for q in parts:
                            if os.path.exists(q):
                                os.remove(q)
            self.tempImageFeature.isBreak = True
            names = ([os.path.join(outDir, "stitching_result_" + str(i) + "." + outputfileExtension)] if len(result) == 1 else
                     [os.path.join(outDir, "stitching_result_" + str(i) + "_" + str(j + 1) + "." + outputfileExtension) for j in range(len(result))])
            streamedParts = iter(parts)
            for j in range(0, len(result)):
                if result[j] is None:                         # streamed while it was assembled: the part file takes the reference's name
                    os.replace(next(streamedParts), names[j])
                else:
                    _imwrite(names[j], result[j])
            endTime = time.time()
            print("Time Consuming for " + fileAddress + " is " + str(endTime - startTime))

    # ------------------------------------------------------------------------------------------------
    def calculateOffsetForPhaseCorrleate(self, dirAddress):
        """Stitcher.py:184-203 dereferences `self.phase`, which the reference never defines (dead code);
        the same AttributeError is raised here.  Use calculateOffsetForPhaseCorrleateIncre."""
        raise AttributeError("'Stitcher' object has no attribute 'phase'")

    # -- device tile cache: consecutive pairs share a tile (B of pair k is A of pair k+1) ---------------
    _TILE_CACHE = 4

    def _tileHandles(self, images):
        """Handles of the tiles of ONE job, resolved together: least-recently-used entries are evicted only after every
        tile of the job has its handle, and never a tile of the job itself.  An entry is keyed by the array object and its
        buffer address / shape / strides (a new array at a recycled id() must not hit); in-place edits of a cached array are
        not detected -- the reference's drivers decode a fresh array per tile."""
        cache = self.__dict__.setdefault("_tiles", [])
        eng = self.engine
        out = []
        for image in images:
            key = (id(image), image.__array_interface__["data"][0], image.shape, image.strides)
            for n, ent in enumerate(cache):
                if ent[0] is image and ent[1] == key:
                    cache.append(cache.pop(n))            # refresh on a hit (LRU)
                    out.append(ent[2])
                    break
            else:
                h = eng.tile_upload(image)
                cache.append((image, key, h))
                out.append(h)
        keep = set(out)
        n = 0
        while len(cache) > max(self._TILE_CACHE, len(keep)) and n < len(cache):
            if cache[n][2] in keep:
                n += 1
                continue
            eng.tile_free(cache.pop(n)[2])
        return out

    def releaseTiles(self):
        """Free the device copies the pair-by-pair methods cached (called at the end of every flowStitch)."""
        for _img, _key, h in self.__dict__.pop("_tiles", []):
            try:
                self.engine.tile_free(h)
            except Exception:
                pass
        if isinstance(self.tempImageFeature.feature, ResidentFeatures):
            # the device copy goes with the tiles; the cache keeps the (downloaded) arrays so that it still reads like the reference's
            rf = self.tempImageFeature.feature
            if rf.handle is not None:
                kps, desc = rf.arrays()
                rf.release()
                self.tempImageFeature.kps, self.tempImageFeature.feature = kps, desc

    def _usesStockOperators(self):
        c = type(self)
        return (c.detectAndDescribe is Utility.Method.detectAndDescribe and c.matchDescriptors is Utility.Method.matchDescriptors
                and c.getOffsetByMode is Utility.Method.getOffsetByMode and (not self.isEnhance or self.featureMethod == "surf")
                and self.featureMethod in ("surf", "orb") and self.offsetCaculate == "mode")

    def _enhanceSpec(self):
        """(mode, clipLimit, tileSize) of Stitcher.py:269-276 / 327-334: 0 none, 1 cv2.equalizeHist, 2 cv2.createCLAHE(...).apply."""
        if not self.isEnhance:
            return (0, 0.0, 0)
        return (2, float(self.clipLimit), int(self.tileSize)) if self.isClahe else (1, 0.0, 0)

    def _enhance(self, image):
        mode, clip, tiles = self._enhanceSpec()
        return image if mode == 0 else self.engine.enhance(np.asarray(image), mode, clip, tiles)

    def _featureAttempt(self, imageA, imageB, direction, searchRatio):
        """One pass of the loop body at Stitcher.py:322-345 -> (status, [dx, dy]) or None when an image has no features."""
        if self._usesStockOperators():
            ra = roi_rect(imageA.shape, direction, "first", searchRatio)
            rb = roi_rect(imageB.shape, direction, "second", searchRatio)
            if ra[2:] == rb[2:] and ra[2] > 0 and ra[3] > 0:
                ha, hb = self._tileHandles([imageA, imageB])
                job = (ha, hb, ra[0], ra[1], rb[0], rb[1], ra[2], ra[3])
                if self.featureMethod == "orb":
                    max_dist = self.orbMaxDistance if self.isGPUAvailable else -1
                    row = self.engine.attempt_orb_batch([job], self._orbParams(), max_dist, self.offsetEvaluate)[0]
                elif self.isEnhance:
                    row = self.engine.attempt_surf_batch_enhanced([job], self._surfParams(), self.searchRatio, self.offsetEvaluate, self._enhanceSpec())[0]
                else:
                    row = self.engine.attempt_surf_batch([job], self._surfParams(), self.searchRatio, self.offsetEvaluate)[0]
                if row[4] == 0 or row[5] == 0:
                    return None                      # featuresA is None or featuresB is None: status untouched
                return (bool(row[0]), [int(row[1]), int(row[2])])
        roiImageA = self.getROIRegionForIncreMethod(imageA, direction=direction, order="first", searchRatio=searchRatio)
        roiImageB = self.getROIRegionForIncreMethod(imageB, direction=direction, order="second", searchRatio=searchRatio)
        if self.isEnhance:                                     # Stitcher.py:327-334
            roiImageA = self._enhance(roiImageA)
            roiImageB = self._enhance(roiImageB)
        kpsA, featuresA = self.detectAndDescribe(roiImageA, featureMethod=self.featureMethod)
        kpsB, featuresB = self.detectAndDescribe(roiImageB, featureMethod=self.featureMethod)
        if featuresA is not None and featuresB is not None:
            matches = self.matchDescriptors(featuresA, featuresB)
            if self.offsetCaculate == "mode":
                return self.getOffsetByMode(kpsA, kpsB, matches, offsetEvaluate=self.offsetEvaluate)
            (status, offset, _adjustH) = self.getOffsetByRansac(kpsA, kpsB, matches, offsetEvaluate=self.offsetEvaluate)
            return (status, offset)
        return None

    def _phaseCorrelate(self, roiImageA, roiImageB):
        """cv2.phaseCorrelate(np.float64(a), np.float64(b)) at Stitcher.py:230 -> ((x, y), response)."""
        return self.engine.phase_correlate(roiImageA, roiImageB)

    def _axisCorrection(self, offset, localDirection, i, imageA, imageB):
        """Stitcher.py:244-251 / 353-360: put the ROI-relative vote back into full-tile coordinates."""
        if localDirection == 1:
            offset[0] = offset[0] + imageA.shape[0] - int(i * self.roiRatio * imageA.shape[0])
        elif localDirection == 2:
            offset[1] = offset[1] + imageA.shape[1] - int(i * self.roiRatio * imageA.shape[1])
        elif localDirection == 3:
            offset[0] = offset[0] - (imageB.shape[0] - int(i * self.roiRatio * imageB.shape[0]))
        elif localDirection == 4:
            offset[1] = offset[1] - (imageB.shape[1] - int(i * self.roiRatio * imageB.shape[1]))
        return offset

    def _maxI(self):
        return int(np.floor(0.5 / self.roiRatio) + 1) + 1     # Stitcher.py:217,316

    def calculateOffsetForPhaseCorrleateIncre(self, images):
        """Stitcher.py:205-258: incremental ROI x direction rotation around FP64 phase correlation.
        offset = [int(y), int(x)] (truncation), accepted when response > phaseResponseThreshold."""
        (imageA, imageB) = images
        offset = [0, 0]
        status = False
        iniDirection = self.direction
        localDirection = iniDirection
        for i in range(1, self._maxI()):
            while True:
                roiImageA = self.getROIRegionForIncreMethod(imageA, direction=localDirection, order="first", searchRatio=i * self.roiRatio)
                roiImageB = self.getROIRegionForIncreMethod(imageB, direction=localDirection, order="second", searchRatio=i * self.roiRatio)
                (offsetTemp, response) = self._phaseCorrelate(roiImageA, roiImageB)
                offset[0] = int(offsetTemp[1])
                offset[1] = int(offsetTemp[0])
                if self.phaseSignFix:                  # opt-in, not the reference's behaviour: see the class attribute
                    offset[0], offset[1] = -offset[0], -offset[1]
                if response > self.phaseResponseThreshold:
                    status = True
                if status == True:
                    break
                else:
                    localDirection = self.directionIncrease(localDirection)
                if localDirection == iniDirection:
                    break
            if status == True:
                offset = self._axisCorrection(offset, localDirection, i, imageA, imageB)
                self.direction = localDirection
                break
        if status == False:
            return (status, CANNOT_MATCH)
        self.printAndWrite("  The offset of stitching: dx is " + str(offset[0]) + " dy is " + str(offset[1]))
        return (status, offset)

    def calculateOffsetForFeatureSearch(self, images):
        """Stitcher.py:260-304: whole-tile features; tile B's features are reused as tile A's of the next pair."""
        (imageA, imageB) = images
        offset = [0, 0]
        status = False
        if self._usesStockOperators() and self.featureMethod == "surf" and hasattr(self.engine, "features_surf"):
            return self._featureSearchResident(imageA, imageB)
        if self.isEnhance == True:                              # Stitcher.py:269-276
            imageA = self._enhance(imageA)
            imageB = self._enhance(imageB)
        if self.tempImageFeature.isBreak == True:
            (kpsA, featuresA) = self.detectAndDescribe(imageA, featureMethod=self.featureMethod)
            (kpsB, featuresB) = self.detectAndDescribe(imageB, featureMethod=self.featureMethod)
        else:
            kpsA = self.tempImageFeature.kps
            featuresA = self.tempImageFeature.feature
            (kpsB, featuresB) = self.detectAndDescribe(imageB, featureMethod=self.featureMethod)
        self.tempImageFeature.isBreak = False
        self.tempImageFeature.kps = kpsB
        self.tempImageFeature.feature = featuresB
        if featuresA is not None and featuresB is not None:
            matches = self.matchDescriptors(featuresA, featuresB)
            if self.offsetCaculate == "mode":
                (status, offset) = self.getOffsetByMode(kpsA, kpsB, matches, offsetEvaluate=self.offsetEvaluate)
            elif self.offsetCaculate == "ransac":
                (status, offset, adjustH) = self.getOffsetByRansac(kpsA, kpsB, matches, offsetEvaluate=self.offsetEvaluate)
        if status == False:
            self.tempImageFeature.isBreak = True
            return (status, CANNOT_MATCH)
        self.tempImageFeature.isBreak = False
        self.printAndWrite("  The offset of stitching: dx is " + str(offset[0]) + " dy is " + str(offset[1]))
        return (status, offset)

    def _featureSearchResident(self, imageA, imageB):
        """calculateOffsetForFeatureSearch with the stock SURF operators, device resident: tile B's keypoints + descriptors stay in
        HBM and become tile A's of the next pair (Stitcher.py:278-290), the optional enhancement (Stitcher.py:269-276) runs on the
        device in front of the detector, and matching + mode vote read both sets where they are -- eight ints return per pair."""
        eng = self.engine
        tf = self.tempImageFeature
        params, enh = self._surfParams(), self._enhanceSpec()
        dim = 128 if params.extended else 64

        def describe(image):
            (h,) = self._tileHandles([image])
            f, n = eng.features_surf(h, (0, 0, image.shape[0], image.shape[1]), params, enh)
            return ResidentFeatures(eng, f, n, dim)
        prev = tf.feature if isinstance(tf.feature, ResidentFeatures) and tf.feature.handle is not None else None
        if tf.isBreak == True or prev is None:
            if prev is not None:
                prev.release()
            featA = describe(imageA)
        else:
            featA = prev
        try:
            featB = describe(imageB)
        except Exception:
            featA.release()                                     # nothing cached may point at a set that is gone
            tf.isBreak, tf.kps, tf.feature = True, None, None
            raise
        tf.isBreak = False
        tf.kps = featB if featB.n else np.float32([])
        tf.feature = featB if featB.n else None            # cv2 returns (kps, None) for an image without keypoints
        status, offset = False, [0, 0]
        try:
            if featA.n and featB.n:
                row = eng.features_match_offset(featA.handle, featB.handle, self.searchRatio, self.offsetEvaluate)
                status, offset = bool(row[0]), [int(row[1]), int(row[2])]
        finally:
            featA.release()                                     # A's set is never needed again (B's lives on in the cache)
            if not featB.n:
                featB.release()
        if status == False:
            tf.isBreak = True
            return (status, CANNOT_MATCH)
        tf.isBreak = False
        self.printAndWrite("  The offset of stitching: dx is " + str(offset[0]) + " dy is " + str(offset[1]))
        return (status, offset)

    def calculateOffsetForFeatureSearchIncre(self, images):
        """Stitcher.py:306-367: for i in 1..maxI-1 { rotate directions until one attempt votes >= offsetEvaluate }."""
        (imageA, imageB) = images
        offset = [0, 0]
        status = False
        iniDirection = self.direction
        localDirection = iniDirection
        for i in range(1, self._maxI()):
            while True:
                res = self._featureAttempt(imageA, imageB, localDirection, i * self.roiRatio)
                if res is not None:
                    (status, offset) = res
                    offset = list(offset)
                if status:
                    break
                else:
                    localDirection = self.directionIncrease(localDirection)
                if localDirection == iniDirection:
                    break
            if status:
                offset = self._axisCorrection(offset, localDirection, i, imageA, imageB)
                self.direction = localDirection
                break
        if status == False:
            return (status, CANNOT_MATCH)
        self.printAndWrite("  The offset of stitching: dx is " + str(offset[0]) + " dy is " + str(offset[1]))
        return (status, offset)

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _layout(shapes, originOffsetList):
        """Stitcher.py:387-431: canvas-relative tile origins, running bounding boxes, canvas size.
        originOffsetList already carries the leading [0, 0]."""
        n = len(originOffsetList)
        offsetList = copy.deepcopy(originOffsetList)
        rangeX = [[0, 0] for _ in range(n)]
        rangeY = [[0, 0] for _ in range(n)]
        resultRow, resultCol = shapes[0][0], shapes[0][1]
        rangeX[0][1] = resultRow
        rangeY[0][1] = resultCol
        dxSum = dySum = 0
        for i in range(1, n):
            th, tw = shapes[i][0], shapes[i][1]
            dxSum += offsetList[i][0]
            dySum += offsetList[i][1]
            for (axis, total, ranges, tlen) in ((0, dxSum, rangeX, th), (1, dySum, rangeY, tw)):
                size = resultRow if axis == 0 else resultCol
                if total <= 0:
                    shift = abs(total)
                    for j in range(0, i):
                        offsetList[j][axis] += shift
                        ranges[j][0] += shift
                        ranges[j][1] += shift
                    size += shift
                    ranges[i][1] = size
                    ranges[i][0] = offsetList[i][axis] = 0
                    if axis == 0:
                        dxSum = 0
                    else:
                        dySum = 0
                else:
                    offsetList[i][axis] = total
                    size = max(size, total + tlen)
                    ranges[i][1] = size
                if axis == 0:
                    resultRow = size
                else:
                    resultCol = size
        return offsetList, rangeX, rangeY, resultRow, resultCol

    def getStitchByOffset(self, fileList, originOffsetList):
        """Stitcher.py:369-486.  NB: like the reference this inserts [0, 0] at the head of the caller's list."""
        color = self.isColorMode
        originOffsetList.insert(0, [0, 0])
        n = len(originOffsetList)
        eng = self.engine
        # tiles the batched registration left in HBM (the gray planes, or the B G R tiles of the same decode): fused from where they are
        resident = self.__dict__.pop("_resident", None) or {}
        device_fuse = self.fuseMethod in ("notFuse", "fadeInAndFadeOut", "trigonometric")
        fmethod = 1 if self.fuseMethod == "trigonometric" else 0
        simple = {"average": 0, "maximum": 1, "minimum": 2}.get(self.fuseMethod)
        if simple is not None and not hasattr(eng, "canvas_blend_tile"):
            simple = None
        handles = imageList = None
        use_res = (device_fuse or (simple is not None and hasattr(eng, "canvas_blend_tile_resident"))) and \
            all(fileList[i] in resident and (len(resident[fileList[i]][1]) == 3) == color for i in range(n))
        try:
            if not use_res and (device_fuse or simple is not None) and hasattr(eng, "tile_reserve") and hasattr(eng, "tile_fill_pair") and \
                    (simple is None or hasattr(eng, "canvas_blend_tile_resident")):
                # not (all) resident -- a custom registration method ran pair by pair on host arrays: the mosaic's tiles go straight from a
                # pool of decoder threads into reserved device tiles, never as a list on the host (the reference holds all of them,
                # Stitcher.py:382-403)
                for h, _shape in resident.values():
                    eng.tile_free(h)
                resident = {}
                resident = self._ingestForMosaic([fileList[i] for i in range(n)], color) or {}
                use_res = bool(resident)
            if use_res:
                shapes = [resident[fileList[i]][1] for i in range(n)]
                handles = [resident[fileList[i]][0] for i in range(n)]
            else:
                imageList = [_imread(fileList[0], color)]
                for i in range(1, n):
                    imageList.append(_imread(fileList[i], Stitcher.isColorMode))
                shapes = [im.shape for im in imageList]
                if device_fuse and hasattr(eng, "tile_upload_color") and len({im.ndim for im in imageList}) == 1:
                    # engines without the ingest entry points: all uploads are queued on the copy stream up front and the per-tile canvas
                    # calls below only enqueue work behind them -- no host synchronisation per tile (Stitcher.py:174-179, 434-483)
                    for h, _shape in resident.values():
                        eng.tile_free(h)
                    resident = {}
                    for i in range(n):
                        im = np.ascontiguousarray(imageList[i])
                        hnd = eng.tile_upload_color(im, asynchronous=True) if im.ndim == 3 else eng.tile_upload_async(im)
                        resident[(i, fileList[i])] = (hnd, im.shape)
                    handles = [resident[(i, fileList[i])][0] for i in range(n)]
                    use_res = True
            offsetList, rangeX, rangeY, resultRow, resultCol = self._layout(shapes, originOffsetList)
            self.printAndWrite("  The rectified offsetList is " + str(offsetList))
            if not device_fuse and simple is None:
                return self._stitchWithHostFuse(fileList, imageList, originOffsetList, offsetList, rangeX, rangeY, resultRow, resultCol)
            ch = 3 if color else 1
            canvas = eng.canvas_create(resultRow, resultCol, ch)
            try:
                one_call = use_res and hasattr(eng, "canvas_assemble_resident")
                if one_call:
                    # every tile is resident: the walk below as ONE library call (the per-tile calls cost the host more than
                    # their two launches cost the device)
                    geom = np.zeros((n, 9), np.int32)
                    for i in range(0, n):
                        self.printAndWrite("  stitching " + str(fileList[i]))
                        th, tw = shapes[i][0], shapes[i][1]
                        oy, ox = offsetList[i][0], offsetList[i][1]
                        if i == 0 or self.fuseMethod == "notFuse":
                            geom[i] = (oy, ox, 0, 0, 0, 0, 0, 0, -1)
                        else:
                            geom[i] = (oy, ox, max(oy, rangeX[i - 1][0]), max(ox, rangeY[i - 1][0]), min(oy + th, rangeX[i - 1][1]),
                                       min(ox + tw, rangeY[i - 1][1]), originOffsetList[i][0], originOffsetList[i][1],
                                       fmethod if simple is None else 2 + simple)
                    eng.canvas_assemble_resident(canvas, handles, geom)
                for i in range(n if one_call else 0, n):
                    self.printAndWrite("  stitching " + str(fileList[i]))
                    th, tw = shapes[i][0], shapes[i][1]
                    oy, ox = offsetList[i][0], offsetList[i][1]
                    if i == 0 or self.fuseMethod == "notFuse":
                        if use_res:
                            eng.canvas_paste_tile(canvas, handles[i], oy, ox)
                        else:
                            eng.canvas_paste(canvas, imageList[i], oy, ox)
                        continue
                    roi = (max(oy, rangeX[i - 1][0]), max(ox, rangeY[i - 1][0]),
                           min(oy + th, rangeX[i - 1][1]), min(ox + tw, rangeY[i - 1][1]))
                    if simple is not None and use_res:
                        eng.canvas_blend_tile_resident(canvas, handles[i], oy, ox, roi, simple)
                    elif simple is not None:
                        eng.canvas_blend_tile(canvas, imageList[i], oy, ox, roi, simple)
                    elif use_res:
                        eng.canvas_fuse_tile_resident(canvas, handles[i], oy, ox, roi, originOffsetList[i][0], originOffsetList[i][1], method=fmethod)
                    else:
                        eng.canvas_fuse_tile(canvas, imageList[i], oy, ox, roi, originOffsetList[i][0], originOffsetList[i][1], method=fmethod)
                sink = getattr(self, "mosaicSink", None)
                if sink is not None and hasattr(eng, "canvas_download_bands"):
                    # streamed write-out: the mosaic leaves the device band by band and is never whole in host memory
                    # (a sink that is done with a band two bands later -- `transient_bands`: JpegBandWriter -- gets views of the engine's
                    #  pinned band ring instead of a fresh array per band: a DMA at link speed, no page faults of a fresh 300 MB array per
                    #  band; VFSMS_PINNED_BANDS=0 turns it off)
                    transient = bool(getattr(sink, "transient_bands", False)) and os.environ.get("VFSMS_PINNED_BANDS", "1") == "1"
                    kw = {"transient": True} if transient else {}
                    for r0, band in eng.canvas_download_bands(canvas, resultRow, resultCol, ch, int(getattr(self, "mosaicBandRows", 4096)), **kw):
                        sink(r0, band, (resultRow, resultCol, ch) if ch > 1 else (resultRow, resultCol))
                    return None
                return eng.canvas_download(canvas, resultRow, resultCol, ch)
            finally:
                eng.canvas_free(canvas)
        finally:
            if hasattr(eng, "sync_uploads"):
                eng.sync_uploads()                           # the host tiles of asynchronous uploads may be released now
            for h, _shape in resident.values():
                eng.tile_free(h)

    def _ingestForMosaic(self, files, color):
        """The mosaic's tiles from their files into reserved device tiles through a pool of decoder threads (one decode per file, the
        colour conversion on the GPU) -> {file: (handle, shape)}."""
        eng = self.engine
        if len(set(files)) != len(files):                    # (a file listed twice: the host path keys nothing by name)
            return None
        shapes = [_imshape(f) for f in files]
        handles = []
        try:
            for s_ in shapes:
                handles.append(eng.tile_reserve_color(s_[0], s_[1], 3) if color else eng.tile_reserve(s_[0], s_[1]))

            def ingest(k):
                try:
                    if _fill_from_jpeg(eng, files[k], 0 if color else handles[k], handles[k] if color else 0):
                        return
                    owner, shape, parts = _decode_once(files[k], color)
                    if tuple(shape) != tuple(shapes[k]):
                        raise ValueError("decoded size %s of %s differs from its header %s" % (shape, files[k], shapes[k]))
                    if parts[0] == "src":
                        eng.tile_fill_pair(0 if color else handles[k], handles[k] if color else 0, parts[1], parts[2], parts[3])
                    else:
                        eng.tile_fill(handles[k], parts[2] if color else parts[1])
                    del owner
                except BaseException:
                    try:
                        eng.tile_fill(handles[k], None)
                    except Exception:
                        pass
                    raise
            nthreads = self._decoderThreads(len(files))
            with _PillowBlocks(color):
                futs = [_decoder_pool(nthreads).submit(ingest, k) for k in range(len(files))]
                first = None
                for fu in futs:                               # every decoder has finished with its handle before anything is given up
                    try:
                        fu.result()
                    except BaseException as e:                 # noqa: PERF203
                        first = first or e
                if first is not None:
                    raise first
        except BaseException:
            for h in handles:
                try:
                    eng.tile_fill(h, None)
                except Exception:
                    pass
                try:
                    eng.tile_free(h)
                except Exception:
                    pass
            raise
        return dict(zip(files, zip(handles, [(s_[0], s_[1], 3) if color else s_ for s_ in shapes])))

    def _stitchWithHostFuse(self, fileList, imageList, originOffsetList, offsetList, rangeX, rangeY, resultRow, resultCol):
        """Fallback for engines without the canvas blend entry points (the CPU test doubles): same int64 / -1 canvas walk as
        Stitcher.py:434-486, blend through self.fuseImage."""
        color = self.isColorMode
        shape = (resultRow, resultCol, 3) if color else (resultRow, resultCol)
        stitchResult = np.zeros(shape, np.int64) - 1
        for i in range(0, len(offsetList)):
            self.printAndWrite("  stitching " + str(fileList[i]))
            tile = imageList[i]
            oy, ox = offsetList[i][0], offsetList[i][1]
            if i == 0:
                stitchResult[oy:oy + tile.shape[0], ox:ox + tile.shape[1]] = tile
                continue
            roi_ltx = max(oy, rangeX[i - 1][0]); roi_lty = max(ox, rangeY[i - 1][0])
            roi_rbx = min(oy + tile.shape[0], rangeX[i - 1][1]); roi_rby = min(ox + tile.shape[1], rangeY[i - 1][1])
            roiImageRegionA = stitchResult[roi_ltx:roi_rbx, roi_lty:roi_rby].copy()
            stitchResult[oy:oy + tile.shape[0], ox:ox + tile.shape[1]] = tile
            roiImageRegionB = stitchResult[roi_ltx:roi_rbx, roi_lty:roi_rby].copy()
            stitchResult[roi_ltx:roi_rbx, roi_lty:roi_rby] = self.fuseImage([roiImageRegionA, roiImageRegionB],
                                                                            originOffsetList[i][0], originOffsetList[i][1])
        stitchResult[stitchResult == -1] = 0
        return stitchResult.astype(np.uint8)

    def fuseImage(self, images, dx, dy):
        """Stitcher.py:488-525: dispatch on fuseMethod (int64 regions with -1 = empty)."""
        self.imageFusion.isColorMode = self.isColorMode
        self.imageFusion._engine = self._engine
        (imageA, imageB) = images
        if self.fuseMethod != "fadeInAndFadeOut" and self.fuseMethod != "trigonometric":
            imageA[imageA == -1] = 0
            imageB[imageB == -1] = 0
            imageA[imageA == 0] = imageB[imageA == 0]
            imageB[imageB == 0] = imageA[imageB == 0]
        if self.fuseMethod == "notFuse":
            return imageB
        if self.fuseMethod == "average":
            return self.imageFusion.fuseByAverage([imageA, imageB])
        if self.fuseMethod == "maximum":
            return self.imageFusion.fuseByMaximum([imageA, imageB])
        if self.fuseMethod == "minimum":
            return self.imageFusion.fuseByMinimum([imageA, imageB])
        if self.fuseMethod == "fadeInAndFadeOut":
            return self.imageFusion.fuseByFadeInAndFadeOut(images, dx, dy)
        if self.fuseMethod == "trigonometric":
            return self.imageFusion.fuseByTrigonometric(images, dx, dy)
        if self.fuseMethod in ("multiBandBlending", "optimalSeamLine"):
            raise NotImplementedError("fuseMethod %r is outside the VFSMS hot path (gray-only / interactive in the reference, ImageFusion.py:296-492)" % self.fuseMethod)
        return np.zeros(imageA.shape, np.uint8)

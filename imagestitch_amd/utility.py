"""Operator layer of the VFSMS hot path -- host-side mirror of the reference's `ImageUtility.Method`.

Same class / attribute / method names, argument meaning and return conventions as
/root/reference/ImageUtility.py (cited per method), so `Stitcher` code written against the reference runs
unchanged; every operator dispatches into libvfsms.so (hand-written HIP for MI355X) through
imagestitch_amd._lib.Engine.  There is no cv2 and no CPU fallback behind these methods.
"""
import math

import numpy as np

from . import _lib


def roi_rect(shape, direction=1, order="first", searchRatio=0.1):
    """(y0, x0, h, w) of the strip Method.getROIRegionForIncreMethod slices (ImageUtility.py:66-101).

    direction 1: A bottom / B top; 2: A right / B left; 3: A top / B bottom; 4: A left / B right, where
    order "first" is image A and "second" image B.  The length is floor(len * searchRatio) evaluated in
    float64 exactly like np.floor(row * searchRatio).astype(int) (3*0.2 = 0.6000000000000001 matters).
    """
    row, col = int(shape[0]), int(shape[1])
    if direction in (1, 3):
        n = int(math.floor(row * searchRatio))
        at_end = (direction == 1) == (order == "first")
        if order not in ("first", "second"):
            return (0, 0, row, col)
        return (row - n, 0, n, col) if at_end else (0, 0, n, col)
    if direction in (2, 4):
        n = int(math.floor(col * searchRatio))
        at_end = (direction == 2) == (order == "first")
        if order not in ("first", "second"):
            return (0, 0, row, col)
        return (0, col - n, row, n) if at_end else (0, 0, row, n)
    return (0, 0, row, col)


class Method():
    # ---- logging (ImageUtility.py:8-12) ----
    outputAddress = "result/"
    isEvaluate = False
    evaluateFile = "evaluate.txt"
    isPrintLog = True

    # ---- feature search (ImageUtility.py:14-17) ----
    featureMethod = "surf"      # "sift", "surf" or "orb"
    roiRatio = 0.1
    searchRatio = 0.75

    # ---- backend switch (ImageUtility.py:19-20).  Both values run on the MI355X here; the flag keeps the
    # reference's meaning of WHICH parameter set is used: False -> cv2 defaults (SURF 100/4/3, 64-d;
    # ORB without distance threshold), True -> the surf*/orb* attributes below (128-d SURF, orbMaxDistance).
    isGPUAvailable = False

    # ---- SURF parameters of the DLL path (ImageUtility.py:22-28) ----
    surfHessianThreshold = 100.0
    surfNOctaves = 4
    surfNOctaveLayers = 3
    surfIsExtended = True
    surfKeypointsRatio = 0.01
    surfIsUpright = False

    # ---- ORB parameters (ImageUtility.py:30-40) ----
    orbNfeatures = 5000
    orbScaleFactor = 1.2
    orbNlevels = 8
    orbEdgeThreshold = 31
    orbFirstLevel = 0
    orbWTA_K = 2
    orbPatchSize = 31
    orbFastThreshold = 20
    orbBlurForDescriptor = False
    orbMaxDistance = 30

    # ---- registration (ImageUtility.py:42-44) ----
    offsetCaculate = "mode"     # "mode" or "ransac"
    offsetEvaluate = 3

    # ---- enhancement (ImageUtility.py:46-50; CLAHE/equalizeHist are out of the hot-path scope) ----
    isEnhance = False
    isClahe = False
    clipLimit = 20
    tileSize = 5

    # engine injection point (tests substitute fakes; production resolves the per-process GPU engine)
    _engine = None

    @property
    def engine(self):
        eng = self._engine
        if eng is None:
            eng = _lib.default_engine()
        return eng

    # ------------------------------------------------------------------------------------------------
    def printAndWrite(self, content):
        """ImageUtility.py:52-64: print if isPrintLog; append to outputAddress+evaluateFile if isEvaluate."""
        if self.isPrintLog:
            print(content)
        if self.isEvaluate:
            with open(self.outputAddress + self.evaluateFile, "a") as f:
                f.write(content)
                f.write("\n")

    def getROIRegionForIncreMethod(self, image, direction=1, order="first", searchRatio=0.1):
        """ImageUtility.py:66-101: the search strip as a numpy VIEW of `image` (no copy)."""
        if direction not in (1, 2, 3, 4) or order not in ("first", "second"):
            return np.zeros(image.shape, np.uint8)          # the reference's untouched initial value
        y0, x0, h, w = roi_rect(image.shape, direction, order, searchRatio)
        return image[y0:y0 + h, x0:x0 + w]

    def getOffsetByMode(self, kpsA, kpsB, matches, offsetEvaluate=10):
        """ImageUtility.py:139-178 -> (status, [dx, dy]).  matches: [(trainIdx, queryIdx)].
        dx/dy = int() of float32 differences (truncation), (0,0) votes dropped, mode with first-seen
        tie-break, status = count >= offsetEvaluate; empty matches -> (False, [0, 0])."""
        if len(matches) == 0:
            return (False, [0, 0])
        status, off, _votes = self.engine.mode_offset(np.asarray(kpsA, np.float32), np.asarray(kpsB, np.float32),
                                                      np.asarray(matches, np.int32), offsetEvaluate)
        return (status, off)

    def getOffsetByRansac(self, kpsA, kpsB, matches, offsetEvaluate=100):
        """ImageUtility.py:180-210 is marked incomplete by the reference (its getAffineTransform call raises
        for != 3 points) and Main.py never selects it; not part of the accelerated path."""
        raise NotImplementedError("offsetCaculate='ransac' is outside the VFSMS hot path (reference: ImageUtility.py:180-210, incomplete)")

    # -- array adapters of the DLL path (ImageUtility.py:212-246): kept for API compatibility ------------
    def npToListForKeypoints(self, array):
        return [[array[i, 0], array[i, 1]] for i in range(array.shape[0])]

    def npToListForMatches(self, array):
        return [(array[i, 0], array[i, 1]) for i in range(array.shape[0])]

    def npToKpsAndDescriptors(self, array):
        """float32[N, D, 2] packing of appendix/myGpuFeatures.cpp:16-51: [i,0,0]=x, [i,1,0]=y, [i,:,1]=descriptor."""
        return ([[array[i, 0, 0], array[i, 1, 0]] for i in range(array.shape[0])], array[:, :, 1])

    # ------------------------------------------------------------------------------------------------
    def _surfParams(self):
        if self.isGPUAvailable:
            return self.engine.surf_params(self.surfHessianThreshold, self.surfNOctaves, self.surfNOctaveLayers,
                                           self.surfIsExtended, self.surfIsUpright)
        return self.engine.surf_params()      # cv2.xfeatures2d.SURF_create() defaults (ImageUtility.py:258)

    def _orbParams(self):
        """cv2.ORB_create(orbNfeatures, orbScaleFactor, orbNlevels, orbEdgeThreshold, orbFirstLevel, orbWTA_K, 0, orbPatchSize,
        orbFastThreshold) -- ImageUtility.py:260 (both backends of the reference pass the same Method.orb* attributes)."""
        return self.engine.orb_params(self.orbNfeatures, self.orbScaleFactor, self.orbNlevels, self.orbEdgeThreshold,
                                      self.orbFirstLevel, self.orbWTA_K, 0, self.orbPatchSize, self.orbFastThreshold)

    def detectAndDescribe(self, image, featureMethod):
        """ImageUtility.py:248-276 -> (kps float32[N,2] of (x, y), features float32[N,D] or None)."""
        if featureMethod == "surf":
            kps, feats = self.engine.surf_detect_describe(np.asarray(image), self._surfParams())
            if len(kps) == 0:
                return (np.float32([]), None)      # cv2 returns ([], None) for an image without keypoints
            return (kps, feats)
        if featureMethod == "orb":
            kps, feats = self.engine.orb_detect_describe(np.asarray(image), self._orbParams())
            if len(kps) == 0:
                return (np.float32([]), None)
            return (kps, feats)
        raise NotImplementedError("featureMethod %r is outside the VFSMS hot path (sift is CPU-only in the reference too)" % (featureMethod,))

    def matchDescriptors(self, featuresA, featuresB):
        """ImageUtility.py:278-309 -> [(trainIdx, queryIdx)] in query order."""
        if self.featureMethod in ("surf", "sift"):
            pairs = self.engine.bf_l2_ratio_matches(featuresA, featuresB, self.searchRatio)
        elif self.featureMethod == "orb":
            max_dist = self.orbMaxDistance if self.isGPUAvailable else -1
            pairs = self.engine.bf_hamming_matches(featuresA, featuresB, max_dist)
        else:
            raise NotImplementedError("featureMethod %r" % (self.featureMethod,))
        return [(int(t), int(q)) for t, q in pairs]

"""Overlap blending -- host-side mirror of the reference's `ImageFusion.ImageFusion`
(/root/reference/ImageFusion.py).  Regions are int64 arrays with -1 = empty, exactly what
Stitcher.getStitchByOffset hands to fuseImage (Stitcher.py:434-436,475-483).

On the hot path (north_star): fuseByFadeInAndFadeOut + getWeightsMatrix -> HIP (csrc/fuse_kernels.hip).
fuseByAverage / Maximum / Minimum / Trigonometric are listed OUT OF SCOPE for kernels in SURVEY section 2
(rows 7-8); as array operators they are kept as numpy one-liners so the `fuseMethod` switch keeps working.  (Inside
Stitcher.getStitchByOffset the first three run on the device canvas: vfsms_canvas_blend_tile.)
"""
import math

import numpy as np

from . import utility as Utility


class ImageFusion(Utility.Method):

    isColorMode = False

    # ---- trivial element-wise fuses (ImageFusion.py:12-41) ------------------------------------------
    def fuseByAverage(self, images):
        (imageA, imageB) = images
        return np.uint8((imageA.astype(int) + imageB.astype(int)) / 2)

    def fuseByMaximum(self, images):
        (imageA, imageB) = images
        return np.maximum(imageA, imageB)

    def fuseByMinimum(self, images):
        (imageA, imageB) = images
        return np.minimum(imageA, imageB)

    # ---- fade in / fade out (ImageFusion.py:192-244) --------------------------------------------------
    def fuseByFadeInAndFadeOut(self, images, dx, dy):
        """Linear-ramp blend: strip mode when more than 65 % of A is occupied (ramp along the short side,
        orientation by sign of dy / dx; the reference's weights sum to (n-1)/n or (n+1)/n, not 1), else
        corner mode through getWeightsMatrix.  float32 weights x int -> float64, clamp, truncate to uint8.
        Like the reference, empty pixels of imageA are filled from imageB IN PLACE."""
        (imageA, imageB) = images
        out = self.engine.fuse_fade_i64(imageA, imageB, dx, dy)
        hole = imageA < 0
        imageA[hole] = imageB[hole]
        return out

    def getWeightsMatrix(self, images):
        """ImageFusion.py:43-190 -> (weightMatA, weightMatB) float32, weightMatB = rowRamp x colRamp."""
        (imageA, _imageB) = images
        ramps, _info = self.engine.fuse_ramps_i64(imageA, 0, 0, force_corner=True)
        wBr, wBc = ramps[1], ramps[3]
        shape = imageA.shape
        if imageA.ndim == 3:
            weightMatB = (wBr[:, None, None] * wBc[None, :, None]) * np.ones(shape, np.float32)
        else:
            weightMatB = wBr[:, None] * wBc[None, :]
        weightMatB = weightMatB.astype(np.float32)
        return (np.float32(1) - weightMatB, weightMatB)

    # ---- trigonometric re-weighting (ImageFusion.py:246-293; "next" row f-4 of the scope table) --------
    def fuseByTrigonometric(self, images, dx, dy):
        (imageA, imageB) = images
        row, col = imageA.shape[:2]
        tail = (1,) * (imageA.ndim - 2)
        if np.count_nonzero(imageA > -1) / imageA.size > 0.65:
            weightMatA = np.ones(imageA.shape, dtype=np.float64)
            if col <= row:
                k = np.arange(col, dtype=np.float64)
                ramp = (k if dy >= 0 else (col - k)) * 1.0 / col
                weightMatA = weightMatA * ramp.reshape((1, col) + tail)
            else:
                k = np.arange(row, dtype=np.float64)
                ramp = (k if dx <= 0 else (row - k)) * 1.0 / row
                weightMatA = weightMatA * ramp.reshape((row, 1) + tail)
        else:
            weightMatA, _ = self.getWeightsMatrix(images)
        weightMatA = np.power(np.sin(weightMatA * math.pi / 2), 2)
        weightMatB = 1 - weightMatA
        hole = imageA < 0
        imageA[hole] = imageB[hole]
        result = weightMatA * imageA.astype(np.int64) + weightMatB * imageB.astype(np.int64)
        result[result < 0] = 0
        result[result > 255] = 255
        return np.uint8(result)

    def fuseByMultiBandBlending(self, images):
        raise NotImplementedError("multi-band blending (ImageFusion.py:296-367) is outside the VFSMS hot path")

    def fuseByOptimalSeamLine(self, images, direction="horizontal"):
        raise NotImplementedError("optimal seam line (ImageFusion.py:377-492, interactive cv2.imshow) is outside the VFSMS hot path")

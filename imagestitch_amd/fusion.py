"""Overlap blending -- host-side mirror of the reference's `ImageFusion.ImageFusion`
(/root/reference/ImageFusion.py).  Regions are int64 arrays with -1 = empty, exactly what
Stitcher.getStitchByOffset hands to fuseImage (Stitcher.py:434-436,475-483).

On the hot path (north_star): fuseByFadeInAndFadeOut + getWeightsMatrix -> HIP (csrc/fuse_kernels.hip); fuseByTrigonometric
shares their kernels (scope row f-4).  fuseByAverage / Maximum / Minimum are numpy one-liners as array operators (inside
Stitcher.getStitchByOffset they run on the device canvas: vfsms_canvas_blend_tile).
"""
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

    # ---- trigonometric re-weighting (ImageFusion.py:246-293) --------------------------------------------
    def fuseByTrigonometric(self, images, dx, dy):
        """wA = sin(w pi / 2)^2 of the fade's ramps (float64 i / n in the strip modes, getWeightsMatrix's float32 matrix in corner
        mode), wB = 1 - wA, on the device (vfsms_fuse_trig_i64).  sin is the library's explicit double-precision routine; the
        reference's bytes hang on numpy's own sin in the last ulp, so single grey levels may differ on < 0.1 % of the pixels.
        Like the reference, empty pixels of imageA are filled from imageB IN PLACE."""
        (imageA, imageB) = images
        out = self.engine.fuse_trig_i64(imageA, imageB, dx, dy)
        hole = imageA < 0
        imageA[hole] = imageB[hole]
        return out

    def fuseByMultiBandBlending(self, images):
        raise NotImplementedError("multi-band blending (ImageFusion.py:296-367) is outside the VFSMS hot path")

    def fuseByOptimalSeamLine(self, images, direction="horizontal"):
        raise NotImplementedError("optimal seam line (ImageFusion.py:377-492, interactive cv2.imshow) is outside the VFSMS hot path")

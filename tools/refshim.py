"""Container-only harness: import the reference's Python modules from /root/reference.

The reference (Keep-Passion/ImageStitch) cannot be imported as-is here:
  * `import cv2` fails (opencv-python 3.3.1 is not installed and cannot be),
  * ImageUtility.py:4 hard-imports the Windows DLL `myGpuFeatures`,
  * it uses `np.int`, removed from numpy >= 1.24.
This module installs three shims (a stub `cv2` carrying only what the probed pure-numpy functions touch,
a stub `myGpuFeatures`, `np.int = int`) and imports Stitcher / ImageFusion / ImageUtility.

It is used ONLY by tools/capture_golden.py, in this container, to produce small data fixtures under
tests/golden/.  Nothing in tests/, bench.py or the product imports it; the reference's files never
travel to the GPU box and are never copied into this repository.
"""
import io
import sys
import types

import numpy as np

REF = "/root/reference"


def install(phase_correlate=None):
    np.int = int  # numpy<1.24 alias: `np.int is int`
    cv2 = types.ModuleType("cv2")
    cv2.INTER_AREA = 3
    cv2.IMREAD_GRAYSCALE = 0
    cv2.IMREAD_COLOR = 1

    def imdecode(buf, flag):
        from PIL import Image
        im = Image.open(io.BytesIO(np.asarray(buf, np.uint8).tobytes()))
        if flag == cv2.IMREAD_GRAYSCALE:
            return np.asarray(im.convert("L"))
        a = np.asarray(im.convert("RGB"))
        return a[:, :, ::-1].copy()  # BGR like OpenCV

    cv2.imdecode = imdecode
    cv2._written = []
    cv2.imwrite = lambda path, img: cv2._written.append((path, np.array(img)))
    if phase_correlate is not None:
        cv2.phaseCorrelate = phase_correlate
    sys.modules["cv2"] = cv2
    gpu = types.ModuleType("myGpuFeatures")
    gpu.myGpuFeatures = types.SimpleNamespace()
    sys.modules["myGpuFeatures"] = gpu
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import ImageFusion  # noqa: E402
    import ImageUtility  # noqa: E402
    import Stitcher  # noqa: E402
    return cv2, Stitcher, ImageFusion, ImageUtility

"""A second, independent float64 restatement of cv2.phaseCorrelate(src1, src2) (Stitcher.py:230; OpenCV 3.3.1 imgproc/src/phasecorr.cpp,
SURVEY.md Appendix A.1) in numpy: half spectra through numpy's pocketfft (rfft2 / irfft2), vectorised per-bin arithmetic, slicing
instead of loops.  TEST INFRASTRUCTURE ONLY: it shares no code with oracle/vfsms_oracle.c (own complex mixed-radix FFT over the full
spectrum) nor with the HIP path (rocFFT), so a mistake in the restated per-bin rules (real-only bins, the unscaled inverse, the
quadrant swap for odd sizes, the clipped 5 x 5 centroid, the (N/2 - cx, M/2 - cy) sign) would have to be made twice to go unnoticed.
It is not the reference (cv2 cannot be installed here): agreement with it pins the oracle's arithmetic, not OpenCV's."""
import numpy as np


def optimal_dft_size(n):
    """smallest 2^a 3^b 5^c >= n (cv::getOptimalDFTSize)"""
    best = None
    p2 = 1
    while p2 < 2 * n:
        p3 = p2
        while p3 < 2 * n:
            p5 = p3
            while p5 < 2 * n:
                if p5 >= n and (best is None or p5 < best):
                    best = p5
                p5 *= 5
            p3 *= 3
        p2 *= 2
    return best


def phase_correlate(a, b):
    """-> ((x, y), response) as cv2.phaseCorrelate(np.float64(a), np.float64(b)) returns them"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape and a.ndim == 2
    h, w = a.shape
    M, N = optimal_dft_size(h), optimal_dft_size(w)
    pa = np.zeros((M, N)); pb = np.zeros((M, N))
    pa[:h, :w] = a; pb[:h, :w] = b                              # copyMakeBorder: zeros on the bottom / right only
    F1, F2 = np.fft.rfft2(pa), np.fft.rfft2(pb)                 # M x (N/2 + 1)
    P = F1 * np.conj(F2)                                        # mulSpectrums(..., conjB = true)
    eps = np.finfo(np.float64).eps
    mag = np.abs(P)                                             # magSpectrums: |P| with a zero imaginary slot ...
    C = P * mag / (mag * mag + eps)                             # divSpectrums by that real array
    # ... except the purely real bins of the packed layout, where magSpectrums stores x * x: C = x * x^2 / (x^4 + eps)
    us = [0] + ([M // 2] if M % 2 == 0 else [])
    vs = [0] + ([N // 2] if N % 2 == 0 else [])
    for u in us:
        for v in vs:
            x = P[u, v].real
            m2 = x * x
            C[u, v] = x / (m2 + eps)
    R = np.fft.irfft2(C, s=(M, N)) * (M * N)                    # idft without DFT_SCALE
    ym, xm = M >> 1, N >> 1                                     # fftShift: quadrants of (N >> 1) x (M >> 1); an odd last row / column stays
    S = R.copy()
    S[:ym, :xm] = R[ym:2 * ym, xm:2 * xm]; S[ym:2 * ym, xm:2 * xm] = R[:ym, :xm]
    S[:ym, xm:2 * xm] = R[ym:2 * ym, :xm]; S[ym:2 * ym, :xm] = R[:ym, xm:2 * xm]
    py, px = np.unravel_index(np.argmax(S), S.shape)            # minMaxLoc: first maximum in row-major order
    r0, r1 = max(py - 2, 0), min(py + 2, M - 1)
    c0, c1 = max(px - 2, 0), min(px + 2, N - 1)
    win = S[r0:r1 + 1, c0:c1 + 1]
    ys, xs = np.mgrid[r0:r1 + 1, c0:c1 + 1]
    s = win.sum()
    cx = (xs * win).sum() / (s + eps); cy = (ys * win).sum() / (s + eps)
    return (N / 2.0 - cx, M / 2.0 - cy), s / (M * N), (int(py), int(px))

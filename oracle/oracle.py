"""ctypes front-end of the CPU ORACLE (oracle/vfsms_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from imagestitch_amd/ (the product).  See vfsms_oracle.h for the
parity status of each function and the reference file:line each one restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvfsms_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("vfsms_oracle.c", "vfsms_oracle_orb.c", "vfsms_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"),
                     ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])

_lib = None
_kind = "portable (-O3)"


def use_native():
    """Switch to the -O3 -march=native build, compiled on THIS machine (oracle/Makefile `native`): bench.py's cpu_baseline leg.
    Same results bit for bit (no fast-math, no FMA contraction)."""
    global _lib, _SO, _kind
    subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
    _SO = os.path.join(_HERE, "_build", "libvfsms_oracle_native.so")
    _kind = "native (-O3 -march=native)"
    _lib = None
    return lib()


def build_kind():
    return _kind


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        L.orc_optimal_dft_size.argtypes = [i32]; L.orc_optimal_dft_size.restype = i32
        L.orc_integral_u8_i32.argtypes = [vp, i32, i32, i32, vp]; L.orc_integral_u8_i32.restype = None
        L.orc_surf_layer.argtypes = [vp, i32, i32, i32, i32, vp, vp]; L.orc_surf_layer.restype = None
        L.orc_surf_detect_describe.argtypes = [vp, i32, i32, i32, f64, i32, i32, i32, i32, vp, vp, i32, i32]
        L.orc_surf_detect_describe.restype = i32
        L.orc_surf_detect.argtypes = [vp, i32, i32, i32, f64, i32, i32, vp, i32]; L.orc_surf_detect.restype = i32
        L.orc_bf_l2_knn2.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, vp, i32]; L.orc_bf_l2_knn2.restype = None
        L.orc_bf_l2_ratio_matches.argtypes = [vp, i32, vp, i32, i32, f64, vp, i32]; L.orc_bf_l2_ratio_matches.restype = i32
        L.orc_bf_hamming_matches.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp]; L.orc_bf_hamming_matches.restype = i32
        L.orc_mode_offset.argtypes = [vp, vp, vp, i32, i32, vp]; L.orc_mode_offset.restype = None
        L.orc_phase_correlate_u8.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]; L.orc_phase_correlate_u8.restype = None
        L.orc_fuse_fade.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]; L.orc_fuse_fade.restype = None
        L.orc_orb_detect_describe.argtypes = [vp, i32, i32, i32, i32, C.c_float, i32, i32, i32, i32, i32, vp, vp, i32]
        L.orc_orb_detect_describe.restype = i32
        L.orc_orb_pattern.argtypes = [i32, i32, vp]; L.orc_orb_pattern.restype = None
        L.orc_corner_ramps.argtypes = [vp, i32, i32, i32, vp, vp, vp]; L.orc_corner_ramps.restype = i32
        L.orc_det_sincos.argtypes = [vp, i32, vp]; L.orc_det_sincos.restype = None
        L.orc_equalize_hist.argtypes = [vp, i32, i32, i32, vp]; L.orc_equalize_hist.restype = None
        L.orc_clahe.argtypes = [vp, i32, i32, i32, f64, i32, i32, vp]; L.orc_clahe.restype = None
        _lib = L
    return _lib


def corner_ramps(A):
    """ImageFusion.getWeightsMatrix -> (wB_r float32[r], wB_c float32[c], info)."""
    A = np.ascontiguousarray(A, np.int64)
    r, c = A.shape[:2]
    ch = 1 if A.ndim == 2 else A.shape[2]
    wr = np.ones(r, np.float32); wc = np.ones(c, np.float32)
    info = np.zeros(4, np.int32)
    if lib().orc_corner_ramps(_p(A), r, c, ch, _p(wr), _p(wc), _p(info)) != 0:
        raise IndexError("reference getWeightsMatrix would raise on this input")
    return wr, wc, info


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8_2d(img):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


def det_sincos(x):
    """The explicit sin/cos both the oracle and the engine evaluate (vfsms_oracle.h) -> (sin, cos) float64 arrays."""
    x = np.ascontiguousarray(x, np.float64).ravel()
    out = np.empty((len(x), 2), np.float64)
    lib().orc_det_sincos(_p(x), len(x), _p(out))
    return out[:, 0].copy(), out[:, 1].copy()


def equalize_hist(img):
    """cv2.equalizeHist(img) (Stitcher.py:275-276)."""
    img = _u8_2d(img)
    out = np.empty(img.shape, np.uint8)
    lib().orc_equalize_hist(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(out))
    return out


def clahe(img, clip_limit=20.0, tile_size=5):
    """cv2.createCLAHE(clipLimit, (tileSize, tileSize)).apply(img) (Stitcher.py:271-273)."""
    img = _u8_2d(img)
    out = np.empty(img.shape, np.uint8)
    lib().orc_clahe(_p(img), img.shape[0], img.shape[1], img.strides[0], float(clip_limit), int(tile_size), int(tile_size), _p(out))
    return out


def optimal_dft_size(n):
    return lib().orc_optimal_dft_size(int(n))


def integral(img):
    img = _u8_2d(img)
    h, w = img.shape
    out = np.empty((h + 1, w + 1), np.int32)
    lib().orc_integral_u8_i32(_p(img), h, w, img.strides[0], _p(out))
    return out


def surf_layer(sum_img, size, step):
    sum_img = np.ascontiguousarray(sum_img, np.int32)
    h, w = sum_img.shape[0] - 1, sum_img.shape[1] - 1
    det = np.empty((h // step, w // step), np.float32)
    tr = np.empty_like(det)
    lib().orc_surf_layer(_p(sum_img), h, w, size, step, _p(det), _p(tr))
    return det, tr


def surf_detect(img, hessian=100.0, n_octaves=4, n_layers=3, cap=None):
    img = _u8_2d(img)
    h, w = img.shape
    cap = cap or (h * w // 8 + 1024)
    kps = np.zeros(cap, KP_DTYPE)
    n = lib().orc_surf_detect(_p(img), h, w, img.strides[0], hessian, n_octaves, n_layers, _p(kps), cap)
    if n < 0:
        raise RuntimeError("oracle surf_detect: capacity exceeded")
    return kps[:n].copy()


def surf_detect_describe(img, hessian=100.0, n_octaves=4, n_layers=3, extended=False, upright=False,
                         cap=None, nthreads=0):
    """cv2.xfeatures2d.SURF_create(...).detectAndCompute(img, None) -> (keypoint records, float32[N,D])."""
    img = _u8_2d(img)
    h, w = img.shape
    cap = cap or (h * w // 8 + 1024)
    d = 128 if extended else 64
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, d), np.float32)
    n = lib().orc_surf_detect_describe(_p(img), h, w, img.strides[0], hessian, n_octaves, n_layers,
                                       int(extended), int(upright), _p(kps), _p(desc), cap, nthreads)
    if n < 0:
        raise RuntimeError("oracle surf: capacity exceeded")
    return kps[:n].copy(), desc[:n].copy()


def bf_l2_knn2(q, t, nthreads=0):
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    nq, nt = len(q), len(t)
    dim = q.shape[1] if nq else (t.shape[1] if nt else 64)
    i1 = np.empty(nq, np.int32); i2 = np.empty(nq, np.int32)
    d1 = np.empty(nq, np.float32); d2 = np.empty(nq, np.float32)
    lib().orc_bf_l2_knn2(_p(q), nq, _p(t), nt, dim, _p(i1), _p(d1), _p(i2), _p(d2), nthreads)
    return i1, d1, i2, d2


def bf_l2_ratio_matches(q, t, ratio=0.75, nthreads=0):
    """ImageUtility.py:288-296 -> int32[M,2] of (trainIdx, queryIdx)."""
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    nq, nt = len(q), len(t)
    dim = q.shape[1] if nq else 64
    pairs = np.empty((max(nq, 1), 2), np.int32)
    m = lib().orc_bf_l2_ratio_matches(_p(q), nq, _p(t), nt, dim, float(ratio), _p(pairs), nthreads)
    return pairs[:m].copy()


def bf_hamming_matches(q, t, max_dist=-1):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq, nt = len(q), len(t)
    nb = q.shape[1] if nq else 32
    pairs = np.empty((max(nq, 1), 2), np.int32)
    dist = np.empty(max(nq, 1), np.int32)
    m = lib().orc_bf_hamming_matches(_p(q), nq, _p(t), nt, nb, int(max_dist), _p(pairs), _p(dist))
    return pairs[:m].copy(), dist[:m].copy()


def mode_offset(kpsA, kpsB, pairs, offset_evaluate=3):
    """Method.getOffsetByMode (ImageUtility.py:139-178) -> (status, [dx, dy], votes)."""
    kpsA = np.ascontiguousarray(kpsA, np.float32).reshape(-1, 2)
    kpsB = np.ascontiguousarray(kpsB, np.float32).reshape(-1, 2)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    out = np.zeros(4, np.int32)
    lib().orc_mode_offset(_p(kpsA), _p(kpsB), _p(pairs), len(pairs), int(offset_evaluate), _p(out))
    return bool(out[0]), [int(out[1]), int(out[2])], int(out[3])


def phase_correlate(a, b):
    """cv2.phaseCorrelate(np.float64(a), np.float64(b)) -> ((x, y), response)  (Stitcher.py:230)."""
    a = _u8_2d(a); b = _u8_2d(b)
    assert a.shape == b.shape
    h, w = a.shape
    out = np.zeros(3, np.float64)
    mn = np.zeros(2, np.int32)
    lib().orc_phase_correlate_u8(_p(a), _p(b), h, w, a.strides[0], b.strides[0], _p(out), _p(mn))
    return (float(out[0]), float(out[1])), float(out[2])


def fuse_fade(A, B, dx, dy, return_info=False):
    """ImageFusion.fuseByFadeInAndFadeOut([A, B], dx, dy) on int64 arrays with -1 = empty."""
    A = np.array(A, dtype=np.int64, order="C", copy=True)
    B = np.ascontiguousarray(B, np.int64)
    assert A.shape == B.shape
    r, c = A.shape[:2]
    ch = 1 if A.ndim == 2 else A.shape[2]
    out = np.zeros(A.shape, np.uint8)
    info = np.zeros(4, np.int32)
    lib().orc_fuse_fade(_p(A), _p(B), r, c, ch, int(dx), int(dy), _p(out), _p(info))
    if info[0] < 0:
        raise IndexError("reference getWeightsMatrix would raise on this input")
    return (out, info) if return_info else out


def orb_pattern(patch_size=31, npoints=512):
    xy = np.zeros((npoints, 2), np.int32)
    lib().orc_orb_pattern(int(patch_size), int(npoints), _p(xy))
    return xy


def orb_detect_describe(img, nfeatures=5000, scale_factor=1.2, nlevels=8, edge_threshold=31, first_level=0,
                        patch_size=31, fast_threshold=20, cap=None):
    """cv2.ORB_create(...).detectAndCompute(img, None) -> (keypoint records, uint8[N,32])  (ImageUtility.py:260,262)."""
    img = _u8_2d(img)
    h, w = img.shape
    cap = cap or (2 * nfeatures + 4096)
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().orc_orb_detect_describe(_p(img), h, w, img.strides[0], nfeatures, float(scale_factor), nlevels, edge_threshold,
                                      first_level, patch_size, fast_threshold, _p(kps), _p(desc), cap)
    if n < 0:
        raise RuntimeError("oracle orb: capacity exceeded")
    return kps[:n].copy(), desc[:n].copy()

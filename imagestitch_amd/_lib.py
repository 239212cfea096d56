"""ctypes binding of libvfsms.so (include/vfsms.h) -- the only compute backend of this package.

There is deliberately NO CPU fallback: if the shared library is missing or no MI355X is visible, every
operator raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VFSMS_LIB: another build of the library (A/B timing of kernel variants: tools/microbench.py, bench.py); the product is the in-tree one
LIB_PATH = os.path.abspath(os.environ["VFSMS_LIB"]) if os.environ.get("VFSMS_LIB") else os.path.join(_HERE, "lib", "libvfsms.so")

VFSMS_OK = 0
VFSMS_ERR_BAD_ARG, VFSMS_ERR_CAPACITY, VFSMS_ERR_UNSUPPORTED = -1, -2, -6
ATTEMPT_INTS = 8


class VfsmsError(RuntimeError):
    pass


class SurfParams(C.Structure):
    _fields_ = [("hessian_threshold", C.c_float), ("n_octaves", C.c_int32), ("n_octave_layers", C.c_int32),
                ("extended", C.c_int32), ("upright", C.c_int32)]


class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32), ("edge_threshold", C.c_int32),
                ("first_level", C.c_int32), ("wta_k", C.c_int32), ("score_type", C.c_int32), ("patch_size", C.c_int32),
                ("fast_threshold", C.c_int32)]


class AttemptKey(C.Structure):
    _fields_ = [("pair", C.c_int32), ("direction", C.c_int32), ("i", C.c_int32)]


class GridParams(C.Structure):
    _fields_ = [("method", C.c_int32), ("offset_evaluate", C.c_int32), ("direct_incre", C.c_int32), ("window", C.c_int32),
                ("orb_max_dist", C.c_int32), ("enhance_mode", C.c_int32), ("tile_grid", C.c_int32), ("reserved", C.c_int32),
                ("roi_ratio", C.c_double), ("search_ratio", C.c_double), ("phase_threshold", C.c_double), ("clip_limit", C.c_double),
                ("surf", SurfParams), ("orb", OrbParams),
                ("path_hint", C.POINTER(C.c_int32)), ("path_hint_len", C.c_int32), ("reserved2", C.c_int32)]


ATTEMPT_EVAL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(AttemptKey), C.c_int, C.POINTER(C.c_int32))


class RoiPair(C.Structure):
    _fields_ = [("tile_a", C.c_int64), ("tile_b", C.c_int64),
                ("ay0", C.c_int32), ("ax0", C.c_int32), ("by0", C.c_int32), ("bx0", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"),
                     ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])

_SIGNATURES = {
    # name: (restype, argtypes)
    "vfsms_version": (C.c_int, []),
    "vfsms_device_count": (C.c_int, []),
    "vfsms_last_error": (C.c_int, [C.c_char_p, C.c_int]),
    "vfsms_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vfsms_ctx_destroy": (C.c_int, [C.c_void_p]),
    "vfsms_ctx_sync": (C.c_int, [C.c_void_p]),
    "vfsms_ctx_sync_uploads": (C.c_int, [C.c_void_p]),
    "vfsms_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "vfsms_ctx_set_keypoint_capacity": (C.c_int, [C.c_void_p, C.c_int]),
    "vfsms_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "vfsms_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "vfsms_tile_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vfsms_tile_upload_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vfsms_tile_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vfsms_tile_fill": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "vfsms_tile_reserve_ch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vfsms_tile_fill_pair": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int]),
    "vfsms_tile_fill_jpeg": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_size_t]),
    "vfsms_jpeg_decode": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vfsms_jpeg_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vfsms_jpeg_join": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vfsms_canvas_blend_tile_resident": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfsms_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vfsms_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vfsms_tile_wrap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vfsms_tile_free": (C.c_int, [C.c_void_p, C.c_int64]),
    "vfsms_integral_u8_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfsms_surf_detect_describe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(SurfParams),
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vfsms_surf_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(SurfParams),
                                    C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vfsms_orb_detect_describe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(OrbParams),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vfsms_attempt_orb_batch": (C.c_int, [C.c_void_p, C.POINTER(RoiPair), C.c_int, C.POINTER(OrbParams), C.c_int, C.c_int, C.c_void_p]),
    "vfsms_bf_l2_knn2_ratio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                         C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vfsms_bf_l2_knn2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "vfsms_bf_hamming_nn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vfsms_mode_offset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p]),
    "vfsms_phase_correlate_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p]),
    "vfsms_fuse_fade_i64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "vfsms_fuse_ramps_i64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "vfsms_attempt_surf_batch": (C.c_int, [C.c_void_p, C.POINTER(RoiPair), C.c_int, C.POINTER(SurfParams), C.c_double,
                                           C.c_int, C.c_void_p]),
    "vfsms_attempt_surf_batch_enhanced": (C.c_int, [C.c_void_p, C.POINTER(RoiPair), C.c_int, C.POINTER(SurfParams), C.c_double,
                                                    C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "vfsms_features_surf": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SurfParams), C.c_int, C.c_double,
                                      C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "vfsms_features_match_offset": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_int, C.c_void_p]),
    "vfsms_features_download": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vfsms_features_free": (C.c_int, [C.c_void_p, C.c_int64]),
    "vfsms_enhance_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "vfsms_pairs_offsets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(GridParams), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "vfsms_pairs_offsets_blind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vfsms_pairs_offsets_blind_eval": (C.c_int, [ATTEMPT_EVAL, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vfsms_pairs_offsets_eval": (C.c_int, [ATTEMPT_EVAL, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(GridParams), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "vfsms_attempt_phase_batch": (C.c_int, [C.c_void_p, C.POINTER(RoiPair), C.c_int, C.c_void_p]),
    "vfsms_phase_plan": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "vfsms_canvas_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vfsms_canvas_free": (C.c_int, [C.c_void_p, C.c_int64]),
    "vfsms_canvas_paste": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vfsms_canvas_blend_tile": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int]),
    "vfsms_canvas_paste_tile": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "vfsms_canvas_fuse_tile_resident": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_void_p]),
    "vfsms_canvas_fuse_tile": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfsms_canvas_assemble_resident": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "vfsms_canvas_fuse_tile_resident_m": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfsms_canvas_fuse_tile_m": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfsms_fuse_trig_i64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vfsms_features_surf_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "vfsms_features_match_offset_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "vfsms_canvas_download": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "vfsms_canvas_download_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "vfsms_tile_upload_ch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
}

_lib = None


def load_library():
    """dlopen libvfsms.so and bind every declared entry point.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VfsmsError("libvfsms.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(make -C imagestitch_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError here == header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_names():
    return sorted(_SIGNATURES)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8_2d(img):
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise ValueError("expected a 2-D uint8 image, got %s %s" % (img.dtype, img.shape))
    if img.shape[1] > 1 and img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    if img.strides[0] < img.shape[1]:
        img = np.ascontiguousarray(img)
    return img


class Engine:
    """One HIP context (one GPU, one stream).  All methods are synchronous."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.vfsms_ctx_create(int(device), C.byref(h))
        self.ctx = h
        self._check(rc)
        self.device = device

    # -- plumbing ---------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != VFSMS_OK:
            buf = C.create_string_buffer(512)
            self.lib.vfsms_last_error(buf, 512)
            raise VfsmsError("libvfsms error %d: %s" % (rc, buf.value.decode(errors="replace")))

    def close(self):
        if getattr(self, "ctx", None):
            for p in self.__dict__.pop("_pinned", []):
                self.lib.vfsms_host_free(self.ctx, C.c_void_p(p))
            self.lib.vfsms_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.lib.vfsms_ctx_sync(self.ctx))
        self.__dict__.pop("_inflight", None)

    def sync_uploads(self):
        """wait for tile_upload_async copies only: their host buffers may be reused"""
        self._check(self.lib.vfsms_ctx_sync_uploads(self.ctx))
        self.__dict__.pop("_inflight", None)

    def stream_handle(self):
        return self.lib.vfsms_ctx_stream(self.ctx)

    def set_keypoint_capacity(self, cap):
        self._check(self.lib.vfsms_ctx_set_keypoint_capacity(self.ctx, int(cap)))

    def profile_enable(self, on=True):
        self._check(self.lib.vfsms_profile_enable(self.ctx, int(bool(on))))

    def profile_read(self, reset=True):
        """-> {stage: (total_ms, launch_groups)} measured with HIP events on the engine's stream."""
        names = C.create_string_buffer(4096)
        ms = np.zeros(64, np.float64); calls = np.zeros(64, np.int64)
        n = C.c_int()
        self._check(self.lib.vfsms_profile_read(self.ctx, names, 4096, _ptr(ms), _ptr(calls), 64, C.byref(n), int(bool(reset))))
        keys = names.value.decode().split(",") if n.value else []
        return {k: (float(ms[i]), int(calls[i])) for i, k in enumerate(keys)}

    # -- tiles ---------------------------------------------------------------------------------------------
    def tile_upload(self, img):
        img = _u8_2d(img)
        h = C.c_int64()
        self._check(self.lib.vfsms_tile_upload(self.ctx, _ptr(img), img.shape[0], img.shape[1], img.strides[0], C.byref(h)))
        return h.value

    def tile_reserve(self, h, w):
        """handle of a gray tile whose pixels a decoder thread delivers later (tile_fill); batch calls wait for exactly the tiles they name"""
        hd = C.c_int64()
        self._check(self.lib.vfsms_tile_reserve(self.ctx, int(h), int(w), C.byref(hd)))
        return hd.value

    def tile_reserve_color(self, h, w, ch=3):
        """tile_reserve for an interleaved tile of `ch` channels (the mosaic's colour tiles)"""
        hd = C.c_int64()
        self._check(self.lib.vfsms_tile_reserve_ch(self.ctx, int(h), int(w), int(ch), C.byref(hd)))
        return hd.value

    def tile_fill(self, handle, img):
        """deliver (img: u8 (h, w) or (h, w, ch) array, C-contiguous rows) or give up on (img is None) a reserved tile; safe from any thread"""
        if img is None:
            self._check(self.lib.vfsms_tile_fill(self.ctx, C.c_int64(handle), None, 0))
            return
        assert img.dtype == np.uint8 and img.ndim in (2, 3) and img.strides[-1] == 1 and (img.ndim == 2 or img.strides[1] == img.shape[2])
        self._check(self.lib.vfsms_tile_fill(self.ctx, C.c_int64(handle), _ptr(img), img.strides[0]))

    SRC_GRAY8, SRC_YCC24, SRC_YCCX32 = 0, 1, 2

    def tile_fill_pair(self, gray_handle, color_handle, address, stride_bytes, fmt):
        """One decoded image -> the reserved gray tile and / or the reserved BGR tile (vfsms_tile_fill_pair; either handle may be 0 / None);
        `address`: raw host address of the decoder's output in format `fmt` (SRC_*), None gives both tiles up.  Safe from any thread."""
        self._check(self.lib.vfsms_tile_fill_pair(self.ctx, C.c_int64(gray_handle or 0), C.c_int64(color_handle or 0),
                                                  C.c_void_p(address) if address is not None else None, int(stride_bytes), int(fmt)))

    def tile_fill_jpeg(self, gray_handle, color_handle, data):
        """A JPEG file's bytes -> the reserved gray tile and / or the reserved BGR tile, decoded by the library itself (vfsms_tile_fill_jpeg:
        libjpeg-turbo into pinned staging, colour conversion on the device).  True when the tiles are filled; False when this decoder does
        not take the file (no libjpeg.so.8, an unusual colour space, a damaged file, a size other than the tiles'): the tiles are STILL
        RESERVED and the caller decodes some other way.  Safe from any thread."""
        rc = self.lib.vfsms_tile_fill_jpeg(self.ctx, C.c_int64(gray_handle or 0), C.c_int64(color_handle or 0), data, len(data))
        if rc in (VFSMS_ERR_UNSUPPORTED, VFSMS_ERR_BAD_ARG):
            return False
        self._check(rc)
        return True

    def tile_fill_ptr(self, handle, address, stride):
        """tile_fill from a raw host address (rows `stride` bytes apart); the caller keeps the memory alive until this returns"""
        self._check(self.lib.vfsms_tile_fill(self.ctx, C.c_int64(handle), C.c_void_p(address), int(stride)))

    def tile_upload_async(self, img):
        """Upload without waiting: the copy overlaps the compute stream; `img` must stay alive and unchanged until sync() or
        the first synchronous call that used the tile (arrays from pinned_empty() make the copy a true asynchronous DMA)."""
        img = _u8_2d(img)
        h = C.c_int64()
        self._check(self.lib.vfsms_tile_upload_async(self.ctx, _ptr(img), img.shape[0], img.shape[1], img.strides[0], C.byref(h)))
        self.__dict__.setdefault("_inflight", []).append(img)
        return h.value

    def tile_upload_color(self, img, asynchronous=False):
        """Interleaved colour tile (h, w, ch) u8 for the mosaic canvas (vfsms_tile_upload_ch); asynchronous like tile_upload_async."""
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim != 3:
            raise ValueError("tile_upload_color takes an (h, w, ch) array")
        h = C.c_int64()
        self._check(self.lib.vfsms_tile_upload_ch(self.ctx, _ptr(img), img.shape[0], img.shape[1], img.shape[2], img.strides[0],
                                                  1 if asynchronous else 0, C.byref(h)))
        if asynchronous:
            self.__dict__.setdefault("_inflight", []).append(img)
        return h.value

    def pinned_empty(self, shape, dtype=np.uint8):
        """numpy array over pinned host memory (vfsms_host_alloc); freed when the engine closes."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self.lib.vfsms_host_alloc(self.ctx, C.c_size_t(max(n, 1)), C.byref(p)))
        self.__dict__.setdefault("_pinned", []).append(p.value)
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def tile_wrap(self, device_ptr, h, w, stride):
        hd = C.c_int64()
        self._check(self.lib.vfsms_tile_wrap(self.ctx, C.c_void_p(int(device_ptr)), int(h), int(w), int(stride), C.byref(hd)))
        return hd.value

    def tile_free(self, handle):
        self._check(self.lib.vfsms_tile_free(self.ctx, C.c_int64(handle)))

    # -- operators -------------------------------------------------------------------------------------------
    def integral(self, img):
        img = _u8_2d(img)
        h, w = img.shape
        out = np.empty((h + 1, w + 1), np.int32)
        self._check(self.lib.vfsms_integral_u8_i32(self.ctx, _ptr(img), h, w, img.strides[0], _ptr(out)))
        return out

    @staticmethod
    def surf_params(hessian=100.0, n_octaves=4, n_layers=3, extended=False, upright=False):
        return SurfParams(float(hessian), int(n_octaves), int(n_layers), int(bool(extended)), int(bool(upright)))

    def surf_detect_describe(self, img, params=None, cap=None, full=False):
        img = _u8_2d(img)
        h, w = img.shape
        params = params or self.surf_params()
        cap = cap or (h * w // 24 + 4096)
        d = 128 if params.extended else 64
        kxy = np.empty((cap, 2), np.float32)
        desc = np.empty((cap, d), np.float32)
        kfull = np.empty(cap, KP_DTYPE) if full else None
        n = C.c_int()
        self._check(self.lib.vfsms_surf_detect_describe(self.ctx, _ptr(img), h, w, img.strides[0], C.byref(params),
                                                        _ptr(kxy), _ptr(desc), _ptr(kfull), cap, C.byref(n)))
        n = n.value
        if full:
            return kxy[:n].copy(), desc[:n].copy(), kfull[:n].copy()
        return kxy[:n].copy(), desc[:n].copy()

    def surf_detect(self, img, params=None, cap=None):
        img = _u8_2d(img)
        h, w = img.shape
        params = params or self.surf_params()
        cap = cap or (h * w // 24 + 4096)
        kfull = np.empty(cap, KP_DTYPE)
        n = C.c_int()
        self._check(self.lib.vfsms_surf_detect(self.ctx, _ptr(img), h, w, img.strides[0], C.byref(params),
                                               _ptr(kfull), cap, C.byref(n)))
        return kfull[:n.value].copy()

    @staticmethod
    def orb_params(nfeatures=5000, scale_factor=1.2, nlevels=8, edge_threshold=31, first_level=0, wta_k=2, score_type=0,
                   patch_size=31, fast_threshold=20):
        return OrbParams(int(nfeatures), float(scale_factor), int(nlevels), int(edge_threshold), int(first_level), int(wta_k),
                         int(score_type), int(patch_size), int(fast_threshold))

    def orb_detect_describe(self, img, params=None, cap=None, full=False):
        img = _u8_2d(img)
        h, w = img.shape
        params = params or self.orb_params()
        cap = cap or (2 * params.n_features + 2048)
        kxy = np.empty((cap, 2), np.float32)
        desc = np.empty((cap, 32), np.uint8)
        kfull = np.empty(cap, KP_DTYPE) if full else None
        n = C.c_int()
        self._check(self.lib.vfsms_orb_detect_describe(self.ctx, _ptr(img), h, w, img.strides[0], C.byref(params),
                                                       _ptr(kxy), _ptr(desc), _ptr(kfull), cap, C.byref(n)))
        n = n.value
        if full:
            return kxy[:n].copy(), desc[:n].copy(), kfull[:n].copy()
        return kxy[:n].copy(), desc[:n].copy()

    def attempt_orb_batch(self, jobs, params=None, max_dist=-1, offset_evaluate=3):
        n = len(jobs)
        out = np.zeros((n, ATTEMPT_INTS), np.int32)
        if n == 0:
            return out
        arr = jobs if isinstance(jobs, C.Array) else self.make_jobs(jobs)
        params = params or self.orb_params()
        self._check(self.lib.vfsms_attempt_orb_batch(self.ctx, arr, n, C.byref(params), int(max_dist), int(offset_evaluate), _ptr(out)))
        return out

    def bf_l2_ratio_matches(self, q, t, ratio=0.75):
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        nq, nt = len(q), len(t)
        if nq == 0 or nt == 0:
            return np.zeros((0, 2), np.int32)
        dim = q.shape[1]
        pairs = np.empty((nq, 2), np.int32)
        m = C.c_int()
        self._check(self.lib.vfsms_bf_l2_knn2_ratio(self.ctx, _ptr(q), nq, _ptr(t), nt, dim, float(ratio), _ptr(pairs), nq, C.byref(m)))
        return pairs[:m.value].copy()

    def bf_l2_knn2(self, q, t):
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        nq, nt = len(q), len(t)
        dim = q.shape[1] if nq else (t.shape[1] if nt else 64)
        i1 = np.empty(nq, np.int32); d1 = np.empty(nq, np.float32); d2 = np.empty(nq, np.float32)
        self._check(self.lib.vfsms_bf_l2_knn2(self.ctx, _ptr(q), nq, _ptr(t), nt, dim, _ptr(i1), _ptr(d1), _ptr(d2)))
        return i1, d1, d2

    def bf_hamming_matches(self, q, t, max_dist=-1):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        nq, nt = len(q), len(t)
        if nq == 0 or nt == 0:
            return np.zeros((0, 2), np.int32)
        pairs = np.empty((nq, 2), np.int32)
        m = C.c_int()
        self._check(self.lib.vfsms_bf_hamming_nn(self.ctx, _ptr(q), nq, _ptr(t), nt, q.shape[1], int(max_dist), _ptr(pairs), nq, C.byref(m)))
        return pairs[:m.value].copy()

    def mode_offset(self, kpsA, kpsB, pairs, offset_evaluate=3):
        kpsA = np.ascontiguousarray(kpsA, np.float32).reshape(-1, 2)
        kpsB = np.ascontiguousarray(kpsB, np.float32).reshape(-1, 2)
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        out = np.zeros(4, np.int32)
        self._check(self.lib.vfsms_mode_offset(self.ctx, _ptr(kpsA), len(kpsA), _ptr(kpsB), len(kpsB), _ptr(pairs), len(pairs),
                                               int(offset_evaluate), _ptr(out)))
        return bool(out[0]), [int(out[1]), int(out[2])], int(out[3])

    def phase_correlate(self, a, b):
        a = _u8_2d(a); b = _u8_2d(b)
        if a.shape != b.shape:
            raise ValueError("phase_correlate: shapes differ")
        out = np.zeros(3, np.float64)
        self._check(self.lib.vfsms_phase_correlate_u8(self.ctx, _ptr(a), _ptr(b), a.shape[0], a.shape[1], a.strides[0], b.strides[0], _ptr(out)))
        return (float(out[0]), float(out[1])), float(out[2])

    def phase_plan(self, h, w):
        """-> dict: how a strip of h x w is correlated (vfsms_phase_plan): the LDS transforms or rocFFT, orientation, padded sizes, workgroup shapes"""
        info = np.zeros(8, np.int32)
        self._check(self.lib.vfsms_phase_plan(int(h), int(w), _ptr(info)))
        keys = ("lds_transforms", "transposed", "M", "N", "columns_per_workgroup", "rows_per_workgroup", "row_threads", "column_threads")
        return {k: int(v) for k, v in zip(keys, info)}

    def fuse_fade_i64(self, A, B, dx, dy, return_info=False):
        A = np.ascontiguousarray(A, np.int64); B = np.ascontiguousarray(B, np.int64)
        if A.shape != B.shape:
            raise ValueError("fuse: shapes differ")
        r, c = A.shape[:2]
        ch = 1 if A.ndim == 2 else A.shape[2]
        out = np.empty(A.shape, np.uint8)
        info = np.zeros(4, np.int32)
        self._check(self.lib.vfsms_fuse_fade_i64(self.ctx, _ptr(A), _ptr(B), r, c, ch, int(dx), int(dy), _ptr(out), _ptr(info)))
        return (out, info) if return_info else out

    def fuse_trig_i64(self, A, B, dx, dy, return_info=False):
        """ImageFusion.fuseByTrigonometric on int64 regions (-1 = empty) -> uint8"""
        A = np.ascontiguousarray(A, np.int64); B = np.ascontiguousarray(B, np.int64)
        if A.shape != B.shape:
            raise ValueError("fuse: shapes differ")
        r, c = A.shape[:2]
        ch = 1 if A.ndim == 2 else A.shape[2]
        out = np.empty(A.shape, np.uint8)
        info = np.zeros(4, np.int32)
        self._check(self.lib.vfsms_fuse_trig_i64(self.ctx, _ptr(A), _ptr(B), r, c, ch, int(dx), int(dy), _ptr(out), _ptr(info)))
        return (out, info) if return_info else out

    def fuse_ramps_i64(self, A, dx, dy, force_corner=False):
        """-> ((wA_r, wB_r, wA_c, wB_c), info): the separable float32 ramps of the fade blend / getWeightsMatrix."""
        A = np.ascontiguousarray(A, np.int64)
        r, c = A.shape[:2]
        ch = 1 if A.ndim == 2 else A.shape[2]
        ramps = np.empty(2 * (r + c), np.float32)
        info = np.zeros(4, np.int32)
        self._check(self.lib.vfsms_fuse_ramps_i64(self.ctx, _ptr(A), r, c, ch, int(dx), int(dy), int(bool(force_corner)),
                                                  _ptr(ramps), _ptr(info)))
        return (ramps[:r].copy(), ramps[r:2 * r].copy(), ramps[2 * r:2 * r + c].copy(), ramps[2 * r + c:].copy()), info

    # -- fused fast path ----------------------------------------------------------------------------------------
    @staticmethod
    def make_jobs(jobs):
        arr = (RoiPair * len(jobs))()
        for k, j in enumerate(jobs):
            arr[k] = RoiPair(*[int(v) for v in j])
        return arr

    def attempt_surf_batch(self, jobs, params=None, ratio=0.75, offset_evaluate=3):
        """jobs: sequence of (tile_a, tile_b, ay0, ax0, by0, bx0, h, w) -> int32[n, 8]
        columns: status, dx, dy, votes, nA, nB, nMatches, 0   (raw vote, before the stitch-axis correction)."""
        n = len(jobs)
        out = np.zeros((n, ATTEMPT_INTS), np.int32)
        if n == 0:
            return out
        arr = jobs if isinstance(jobs, C.Array) else self.make_jobs(jobs)
        params = params or self.surf_params()
        self._check(self.lib.vfsms_attempt_surf_batch(self.ctx, arr, n, C.byref(params), float(ratio), int(offset_evaluate), _ptr(out)))
        return out

    def attempt_surf_batch_enhanced(self, jobs, params=None, ratio=0.75, offset_evaluate=3, enhance=(0, 0.0, 0)):
        """attempt_surf_batch with every ROI strip equalised (enhance = (1, 0, 0)) or CLAHE'd (enhance = (2, clipLimit, tileSize))
        first, as Stitcher.py:327-334 does when Method.isEnhance is set."""
        n = len(jobs)
        out = np.zeros((n, ATTEMPT_INTS), np.int32)
        if n == 0:
            return out
        arr = jobs if isinstance(jobs, C.Array) else self.make_jobs(jobs)
        params = params or self.surf_params()
        self._check(self.lib.vfsms_attempt_surf_batch_enhanced(self.ctx, arr, n, C.byref(params), float(ratio), int(offset_evaluate),
                                                               int(enhance[0]), float(enhance[1]), int(enhance[2]), _ptr(out)))
        return out

    # -- whole shooting paths ------------------------------------------------------------------------------------------------
    @staticmethod
    def grid_params(method="surf", roiRatio=0.2, searchRatio=0.75, offsetEvaluate=3, directIncre=1, window=24, surf=None, orb=None,
                    phaseResponseThreshold=0.15, orbMaxDistance=-1, enhance=(0, 0.0, 0), hint=None):
        p = GridParams()
        if hint is not None and len(hint):
            arr = np.ascontiguousarray(hint, np.int32)
            p._hint_keepalive = arr                          # the struct only holds the address
            p.path_hint = arr.ctypes.data_as(C.POINTER(C.c_int32))
            p.path_hint_len = len(arr)
        p.method = {"surf": 0, "orb": 1, "phase": 2}[method]
        p.offset_evaluate, p.direct_incre, p.window, p.orb_max_dist = int(offsetEvaluate), int(directIncre), int(window), int(orbMaxDistance)
        p.enhance_mode, p.clip_limit, p.tile_grid = int(enhance[0]), float(enhance[1]), int(enhance[2])
        p.roi_ratio, p.search_ratio, p.phase_threshold = float(roiRatio), float(searchRatio), float(phaseResponseThreshold)
        p.surf = surf or Engine.surf_params()
        p.orb = orb or Engine.orb_params()
        return p

    def pairs_offsets(self, handles, shapes, params, first=0, last=None, direction=1, midpath=False, stop_on_fail=False):
        """vfsms_pairs_offsets: every pair of [first, last) of the path through the library's own candidate state machine.
        -> (int32[last-first, 6] = status, dx, dy, direction, i, votes; direction out; (attempts, batches, capacity retries))."""
        n = len(shapes)
        last = n - 1 if last is None else last
        hs = np.array([h if h is not None else 0 for h in handles], np.int64)
        sh = np.ascontiguousarray(np.array([[s[0], s[1]] for s in shapes], np.int32))
        out = np.zeros((max(last - first, 0), 6), np.int32)
        d_out = C.c_int32()
        stats = np.zeros(8, np.int64)
        self._check(self.lib.vfsms_pairs_offsets(self.ctx, _ptr(hs), _ptr(sh), n, int(first), int(last), int(direction), int(bool(midpath)),
                                                 int(bool(stop_on_fail)), C.byref(params), _ptr(out), C.byref(d_out), _ptr(stats)))
        return out, d_out.value, tuple(int(v) for v in stats)

    def pairs_offsets_blind(self, handles, shapes, params, first, last, per):
        """vfsms_pairs_offsets_blind: the chunk [first, last) for every possible incoming direction -> (int32[4, per, 6], int32[4] directions out, stats)."""
        n = len(shapes)
        hs = np.array([h if h is not None else 0 for h in handles], np.int64)
        sh = np.ascontiguousarray(np.array([[s[0], s[1]] for s in shapes], np.int32))
        out = np.zeros((4, max(per, 1), 6), np.int32)
        d_out = np.zeros(4, np.int32)
        stats = np.zeros(8, np.int64)
        self._check(self.lib.vfsms_pairs_offsets_blind(self.ctx, _ptr(hs), _ptr(sh), n, int(first), int(last), int(max(per, 1)), C.byref(params),
                                                       _ptr(out), _ptr(d_out), _ptr(stats)))
        return out[:, :per], d_out, tuple(int(v) for v in stats)

    # -- resident feature sets (Stitcher.tempImageFeature's payload kept in HBM) -----------------------------------------------
    def features_surf(self, tile_handle, rect, params=None, enhance=(0, 0.0, 0)):
        """SURF of rect = (y0, x0, h, w) of a resident tile -> (feature handle, n keypoints); nothing returns to the host."""
        params = params or self.surf_params()
        f, n = C.c_int64(), C.c_int()
        y0, x0, h, w = [int(v) for v in rect]
        self._check(self.lib.vfsms_features_surf(self.ctx, C.c_int64(tile_handle), y0, x0, h, w, C.byref(params), int(enhance[0]),
                                                 float(enhance[1]), int(enhance[2]), C.byref(f), C.byref(n)))
        return f.value, n.value

    def features_surf_batch(self, tile_handles, params=None, enhance=(0, 0.0, 0)):
        """whole-tile SURF of many resident tiles in fused launches -> (feature handles, keypoint counts)"""
        params = params or self.surf_params()
        n = len(tile_handles)
        th = np.ascontiguousarray(tile_handles, np.int64)
        feats = np.zeros(max(n, 1), np.int64); counts = np.zeros(max(n, 1), np.int32)
        self._check(self.lib.vfsms_features_surf_batch(self.ctx, _ptr(th), n, C.byref(params), int(enhance[0]), float(enhance[1]), int(enhance[2]),
                                                       _ptr(feats), _ptr(counts)))
        return [int(f) for f in feats[:n]], [int(c) for c in counts[:n]]

    def features_match_offset_batch(self, feats_a, feats_b, ratio=0.75, offset_evaluate=3):
        """matchDescriptors + getOffsetByMode of n (A, B) jobs in one batch -> int32[n][8]"""
        n = len(feats_a)
        fa = np.ascontiguousarray(feats_a, np.int64); fb = np.ascontiguousarray(feats_b, np.int64)
        out = np.zeros((max(n, 1), ATTEMPT_INTS), np.int32)
        self._check(self.lib.vfsms_features_match_offset_batch(self.ctx, _ptr(fa), _ptr(fb), n, float(ratio), int(offset_evaluate), _ptr(out)))
        return out[:n]

    def features_match_offset(self, feat_a, feat_b, ratio=0.75, offset_evaluate=3):
        """matchDescriptors + getOffsetByMode on two resident sets -> int32[8] = status, dx, dy, votes, nA, nB, nMatches, 0."""
        out = np.zeros(ATTEMPT_INTS, np.int32)
        self._check(self.lib.vfsms_features_match_offset(self.ctx, C.c_int64(feat_a), C.c_int64(feat_b), float(ratio), int(offset_evaluate), _ptr(out)))
        return out

    def features_download(self, feat, n, dim=64):
        kxy = np.empty((max(n, 1), 2), np.float32); desc = np.empty((max(n, 1), dim), np.float32)
        nn, dd = C.c_int(), C.c_int()
        self._check(self.lib.vfsms_features_download(self.ctx, C.c_int64(feat), _ptr(kxy), _ptr(desc), max(n, 1), C.byref(nn), C.byref(dd)))
        return kxy[:nn.value].copy(), desc[:nn.value].copy()

    def features_free(self, feat):
        self._check(self.lib.vfsms_features_free(self.ctx, C.c_int64(feat)))

    def enhance(self, img, mode, clip_limit=20.0, tile_size=5):
        """mode 1: cv2.equalizeHist(img); mode 2: cv2.createCLAHE(clip_limit, (tile_size, tile_size)).apply(img)."""
        img = _u8_2d(img)
        out = np.empty(img.shape, np.uint8)
        self._check(self.lib.vfsms_enhance_u8(self.ctx, _ptr(img), img.shape[0], img.shape[1], img.strides[0], int(mode), float(clip_limit),
                                              int(tile_size), _ptr(out)))
        return out

    def attempt_phase_batch(self, jobs):
        n = len(jobs)
        out = np.zeros((n, 3), np.float64)
        if n == 0:
            return out
        arr = jobs if isinstance(jobs, C.Array) else self.make_jobs(jobs)
        self._check(self.lib.vfsms_attempt_phase_batch(self.ctx, arr, n, _ptr(out)))
        return out

    # -- canvas ----------------------------------------------------------------------------------------------------
    def canvas_create(self, rows, cols, ch):
        h = C.c_int64()
        self._check(self.lib.vfsms_canvas_create(self.ctx, int(rows), int(cols), int(ch), C.byref(h)))
        return h.value

    def canvas_free(self, handle):
        self._check(self.lib.vfsms_canvas_free(self.ctx, C.c_int64(handle)))

    def canvas_paste(self, handle, tile, y0, x0):
        tile = np.ascontiguousarray(tile, np.uint8)
        self._check(self.lib.vfsms_canvas_paste(self.ctx, C.c_int64(handle), _ptr(tile), tile.shape[0], tile.shape[1], int(y0), int(x0)))

    def canvas_fuse_tile(self, handle, tile, y0, x0, roi, dx, dy, method=0):
        """method 0: fadeInAndFadeOut, 1: trigonometric"""
        tile = np.ascontiguousarray(tile, np.uint8)
        info = np.zeros(4, np.int32)
        ry0, rx0, ry1, rx1 = [int(v) for v in roi]
        self._check(self.lib.vfsms_canvas_fuse_tile_m(self.ctx, C.c_int64(handle), _ptr(tile), tile.shape[0], tile.shape[1],
                                                      int(y0), int(x0), ry0, rx0, ry1, rx1, int(dx), int(dy), int(method), _ptr(info)))
        return info

    def canvas_blend_tile(self, handle, tile, y0, x0, roi, mode):
        """fuseMethod average / maximum / minimum (mode 0 / 1 / 2) of a host tile into the canvas."""
        tile = np.ascontiguousarray(tile, np.uint8)
        ry0, rx0, ry1, rx1 = [int(v) for v in roi]
        self._check(self.lib.vfsms_canvas_blend_tile(self.ctx, C.c_int64(handle), _ptr(tile), tile.shape[0], tile.shape[1],
                                                     int(y0), int(x0), ry0, rx0, ry1, rx1, int(mode)))

    def canvas_blend_tile_resident(self, handle, tile_handle, y0, x0, roi, mode):
        """canvas_blend_tile with a tile that is already resident; enqueue only"""
        ry0, rx0, ry1, rx1 = [int(v) for v in roi]
        self._check(self.lib.vfsms_canvas_blend_tile_resident(self.ctx, C.c_int64(handle), C.c_int64(tile_handle), int(y0), int(x0),
                                                              ry0, rx0, ry1, rx1, int(mode)))

    def canvas_paste_tile(self, handle, tile_handle, y0, x0):
        """paste of a single-channel tile that is already resident in HBM (tile_upload handle)."""
        self._check(self.lib.vfsms_canvas_paste_tile(self.ctx, C.c_int64(handle), C.c_int64(tile_handle), int(y0), int(x0)))

    def canvas_fuse_tile_resident(self, handle, tile_handle, y0, x0, roi, dx, dy, want_info=False, method=0):
        """want_info=False: the call only enqueues work; geometry errors surface in canvas_download.  method 0: fadeInAndFadeOut, 1: trigonometric"""
        info = np.zeros(4, np.int32) if want_info else None
        ry0, rx0, ry1, rx1 = [int(v) for v in roi]
        self._check(self.lib.vfsms_canvas_fuse_tile_resident_m(self.ctx, C.c_int64(handle), C.c_int64(tile_handle), int(y0), int(x0),
                                                               ry0, rx0, ry1, rx1, int(dx), int(dy), int(method), _ptr(info) if want_info else None))
        return info

    def canvas_assemble_resident(self, handle, tile_handles, geom):
        """The mosaic walk over resident tiles as one call.  geom: int32 [n][9] = y0, x0, ry0, rx0, ry1, rx1, dx, dy, mode
        (mode -1 paste, 0 fadeInAndFadeOut, 1 trigonometric, 2 / 3 / 4 average / maximum / minimum); enqueue only, geometry errors surface in canvas_download."""
        th = np.ascontiguousarray(tile_handles, np.int64)
        g = np.ascontiguousarray(geom, np.int32).reshape(-1, 9)
        if len(th) != len(g):
            raise ValueError("canvas_assemble_resident: one geometry row per tile")
        self._check(self.lib.vfsms_canvas_assemble_resident(self.ctx, C.c_int64(handle), len(th), _ptr(th), _ptr(g)))

    def canvas_download(self, handle, rows, cols, ch):
        out = np.empty((rows, cols, ch) if ch > 1 else (rows, cols), np.uint8)
        self._check(self.lib.vfsms_canvas_download(self.ctx, C.c_int64(handle), _ptr(out)))
        return out

    def canvas_download_rows(self, handle, row0, nrows, cols, ch, out=None):
        """Rows [row0, row0 + nrows) of the mosaic (vfsms_canvas_download_rows); `out` may be a preallocated band buffer."""
        shape = (nrows, cols, ch) if ch > 1 else (nrows, cols)
        if out is None:
            out = np.empty(shape, np.uint8)
        if out.shape != shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("canvas_download_rows: `out` must be a C-contiguous u8 array of shape %r" % (shape,))
        self._check(self.lib.vfsms_canvas_download_rows(self.ctx, C.c_int64(handle), int(row0), int(nrows), _ptr(out)))
        return out

    BAND_RING = 3

    def canvas_download_bands(self, handle, rows, cols, ch, band_rows=4096, transient=False):
        """Generator of (row0, band) over the whole mosaic, `band_rows` rows at a time (streamed write-out of large mosaics).  With
        `transient` the bands are views of a ring of BAND_RING pinned buffers kept by the engine (the copy is a DMA at link speed, no
        page faults of a fresh array per band): a band is valid until BAND_RING - 1 further bands have been requested -- for consumers
        that are done with a band by then (JpegBandWriter declares it: `transient_bands`)."""
        if not transient:
            for r0 in range(0, rows, band_rows):
                yield r0, self.canvas_download_rows(handle, r0, min(band_rows, rows - r0), cols, ch)
            return
        need = min(band_rows, rows) * cols * ch
        ring = self.__dict__.get("_band_ring")
        if ring is None or ring[0].size < need:
            ring = self.__dict__["_band_ring"] = [self.pinned_empty((need,)) for _ in range(self.BAND_RING)]      # (an outgrown ring is freed with the engine)
        for k, r0 in enumerate(range(0, rows, band_rows)):
            n = min(band_rows, rows - r0)
            out = ring[k % self.BAND_RING][:n * cols * ch].reshape((n, cols, ch) if ch > 1 else (n, cols))
            yield r0, self.canvas_download_rows(handle, r0, n, cols, ch, out=out)


_default_engine = None


def jpeg_decode(data, want_planes=False):
    """vfsms_jpeg_decode: the library's own JPEG decode (system libjpeg-turbo), no GPU involved -> u8 (h, w) -- the grayscale decode -- or,
    with want_planes and a three-component file, u8 (h, w, 3) Y Cb Cr.  None when the library does not take the file (see tile_fill_jpeg)."""
    lib = load_library()
    h, w, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib.vfsms_jpeg_decode(data, len(data), int(bool(want_planes)), None, 0, C.byref(h), C.byref(w), C.byref(c))
    if rc != VFSMS_ERR_CAPACITY:
        return None
    out = np.empty((h.value, w.value, c.value), np.uint8)
    rc = lib.vfsms_jpeg_decode(data, len(data), int(bool(want_planes)), out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(h), C.byref(w), C.byref(c))
    if rc != VFSMS_OK:
        return None
    return out[:, :, 0] if c.value == 1 else out


def jpeg_decode_raw420(data):
    """vfsms_jpeg_decode(want_planes = 2): the DOWNSAMPLED planes of a 4:2:0 Y Cb Cr file as the entropy decode + IDCT leave them (what
    vfsms_tile_fill_jpeg stages for the device's upsampler) -> (Y u8 (ph, pw), Cb u8 (ph / 2, pw / 2), Cr, h, w) on the iMCU grid (pw, ph =
    w, h rounded up to 16), or None when the file is not 4:2:0 / not taken."""
    lib = load_library()
    h, w, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib.vfsms_jpeg_decode(data, len(data), 2, None, 0, C.byref(h), C.byref(w), C.byref(c))
    if rc != VFSMS_ERR_CAPACITY:
        return None
    pw, ph = (w.value + 15) & ~15, (h.value + 15) & ~15
    out = np.empty(pw * ph * 3 // 2, np.uint8)
    rc = lib.vfsms_jpeg_decode(data, len(data), 2, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(h), C.byref(w), C.byref(c))
    if rc != VFSMS_OK:
        return None
    n = pw * ph
    return out[:n].reshape(ph, pw), out[n:n + n // 4].reshape(ph // 2, pw // 2), out[n + n // 4:].reshape(ph // 2, pw // 2), h.value, w.value


def _last_error(lib):
    buf = C.create_string_buffer(512)
    lib.vfsms_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def jpeg_encode(img, bgr=True, quality=95):
    """vfsms_jpeg_encode: u8 (h, w) or (h, w, 3) rows (B G R when `bgr`, the canvas order) -> a complete baseline JPEG stream as a u8 array,
    libjpeg defaults + `quality` (cv2.imwrite's: 95) -- byte for byte what Pillow / cv2 write for the same pixels.  None when the host has
    no libjpeg.so.8 (the caller writes through Pillow).  No GPU involved; releases the interpreter lock."""
    lib = load_library()
    assert img.dtype == np.uint8 and img.ndim in (2, 3) and img.strides[-1] == 1 and (img.ndim == 2 or (img.shape[2] == 3 and img.strides[1] == 3))
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else 3
    cap = max(1 << 16, h * w * ch // 4)
    n = C.c_size_t()
    for _ in range(2):
        out = np.empty(cap, np.uint8)
        rc = lib.vfsms_jpeg_encode(img.ctypes.data_as(C.c_void_p), h, w, ch, img.strides[0], int(bool(bgr)), int(quality),
                                   out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        if rc == VFSMS_OK:
            return out[:n.value]
        if rc == VFSMS_ERR_UNSUPPORTED:
            return None
        if rc != VFSMS_ERR_CAPACITY:
            raise VfsmsError("libvfsms error %d: %s" % (rc, _last_error(lib)))
        cap = n.value
    raise VfsmsError("jpeg_encode: the stream did not fit the size the library asked for")


def jpeg_join(parts, stripe_rows, total_rows):
    """vfsms_jpeg_join: the streams of consecutive stripes (jpeg_encode of rows [k * stripe_rows, (k + 1) * stripe_rows) of one image) -> ONE
    JPEG whose restart intervals are the stripes; u8 array."""
    lib = load_library()
    n = len(parts)
    ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in parts])
    sizes = (C.c_size_t * n)(*[p.size for p in parts])
    need = C.c_size_t()
    rc = lib.vfsms_jpeg_join(ptrs, sizes, n, int(stripe_rows), int(total_rows), None, 0, C.byref(need))
    if rc == VFSMS_OK:
        out = np.empty(need.value, np.uint8)
        rc = lib.vfsms_jpeg_join(ptrs, sizes, n, int(stripe_rows), int(total_rows), out.ctypes.data_as(C.c_void_p), out.size, C.byref(need))
    if rc != VFSMS_OK:
        raise VfsmsError("libvfsms error %d: %s" % (rc, _last_error(lib)))
    return out


def default_engine():
    """Process-wide engine on the GPU selected by LOCAL_RANK (one process per GPU) or device 0."""
    global _default_engine
    if _default_engine is None:
        dev = int(os.environ.get("VFSMS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        lib = load_library()
        n = lib.vfsms_device_count()
        if n <= 0:
            raise VfsmsError("no HIP device visible: imagestitch_amd needs an MI355X (there is no CPU fallback)")
        _default_engine = Engine(dev % n)
    return _default_engine


def pairs_offsets_blind_eval(attempts, shapes, params, first, last, per):
    """vfsms_pairs_offsets_blind_eval over a Python evaluator (needs no GPU) -> (int32[4, per, 6], int32[4], stats)."""
    lib = load_library()
    n = len(shapes)
    err = []

    def cb(_user, items, count, rows):
        try:
            res = attempts([(items[k].pair, items[k].direction, items[k].i) for k in range(count)])
            for k, r in enumerate(res):
                for c in range(min(len(r), ATTEMPT_INTS)):
                    rows[k * ATTEMPT_INTS + c] = int(r[c])
            return 0
        except Exception as e:            # never let an exception cross the C frame
            err.append(e)
            return -1
    sh = np.ascontiguousarray(np.array([[s[0], s[1]] for s in shapes], np.int32))
    out = np.zeros((4, max(per, 1), 6), np.int32)
    d_out = np.zeros(4, np.int32)
    stats = np.zeros(8, np.int64)
    rc = lib.vfsms_pairs_offsets_blind_eval(ATTEMPT_EVAL(cb), None, _ptr(sh), n, int(first), int(last), int(max(per, 1)), C.byref(params),
                                            _ptr(out), _ptr(d_out), _ptr(stats))
    if err:
        raise err[0]
    if rc != VFSMS_OK:
        buf = C.create_string_buffer(512)
        lib.vfsms_last_error(buf, 512)
        raise VfsmsError("libvfsms error %d: %s" % (rc, buf.value.decode(errors="replace")))
    return out[:, :per], d_out, tuple(int(v) for v in stats)


def pairs_offsets_eval(attempts, shapes, params, first=0, last=None, direction=1, midpath=False, stop_on_fail=False):
    """vfsms_pairs_offsets_eval: the library's candidate state machine over a Python evaluator of fused batches (needs no GPU).
    attempts(items) with items = [(pair, direction, i), ...] -> rows [[status, raw dx, raw dy, votes, nA, nB, ...], ...]."""
    lib = load_library()
    n = len(shapes)
    last = n - 1 if last is None else last
    err = []

    def cb(_user, items, count, rows):
        try:
            res = attempts([(items[k].pair, items[k].direction, items[k].i) for k in range(count)])
            for k, r in enumerate(res):
                for c in range(min(len(r), ATTEMPT_INTS)):
                    rows[k * ATTEMPT_INTS + c] = int(r[c])
            return 0
        except Exception as e:            # never let an exception cross the C frame
            err.append(e)
            return -1
    sh = np.ascontiguousarray(np.array([[s[0], s[1]] for s in shapes], np.int32))
    out = np.zeros((max(last - first, 0), 6), np.int32)
    d_out = C.c_int32()
    stats = np.zeros(8, np.int64)
    rc = lib.vfsms_pairs_offsets_eval(ATTEMPT_EVAL(cb), None, _ptr(sh), n, int(first), int(last), int(direction), int(bool(midpath)),
                                      int(bool(stop_on_fail)), C.byref(params), _ptr(out), C.byref(d_out), _ptr(stats))
    if err:
        raise err[0]
    if rc != VFSMS_OK:
        buf = C.create_string_buffer(512)
        lib.vfsms_last_error(buf, 512)
        raise VfsmsError("libvfsms error %d: %s" % (rc, buf.value.decode(errors="replace")))
    return out, d_out.value, tuple(int(v) for v in stats)
